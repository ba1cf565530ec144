"""Shared by tests/test_reference_driver_source.py, tests/golden/make_eval_driver_trace.py and
tests/test_gpu_eval_driver_trace.py: the synthetic working directory the reference's
exp_clevr/eval_clevr.py is executed in, and the import map that answers its imports with the drop-in.

Image features are a pure function of the question index (`feature_of`), so the GPU replay of the
recorded run can rebuild them without the 1.4 MB-per-question arrays travelling in a fixture."""
import os
import shutil
import sys
import types

import numpy as np

from n2nmn_amd import synth
from n2nmn_amd.spec import Dims

REF = '/root/reference'
SCRIPT = os.path.join(REF, 'exp_clevr', 'eval_clevr.py')
N_QUESTIONS = 70            # one full batch of 64 and a short one of 6
ARGV = ['eval_clevr.py', '--exp_name', 'exp0', '--snapshot_name', '00050000', '--test_split', 'syn']


def feature_of(i: int, d: Dims) -> np.ndarray:
    """pool5-shaped features of synthetic question i: [1, H, W, D] float32, >= 0."""
    rng = np.random.default_rng(7000 + i)
    return np.maximum(rng.standard_normal((1, d.H, d.W, d.D)), 0).astype(np.float32)


def module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def build_scratch(tmp_path, d: Dims):
    """exp_clevr/data/{vocabulary files, imdb/imdb_syn.npy}, feature files, an .npz 'snapshot' of
    seeded weights under exp_clevr/tfmodel/exp0/.  Returns (data dir, words, answers, weights)."""
    data = tmp_path / 'exp_clevr' / 'data'
    (data / 'imdb').mkdir(parents=True)
    for f in ('vocabulary_clevr.txt', 'vocabulary_layout.txt', 'answers_clevr.txt'):
        shutil.copy(os.path.join(REF, 'exp_clevr', 'data', f), data / f)      # data files, scratch only
    words = [l.strip() for l in open(data / 'vocabulary_clevr.txt')]
    answers = [l.strip() for l in open(data / 'answers_clevr.txt')]
    assert (len(words), len(answers)) == (d.num_vocab_txt, d.num_choices)
    rng = np.random.default_rng(5)
    feat_dir = tmp_path / 'feat'
    feat_dir.mkdir()
    imdb = []
    for i in range(N_QUESTIONS):
        fp = str(feat_dir / ('%03d.npy' % i))
        np.save(fp, feature_of(i, d))
        L = int(rng.integers(3, d.T_encoder + 1))
        imdb.append(dict(image_path='CLEVR_syn_%06d.png' % i, feature_path=fp,
                         question_tokens=[words[int(rng.integers(0, len(words)))] for _ in range(L)],
                         answer=answers[int(rng.integers(0, len(answers)))],
                         gt_layout_tokens=list(synth.CLEVR_LAYOUT_TEMPLATES[i % 10])))
    np.save(data / 'imdb' / 'imdb_syn.npy', np.array(imdb, dtype=object), allow_pickle=True)
    w = synth.make_weights(d, seed=0)
    snap = tmp_path / 'exp_clevr' / 'tfmodel' / 'exp0'
    snap.mkdir(parents=True)
    np.savez(snap / '00050000.npz', **w)
    return data, words, answers, w


def import_map():
    """sys.modules entries that answer the driver's imports with the drop-in."""
    from n2nmn_amd import data_reader, nmn3_assembler, nmn3_model, runtime
    return {
        'tensorflow': module('tensorflow', **runtime.tf.__dict__),
        'models_clevr': module('models_clevr'),
        'models_clevr.nmn3_assembler': module('models_clevr.nmn3_assembler', Assembler=nmn3_assembler.Assembler),
        'models_clevr.nmn3_model': module('models_clevr.nmn3_model', NMN3Model=nmn3_model.NMN3Model),
        'util': module('util'), 'util.clevr_train': module('util.clevr_train'),
        'util.clevr_train.data_reader': module('util.clevr_train.data_reader',
                                               DataReader=data_reader.DataReader),
    }


def fetch_name(f):
    """a fetched graph node by the name the reference's scripts give it"""
    from n2nmn_amd import runtime, runtime_train
    if isinstance(f, runtime.Fetch):
        return f.name
    if isinstance(f, runtime_train.Const):
        return 'train_step' if f.deps else 'const'
    if isinstance(f, runtime_train.Op):
        return 'avg_sample_loss' if f.kind == 'mean' else f.kind
    return type(f).__name__


class SessionRecorder:
    """Wraps n2nmn_amd.runtime.Session.partial_run and NMN3Model.__init__ while the reference's script
    runs: what the script ASKED of the drop-in (constructor keywords, every partial_run with its feeds)
    and what it got back.  Placeholders are recorded by name; image features by question index."""

    def __init__(self, d: Dims = None, model_cls=None, feature_fn=None, n_questions=N_QUESTIONS):
        """model_cls: the drop-in model class whose constructor the script calls (default: the models_clevr
        NMN3Model); feature_fn(i) -> the image feed row(s) of question i, [1, ...] (default: feature_of);
        None: image feeds are recorded as arrays"""
        self.d = d
        self.model_cls = model_cls
        self.feature_fn = feature_fn if feature_fn is not None else ((lambda i: feature_of(i, d)) if d is not None else None)
        self.n_questions = n_questions
        self.model_kwargs = None
        self.calls = []
        self.setups = {}
        self.placeholders = {}
        self._feats = None

    def _image_ids(self, feat):
        if self._feats is None:
            self._feats = [self.feature_fn(i)[0] for i in range(self.n_questions)]
        ids = []
        for row in np.asarray(feat):
            hit = [i for i, f in enumerate(self._feats) if np.array_equal(f, row)]
            ids.append(hit[0] if hit else -1)          # -1: a padding row of a short batch
        return ids

    def role(self, ph):
        """name of a placeholder by what the script uses it for (Placeholder names count per process)"""
        for k, v in self.placeholders.items():
            if v[0] is ph:
                return k
        if getattr(ph, 'name', None) == 'loom_input_tensor':
            return 'loom_input_tensor'
        return 'extra:%s:%s' % (ph.dtype, list(ph.shape) if ph.shape is not None else None)

    def install(self, monkeypatch):
        from n2nmn_amd import nmn3_model, runtime
        from n2nmn_amd.nmn3_assembler import PackedLayouts
        rec = self
        model_cls = self.model_cls if self.model_cls is not None else nmn3_model.NMN3Model
        init0 = model_cls.__init__
        run0 = runtime.Session.partial_run
        setup0 = runtime.Session.partial_run_setup

        def init(self, image_feat_grid, text_seq_batch, seq_length_batch, **kw):
            rec.model_kwargs = {k: (v if isinstance(v, (int, float, bool, str, type(None))) else type(v).__name__)
                                for k, v in kw.items()}
            rec.placeholders = dict(image_feat_grid=(image_feat_grid, image_feat_grid.dtype, image_feat_grid.shape),
                                    text_seq_batch=(text_seq_batch, text_seq_batch.dtype, text_seq_batch.shape),
                                    seq_length_batch=(seq_length_batch, seq_length_batch.dtype,
                                                      seq_length_batch.shape))
            init0(self, image_feat_grid, text_seq_batch, seq_length_batch, **kw)

        def partial_run_setup(self, fetches, feeds=None):
            h = setup0(self, fetches, feeds)
            rec.setups[id(h)] = dict(fetches=[fetch_name(f) for f in h.allowed_fetches],
                                     feeds=[rec.role(p) for p in h.allowed_feeds], keep=h)   # (keeps id(h) unique)
            return h

        def partial_run(self, handle, fetches, feed_dict=None):
            out = run0(self, handle, fetches, feed_dict)
            feeds = {}
            for k, v in (feed_dict or {}).items():
                role = rec.role(k)
                if isinstance(v, PackedLayouts):
                    feeds[role] = ('packed', v)
                elif role == 'image_feat_grid' and rec.feature_fn is not None:
                    feeds[role] = ('image_ids', rec._image_ids(v))
                else:
                    feeds[role] = ('array', np.asarray(v))
            # (a copy: exp_vqa/eval_vqa2.py:137 writes into the array it got back)
            if isinstance(fetches, (list, tuple)):     # the training drivers fetch tuples (train_clevr_gt_layout.py:171,190)
                rec.calls.append(dict(fetch='(%s)' % ', '.join(fetch_name(f) for f in fetches), feeds=feeds,
                                      handle=id(handle), result_list=[np.array(o, copy=True) for o in out]))
            else:
                rec.calls.append(dict(fetch=fetches.name, feeds=feeds, handle=id(handle), result=np.array(out, copy=True)))
            return out

        monkeypatch.setattr(model_cls, '__init__', init)
        monkeypatch.setattr(runtime.Session, 'partial_run', partial_run)
        monkeypatch.setattr(runtime.Session, 'partial_run_setup', partial_run_setup)


def run_reference_script(tmp_path, monkeypatch, engine_cls, recorder=None):
    """Executes the reference's eval_clevr.py, every line of it, in a scratch tree; the drop-in answers
    its imports; `engine_cls` replaces n2nmn_amd.engine.Engine behind the drop-in's Python face."""
    import runpy
    from n2nmn_amd import nmn3_model, runtime
    sys.dont_write_bytecode = True
    d = Dims()
    data, words, answers, w = build_scratch(tmp_path, d)
    monkeypatch.setattr(nmn3_model, 'Engine', engine_cls)
    monkeypatch.setattr(runtime, '_MODELS', [])
    if recorder is not None:
        recorder.install(monkeypatch)
    for name, mod in import_map().items():
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.setattr(sys, 'argv', list(ARGV))
    monkeypatch.chdir(tmp_path)
    g = runpy.run_path(SCRIPT, run_name='__main__')
    return g, d, data, words, answers, w
