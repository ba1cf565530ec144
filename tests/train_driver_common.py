"""Shared by tests/test_reference_train_driver_source.py, tests/golden/make_train_driver_trace.py and
tests/test_gpu_train_driver_trace.py: the scratch tree the reference's TRAINING drivers are executed in
(exp_clevr/train_clevr_gt_layout.py, exp_clevr/train_clevr_rl_gt_layout.py), and the import map that answers
their imports with the drop-in.

The scripts hard-code their hyper-parameters (max_iter = 80000, N = 64, snapshot_interval = 10000): they run
UNMODIFIED, so the scratch side decides how long and how wide a run is -- the data reader the import map hands
out delivers `BATCH` questions per batch (whatever batch_size the script asks for) and stops after `N_ITERS`
batches, which ends the script's `for n_iter, batch in enumerate(dataset_trn.batches())` loop the way an
exhausted one-pass reader would."""
import os
import shutil
import sys

import numpy as np

import eval_driver_common as EC
from n2nmn_amd import synth
from n2nmn_amd.spec import Dims

REF = EC.REF
SCRIPT_GT = os.path.join(REF, 'exp_clevr', 'train_clevr_gt_layout.py')
SCRIPT_RL = os.path.join(REF, 'exp_clevr', 'train_clevr_rl_gt_layout.py')
SCRIPT_SCRATCH = os.path.join(REF, 'exp_clevr', 'train_clevr_scratch.py')
N_QUESTIONS = 24
BATCH = 6                  # questions per batch delivered to the script
N_ITERS = 21               # log_interval = 20: iteration 20 writes the TensorBoard summary
T_DECODER = 10             # train_clevr_gt_layout.py:35
T_DECODER_SCRATCH = 6       # train_clevr_scratch.py:35


def train_dims(t_decoder=None):
    return Dims(T_decoder=t_decoder or T_DECODER)


def build_scratch(tmp_path, d: Dims, with_snapshot=False):
    """exp_clevr/data/{vocabulary files, imdb/imdb_trn.npy}, feature files; with_snapshot: a TensorFlow-format
    checkpoint of seeded weights at the RL script's default --pretrained_model path."""
    data = tmp_path / 'exp_clevr' / 'data'
    (data / 'imdb').mkdir(parents=True)
    for f in ('vocabulary_clevr.txt', 'vocabulary_layout.txt', 'answers_clevr.txt'):
        shutil.copy(os.path.join(REF, 'exp_clevr', 'data', f), data / f)      # data files, scratch only
    words = [l.strip() for l in open(data / 'vocabulary_clevr.txt')]
    answers = [l.strip() for l in open(data / 'answers_clevr.txt')]
    rng = np.random.default_rng(17)
    feat_dir = tmp_path / 'feat'
    feat_dir.mkdir()
    imdb = []
    for i in range(N_QUESTIONS):
        fp = str(feat_dir / ('%03d.npy' % i))
        np.save(fp, EC.feature_of(i, d))
        L = int(rng.integers(3, 20))
        imdb.append(dict(image_path='CLEVR_syn_%06d.png' % i, feature_path=fp,
                         question_tokens=[words[int(rng.integers(0, len(words)))] for _ in range(L)],
                         answer=answers[int(rng.integers(0, len(answers)))],
                         gt_layout_tokens=list(synth.CLEVR_LAYOUT_TEMPLATES[i % 10])))
    np.save(data / 'imdb' / 'imdb_trn.npy', np.array(imdb, dtype=object), allow_pickle=True)
    w = synth.make_weights(d, seed=3)
    if with_snapshot:
        from n2nmn_amd import tf_checkpoint
        snap = tmp_path / 'exp_clevr' / 'tfmodel' / 'clevr_gt_layout'
        snap.mkdir(parents=True)
        tf_checkpoint.write_checkpoint(str(snap / '00050000'), w)
    return data, words, answers, w


def short_reader(batches_seen):
    """the DataReader the import map hands to the script: BATCH questions per batch, N_ITERS batches, every
    delivered batch appended to `batches_seen`"""
    from n2nmn_amd import data_reader

    class ShortReader(data_reader.DataReader):
        def __init__(self, imdb_file, **kw):
            kw['batch_size'] = BATCH
            kw['shuffle'] = False          # (deterministic order: the recording is replayed on another box)
            super().__init__(imdb_file, **kw)

        def batches(self):
            for i, b in enumerate(super().batches()):
                if i >= N_ITERS:
                    return
                batches_seen.append({k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v)
                                     for k, v in b.items()})
                yield b
    return ShortReader


def run_train_script(script, tmp_path, monkeypatch, engine_cls, trainer_cls, recorder=None, argv=None,
                     with_snapshot=False, seed_weights=True, t_decoder=None):
    """Executes the reference's training script, every line of it, in a scratch tree.  engine_cls / trainer_cls
    replace n2nmn_amd.engine.Engine / n2nmn_amd.train.Trainer behind the drop-in's Python face (None: the HIP
    ones).  seed_weights: `sess.run(tf.global_variables_initializer())` loads the scratch's seeded weights
    instead of fresh draws, so that two runs (oracle double here, HIP engine on the GPU box) start equal."""
    import runpy
    from n2nmn_amd import nmn3_model, runtime, runtime_train, train
    sys.dont_write_bytecode = True
    d = train_dims(t_decoder)
    data, words, answers, w = build_scratch(tmp_path, d, with_snapshot)
    if engine_cls is not None:
        monkeypatch.setattr(nmn3_model, 'Engine', engine_cls)
    if trainer_cls is not None:
        monkeypatch.setattr(train, 'Trainer', trainer_cls)
    monkeypatch.setattr(runtime, '_MODELS', [])
    monkeypatch.setattr(runtime_train, '_GLOBALS', [])
    if seed_weights:
        monkeypatch.setattr(runtime_train, 'initial_weights', lambda shapes, seed=0: {k: w[k] for k in shapes})
    if recorder is not None:
        recorder.install(monkeypatch)
    seen = []
    mods = EC.import_map()
    mods['util.clevr_train.data_reader'] = EC.module('util.clevr_train.data_reader', DataReader=short_reader(seen))
    for name, mod in mods.items():
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.setattr(sys, 'argv', list(argv or [os.path.basename(script)]))
    monkeypatch.chdir(tmp_path)
    g = runpy.run_path(script, run_name='__main__')
    return g, d, seen, w
