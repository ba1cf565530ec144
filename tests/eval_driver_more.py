"""The reference's OTHER two inference drivers, unmodified, against the drop-in (VERDICT r4 "missing" #3):

    exp_vqa/eval_vqa2.py      (BASELINE.json configs[4]; incl. the `scores_val[:, 0] = -1e10` step, :137)
    exp_shapes/eval_shapes.py (BASELINE.json configs[0]: the reference's own CPU-runnable plumbing case)

Same arrangement as tests/eval_driver_common.py has for exp_clevr/eval_clevr.py: the script is executed as
`__main__` in a scratch tree, its imports are answered by the drop-in's Python faces
(n2nmn_amd.models_vqa / n2nmn_amd.models_shapes / n2nmn_amd.runtime.tf), the compute behind the faces is a
CPU test double on the build box (tests/oracle_engine.py) and the HIP engine on the GPU box, and what the
script ASKED of the faces is recorded for the GPU replay (tests/test_gpu_eval_driver_trace.py).

VQA scratch: the reference's own vocabulary files (17 742 words, 3 001 answers, 5 layout tokens), a
synthetic imdb of 53 questions (one batch of 50 and a short one of 3) with 14 x 14 x 2048 features that are a
pure function of the question index.  SHAPES scratch: the reference's own dataset files (`train.tiny`: 64
images) and vocabularies."""
import os
import shutil
import sys

import numpy as np

import eval_driver_common as EC

REF = EC.REF
VQA_SCRIPT = os.path.join(REF, 'exp_vqa', 'eval_vqa2.py')
SHAPES_SCRIPT = os.path.join(REF, 'exp_shapes', 'eval_shapes.py')
VQA_ARGV = ['eval_vqa2.py', '--exp_name', 'exp0', '--snapshot_name', '00040000', '--test_split', 'syn']
SHAPES_ARGV = ['eval_shapes.py', '--exp_name', 'exp0', '--snapshot_name', '00040000', '--test_split', 'train.tiny']
VQA_N = 53


def vqa_dims():
    from n2nmn_amd.vqa import VQADims
    return VQADims(N=64)            # the face's default capacity (the script's batch is 50)


def vqa_feature_of(i: int) -> np.ndarray:
    """ResNet-shaped features of synthetic question i: [1, 14, 14, 2048] float32, >= 0, sparse"""
    rng = np.random.default_rng(9000 + i)
    return np.maximum(rng.standard_normal((1, 14, 14, 2048), dtype=np.float32) - 0.5, 0)


def vqa_weights():
    from n2nmn_amd import synth
    from n2nmn_amd.vqa import vqa_variable_shapes
    return synth.make_weights_from_shapes(vqa_variable_shapes(vqa_dims()), seed=0)


def build_vqa_scratch(tmp_path, imdb_dir='imdb_vqa_v2'):
    """imdb_dir: 'imdb_vqa_v2' for eval_vqa2.py (:51), 'imdb' for eval_vqa.py (:51) -- all the two scripts differ in"""
    data = tmp_path / 'exp_vqa' / 'data'
    (data / imdb_dir).mkdir(parents=True)
    for f in ('vocabulary_vqa.txt', 'vocabulary_layout.txt', 'answers_vqa.txt'):
        shutil.copy(os.path.join(REF, 'exp_vqa', 'data', f), data / f)          # data files, scratch only
    words = [l.strip() for l in open(data / 'vocabulary_vqa.txt')]
    answers = [l.strip() for l in open(data / 'answers_vqa.txt')]
    rng = np.random.default_rng(11)
    feat_dir = tmp_path / 'feat'
    feat_dir.mkdir()
    layouts = (['_Find', '_Describe'], ['_Find', '_Find', '_And', '_Describe'], ['_Find', '_Transform', '_Describe'])
    imdb = []
    for i in range(VQA_N):
        fp = str(feat_dir / ('%03d.npy' % i))
        np.save(fp, vqa_feature_of(i))
        L = int(rng.integers(2, 27))
        imdb.append(dict(image_path='COCO_syn_%06d.jpg' % i, feature_path=fp, question_id=1000 + i,
                         question_str='synthetic question %d' % i,
                         question_tokens=[words[int(rng.integers(0, len(words)))] for _ in range(L)],
                         gt_layout_tokens=list(layouts[i % 3])))
    np.save(data / imdb_dir / 'imdb_syn.npy', np.array(imdb, dtype=object), allow_pickle=True)
    w = vqa_weights()
    snap = tmp_path / 'exp_vqa' / 'tfmodel' / 'exp0'
    snap.mkdir(parents=True)
    np.savez(snap / '00040000.npz', **w)
    return data, words, answers, w


def vqa_import_map():
    from n2nmn_amd import models_vqa, runtime
    return {
        'tensorflow': EC.module('tensorflow', **runtime.tf.__dict__),
        'models_vqa': EC.module('models_vqa'),
        'models_vqa.nmn3_assembler': EC.module('models_vqa.nmn3_assembler', Assembler=models_vqa.Assembler),
        'models_vqa.nmn3_model': EC.module('models_vqa.nmn3_model', NMN3Model=models_vqa.NMN3Model),
        'util': EC.module('util'), 'util.vqa_train': EC.module('util.vqa_train'),
        'util.vqa_train.data_reader': EC.module('util.vqa_train.data_reader', DataReader=models_vqa.DataReader),
    }


def run_vqa_script(tmp_path, monkeypatch, engine_cls, recorder=None, script='eval_vqa2.py'):
    """exp_vqa/eval_vqa2.py (or eval_vqa.py, the VQAv1 form), every line of it; `engine_cls` replaces
    n2nmn_amd.vqa.VQAEngine behind the face"""
    import runpy
    from n2nmn_amd import models_vqa, runtime
    sys.dont_write_bytecode = True
    data, words, answers, w = build_vqa_scratch(tmp_path, 'imdb_vqa_v2' if script == 'eval_vqa2.py' else 'imdb')
    monkeypatch.setattr(models_vqa, 'VQAEngine', engine_cls)
    monkeypatch.setattr(runtime, '_MODELS', [])
    if recorder is not None:
        recorder.install(monkeypatch)
    for name, mod in vqa_import_map().items():
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.setattr(sys, 'argv', [script] + list(VQA_ARGV[1:]))
    monkeypatch.chdir(tmp_path)
    g = runpy.run_path(os.path.join(REF, 'exp_vqa', script), run_name='__main__')
    return g, data, words, answers, w


# ---- models_vqa TRAINING drivers (exp_vqa/train_vqa2_gt_layout.py, train_vqa2_rl_gt_layout.py, and the VQAv1 forms) ----
VQA_TRAIN_N = 24            # questions in the scratch imdb
VQA_TRAIN_BATCH = 6         # questions per batch the reader delivers (the scripts ask for N = 64)
VQA_TRAIN_ITERS = 21        # log_interval = 20: iteration 20 writes the summary


def vqa_train_dims():
    from n2nmn_amd.vqa import VQADims
    return VQADims(N=64, lstm_dim=1000)          # exp_vqa/train_vqa2_gt_layout.py:29: lstm_dim = 1000


def vqa_train_weights():
    from n2nmn_amd import synth
    from n2nmn_amd.vqa import vqa_variable_shapes
    return synth.make_weights_from_shapes(vqa_variable_shapes(vqa_train_dims()), seed=5)


def build_vqa_train_scratch(tmp_path, imdb_dir='imdb_vqa_v2', snapshot=None):
    """exp_vqa/data/{vocabulary files, vocabulary_vqa_glove.npy, <imdb_dir>/imdb_trainval2014.npy}, feature files;
    snapshot: '<experiment>/<iteration>' under exp_vqa/tfmodel/ that gets a TensorFlow-format checkpoint of the seeded
    weights (the RL scripts' --pretrained_model defaults: vqa2_gt_layout/00080000, vqa_gt_layout/00040000)."""
    data = tmp_path / 'exp_vqa' / 'data'
    (data / imdb_dir).mkdir(parents=True)
    for f in ('vocabulary_vqa.txt', 'vocabulary_layout.txt', 'answers_vqa.txt'):
        shutil.copy(os.path.join(REF, 'exp_vqa', 'data', f), data / f)          # data files, scratch only
    words = [l.strip() for l in open(data / 'vocabulary_vqa.txt')]
    answers = [l.strip() for l in open(data / 'answers_vqa.txt')]
    rng = np.random.default_rng(23)
    # (the real file holds GloVe rows; any [num_vocab_txt, 300] array exercises `tf.assign(embedding_mat, glove_mat)`)
    glove = (0.1 * rng.standard_normal((len(words), 300))).astype(np.float32)
    np.save(data / 'vocabulary_vqa_glove.npy', glove)
    feat_dir = tmp_path / 'feat'
    feat_dir.mkdir()
    layouts = (['_Find', '_Describe'], ['_Find', '_Find', '_And', '_Describe'], ['_Find', '_Transform', '_Describe'])
    imdb = []
    for i in range(VQA_TRAIN_N):
        fp = str(feat_dir / ('%03d.npy' % i))
        np.save(fp, vqa_feature_of(i))
        L = int(rng.integers(2, 27))
        imdb.append(dict(image_path='COCO_syn_%06d.jpg' % i, feature_path=fp, question_id=2000 + i,
                         question_str='synthetic question %d' % i,
                         question_tokens=[words[int(rng.integers(0, len(words)))] for _ in range(L)],
                         # ONE valid answer per question: the reader's np.random.choice among them is then deterministic
                         valid_answers=[answers[1 + int(rng.integers(0, len(answers) - 1))]],
                         gt_layout_tokens=list(layouts[i % 3])))
    np.save(data / imdb_dir / 'imdb_trainval2014.npy', np.array(imdb, dtype=object), allow_pickle=True)
    w = vqa_train_weights()
    if snapshot:
        from n2nmn_amd import tf_checkpoint
        snap = tmp_path / 'exp_vqa' / 'tfmodel' / snapshot
        snap.parent.mkdir(parents=True)
        tf_checkpoint.write_checkpoint(str(snap), w)
    return data, words, answers, w, glove


def vqa_short_reader(batches_seen):
    from n2nmn_amd import models_vqa

    class ShortReader(models_vqa.DataReader):
        def __init__(self, imdb_file, **kw):
            kw['batch_size'] = VQA_TRAIN_BATCH
            kw['shuffle'] = False
            super().__init__(imdb_file, **kw)

        def batches(self):
            for i, b in enumerate(super().batches()):
                if i >= VQA_TRAIN_ITERS:
                    return
                batches_seen.append({k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v)
                                     for k, v in b.items()})
                yield b
    return ShortReader


def run_vqa_train_script(script, tmp_path, monkeypatch, engine_cls, trainer_cls, recorder=None, snapshot=None,
                         seed_weights=True):
    """exp_vqa/<script>, every line of it, in a scratch tree; engine_cls / trainer_cls replace VQAEngine / VQATrainer
    behind the face (None: the HIP ones)"""
    import runpy
    from n2nmn_amd import models_vqa, runtime, runtime_train, vqa
    sys.dont_write_bytecode = True
    data, words, answers, w, glove = build_vqa_train_scratch(
        tmp_path, 'imdb_vqa_v2' if 'vqa2' in script else 'imdb', snapshot)
    if engine_cls is not None:
        monkeypatch.setattr(models_vqa, 'VQAEngine', engine_cls)
    if trainer_cls is not None:
        monkeypatch.setattr(vqa, 'VQATrainer', trainer_cls)
    monkeypatch.setattr(runtime, '_MODELS', [])
    monkeypatch.setattr(runtime_train, '_GLOBALS', [])
    if seed_weights:
        monkeypatch.setattr(runtime_train, 'initial_weights', lambda shapes, seed=0: {k: w[k] for k in shapes})
    if recorder is not None:
        recorder.install(monkeypatch)
    seen = []
    mods = vqa_import_map()
    mods['util.vqa_train.data_reader'] = EC.module('util.vqa_train.data_reader', DataReader=vqa_short_reader(seen))
    for name, mod in mods.items():
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.setattr(sys, 'argv', [script])
    monkeypatch.chdir(tmp_path)
    g = runpy.run_path(os.path.join(REF, 'exp_vqa', script), run_name='__main__')
    return g, seen, w, glove, answers


# ---- SHAPES ---------------------------------------------------------------------------------------------
def shapes_weights():
    """Seeded weights whose GREEDY layouts are valid.  models_shapes' decoder has no validity automaton, so
    random weights decode garbage and the script would only ever walk its INVALID_EXPR branch.  The token
    classifier (token_prediction/{weights,biases}: one linear layer on [h, context]) is therefore fitted by
    least squares to the data set's own ground-truth layouts on the teacher-forced decoder states of the 64
    `train.tiny` questions: the greedy decoder then reproduces most of them (where it does not, the layout
    is simply invalid -- both branches of the script's loop run)."""
    from n2nmn_amd import synth
    from oracle import n2nmn_oracle as O
    from oracle import n2nmn_oracle_shapes as S
    w = synth.make_weights_from_shapes(S.variable_shapes(14, 5), seed=0)
    d = S.load_split(REF, 'train.tiny')
    enc = O.encoder_forward(w, d['text_seq'], d['seq_length'], np.float64)
    eos = list(S.SHAPES_MODULE_NAMES).index('<eos>')
    dec = S.decoder_forward(w, enc, S.DIMS['T_decoder'], eos, np.float64, True, d['gt_layout'])
    X = dec['token_features'].reshape(-1, dec['token_features'].shape[-1])
    X1 = np.concatenate([X, np.ones((X.shape[0], 1))], axis=1)
    Y = 8.0 * np.eye(5)[d['gt_layout'].reshape(-1)]
    sol = np.linalg.lstsq(X1, Y, rcond=1e-6)[0]
    w[O._DEC + 'token_prediction/weights'] = sol[:-1].astype(np.float32)
    w[O._DEC + 'token_prediction/biases'] = sol[-1].astype(np.float32)
    return w


def build_shapes_scratch(tmp_path):
    ds, data = tmp_path / 'exp_shapes' / 'shapes_dataset', tmp_path / 'exp_shapes' / 'data'
    ds.mkdir(parents=True)
    data.mkdir(parents=True)
    for f in ('train.tiny.query_str.txt', 'train.tiny.input.npy', 'train.tiny.output'):
        shutil.copy(os.path.join(REF, 'exp_shapes', 'shapes_dataset', f), ds / f)        # data files, scratch only
    for f in ('vocabulary_shape.txt', 'vocabulary_layout.txt', 'train.tiny.query_layout_symbols.json',
              'image_mean.npy'):
        shutil.copy(os.path.join(REF, 'exp_shapes', 'data', f), data / f)
    w = shapes_weights()
    snap = tmp_path / 'exp_shapes' / 'tfmodel' / 'exp0'
    snap.mkdir(parents=True)
    np.savez(snap / '00040000.npz', **w)
    return w


def shapes_import_map():
    from n2nmn_amd import models_shapes, runtime
    return {
        'tensorflow': EC.module('tensorflow', **runtime.tf.__dict__),
        'models_shapes': EC.module('models_shapes'),
        'models_shapes.nmn3_assembler': EC.module('models_shapes.nmn3_assembler', Assembler=models_shapes.Assembler),
        'models_shapes.nmn3_model': EC.module('models_shapes.nmn3_model', NMN3ModelAtt=models_shapes.NMN3ModelAtt),
    }


def run_shapes_script(tmp_path, monkeypatch, engine_cls, recorder=None):
    """exp_shapes/eval_shapes.py, every line of it; `engine_cls` replaces n2nmn_amd.engine.Engine behind the face"""
    import runpy
    from n2nmn_amd import models_shapes, runtime
    sys.dont_write_bytecode = True
    w = build_shapes_scratch(tmp_path)
    monkeypatch.setattr(models_shapes, 'Engine', engine_cls)
    monkeypatch.setattr(runtime, '_MODELS', [])
    if recorder is not None:
        recorder.install(monkeypatch)
    for name, mod in shapes_import_map().items():
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.setattr(sys, 'argv', list(SHAPES_ARGV))
    monkeypatch.chdir(tmp_path)
    g = runpy.run_path(SHAPES_SCRIPT, run_name='__main__')
    return g, w
