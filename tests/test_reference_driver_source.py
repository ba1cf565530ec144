"""The reference's own driver, UNMODIFIED, against the drop-in (north star: "drops
in under exp_clevr/eval_clevr.py").

`/root/reference/exp_clevr/eval_clevr.py` is executed as a script (runpy, `__main__`): its imports are
answered by the drop-in --

    import tensorflow as tf                          -> n2nmn_amd.runtime.tf  (Session, placeholder, train.Saver)
    from models_clevr.nmn3_assembler import Assembler -> n2nmn_amd.nmn3_assembler.Assembler
    from models_clevr.nmn3_model import NMN3Model     -> n2nmn_amd.nmn3_model.NMN3Model
    from util.clevr_train.data_reader import DataReader -> n2nmn_amd.data_reader.DataReader

-- on a synthetic imdb in a scratch directory (the reference's own vocabulary files, random pool5
features, seeded weights in an .npz "snapshot").  Every line of the script runs: argument parsing, model
construction with the reference's keyword arguments, `snapshot_saver.restore`, the partial_run loop
(phase 1 -> `assembler.assemble` -> `compiler.build_feed_dict` -> phase 2), accuracy bookkeeping and the
result files.  The answers it writes and the `scores_val` of its last batch are compared with the
oracle's.

The checkout exists only in the build container and the GPU only on the gpurun box, so the compute
behind the drop-in's Python face is the CPU oracle here (tests/oracle_engine.py, a test double that
replaces n2nmn_amd.engine.Engine for this test only); the same loop over the HIP engine runs on the
GPU in tests/test_gpu_end2end.py::test_reference_shaped_session_loop."""
import os
import runpy
import shutil
import sys
import types

import numpy as np
import pytest

from oracle import n2nmn_oracle as O
from n2nmn_amd import synth
from n2nmn_amd.spec import Dims

REF = '/root/reference'
SCRIPT = os.path.join(REF, 'exp_clevr', 'eval_clevr.py')
pytestmark = pytest.mark.skipif(not os.path.exists(SCRIPT), reason='reference checkout not present')


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


def test_eval_clevr_script_runs_unmodified_against_the_drop_in(tmp_path, monkeypatch):
    sys.dont_write_bytecode = True
    from n2nmn_amd import data_reader, nmn3_assembler, nmn3_model, runtime
    from oracle_engine import OracleEngine
    d = Dims()
    # ---- scratch tree shaped like the reference's working directory ------------------------------
    data = tmp_path / 'exp_clevr' / 'data'
    (data / 'imdb').mkdir(parents=True)
    for f in ('vocabulary_clevr.txt', 'vocabulary_layout.txt', 'answers_clevr.txt'):
        shutil.copy(os.path.join(REF, 'exp_clevr', 'data', f), data / f)      # data files, scratch only
    words = [l.strip() for l in open(data / 'vocabulary_clevr.txt')]
    answers = [l.strip() for l in open(data / 'answers_clevr.txt')]
    assert (len(words), len(answers)) == (d.num_vocab_txt, d.num_choices)
    rng = np.random.default_rng(5)
    n_q = 70                                            # one full batch of 64 and a short one of 6
    feat_dir = tmp_path / 'feat'
    feat_dir.mkdir()
    imdb = []
    for i in range(n_q):
        fp = str(feat_dir / ('%03d.npy' % i))
        np.save(fp, np.maximum(rng.standard_normal((1, d.H, d.W, d.D)), 0).astype(np.float32))
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

    # ---- the drop-in answers the driver's imports ---------------------------------------------------
    monkeypatch.setattr(nmn3_model, 'Engine', OracleEngine)
    monkeypatch.setattr(runtime, '_MODELS', [])
    for name, mod in {
        'tensorflow': _module('tensorflow', **runtime.tf.__dict__),
        'models_clevr': _module('models_clevr'),
        'models_clevr.nmn3_assembler': _module('models_clevr.nmn3_assembler', Assembler=nmn3_assembler.Assembler),
        'models_clevr.nmn3_model': _module('models_clevr.nmn3_model', NMN3Model=nmn3_model.NMN3Model),
        'util': _module('util'), 'util.clevr_train': _module('util.clevr_train'),
        'util.clevr_train.data_reader': _module('util.clevr_train.data_reader',
                                                DataReader=data_reader.DataReader),
    }.items():
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.setattr(sys, 'argv', ['eval_clevr.py', '--exp_name', 'exp0', '--snapshot_name', '00050000',
                                      '--test_split', 'syn'])
    monkeypatch.chdir(tmp_path)
    g = runpy.run_path(SCRIPT, run_name='__main__')         # the reference's file, every line of it

    # ---- what the script computed ---------------------------------------------------------------------
    model, sess, asm = g['nmn3_model_tst'], g['sess'], g['assembler']
    assert model.engine.calls == dict(seq2seq=2, execute=2)          # two batches, two phases each
    pred_file = tmp_path / 'exp_clevr' / 'eval_outputs' / 'exp0' / '00050000.syn.txt'
    written = [l.strip() for l in open(pred_file)]
    assert len(written) == n_q
    # oracle on the same questions, batch by batch like the reader delivers them
    reader = data_reader.DataReader(str(data / 'imdb' / 'imdb_syn.npy'), shuffle=False, one_pass=True,
                                    batch_size=64, T_encoder=d.T_encoder, T_decoder=d.T_decoder,
                                    assembler=asm, vocab_question_file=str(data / 'vocabulary_clevr.txt'),
                                    vocab_answer_file=str(data / 'answers_clevr.txt'),
                                    prune_filter_module=True)
    w64 = {k: v.astype(np.float64) for k, v in w.items()}
    want, last = [], None
    for batch in reader.batches():
        last = O.forward(w64, list(asm.module_names), batch, d.T_decoder, d.num_choices, np.float64)
        want += [answers[p] for p in np.argmax(last['scores'], axis=1)]
    assert written == want
    assert np.abs(sess.last['scores'] - last['scores']).max() < 1e-10
    assert np.array_equal(sess.last['predicted_tokens'], last['dec']['predicted_tokens'])
    result = open(tmp_path / 'exp_clevr' / 'results' / 'exp0' / '00050000.syn.txt').read()
    assert 'answer accuracy' in result and 'layout validity = 1.0' in result
