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
replaces n2nmn_amd.engine.Engine for this test only).  What the script ASKED of the drop-in -- constructor
keywords, every partial_run with its feeds -- is recorded (tests/golden/eval_driver_trace.npz, checked here
against a fresh recording) and replayed on the same drop-in objects over the HIP engine on the GPU box:
tests/test_gpu_eval_driver_trace.py."""
import os
import sys

import numpy as np
import pytest

from oracle import n2nmn_oracle as O

import eval_driver_common as EC

pytestmark = pytest.mark.skipif(not os.path.exists(EC.SCRIPT), reason='reference checkout not present')


def test_eval_clevr_script_runs_unmodified_against_the_drop_in(tmp_path, monkeypatch):
    from n2nmn_amd import data_reader
    from oracle_engine import OracleEngine
    rec = EC.SessionRecorder(EC.Dims())
    # the reference's file, every line of it (scratch tree + import map: tests/eval_driver_common.py)
    g, d, data, words, answers, w = EC.run_reference_script(tmp_path, monkeypatch, OracleEngine, rec)
    n_q = EC.N_QUESTIONS

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

    # ---- the committed recording of this run (replayed over the HIP engine on the GPU box:
    # tests/test_gpu_eval_driver_trace.py) is still what the reference's script does -------------------
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import make_eval_driver_trace as MT
    z = np.load(MT.OUT)
    assert MT.same(MT.pack_trace(rec, written, answers), {k: z[k] for k in z.files}) is None
