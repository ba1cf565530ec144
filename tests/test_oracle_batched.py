"""The batched torch-CPU port used as bench.py's cpu_baseline (oracle/n2nmn_oracle_batched.py,
Fold-style per-(module, depth) batching) against the numpy oracle, which is pinned to the
reference's own code."""
import numpy as np
import torch

from oracle import n2nmn_oracle as O
from oracle import n2nmn_oracle_batched as OB
from n2nmn_amd import synth
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES

NAMES = list(CLEVR_MODULE_NAMES)


def _check(d, w, batch, **kw):
    ref = O.forward(w, NAMES, batch, d.T_decoder, d.num_choices, np.float64, **kw)
    got = OB.forward(OB.to_torch(w, torch.float64), NAMES, batch, d.T_decoder, d.num_choices,
                     kw.get('use_gt_layout', False), kw.get('gt_layout'))
    assert np.array_equal(got['predicted_tokens'], ref['dec']['predicted_tokens'])
    assert np.array_equal(got['validity'], ref['validity'])
    assert np.abs(got['scores'] - ref['scores']).max() < 1e-10
    assert np.abs(got['s2s']['word_vecs'].numpy() - ref['dec']['word_vecs']).max() < 1e-12


def test_batched_port_matches_oracle_on_templates_and_random_trees():
    d = Dims(N=20, T_encoder=12)
    w = synth.make_weights(d, seed=0, dtype=np.float64)
    batch = synth.make_inputs(d, seed=2, min_len=1)
    _check(d, w, batch, use_gt_layout=True, gt_layout=synth.template_layout_batch(d))
    P, Wv, bv = O.build_validity_mats(NAMES)
    toks = synth.random_valid_layouts(d, P, Wv, bv, seed=4, max_len=7)
    _check(d, w, batch, use_gt_layout=True, gt_layout=toks)


def test_batched_port_greedy_decoder():
    d = Dims(N=6, T_encoder=9, T_decoder=10)
    w = synth.make_weights(d, seed=0, dtype=np.float64)
    _check(d, w, synth.make_inputs(d, seed=3, min_len=1))
