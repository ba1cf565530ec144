"""Seeded synthetic weights / inputs for the CLEVR hot path.

There is no network for CLEVR data or the published snapshots, so every parity test and the
benchmark run on these tensors.  Distributions follow the reference's initialisers:

* fc / conv / 1x1 weights: Xavier-uniform U(+-sqrt(6/(fan_in+fan_out)))
  (util/cnn.py:14,101; util/empty_safe_conv.py:22)
* LSTM and embedding matrices: Glorot-uniform (TF 1.0.0 scope default is unpinned, Appendix A.7)
* biases: U(-0.1, 0.1) instead of the reference's zeros so the bias paths are exercised
* v ~ U(+-sqrt(3/L))
* features = max(0, N(0,1)) (pool5 is post-ReLU), tokens uniform in [0, V), lengths uniform in
  [5, T_enc], zero padded like util/clevr_train/data_reader.py:43,56.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np

from .spec import (Dims, variable_shapes, CLEVR_MODULE_NAMES, CLEVR_LAYOUT_TEMPLATES)


def make_weights(d: Dims, seed: int = 0, dtype=np.float32) -> Dict[str, np.ndarray]:
    return make_weights_from_shapes(variable_shapes(d), seed, dtype)


def make_weights_from_shapes(shapes: Dict[str, tuple], seed: int = 0,
                             dtype=np.float32) -> Dict[str, np.ndarray]:
    """Same distributions for any name -> shape table (e.g. n2nmn_amd.vqa.vqa_variable_shapes)."""
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape in shapes.items():
        if name.endswith('/biases'):
            w = rng.uniform(-0.1, 0.1, size=shape)
        elif name.endswith('/v'):
            lim = np.sqrt(3.0 / shape[0])
            w = rng.uniform(-lim, lim, size=shape)
        else:
            if len(shape) == 4:          # conv [kh, kw, in, out]
                rf = shape[0] * shape[1]
                fan_in, fan_out = rf * shape[2], rf * shape[3]
            elif len(shape) == 2:
                fan_in, fan_out = shape
            else:
                fan_in = fan_out = shape[0]
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            w = rng.uniform(-lim, lim, size=shape)
        out[name] = np.ascontiguousarray(w.astype(dtype))
    return out


def make_inputs(d: Dims, seed: int = 0, n: int | None = None, min_len: int = 5):
    """Batch dict with the keys of util/clevr_train/data_reader.py:74-82."""
    n = d.N if n is None else n
    rng = np.random.default_rng(seed + 1000003)
    feats = np.maximum(rng.standard_normal((n, d.H, d.W, d.D)), 0.0).astype(np.float32)
    lens = rng.integers(min_len, d.T_encoder + 1, size=n).astype(np.int32)
    seq = rng.integers(0, d.num_vocab_txt, size=(d.T_encoder, n)).astype(np.int32)
    seq[np.arange(d.T_encoder)[:, None] >= lens[None, :]] = 0
    labels = rng.integers(0, d.num_choices, size=n).astype(np.int32)
    return dict(input_seq_batch=seq, seq_length_batch=lens, image_feat_batch=feats,
                answer_label_batch=labels)


def module_list2tokens(module_list: Sequence[str], T: int,
                       names: Sequence[str] = CLEVR_MODULE_NAMES) -> List[int]:
    """Same contract as Assembler.module_list2tokens (models_clevr/nmn3_assembler.py:137-143)."""
    idx = {s: i for i, s in enumerate(names)}
    if len(module_list) >= T:
        raise ValueError('Not enough time steps to add <eos>')
    eos = idx['<eos>']
    return [idx[m] for m in module_list] + [eos] * (T - len(module_list))


def template_layout_batch(d: Dims, n: int | None = None, offset: int = 0) -> np.ndarray:
    """gt_layout_batch [T_decoder, N] int32: template (i + offset) mod 10 for question i."""
    n = d.N if n is None else n
    out = np.zeros((d.T_decoder, n), np.int32)
    for i in range(n):
        tpl = CLEVR_LAYOUT_TEMPLATES[(i + offset) % len(CLEVR_LAYOUT_TEMPLATES)]
        out[:, i] = module_list2tokens(tpl, d.T_decoder)
    return out


def random_valid_layouts(d: Dims, P: np.ndarray, W: np.ndarray, b: np.ndarray,
                         seed: int = 0, n: int | None = None, T: int | None = None,
                         max_len: int | None = None) -> np.ndarray:
    """Random walks under the reference's validity automaton (X.W - b >= 0, X += P[token]):
    every column is a layout the greedy decoder could emit.  Returns [T, N] int32.
    `max_len` biases the walk towards ending early (prefers answer modules once reached)."""
    n = d.N if n is None else n
    T = d.T_decoder if T is None else T
    rng = np.random.default_rng(seed + 7919)
    toks = np.zeros((T, n), np.int32)
    for i in range(n):
        X = np.array([0, 0, T], np.int64)
        for t in range(T):
            cons = np.tensordot(X, W, axes=1) - b          # [V, 4]
            valid = np.nonzero(np.all(cons >= 0, axis=1))[0]
            assert valid.size > 0
            if max_len is not None and t >= max_len:
                ans = [v for v in valid if P[v, 1] == 1]
                if ans:
                    valid = np.array(ans)
            tok = int(rng.choice(valid))
            toks[t, i] = tok
            X = X + P[tok]
    return toks


# ---- CLEVR-like layouts of realistic length -----------------------------------------------------------
# The ten templates above average 3.2 tokens; real CLEVR programs are longer (up to 19 layout tokens at
# T_dec = 20).  The dataset is not here, so the second layout mix of the eos_retire measurement is BUILT
# from the reference's own linearisation rules (exp_clevr/data/get_ground_truth_layout.py:4-37 maps
# functions to modules, :49-65 prunes count / query_* under comparisons, :90-96 turns scene + filter into
# _Find and every further filter_* into _Filter, `unique` vanishes) applied to the CLEVR question families
# (zero- to three-hop chains, single-and / single-or, integer and attribute comparisons, same-relate), an
# object description being 1 to 4 filters as in the CLEVR generator.  Post-order (Reverse Polish) token
# order, every layout checked against the validity automaton by the tests.
def _obj(rng, lo=1, hi=4):
    """one object description: scene + k filters -> _Find, _Filter x (k - 1)"""
    return ['_Find'] + ['_Filter'] * (int(rng.integers(lo, hi + 1)) - 1)


def _refine(rng, lo=1, hi=3):
    """filters applied to a relate / same_* / intersect result"""
    return ['_Filter'] * int(rng.integers(lo, hi + 1))


def _chain(rng, hops):
    out = _obj(rng)
    for _ in range(hops):
        out += ['_Transform'] + _refine(rng)
    return out


def clevr_like_layout(rng, T: int) -> List[str]:
    """one layout (module names, no <eos>) of at most T - 1 tokens"""
    while True:
        fam = int(rng.integers(0, 9))
        ans1 = ['_Count', '_Exist', '_Describe'][int(rng.integers(0, 3))]
        if fam <= 2:                         # zero- / one- / two- / three-hop
            lay = _chain(rng, int(rng.integers(0, 4))) + [ans1]
        elif fam == 3:                       # single-and: two one-hop chains intersected
            lay = _chain(rng, 1) + _chain(rng, 1) + ['_And'] + _refine(rng, 0, 2) + [ans1]
        elif fam == 4:                       # single-or
            lay = _obj(rng) + _obj(rng) + ['_Or'] + [['_Count', '_Exist'][int(rng.integers(0, 2))]]
        elif fam == 5:                       # compare integer (count pruned under the comparison)
            lay = _chain(rng, int(rng.integers(0, 2))) + _chain(rng, int(rng.integers(0, 2))) + \
                [['_EqualNum', '_MoreNum', '_LessNum'][int(rng.integers(0, 3))]]
        elif fam == 6:                       # compare attribute (query_* pruned)
            lay = _chain(rng, int(rng.integers(0, 2))) + _chain(rng, int(rng.integers(0, 2))) + ['_SameProperty']
        elif fam == 7:                       # same-relate
            lay = _chain(rng, int(rng.integers(0, 2))) + ['_FindSameProperty'] + _refine(rng, 0, 3) + [ans1]
        else:                                # whole-scene questions
            lay = ['_Scene', ['_Count', '_Exist'][int(rng.integers(0, 2))]]
        if len(lay) <= T - 1:
            return lay


def clevr_like_layout_batch(d: Dims, n: int | None = None, seed: int = 0) -> np.ndarray:
    """gt_layout_batch [T_decoder, N] int32 of CLEVR-like layouts (see above); mean length ~8.5"""
    n = d.N if n is None else n
    rng = np.random.default_rng(seed + 104729)
    out = np.zeros((d.T_decoder, n), np.int32)
    for i in range(n):
        out[:, i] = module_list2tokens(clevr_like_layout(rng, d.T_decoder), d.T_decoder)
    return out
