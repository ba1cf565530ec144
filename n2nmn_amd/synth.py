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
