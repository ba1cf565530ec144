"""Super-bucketing: K in-flight batches of `Nb` questions executed as ONE pass of the hot path
(SURVEY.md section 8(f) rank 2 -- "on-device assemble/pack + multi-batch super-bucketing").

The reference runs one `sess.partial_run` pair per batch of 64 (exp_clevr/eval_clevr.py:103-135).  On
MI355X a batch of 64 cannot fill the chip: the recurrent step kernels re-stream 12.6 MB of weights
for 64 rows, the layout walker occupies 64 of 256 CUs, a pooling launch carries ~15 jobs.  Questions
are independent, so the runtime batches dynamically instead: the engine owns device
slabs with K slots, a client writes its batch (question tokens, lengths, features, optionally the
ground-truth layout) into slot k, and one pass serves all K batches -- every launch of both phases
carries K * Nb rows, and each client reads its own rows of the answer logits.  Nothing in the model
couples questions (the decoder's global `min(token_scores)` only has to be below every valid score,
models_clevr/nmn3_netgen_att.py:228,234), so a question's logits do not depend on its slot.

Slabs (device, allocated once):
    input_seq  [T_enc, K*Nb] int32   slot k = columns k*Nb .. (k+1)*Nb   (time-major like the reference)
    seq_length [K*Nb]        int32
    image_feat [K*Nb, H, W, D] f32   slot k = one contiguous block
    gt_layout  [T_dec, K*Nb] int32   (teacher forcing only)
    scores     [K*Nb, C], validity [K*Nb], predicted_tokens [T_dec, K*Nb]   results, same slotting
"""
from __future__ import annotations

import dataclasses
from typing import Optional

import numpy as np

from .engine import Engine, _torch
from .nmn3_assembler import Assembler
from .spec import Dims


class SuperBucket:
    def __init__(self, dims: Dims, assembler: Assembler, K: int, device: int = 0,
                 engine: Optional[Engine] = None):
        """dims.N = questions per client batch (64); the engine is created with capacity K * dims.N
        (or pass one that already has it)."""
        torch = _torch()
        if K < 1:
            raise ValueError('K >= 1')
        self.K, self.Nb = int(K), int(dims.N)
        self.dims = dims
        self.big = dataclasses.replace(dims, N=self.K * self.Nb)
        self.engine = engine if engine is not None else Engine(self.big, assembler, device=device)
        if self.engine.dims.N < self.big.N:
            raise ValueError('engine capacity %d < K * N = %d' % (self.engine.dims.N, self.big.N))
        dev = self.engine.device
        N, d = self.big.N, dims
        self.input_seq = torch.zeros((d.T_encoder, N), dtype=torch.int32, device=dev)
        self.seq_length = torch.ones((N,), dtype=torch.int32, device=dev)
        self.image_feat = torch.zeros((N, d.H, d.W, d.D), dtype=torch.float32, device=dev)
        self.gt_layout = torch.zeros((d.T_decoder, N), dtype=torch.int32, device=dev)
        # host copy of the lengths of the slots that were filled from host arrays (clients hand over
        # host batches, util/clevr_train/data_reader.py:74-82): n2nmn_seq2seq_io.seq_length_host
        self.seq_length_host = np.ones((N,), np.int32)
        self._len_known = [False] * self.K
        # host copy of the layout lengths of the slots whose ground-truth layout came as a host array
        # (n2nmn_seq2seq_io.gt_length_host, eos_retire)
        self.gt_length_host = np.full((N,), d.T_decoder, np.int32)
        self._glen_known = [False] * self.K
        # ... and how deep each such slot's layouts nest Transform / FindSameProperty (Engine.layout_nesting):
        # the walker then launches exactly that many levels and no fall-back walker
        self.gt_nesting_host = [0] * self.K
        # results live in tensors of the bucket's own: two buckets may share one engine (a worker
        # alternates between them), and the engine's reuse buffers belong to whoever ran last
        self._res = {}
        self.scores = None
        self.validity = None
        self.tokens = None
        self.n_run = 0

    def load_weights(self, weights):
        self.engine.load_weights(weights)

    def _cols(self, k: int):
        if not 0 <= k < self.K:
            raise ValueError('slot %d out of range [0, %d)' % (k, self.K))
        return slice(k * self.Nb, (k + 1) * self.Nb)

    def slot(self, k: int):
        """Views a client writes its batch into (keys of util/clevr_train/data_reader.py:74-82)."""
        c = self._cols(k)
        self._len_known[k] = False        # written behind our back: no host copy of the lengths
        self._glen_known[k] = False
        return dict(input_seq_batch=self.input_seq[:, c], seq_length_batch=self.seq_length[c],
                    image_feat_batch=self.image_feat[c], gt_layout_batch=self.gt_layout[:, c])

    def fill(self, k: int, batch, gt_layout=None):
        """Copy a client batch (numpy or tensors, any device) into slot k; asynchronous on the
        current stream for device tensors."""
        torch = _torch()
        v = self.slot(k)
        for key in ('input_seq_batch', 'seq_length_batch', 'image_feat_batch'):
            v[key].copy_(torch.as_tensor(batch[key]), non_blocking=True)
        lens = batch['seq_length_batch']
        self._len_known[k] = isinstance(lens, np.ndarray)
        if self._len_known[k]:
            self.seq_length_host[self._cols(k)] = lens
        if gt_layout is not None:
            v['gt_layout_batch'].copy_(torch.as_tensor(gt_layout), non_blocking=True)
            self._glen_known[k] = isinstance(gt_layout, np.ndarray)
            if self._glen_known[k]:
                self.gt_length_host[self._cols(k)] = self.engine.layout_lengths(gt_layout)
                self.gt_nesting_host[k] = self.engine.layout_nesting(gt_layout)

    def set_layout(self, k: int, gt_layout):
        """replace only the ground-truth layout of slot k (host array or tensor)"""
        torch = _torch()
        c = self._cols(k)
        self.gt_layout[:, c].copy_(torch.as_tensor(gt_layout), non_blocking=True)
        self._glen_known[k] = isinstance(gt_layout, np.ndarray)
        if self._glen_known[k]:
            self.gt_length_host[c] = self.engine.layout_lengths(gt_layout)
            self.gt_nesting_host[k] = self.engine.layout_nesting(gt_layout)

    def _results(self, n: int, Td: int):
        torch = _torch()
        key = (n, Td)
        r = self._res.get(key)
        if r is None:
            dev, rows = self.engine.device, n * self.Nb
            r = (torch.empty((rows, self.dims.num_choices), dtype=torch.float32, device=dev),
                 torch.empty((Td, rows), dtype=torch.int32, device=dev),
                 torch.empty((rows,), dtype=torch.int32, device=dev))
            self._res[key] = r
        return r

    def run(self, use_gt_layout: bool = False, sample_uniforms=None, T_dec: Optional[int] = None,
            n_slots: Optional[int] = None, host_assemble: bool = False, eos_retire: bool = False):
        """One pass over the first n_slots slots (default: all K).  Returns (scores [n*Nb, C], tokens
        [T_dec, n*Nb], validity [n*Nb]) device tensors owned by this bucket (one set per pass width,
        overwritten by the next pass of that width); nothing synchronises.  eos_retire: Engine.forward's
        inference option (rows leave the decoder at their layout's first <eos>)."""
        n = self.K if n_slots is None else int(n_slots)
        if not 1 <= n <= self.K:
            raise ValueError('n_slots %d out of range [1, %d]' % (n, self.K))
        rows = n * self.Nb
        Td = self.dims.T_decoder if T_dec is None else int(T_dec)
        full = n == self.K
        batch = dict(input_seq_batch=self.input_seq if full else self.input_seq[:, :rows].contiguous(),
                     seq_length_batch=self.seq_length[:rows], image_feat_batch=self.image_feat[:rows])
        if all(self._len_known[:n]):
            batch['seq_length_host'] = self.seq_length_host[:rows]
        gt = None
        if use_gt_layout:
            gt = self.gt_layout if full else self.gt_layout[:, :rows].contiguous()
            if eos_retire and all(self._glen_known[:n]) and (T_dec is None or Td == self.dims.T_decoder):
                batch['gt_length_host'] = self.gt_length_host[:rows]
            if all(self._glen_known[:n]) and (T_dec is None or Td == self.dims.T_decoder):
                batch['gt_nesting_host'] = max(self.gt_nesting_host[:n])
        if host_assemble or not self.engine.walk_supported():
            self.scores, self.tokens, self.validity = self.engine.forward(
                batch, T_dec=T_dec, use_gt_layout=use_gt_layout, gt_layout=gt,
                sample_uniforms=sample_uniforms, fetch=False, host_assemble=True)
        else:
            self.scores, self.tokens, self.validity = self.engine.forward(
                batch, T_dec=T_dec, use_gt_layout=use_gt_layout, gt_layout=gt,
                sample_uniforms=sample_uniforms, fetch=False, out=self._results(n, Td),
                eos_retire=eos_retire)
        self.n_run = n
        return self.scores, self.tokens, self.validity

    def result(self, k: int):
        """(scores [Nb, C], tokens [T_dec, Nb], validity [Nb]) views of slot k after run()."""
        if not 0 <= k < self.n_run:
            raise ValueError('slot %d was not part of the last pass (%d slots)' % (k, self.n_run))
        c = self._cols(k)
        return self.scores[c], self.tokens[:, c], self.validity[c]


def host_batches_to_bucket(bucket: SuperBucket, batches, gts=None):
    """Convenience for tests / benchmarks: fill slot k with batches[k] (and gts[k])."""
    for k, b in enumerate(batches):
        bucket.fill(k, b, None if gts is None else gts[k])
    return bucket
