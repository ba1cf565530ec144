"""Host side of the training step (BASELINE.json configs[3]): the loop body of
exp_clevr/train_clevr_gt_layout.py:150-200 -- forward of the behavioural-cloning objective with
ground-truth layouts, backward, per-tensor clip-by-norm, Adam -- on top of the C-ABI
(include/n2nmn.h section 6), plus the data-parallel gradient exchange the reference does not have
(SURVEY.md 8e): one flat fp32 gradient buffer, all-reduced in two buckets so that the decoder +
module bucket travels over RCCL while the encoder's backward pass is still running.

PyTorch is plumbing only (device buffers, streams, torch.distributed); all arithmetic is in
libn2nmn_hip.so and there is no fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np

from . import _lib
from .engine import Engine


def _torch():
    import torch
    return torch


BWD_DEFER_JOIN = 0x10          # include/n2nmn.h


class GradBuckets:
    """Two-bucket all-reduce of a flat gradient vector (sum over ranks; the optimiser applies the
    1/world scale).  `late` = [split, numel) is ready first (decoder + modules), `early` =
    [0, split) (encoder) last.  Works on any torch.distributed backend (RCCL on GPUs, gloo in
    the CPU tests); without a process group it is a no-op."""

    def __init__(self, flat, split: int, dist=None):
        self.flat = flat
        self.split = int(split)
        self.dist = dist
        self._pending = []

    @property
    def world(self) -> int:
        return self.dist.get_world_size() if self.dist is not None else 1

    def reduce_late(self):
        if self.dist is not None:          # (a 1-rank group still goes through the collective)
            self._pending.append(self.dist.all_reduce(self.flat[self.split:], async_op=True))

    def reduce_early(self):
        if self.dist is not None:
            self._pending.append(self.dist.all_reduce(self.flat[:self.split], async_op=True))

    def wait(self) -> float:
        """Blocks the current stream on both collectives; returns the scale that turns the sum into
        the mean over ranks."""
        for w in self._pending:
            w.wait()
        self._pending = []
        return 1.0 / self.world


class RcclBuckets:
    """The same two buckets through the C-ABI's own RCCL communicator (n2nmn_comm_create /
    n2nmn_allreduce_grads / n2nmn_allreduce_wait, include/n2nmn.h section 6b): the collectives run
    on a library-owned side stream forked from and joined to the caller's stream with events.
    torch.distributed is used for ONE thing only: handing rank 0's 128-byte ncclUniqueId to the other
    ranks.  dist=None: a 1-rank communicator (the code path of the 8-GPU run on one GPU)."""

    def __init__(self, engine: Engine, flat, dist=None):
        torch = _torch()
        self.engine, self.flat, self.dist = engine, flat, dist
        self._lib = engine._lib
        self.rank = dist.get_rank() if dist is not None else 0
        self._world = dist.get_world_size() if dist is not None else 1
        uid = (C.c_char * 128)()
        if self.rank == 0:
            _lib.check(self._lib.n2nmn_comm_unique_id(uid))
        if dist is not None and self._world > 1:
            box = [bytes(uid.raw) if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0, device=engine.device)
            C.memmove(uid, box[0], 128)
        self._comm = C.c_void_p()
        _lib.check(self._lib.n2nmn_comm_create(uid, self.rank, self._world, engine.device.index,
                                               C.byref(self._comm)))
        torch.cuda.synchronize(engine.device)
        self.verified_world = self._self_test()

    def _self_test(self) -> int:
        """One all-reduce of both buckets over a buffer of ones, through the very calls a step uses:
        every element must come back as the number of ranks.  Proves at construction that the
        communicator spans `world` ranks (the figure bench.py prints as `rccl_ranks`) instead of
        assuming it from WORLD_SIZE; leaves the gradient buffer zeroed."""
        torch = _torch()
        self.flat.fill_(1.0)
        self.reduce_late()
        self.reduce_early()
        self.wait()
        torch.cuda.synchronize(self.engine.device)
        lo, hi = float(self.flat.min().item()), float(self.flat.max().item())
        self.flat.zero_()
        if lo != hi or int(lo) != self.comm_world():
            raise RuntimeError('RCCL self-test: all-reduce of ones over %d ranks returned [%g, %g]' %
                               (self._world, lo, hi))
        return int(lo)

    def comm_world(self) -> int:
        """ranks of the C-ABI communicator as the library reports them (n2nmn_comm_world)"""
        return int(_lib.check(self._lib.n2nmn_comm_world(self._comm)))

    @property
    def world(self) -> int:
        return self._world

    def reduce_late(self):
        _lib.check(self._lib.n2nmn_allreduce_grads(self.engine._ctx, self._comm, 0,
                                                   self.flat.data_ptr(), self.engine.stream()))

    def reduce_early(self):
        _lib.check(self._lib.n2nmn_allreduce_grads(self.engine._ctx, self._comm, 1,
                                                   self.flat.data_ptr(), self.engine.stream()))

    def wait(self) -> float:
        _lib.check(self._lib.n2nmn_allreduce_wait(self._comm, self.engine.stream()))
        return 1.0 / self._world

    def close(self):
        comm, self._comm = self._comm, None
        if comm:
            self._lib.n2nmn_comm_destroy(comm)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_rccl(dist) -> bool:
    """Which all-reduce a Trainer uses when the caller does not say: the C-ABI communicator
    (side-stream all-reduce under the encoder's backward pass) on every nccl (= RCCL) group, one
    rank or many -- tests/test_gpu_rccl_multi.py checks gradient equality across ranks on it wherever
    a node has >= 2 devices, and its constructor proves the rank count with an all-reduce of ones.
    N2NMN_RCCL_BUCKETS=0 selects torch.distributed's all_reduce instead; gloo groups always use it."""
    import os
    return dist is not None and dist.get_backend() == 'nccl' and \
        os.environ.get('N2NMN_RCCL_BUCKETS', '1') != '0'


class Trainer:
    """One model replica on one GPU.  `step(batch, gt_layout)` = one iteration of
    train_clevr_gt_layout.py: returns the losses of that iteration (device tensor of 4 floats:
    avg_sample_loss, seq_likelihood_loss, l2_reg, total_loss)."""

    def __init__(self, engine: Engine, weight_decay: float = 5e-6, lr: float = 1e-3,
                 beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
                 max_grad_l2_norm: float = 10.0, dist=None, rccl: Optional[bool] = None):
        """dist: an initialised torch.distributed module (or None).  rccl: all-reduce through the
        C-ABI's own RCCL communicator (n2nmn_comm_* / n2nmn_allreduce_grads); False / gloo:
        torch.distributed collectives.  Default: the C-ABI communicator on every nccl group
        (N2NMN_RCCL_BUCKETS=0 opts out); `bucket_impl` / `rccl_ranks` say what a trainer runs on."""
        torch = _torch()
        if engine._parent is not None:
            raise ValueError('train on the root engine, not on a fork')
        self.engine = engine
        self._lib = engine._lib
        self._ctx = engine._ctx
        _lib.check(self._lib.n2nmn_train_enable(self._ctx))
        self.numel = int(self._lib.n2nmn_grad_numel(self._ctx))
        self.split = int(self._lib.n2nmn_grad_split(self._ctx))
        self.weight_decay = float(weight_decay)
        self.hyper = dict(lr=float(lr), beta1=float(beta1), beta2=float(beta2), eps=float(eps),
                          max_grad_l2_norm=float(max_grad_l2_norm))
        self.grads = torch.zeros(self.numel, dtype=torch.float32, device=engine.device)
        # avg_sample_loss, seq_likelihood | policy_gradient loss, l2_reg, total_loss, entropy_reg, -
        self.losses = torch.zeros(8, dtype=torch.float32, device=engine.device)
        # policy gradient (train_clevr_rl_gt_layout.py:38-42,120): baseline starts at
        # invalid_expr_loss and is an EMA of avg_sample_loss, kept on the device
        self.rl = dict(invalid_expr_loss=0.5, lambda_entropy=0.005, baseline_decay=0.99)
        self.baseline = torch.full((1,), self.rl['invalid_expr_loss'], dtype=torch.float32,
                                   device=engine.device)
        self.scores = None
        self.last_validity = None
        if rccl is None:
            rccl = default_rccl(dist)
        self.buckets = RcclBuckets(engine, self.grads, dist) if rccl else \
            GradBuckets(self.grads, self.split, dist)
        self.bucket_impl = 'n2nmn_comm (C-ABI RCCL communicator, library side stream)' if rccl else \
            ('torch.distributed.all_reduce (%s)' % dist.get_backend() if dist is not None else 'none (one process)')
        self.rccl_ranks = self.buckets.verified_world if rccl else None
        self.iteration = 0
        _lib.check(self._lib.n2nmn_train_reset_optimizer(self._ctx, engine.stream()))
        self.layout: Dict[str, tuple] = {}
        names = list(engine.variable_names().items())
        for i, (name, shape) in enumerate(names):
            off, n = C.c_int64(), C.c_int64()
            _lib.check(self._lib.n2nmn_grad_layout(self._ctx, i, C.byref(off), C.byref(n)))
            self.layout[name] = (int(off.value), int(n.value), shape)
        self._keep = None

    # ------------------------------------------------------------------------------------
    def _io(self, batch, gt_layout, objective: int = 0):
        torch = _torch()
        e = self.engine
        d = e.dims
        seq = e._dev(batch['input_seq_batch'], torch.int32)
        lens = e._dev(batch['seq_length_batch'], torch.int32)
        feat = e._dev(batch['image_feat_batch'], torch.float32)
        labels = e._dev(batch['answer_label_batch'], torch.int32)
        gt_host = np.ascontiguousarray(np.asarray(
            gt_layout.cpu().numpy() if hasattr(gt_layout, 'cpu') else gt_layout), np.int32)
        gt = e.upload_i32(gt_host)       # pinned slot: no stream synchronisation per step
        T, N = seq.shape
        Td = gt.shape[0]
        # teacher forcing: the layout is known before the forward, so the program is assembled
        # up front and there is no host sync inside the step
        packed, validity = e.assembler.assemble_packed(gt_host)
        if self.scores is None or tuple(self.scores.shape) != (N, d.num_choices):
            self.scores = torch.empty((N, d.num_choices), dtype=torch.float32, device=e.device)
        io = _lib.TrainIO()
        io.input_seq = seq.data_ptr(); io.seq_length = lens.data_ptr()
        io.T_enc = T; io.N = N; io.T_dec = Td
        io.gt_layout = gt.data_ptr(); io.image_feat = feat.data_ptr()
        io.answer_labels = labels.data_ptr(); io.weight_decay = self.weight_decay
        io.scores = self.scores.data_ptr(); io.losses = self.losses.data_ptr()
        io.grads = self.grads.data_ptr()
        val = None
        if objective == 1:
            val = e.upload_i32(np.asarray(validity, np.int32))
            io.objective = 1
            io.expr_validity = val.data_ptr()
            io.invalid_expr_loss = self.rl['invalid_expr_loss']
            io.lambda_entropy = self.rl['lambda_entropy']
            io.baseline_decay = self.rl['baseline_decay']
            io.baseline = self.baseline.data_ptr()
        self._keep = (seq, lens, feat, labels, gt, packed, val)
        self.last_validity = np.asarray(validity, bool)
        return io, packed, validity

    def forward_backward(self, batch, gt_layout, reduce: bool = True, objective: int = 0) -> float:
        """Forward + both backward phases (+ the bucketed all-reduce).  Afterwards self.grads is
        d total_loss / d variables summed over ranks; returns the 1/world scale.
        objective 0: behavioural cloning on ground-truth layouts; 1: policy gradient, gt_layout
        then being the layout the decoder sampled for this batch (see step_rl)."""
        io, packed, _ = self._io(batch, gt_layout, objective)
        s = self.engine.stream()
        _lib.check(self._lib.n2nmn_train_forward(self._ctx, C.byref(io), packed.handle, s))
        # the decoder's weight gradients and the finish of the late bucket stay on the library's side
        # stream under the encoder's backward pass (N2NMN_BWD_DEFER_JOIN, include/n2nmn.h).  A
        # torch.distributed all-reduce knows nothing of that stream: the caller's stream joins it first.
        torch_reduce = reduce and isinstance(self.buckets, GradBuckets) and self.buckets.dist is not None
        _lib.check(self._lib.n2nmn_train_backward(self._ctx, C.byref(io), packed.handle,
                                                  0 | BWD_DEFER_JOIN, s))
        if torch_reduce:
            _lib.check(self._lib.n2nmn_train_join(self._ctx, s))
        if reduce:
            self.buckets.reduce_late()       # overlaps the encoder's backward pass
        _lib.check(self._lib.n2nmn_train_backward(self._ctx, C.byref(io), packed.handle, 1, s))
        if reduce:
            self.buckets.reduce_early()
            return self.buckets.wait()
        return 1.0

    def set_baseline(self, value: float):
        """the REINFORCE baseline (train_clevr_rl_gt_layout.py:120: tf.Variable(invalid_expr_loss))"""
        self.baseline.fill_(float(value))

    def get_baseline(self) -> float:
        return float(self.baseline.item())

    def apply(self, scale: float = 1.0):
        self.iteration += 1
        h = self.hyper
        _lib.check(self._lib.n2nmn_adam_step(self._ctx, self.grads.data_ptr(), scale, h['lr'],
                                             h['beta1'], h['beta2'], h['eps'],
                                             h['max_grad_l2_norm'], self.iteration,
                                             self.engine.stream()))

    def step(self, batch, gt_layout):
        scale = self.forward_backward(batch, gt_layout)
        self.apply(scale)
        return self.losses

    def step_rl(self, batch, sample_uniforms, lr: Optional[float] = 1e-4, update: bool = True):
        """One iteration of exp_clevr/train_clevr_rl_gt_layout.py:183-214: the decoder samples a
        layout per question (sample_uniforms [T_dec, N] in [0,1) replace tf.multinomial's RNG),
        the tokens are fetched and assembled on the host (the reference's partial_run does the
        same), then forward + backward of the policy-gradient loss and an Adam step with the
        fine-tuning learning rate (lr=None: the trainer's own).  update=False computes the losses and
        gradients only (no Adam step).  Returns (losses, tokens, expr_validity)."""
        e = self.engine
        s2s = e.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'],
                        sample_uniforms=sample_uniforms)
        tokens = s2s['predicted_tokens'].cpu().numpy()
        scale = self.forward_backward(batch, tokens, objective=1)
        if update:               # lr=None: the trainer's own learning rate
            saved = self.hyper['lr']
            self.hyper['lr'] = saved if lr is None else float(lr)
            self.apply(scale)
            self.hyper['lr'] = saved
        validity = self._keep[6].cpu().numpy().astype(bool) if self._keep[6] is not None else None
        return self.losses, tokens, validity

    # ------------------------------------------------------------------------------------
    def gradients(self) -> Dict[str, object]:
        """name -> view of the flat gradient buffer in the variable's reference shape."""
        return {name: self.grads[off:off + n].view(shape)
                for name, (off, n, shape) in self.layout.items()}

    def get_weights(self) -> Dict[str, object]:
        torch = _torch()
        out = {}
        for name, (off, n, shape) in self.layout.items():
            t = torch.empty(shape, dtype=torch.float32, device=self.engine.device)
            _lib.check(self._lib.n2nmn_get_weight(self._ctx, name.encode(), t.data_ptr(),
                                                  self.engine.stream()))
            out[name] = t
        return out

    def debug_tensor(self, name: str, shape):
        torch = _torch()
        n = int(np.prod(shape))
        t = torch.empty(n, dtype=torch.float32, device=self.engine.device)
        got = self._lib.n2nmn_train_debug_tensor(self._ctx, name.encode(), t.data_ptr(), n,
                                                 self.engine.stream())
        _lib.check(int(got))
        return t[:int(got)].view(shape) if int(got) == n else t[:int(got)]
