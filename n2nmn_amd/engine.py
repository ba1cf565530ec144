"""Thin host-side engine over the C-ABI: one context (weights + workspace) on one GPU.

PyTorch is used only as plumbing here: device memory (`torch.empty(..., device='cuda')`), the current
HIP stream and D2H copies.  All arithmetic runs in libn2nmn_hip.so; there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from . import _lib
from .nmn3_assembler import Assembler, PackedLayouts
from .spec import Dims, OP_CODE, MODULE_INPUT_NUM, MODULE_OUTPUT_TYPE


def _torch():
    import torch
    return torch


class Engine:
    _latest = None          # weak reference to the most recently built root engine (Modules() without engine=)

    @classmethod
    def latest(cls):
        return cls._latest() if cls._latest is not None else None

    def __init__(self, dims: Dims, assembler: Assembler, device: int = 0, _parent=None):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError('n2nmn_amd.Engine needs a HIP device (no CPU fallback exists)')
        if assembler.num_vocab_nmn != dims.num_vocab_nmn:
            raise ValueError('assembler vocabulary size != dims.num_vocab_nmn')
        self.dims = dims
        self.assembler = assembler
        self.device = torch.device('cuda', device)
        self._lib = _lib.lib()
        cd = _lib.Dims(**{k: int(v) for k, v in dims.asdict().items()})
        self._ctx = C.c_void_p()
        self._bufs: Dict[tuple, object] = {}
        self._ring = [None] * 4       # pinned staging slots for per-step host inputs (upload_i32)
        self._ring_i = 0
        self._side = None             # side stream + event for the hoisted conv_image GEMMs
        self._side_ev = None
        # overlap_conv = True: hoisted conv_image GEMMs on a side stream beside phase 1.  Off: measured on
        # MI355X the chip-filling recurrent steps slow down by more than the GEMMs take (0.882 ms per batch
        # overlapped vs 0.812 ms in line, profiles/r02_notes.md).  (An attribute, not an environment switch.)
        self.overlap_conv = False
        self.tokens_via_levels = False      # set_tokens_via_levels()
        self._parent = _parent
        if _parent is not None:
            _lib.check(self._lib.n2nmn_ctx_fork(_parent._ctx, C.byref(self._ctx)))
            # a fork takes the phase-2 path of its parent (set_tokens_via_levels is per context)
            self.tokens_via_levels = _parent.tokens_via_levels
            if self.tokens_via_levels:
                _lib.check(self._lib.n2nmn_set_tokens_via_levels(self._ctx, 1))
            return
        _lib.check(self._lib.n2nmn_ctx_create(C.byref(cd), device, C.byref(self._ctx)))
        P = np.ascontiguousarray(assembler.P, np.int32)
        W = np.ascontiguousarray(assembler.W, np.int32)
        b = np.ascontiguousarray(assembler.b, np.int32)
        _lib.check(self._lib.n2nmn_set_validity_tables(self._ctx, P.ctypes.data, W.ctypes.data,
                                                       b.ctypes.data))
        tok_op = np.ascontiguousarray(assembler._token_op, np.int32)
        _lib.check(self._lib.n2nmn_set_token_ops(self._ctx, tok_op.ctypes.data, tok_op.shape[0]))
        self._bufs: Dict[tuple, object] = {}
        self.all_tokens_valid = not (P.any() or W.any() or b.any())
        import weakref
        Engine._latest = weakref.ref(self)

    def __del__(self):
        ctx, self._ctx = getattr(self, '_ctx', None), None
        if ctx:
            try:
                self._lib.n2nmn_ctx_destroy(ctx)
            except Exception:
                pass

    def set_mode(self, mode: str):
        """'latency' (default) or 'throughput': workgroup tile shape of the recurrent step kernels
        (n2nmn_ctx_set_mode) -- use 'throughput' when several batches are in flight on forks.
        'throughput_bf16x3' (opt-in): 'throughput' with the recurrent contraction on bf16 MFMAs over
        three-way split operands (fp32-class accuracy, not the fp32 kernels' bits)."""
        self.mode = mode
        _lib.check(self._lib.n2nmn_ctx_set_mode(self._ctx, {'latency': 0, 'throughput': 1, 'throughput_ksplit': 2, 'throughput_bf16x3': 3}[mode]))

    def debug_set(self, key: str, value=None):
        """an A/B switch of this context (include/n2nmn.h section 7: n2nmn_debug_set); None removes it"""
        _lib.check(self._lib.n2nmn_debug_set(self._ctx, key.encode(), None if value is None else str(value).encode()))

    def set_tokens_via_levels(self, on: bool):
        """execute_tokens through the device-scheduled level path even where the layout walker applies."""
        self.tokens_via_levels = bool(on)
        _lib.check(self._lib.n2nmn_set_tokens_via_levels(self._ctx, int(bool(on))))

    def fork(self) -> 'Engine':
        """A sibling engine sharing this engine's weights with its own workspace, for running
        another batch concurrently on another stream / host thread."""
        return Engine(self.dims, self.assembler, self.device.index, _parent=self)

    # ------------------------------------------------------------------------------------
    def stream(self) -> int:
        return _torch().cuda.current_stream(self.device).cuda_stream

    def variable_names(self):
        out = {}
        n = _lib.check(self._lib.n2nmn_num_variables(self._ctx))
        for i in range(n):
            name = C.c_char_p()
            shape = (C.c_int64 * 4)()
            nd = C.c_int()
            _lib.check(self._lib.n2nmn_variable_info(self._ctx, i, C.byref(name), shape,
                                                     C.byref(nd)))
            out[name.value.decode()] = tuple(int(shape[k]) for k in range(nd.value))
        return out

    def load_weights(self, weights: Dict[str, object], strict: bool = True):
        """weights: reference variable name -> numpy array or torch tensor (any device).
        strict=False skips names that are not variables of the model (e.g. the optimiser slots and
        the `baseline` of a TF checkpoint read with n2nmn_amd.tf_checkpoint.read_checkpoint)."""
        torch = _torch()
        known = None if strict else set(self.variable_names())
        for name, w in weights.items():
            if known is not None and name not in known:
                continue
            t = torch.as_tensor(w).to(device=self.device, dtype=torch.float32).contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            _lib.check(self._lib.n2nmn_set_weight(self._ctx, name.encode(), t.data_ptr(), shape,
                                                  t.dim()))
        torch.cuda.synchronize(self.device)
        _lib.check(self._lib.n2nmn_commit_weights(self._ctx, self.stream()))

    def get_weights(self) -> Dict[str, object]:
        """reference variable name -> device tensor holding the current value (n2nmn_get_weight)"""
        torch = _torch()
        out = {}
        for name, shape in self.variable_names().items():
            t = torch.empty(shape, dtype=torch.float32, device=self.device)
            _lib.check(self._lib.n2nmn_get_weight(self._ctx, name.encode(), t.data_ptr(), self.stream()))
            out[name] = t
        torch.cuda.synchronize(self.device)
        return out

    def load_tf_checkpoint(self, prefix: str, verify: bool = True):
        """Load a TF V2 checkpoint (`<prefix>.index` + `.data-*`, README.md:75-79 of the reference)
        reading ONLY this model's variables (not the optimiser slots).  Returns the names found."""
        from .tf_checkpoint import read_checkpoint
        w = read_checkpoint(prefix, names=list(self.variable_names()), verify=verify,
                            skip_missing=True)
        self.load_weights(w)
        return sorted(w)

    # ------------------------------------------------------------------------------------
    def _buf(self, key, shape, dtype):
        torch = _torch()
        k = (key, tuple(shape), dtype)
        t = self._bufs.get(k)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._bufs[k] = t
        return t

    def _dev(self, x, dtype):
        torch = _torch()
        if x is None:
            return None
        t = torch.as_tensor(x)
        if t.device != self.device or t.dtype != dtype or not t.is_contiguous():
            t = t.to(device=self.device, dtype=dtype).contiguous()
        return t

    def upload_i32(self, host):
        """Host int32 array -> device through a pinned slot, WITHOUT synchronising the stream (a
        plain .to(device) from pageable memory waits for everything queued before it).  A slot is
        rewritten only after the copy that last read it has completed; the returned tensor stays
        valid until the call after next-but-two on this engine."""
        torch = _torch()
        host = np.ascontiguousarray(host, np.int32)
        i = self._ring_i % len(self._ring)
        self._ring_i += 1
        slot = self._ring[i]
        if slot is None or tuple(slot[0].shape) != host.shape:
            pin = torch.empty(host.shape, dtype=torch.int32).pin_memory()
            dev = torch.empty(host.shape, dtype=torch.int32, device=self.device)
            slot = self._ring[i] = (pin, dev, torch.cuda.Event())
        pin, dev, ev = slot
        ev.synchronize()
        pin.numpy()[...] = host
        dev.copy_(pin, non_blocking=True)
        ev.record(torch.cuda.current_stream(self.device))
        return dev

    def seq2seq(self, input_seq, seq_len, T_dec: Optional[int] = None, use_gt_layout: bool = False,
                gt_layout=None, sample_uniforms=None, forced_tokens=None, debug: bool = False,
                reuse_buffers: bool = True, phase: str = 'both', word_vecs: bool = True,
                image_feat=None, dropout=None, out_tokens=None, seq_len_host=None,
                eos_retire: bool = False, gt_len_host=None):
        """Phase 1.  input_seq [T,N] int32, seq_len [N] int32 (device tensors or anything
        convertible).  Returns a dict of device tensors named like the reference attributes
        (models_clevr/nmn3_netgen_att.py:305-322).  With reuse_buffers the outputs are views of
        engine-owned buffers that the next call overwrites (out_tokens: a caller-owned contiguous
        int32 [T_dec, N] device tensor for predicted_tokens instead).  image_feat (optional): the hoisted
        conv_image GEMMs of the batch are issued by this call too (n2nmn_seq2seq_io.image_feat) and
        walk() / execute_tokens(conv_done=True) can follow directly.  dropout (optional, models_vqa):
        (enc0, dec0) multiplier tensors [T, N, L] / [T_dec, N, L] for the output of LSTM layer 0
        (n2nmn_seq2seq_io.drop_enc0 / drop_dec0), either may be None.  seq_len_host (optional): a
        host int32 copy of seq_len (n2nmn_seq2seq_io.seq_length_host: per-step tile choice).
        eos_retire (inference, teacher-forced passes of >= 128 rows in a throughput mode):
        N2NMN_S2S_EOS_RETIRE -- rows leave the decoder at their layout's first <eos>; predicted_tokens are
        complete, atts / token_probs hold the live (row, step) pairs only (call again with phase='decoder'
        and eos_retire=False for every step).  gt_len_host: host int32 [N] layout lengths (tokens in front
        of the first <eos>, `layout_lengths`), optional."""
        torch = _torch()
        d = self.dims
        seq = self._dev(input_seq, torch.int32)
        lens = self._dev(seq_len, torch.int32)
        T, N = seq.shape
        Td = d.T_decoder if T_dec is None else int(T_dec)
        mk = (lambda k, s, dt: self._buf(k, s, dt)) if reuse_buffers else \
            (lambda k, s, dt: torch.empty(s, dtype=dt, device=self.device))
        out = {
            'predicted_tokens': mk('tok', (Td, N), torch.int32),
            'token_probs': mk('tp', (Td, N), torch.float32),
            'neg_entropy': mk('ne', (N,), torch.float32),
            'atts': mk('att', (Td, T, N), torch.float32),
            'word_vecs': mk('wv', (Td, N, d.embed_dim_txt), torch.float32),
            'log_seq_prob': mk('lsp', (N,), torch.float32),
        }
        if out_tokens is not None:
            if tuple(out_tokens.shape) != (Td, N) or out_tokens.dtype != torch.int32 or \
                    not out_tokens.is_contiguous() or out_tokens.device != self.device:
                raise ValueError('out_tokens must be a contiguous int32 [T_dec, N] tensor on the engine device')
            out['predicted_tokens'] = out_tokens
        if debug:
            out['token_scores'] = mk('ts', (Td, N, d.num_vocab_nmn), torch.float32)
            out['encoder_outputs'] = mk('eo', (T, N, d.lstm_dim), torch.float32)
            out['encoder_h_transformed'] = mk('eh', (T, N, d.lstm_dim), torch.float32)
            out['encoder_states'] = mk('es', (2, 2, N, d.lstm_dim), torch.float32)
        gt = self._dev(gt_layout, torch.int32)
        uni = self._dev(sample_uniforms, torch.float32)
        forced = self._dev(forced_tokens, torch.int32)
        io = _lib.Seq2SeqIO()
        io.input_seq = seq.data_ptr(); io.seq_length = lens.data_ptr()
        io.T_enc = T; io.N = N; io.T_dec = Td
        io.use_gt_layout = 1 if use_gt_layout else 0
        io.gt_layout = gt.data_ptr() if gt is not None else None
        io.sample_uniforms = uni.data_ptr() if uni is not None else None
        io.forced_tokens = forced.data_ptr() if forced is not None else None
        for k, t in out.items():
            setattr(io, k, t.data_ptr())
        feat = self._dev(image_feat, torch.float32)
        if feat is not None:
            if phase == 'encoder':
                raise ValueError('image_feat is read by the decoder half')
            io.image_feat = feat.data_ptr()
        drops = tuple(self._dev(x, torch.float32) for x in (dropout or (None, None)))
        if drops[0] is not None:
            io.drop_enc0 = drops[0].data_ptr()
        if drops[1] is not None:
            io.drop_dec0 = drops[1].data_ptr()
        lens_host = None
        if seq_len_host is not None:
            lens_host = np.ascontiguousarray(np.asarray(seq_len_host), np.int32)
            if lens_host.shape != (N,):
                raise ValueError('seq_len_host must hold N lengths')
            io.seq_length_host = lens_host.ctypes.data
        elif isinstance(seq_len, np.ndarray):
            lens_host = np.ascontiguousarray(seq_len, np.int32)
            io.seq_length_host = lens_host.ctypes.data
        glen_host = None
        if eos_retire:
            io.flags |= 2        # N2NMN_S2S_EOS_RETIRE (word_vecs: rows of live steps only; no neg_entropy / log_seq_prob)
            if gt_len_host is not None:
                glen_host = np.ascontiguousarray(np.asarray(gt_len_host), np.int32)
                if glen_host.shape != (N,):
                    raise ValueError('gt_len_host must hold N layout lengths')
                io.gt_length_host = glen_host.ctypes.data
        if not word_vecs:        # N2NMN_S2S_NO_WORD_VECS: word_vecs / neg_entropy / log_seq_prob not computed
            io.flags |= 1
            for k in ('word_vecs', 'neg_entropy', 'log_seq_prob'):
                out.pop(k)
        fn = {'both': self._lib.n2nmn_seq2seq_forward, 'encoder': self._lib.n2nmn_encoder_forward,
              'decoder': self._lib.n2nmn_decoder_forward}[phase]
        _lib.check(fn(self._ctx, C.byref(io), self.stream()))
        out['_keepalive'] = (seq, lens, gt, uni, forced, feat, drops)
        out['_input_seq'], out['_seq_length'] = seq, lens
        return out

    def execute(self, packed: PackedLayouts, image_feat, word_vecs, reuse_buffers: bool = True):
        """Phase 2: scores [num_rows, num_choices] (device tensor) for a packed program."""
        torch = _torch()
        feat = self._dev(image_feat, torch.float32)
        wv = self._dev(word_vecs, torch.float32)
        n_full = feat.shape[0]
        if wv.shape[1] != n_full:
            raise ValueError('word_vecs [T_dec, N, E] and image_feat [N, H, W, D] disagree on N')
        rows = packed.num_rows
        shape = (rows, self.dims.num_choices)
        scores = self._buf('scores', shape, torch.float32) if reuse_buffers else \
            torch.empty(shape, dtype=torch.float32, device=self.device)
        _lib.check(self._lib.n2nmn_execute_program(self._ctx, packed.handle, feat.data_ptr(),
                                                   wv.data_ptr(), n_full, scores.data_ptr(),
                                                   self.stream()))
        return scores

    # ---- phase 2 without the host hop (include/n2nmn.h section 4b) ---------------------------
    def set_defer_pool(self, mode: int):
        """-1 auto (>= 128 questions per launch), 0 pooling answers inside the walker, 1 as separate
        chip-wide launches (n2nmn_walk_set_defer_pool)."""
        _lib.check(self._lib.n2nmn_walk_set_defer_pool(self._ctx, int(mode)))

    def set_front_end(self, mode: int):
        """-1 auto, 0 text maps / Find epilogues inside the walker, 1 as chip-wide launches ahead of it
        (n2nmn_walk_set_front_end)."""
        _lib.check(self._lib.n2nmn_walk_set_front_end(self._ctx, int(mode)))

    def set_staged(self, mode: int):
        """-1 / 1: with the chip-wide front end and deferred pooling, Transform / FindSameProperty nodes
        run as chip-wide jobs and a light per-question kernel finishes the tree; 0: the one-workgroup
        walker serves every question (n2nmn_walk_set_staged)."""
        _lib.check(self._lib.n2nmn_walk_set_staged(self._ctx, int(mode)))

    def set_walk_levels(self, levels: int):
        """0 (default): the staged walker launches every nesting level of Transform / FindSameProperty a layout
        of T_dec tokens can reach (empty level launches leave at once) -- a question's route depends on its own
        layout only and repeated passes return the same bits; >= 1: exactly that many, deeper layouts on the
        one-workgroup walker (logits within 1e-5); -1: adaptive, as many as the last two passes needed: fewer
        empty launches, history-dependent last bits for nested layouts (n2nmn_walk_set_levels)."""
        _lib.check(self._lib.n2nmn_walk_set_levels(self._ctx, int(levels)))

    def set_nesting_bound(self, bound: int):
        """promise for the NEXT walker pass only: no layout nests Transform / FindSameProperty deeper than
        `bound` levels -> exactly that many level launches and no fall-back walker launch; -1: none
        (n2nmn_walk_set_nesting_bound).  `layout_nesting` computes it from a host layout array."""
        _lib.check(self._lib.n2nmn_walk_set_nesting_bound(self._ctx, int(bound)))

    def layout_nesting(self, gt_layout) -> int:
        """deepest nesting of _Transform / _FindSameProperty nodes in a HOST layout array [T_dec, N]: the
        stack machine of models_clevr/nmn3_assembler.py:153-222 on nesting depths only (a node's depth = the
        deepest of its inputs', + 1 if it is one of the two), all columns at once.  Columns that run out
        of stack stop counting; type errors are ignored, so the result is an upper bound of what the
        device's decoder (plan_layout, kernels_walk.hip) finds for the VALID layouts."""
        g = np.asarray(gt_layout)
        T, N = g.shape
        tok_op = np.asarray(self.assembler._token_op, np.int64)
        ok_tok = (g >= 0) & (g < tok_op.shape[0])
        ops = np.where(ok_tok, tok_op[np.clip(g, 0, tok_op.shape[0] - 1)], -2)
        OP_ARITY_MAX = 64
        arity = np.full(OP_ARITY_MAX, -1, np.int64)
        for code, name in self.assembler._op_name.items():
            arity[code] = self.assembler._input_num[name]
        heavy_codes = (OP_CODE['_Transform'], OP_CODE['_FindSameProperty'])
        stack = np.zeros((N, T + 2), np.int64)
        sp = np.zeros(N, np.int64)
        live = np.ones(N, bool)
        deepest = np.zeros(N, np.int64)
        cols = np.arange(N)
        for t in range(T):
            op = ops[t]
            live &= op >= 0
            k = np.where(live, arity[np.clip(op, 0, OP_ARITY_MAX - 1)], 0)
            live &= (k >= 0) & (sp >= k)
            k = np.where(live, k, 0)
            top = stack[cols, np.maximum(sp - 1, 0)]
            sec = stack[cols, np.maximum(sp - 2, 0)]
            hd = np.where(k >= 1, top, 0)
            hd = np.where(k >= 2, np.maximum(hd, sec), hd) + np.isin(op, heavy_codes)
            nsp = sp - k
            stack[cols[live], nsp[live]] = hd[live]
            sp = np.where(live, nsp + 1, sp)
            deepest = np.where(live, np.maximum(deepest, hd), deepest)
        return int(deepest.max()) if N else 0

    def walk_supported(self) -> bool:
        return bool(self._lib.n2nmn_walk_supported(self._ctx))

    def conv_image(self, image_feat, tokens=None, T_dec: Optional[int] = None,
                   find: bool = True, fsp: bool = True):
        """Hoisted conv_image GEMMs of all images into this engine's workspace (needs only the
        features: callers run it on a side stream beside phase 1).  tokens (device [T_dec, N]): the
        FindSameProperty map is computed only for images whose layout uses that operator."""
        torch = _torch()
        feat = self._dev(image_feat, torch.float32)
        tok = self._dev(tokens, torch.int32)
        Td = 0 if tok is None else (tok.shape[0] if T_dec is None else int(T_dec))
        _lib.check(self._lib.n2nmn_conv_image(
            self._ctx, feat.data_ptr(), feat.shape[0], (1 if find else 0) | (2 if fsp else 0),
            tok.data_ptr() if tok is not None else None, Td, self.stream()))
        return feat

    def walk(self, jobs, N: int, T_dec: int, conv_inline: bool = False):
        """One walker launch over K in-flight batches.  jobs: list of (engine, tokens, image_feat,
        word_vecs, scores, validity) device tensors; `engine` is the (fork of this) engine whose
        conv_image() was called for that batch -- or, with conv_inline, nobody's: the walker call computes
        the maps itself right in front of their reader (n2nmn_walk_set_conv_inline)."""
        if conv_inline:
            _lib.check(self._lib.n2nmn_walk_set_conv_inline(self._ctx, 1))
        arr = (_lib.WalkBatch * len(jobs))()
        T_enc = 0
        for i, job in enumerate(jobs):
            eng, tok, feat, wv, sc, val = job[:6]
            arr[i].ctx = eng._ctx
            arr[i].tokens = tok.data_ptr(); arr[i].image_feat = feat.data_ptr()
            arr[i].word_vecs = wv.data_ptr() if wv is not None else None
            arr[i].scores = sc.data_ptr()
            arr[i].validity = val.data_ptr() if val is not None else None
            if len(job) > 6 and job[6] is not None:      # (atts [T_dec,T_enc,N], input_seq, seq_length)
                atts, seq, lens = job[6]
                arr[i].atts = atts.data_ptr(); arr[i].input_seq = seq.data_ptr()
                arr[i].seq_length = lens.data_ptr()
                T_enc = int(atts.shape[1])
        _lib.check(self._lib.n2nmn_walk_layouts(self._ctx, arr, len(jobs), int(T_dec), T_enc, int(N),
                                                self.stream()))

    def execute_tokens(self, tokens, image_feat, word_vecs, reuse_buffers: bool = True,
                       conv_done: bool = False, atts=None, out=None):
        """Phase 2 straight from DEVICE tokens [T_dec, N]: no token fetch, no host assembly, no
        program upload.  Returns (scores [N, C], validity [N] int32) device tensors; out = (scores,
        validity): caller-owned contiguous tensors of those shapes to write instead."""
        torch = _torch()
        tok = self._dev(tokens, torch.int32)
        feat = self._dev(image_feat, torch.float32)
        wv = self._dev(word_vecs, torch.float32)
        Td, N = tok.shape
        mk = (lambda k, s, dt: self._buf(k, s, dt)) if reuse_buffers else \
            (lambda k, s, dt: torch.empty(s, dtype=dt, device=self.device))
        if out is not None:
            scores, validity = out
            if tuple(scores.shape) != (N, self.dims.num_choices) or tuple(validity.shape) != (N,) or \
                    scores.dtype != torch.float32 or validity.dtype != torch.int32 or \
                    not (scores.is_contiguous() and validity.is_contiguous()):
                raise ValueError('out = (float32 [N, C], int32 [N]) contiguous device tensors')
        else:
            scores = mk('wscores', (N, self.dims.num_choices), torch.float32)
            validity = mk('wvalid', (N,), torch.int32)
        if not self.walk_supported() or self.tokens_via_levels:
            # no layout walker for these dimensions (models_vqa): the level path, assembled and scheduled on
            # the device (n2nmn_execute_tokens); the conv_image GEMMs are part of the call
            _lib.check(self._lib.n2nmn_execute_tokens(self._ctx, tok.data_ptr(), Td, N, feat.data_ptr(),
                                                      wv.data_ptr(), scores.data_ptr(), validity.data_ptr(),
                                                      self.stream()))
            return scores, validity
        # atts = (atts [T_dec, T_enc, N], input_seq [T_enc, N], seq_length [N]): text maps from the
        # decoder's attention and the commit-time (embedding . W_txt) tables; word_vecs not needed
        self.walk([(self, tok, feat, wv, scores, validity, atts)], N, Td, conv_inline=not conv_done)
        return scores, validity

    def module_forward(self, name: str, inputs, time_idx, batch_idx, image_feat, word_vecs):
        """One module operator on explicit inputs (Modules.<X>Module)."""
        torch = _torch()
        if name not in OP_CODE:
            raise KeyError(name)
        arity = MODULE_INPUT_NUM[name]
        if len(inputs) != arity:
            raise ValueError('%s takes %d attention input(s)' % (name, arity))
        t_idx = np.ascontiguousarray(np.asarray(_to_host(time_idx)).reshape(-1), np.int32)
        b_idx = np.ascontiguousarray(np.asarray(_to_host(batch_idx)).reshape(-1), np.int32)
        nb = t_idx.shape[0]
        if b_idx.shape[0] != nb:
            raise ValueError('time_idx and batch_idx must have the same length')
        d = self.dims
        feat = self._dev(image_feat, torch.float32)
        wv = self._dev(word_vecs, torch.float32)
        ins = [self._dev(x, torch.float32).reshape(nb, d.H * d.W) for x in inputs]
        if MODULE_OUTPUT_TYPE[name] == 'att':
            out = torch.empty((nb, d.H, d.W, 1), dtype=torch.float32, device=self.device)
        else:
            out = torch.empty((nb, d.num_choices), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.n2nmn_module_forward(
            self._ctx, OP_CODE[name], nb,
            ins[0].data_ptr() if arity >= 1 else None, ins[1].data_ptr() if arity >= 2 else None,
            t_idx.ctypes.data, b_idx.ctypes.data, feat.data_ptr(), wv.data_ptr(), feat.shape[0],
            out.data_ptr(), self.stream()))
        return out

    def profile_begin(self):
        _lib.check(self._lib.n2nmn_profile_begin(self._ctx))

    def event_overhead_us(self, iters: int = 200) -> float:
        """fixed cost of one profiler entry (event pair around an empty kernel), microseconds"""
        us = C.c_double()
        _lib.check(self._lib.n2nmn_debug_event_overhead(self._ctx, iters, C.byref(us), self.stream()))
        return us.value

    def walk_replay_us(self, which: int, iters: int = 50) -> float:
        """average us per launch of the last walker launch's kernel (0 walker, 1 pool, 2 heads),
        `iters` back-to-back launches inside one event pair (n2nmn_debug_walk_replay)"""
        us = C.c_double()
        _lib.check(self._lib.n2nmn_debug_walk_replay(self._ctx, which, iters, C.byref(us), self.stream()))
        return us.value

    def profile_end(self):
        """-> list of dicts {name, launches, total_ms, flops, bytes} per kernel family."""
        _lib.check(self._lib.n2nmn_profile_end(self._ctx, self.stream()))
        out = []
        for i in range(self._lib.n2nmn_profile_num_families()):
            name = C.c_char_p(); n = C.c_int64()
            ms, fl, by = C.c_double(), C.c_double(), C.c_double()
            _lib.check(self._lib.n2nmn_profile_get(self._ctx, i, C.byref(name), C.byref(n),
                                                   C.byref(ms), C.byref(fl), C.byref(by)))
            out.append(dict(name=name.value.decode(), launches=n.value, total_ms=ms.value,
                            flops=fl.value, bytes=by.value))
        return out

    def fc(self, A, W, bias=None, relu: bool = False):
        """out = [relu](A . W + bias) through n2nmn_fc_forward (util/cnn.py:87-126; on im2col rows the
        VALID strided convolutions of models_shapes/shapes_convnet.py).  Device tensor [M, N]."""
        torch = _torch()
        A = self._dev(A, torch.float32); W = self._dev(W, torch.float32)
        bias = self._dev(bias, torch.float32)
        M, K = A.shape
        N = W.shape[1]
        out = torch.empty((M, N), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.n2nmn_fc_forward(self._ctx, A.data_ptr(), W.data_ptr(),
                                              bias.data_ptr() if bias is not None else None,
                                              out.data_ptr(), M, N, K, 1 if relu else 0, self.stream()))
        return out

    def set_validity_tables(self, P, W, b):
        """replace the decoder's validity automaton (n2nmn_set_validity_tables): all-zero tables make every
        token valid at every step -- the decoder of models_shapes/nmn3_netgen_att.py has no automaton"""
        P = np.ascontiguousarray(P, np.int32); W = np.ascontiguousarray(W, np.int32)
        b = np.ascontiguousarray(b, np.int32)
        _lib.check(self._lib.n2nmn_set_validity_tables(self._ctx, P.ctypes.data, W.ctypes.data, b.ctypes.data))
        self.all_tokens_valid = not (P.any() or W.any() or b.any())

    def gemm(self, A, B, bias=None):
        torch = _torch()
        A = self._dev(A, torch.float32); B = self._dev(B, torch.float32)
        bias = self._dev(bias, torch.float32)
        M, K = A.shape
        N = B.shape[1]
        Cm = torch.empty((M, N), dtype=torch.float32, device=self.device)
        _lib.check(self._lib.n2nmn_debug_gemm(self._ctx, A.data_ptr(), B.data_ptr(),
                                              bias.data_ptr() if bias is not None else None,
                                              Cm.data_ptr(), M, N, K, self.stream()))
        return Cm

    # ------------------------------------------------------------------------------------
    def layout_lengths(self, gt_layout) -> np.ndarray:
        """host int32 [N]: tokens of each column of a HOST layout array [T_dec, N] in front of its first
        <eos> (T_dec when it has none) -- n2nmn_seq2seq_io.gt_length_host"""
        g = np.asarray(gt_layout)
        eos = g == self.assembler.EOS_idx
        return np.where(eos.any(0), eos.argmax(0), g.shape[0]).astype(np.int32)

    def forward(self, batch, T_dec: Optional[int] = None, use_gt_layout: bool = False,
                gt_layout=None, sample_uniforms=None, host_assemble: bool = False,
                fetch: bool = True, out=None, eos_retire: bool = False):
        """The whole hot path of exp_clevr/eval_clevr.py:103-135 for one batch:
        phase 1 -> token fetch (the one host sync) -> C++ assemble/pack -> phase 2.
        Returns (scores device tensor, tokens numpy [T_dec,N], validity numpy [N]).

        Default (dimensions the walker supports): phase 1 -> n2nmn_execute_tokens; the layouts are
        decoded on the device, nothing synchronises between the phases, and with fetch=False the
        tokens / validity are returned as device tensors (no synchronisation at all); they are views
        of engine-owned buffers the next call overwrites unless out = (scores [N, C] f32, tokens
        [T_dec, N] i32, validity [N] i32) names caller-owned tensors (device path only).
        host_assemble=True keeps the reference's flow (token fetch, C++ Assembler, level scheduler).

        With use_gt_layout and a HOST gt_layout (numpy, as the reference's data reader delivers
        it, util/clevr_train/data_reader.py:74-82) the predicted tokens are the ground-truth layout
        by construction (models_clevr/nmn3_netgen_att.py:236-238), so the program is assembled
        from the host copy up front and the step has no host synchronisation at all.

        eos_retire (device path): inference option N2NMN_S2S_EOS_RETIRE -- the decoder
        runs a row only up to its layout's first <eos> (teacher-forced layouts: rows ranked by length up
        front; layouts the decoder chooses: rows leave the recurrence as they emit <eos>); scores / tokens / validity are bit-identical to
        the full decoder's (the fetches of exp_clevr/eval_clevr.py:103-135), the decoder's own outputs for
        the steps behind it are computed on demand (`decoder_outputs`)."""
        if self.walk_supported() and not host_assemble:
            # device path: the walker decodes the layouts itself (no sync between the phases).  The
            # hoisted conv_image GEMMs ride in phase 1's own GEMM launch (with encoder_h_transform
            # and q); engine.overlap_conv = True puts them on a side stream beside the recurrent chain
            # instead (measured slower: they take CUs from the chip-filling step kernels)
            torch = _torch()
            gt_dev = self.upload_i32(gt_layout) if isinstance(gt_layout, np.ndarray) else \
                self._dev(gt_layout, torch.int32)
            feat = self._dev(batch['image_feat_batch'], torch.float32)
            cur = torch.cuda.current_stream(self.device)
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device)
                self._side_ev = torch.cuda.Event()
            known = use_gt_layout and gt_dev is not None     # layouts known before phase 1
            if self.overlap_conv:
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side):
                    self.conv_image(feat, gt_dev if known else None, T_dec, find=True, fsp=known)
                    self._side_ev.record(self._side)
            table = self.dims.num_vocab_txt <= 4096
            # passes of many questions: the conv_image GEMM leaves phase 1's merged launch and runs inside the
            # walker call, right in front of walk_find (its maps then come back from the Infinity Cache); one
            # batch of 64 keeps the merged launch (one launch fewer on the latency path)
            conv_late = (not self.overlap_conv) and feat.shape[0] >= 128
            # (teacher-forced: rows ranked by layout length up front; greedy / sampled: rows leave as they emit <eos>)
            retire = bool(eos_retire) and table and (known or not use_gt_layout)
            glen = batch.get('gt_length_host') if (retire and known) else None
            if retire and known and glen is None and isinstance(gt_layout, np.ndarray):
                glen = self.layout_lengths(gt_layout)
            s2s = self.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], T_dec,
                               use_gt_layout, gt_dev, sample_uniforms, word_vecs=not table,
                               image_feat=None if (self.overlap_conv or conv_late) else feat,
                               out_tokens=None if out is None else out[1],
                               seq_len_host=batch.get('seq_length_host'),
                               eos_retire=retire, gt_len_host=glen)
            # (device tensors of this call: the context keeps pointers to them until the next encoder call)
            self._last_s2s_args = (s2s['_input_seq'], s2s['_seq_length'], T_dec, use_gt_layout, gt_dev,
                                   sample_uniforms)
            if self.overlap_conv and not known:
                self.conv_image(feat, s2s['predicted_tokens'], find=False, fsp=True)
            if self.overlap_conv:
                cur.wait_event(self._side_ev)
            # layouts held on the host: the walker is told how deep they nest (exact level launches, no
            # fall-back launch); a SuperBucket passes the figure it computed when its slots were filled
            nest = batch.get('gt_nesting_host') if known else None
            if nest is None and known and isinstance(gt_layout, np.ndarray):
                nest = self.layout_nesting(gt_layout)
            if nest is not None:
                self.set_nesting_bound(int(nest))
            scores, validity = self.execute_tokens(
                s2s['predicted_tokens'], feat, s2s.get('word_vecs'), conv_done=not conv_late,
                atts=(s2s['atts'], s2s['_input_seq'], s2s['_seq_length']) if table else None,
                out=None if out is None else (out[0], out[2]))
            if not fetch:
                return scores, s2s['predicted_tokens'], validity
            return scores, s2s['predicted_tokens'].cpu().numpy(), validity.cpu().numpy().astype(bool)
        if out is not None:
            raise ValueError('out= needs the device path (walker dimensions, host_assemble=False)')
        if use_gt_layout and isinstance(gt_layout, np.ndarray):
            tokens = np.ascontiguousarray(gt_layout, np.int32)
            packed, validity = self.assembler.assemble_packed(tokens)
            s2s = self.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], T_dec,
                               True, self.upload_i32(tokens), sample_uniforms)
            scores = self.execute(packed, batch['image_feat_batch'], s2s['word_vecs'])
            return scores, tokens, validity
        s2s = self.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], T_dec,
                           use_gt_layout, gt_layout, sample_uniforms)
        tokens = s2s['predicted_tokens'].cpu().numpy()
        packed, validity = self.assembler.assemble_packed(tokens)
        scores = self.execute(packed, batch['image_feat_batch'], s2s['word_vecs'])
        return scores, tokens, validity


    def decoder_outputs(self):
        """The decoder's outputs of the LAST forward() at EVERY step -- predicted_tokens, token_probs,
        atts, neg_entropy, word_vecs, log_seq_prob -- recomputed from the encoder results the context
        still holds (n2nmn_decoder_forward without N2NMN_S2S_EOS_RETIRE): the on-demand half of the
        eos_retire contract (training and the debug fetches need all T_dec steps).  Call it before the
        next forward() of this engine."""
        args = getattr(self, '_last_s2s_args', None)
        if args is None:
            raise RuntimeError('decoder_outputs: no forward() on this engine yet')
        seq, lens, T_dec, use_gt, gt_dev, uni = args
        return self.seq2seq(seq, lens, T_dec, use_gt, gt_dev, uni, phase='decoder', reuse_buffers=False)


def _to_host(x):
    if hasattr(x, 'detach'):
        return x.detach().cpu().numpy()
    return x
