"""models_vqa variant (BASELINE.json configs[4]): the VQA / VQAv2 model of the reference
(`models_vqa/nmn3_model.py`, `models_vqa/nmn3_modules.py`, `models_vqa/question_prior_net.py`,
`exp_vqa/eval_vqa2.py:27-39`) on the same gfx950 kernels as the CLEVR path.

What is different from models_clevr, and how it maps onto the C-ABI:

* four modules: `_Find` (= FindModule), `_Transform` (attention-pooled three-way product: the
  arithmetic of models_clevr's FindSamePropertyModule, so it runs under that operator code with the
  variables of scope `TransformModule`), `_And`, `_Describe`;
* the feature grid gets two coordinate channels (`add_spatial_coordinate_map`): built on the GPU by
  `n2nmn_add_coords` into a buffer whose depth is padded to a multiple of 16 (2050 -> 2064; the
  padded channels are zero and so are the matching weight rows);
* `lstm_dim = 1000` is not a multiple of the 128-wide K split of the LSTM kernels: the context is
  created with 1024 and the variables are zero-padded when they are loaded (padded hidden units have
  zero weights and biases, so c = h = 0 for them forever and every real output is unchanged);
* `scores = scores_nmn + question_prior_net(encoder_states)` (`n2nmn_question_prior_add`).

The reference-shaped variable names / shapes are `vqa_variable_shapes`; `VQAEngine.load_weights`
takes exactly those.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, asdict
from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib
from .engine import Engine
from .train import Trainer
from .nmn3_assembler import Assembler
from .spec import Dims, lstm_var, PREFIX, _ENC, _DEC, _MOD

# exp_vqa/data/vocabulary_layout.txt
VQA_MODULE_NAMES: Tuple[str, ...] = ('_Find', '_Transform', '_And', '_Describe', '<eos>')
# C-ABI operator code of each models_vqa module (include/n2nmn.h, enum n2nmn_op)
VQA_OP_CODE: Dict[str, int] = {'_Find': 1, '_Transform': 3, '_And': 5, '_Describe': 13}
_QPN = PREFIX + 'question_prior_net/'


@dataclass(frozen=True)
class VQADims:
    """exp_vqa/eval_vqa2.py:27-39 (reference-side dimensions; D = image features without coords)."""
    H: int = 14
    W: int = 14
    D: int = 2048
    map_dim: int = 1024            # models_vqa/nmn3_modules.py:82,123,193
    embed_dim_txt: int = 300
    embed_dim_nmn: int = 300
    lstm_dim: int = 1000
    num_layers: int = 2
    num_vocab_txt: int = 17742     # exp_vqa/data/vocabulary_vqa.txt
    num_vocab_nmn: int = 5
    num_choices: int = 3001        # exp_vqa/data/answers_vqa.txt
    T_encoder: int = 26
    T_decoder: int = 13
    N: int = 128                   # BASELINE.json configs[4] (the reference itself uses 50 / 64)
    qpn_hidden: int = 500          # question_prior_net(hidden_dim=500); 0 = use_qpn False

    def asdict(self):
        return asdict(self)


def _round_up(x, m):
    return (x + m - 1) // m * m


def internal_dims(d: VQADims) -> Dims:
    """Dimensions of the GPU context: lstm_dim padded to 128, feature depth (+2 coords) to 16."""
    return Dims(H=d.H, W=d.W, D=_round_up(d.D + 2, 16), map_dim=d.map_dim,
                embed_dim_txt=d.embed_dim_txt, embed_dim_nmn=d.embed_dim_nmn,
                lstm_dim=_round_up(d.lstm_dim, 128), num_layers=d.num_layers,
                num_vocab_txt=d.num_vocab_txt, num_vocab_nmn=d.num_vocab_nmn,
                num_choices=d.num_choices, T_encoder=d.T_encoder, T_decoder=d.T_decoder, N=d.N,
                kernel_size=5, variant=1, qpn_hidden=d.qpn_hidden)


def vqa_variable_shapes(d: VQADims) -> Dict[str, Tuple[int, ...]]:
    """name -> shape of every variable of the models_vqa graph, reference layout."""
    L, E, En, M, C = d.lstm_dim, d.embed_dim_txt, d.embed_dim_nmn, d.map_dim, d.num_choices
    Dc = d.D + 2
    s: Dict[str, Tuple[int, ...]] = {}
    s[_ENC + 'embedding_mat'] = (d.num_vocab_txt, E)
    s[lstm_var('encoder', 0, 'weights')] = (E + L, 4 * L)
    s[lstm_var('encoder', 0, 'biases')] = (4 * L,)
    s[lstm_var('encoder', 1, 'weights')] = (2 * L, 4 * L)
    s[lstm_var('encoder', 1, 'biases')] = (4 * L,)
    s[_ENC + 'encoder_h_transform/weights'] = (L, L)
    s[_ENC + 'encoder_h_transform/biases'] = (L,)
    s[_DEC + 'embedding_mat'] = (d.num_vocab_nmn, En)
    s[_DEC + 'go_embedding'] = (1, En)
    s[_DEC + 'att_prediction/v'] = (L,)
    s[_DEC + 'att_prediction/weights'] = (L, L)
    s[_DEC + 'att_prediction/biases'] = (L,)
    s[_DEC + 'token_prediction/weights'] = (2 * L, d.num_vocab_nmn)
    s[_DEC + 'token_prediction/biases'] = (d.num_vocab_nmn,)
    s[lstm_var('decoder', 0, 'weights')] = (En + L, 4 * L)
    s[lstm_var('decoder', 0, 'biases')] = (4 * L,)
    s[lstm_var('decoder', 1, 'weights')] = (2 * L, 4 * L)
    s[lstm_var('decoder', 1, 'biases')] = (4 * L,)

    def layer(scope, name, shape):
        s[_MOD + scope + '/' + name + '/weights'] = shape
        s[_MOD + scope + '/' + name + '/biases'] = (shape[-1],)

    layer('FindModule', 'conv_image', (Dc, M))
    layer('FindModule', 'fc_text', (E, M))
    layer('FindModule', 'conv_eltwise', (M, 1))
    layer('TransformModule', 'conv_image', (Dc, M))
    layer('TransformModule', 'fc_text', (E, M))
    layer('TransformModule', 'fc_att', (Dc, M))
    layer('TransformModule', 'conv_eltwise', (M, 1))
    layer('DescribeModule', 'fc_text', (E, M))
    layer('DescribeModule', 'fc_att', (Dc, M))
    layer('DescribeModule', 'fc_eltwise', (M, C))
    if d.qpn_hidden > 0:
        s[_QPN + 'fc1/weights'] = (2 * L, d.qpn_hidden)
        s[_QPN + 'fc1/biases'] = (d.qpn_hidden,)
        s[_QPN + 'fc2/weights'] = (d.qpn_hidden, C)
        s[_QPN + 'fc2/biases'] = (C,)
    return s


def pad_variable(name: str, w: np.ndarray, d: VQADims, di: Dims) -> np.ndarray:
    """Reference-shaped variable -> the (zero padded) shape the GPU context expects."""
    L, Lp = d.lstm_dim, di.lstm_dim
    Dc, Dp = d.D + 2, di.D
    w = np.asarray(w, np.float32)

    def pad_gate_cols(x):                         # [..., 4L] -> [..., 4Lp], gate-major columns
        out = np.zeros(x.shape[:-1] + (4 * Lp,), np.float32)
        for g in range(4):
            out[..., g * Lp:g * Lp + L] = x[..., g * L:(g + 1) * L]
        return out

    def pad_rows_blocks(x, blocks):               # blocks: [(src0, src1, dst0)], total rows given
        total = blocks[-1][3]
        out = np.zeros((total,) + x.shape[1:], np.float32)
        for s0, s1, d0, _ in blocks:
            out[d0:d0 + (s1 - s0)] = x[s0:s1]
        return out

    if 'basic_lstm_cell/weights' in name:
        x = pad_gate_cols(w)
        if 'cell_0' in name:                      # [E + L] rows: input rows stay, hidden rows padded
            E = w.shape[0] - L
            return pad_rows_blocks(x, [(0, E + L, 0, E + Lp)])
        return pad_rows_blocks(x, [(0, L, 0, 2 * Lp), (L, 2 * L, Lp, 2 * Lp)])
    if 'basic_lstm_cell/biases' in name:
        return pad_gate_cols(w)
    if name.endswith('encoder_h_transform/weights') or name.endswith('att_prediction/weights'):
        out = np.zeros((Lp, Lp), np.float32); out[:L, :L] = w
        return out
    if name.endswith('encoder_h_transform/biases') or name.endswith('att_prediction/biases') \
            or name.endswith('att_prediction/v'):
        out = np.zeros((Lp,), np.float32); out[:L] = w
        return out
    if name.endswith('token_prediction/weights') or name.endswith('question_prior_net/fc1/weights'):
        return pad_rows_blocks(w, [(0, L, 0, 2 * Lp), (L, 2 * L, Lp, 2 * Lp)])
    if name.endswith('conv_image/weights') or name.endswith('fc_att/weights'):
        out = np.zeros((Dp, w.shape[1]), np.float32); out[:Dc] = w
        return out
    return w


class VQAEngine:
    """GPU context of the models_vqa variant + the eval-loop forward of exp_vqa/eval_vqa2.py."""

    def __init__(self, dims: VQADims, device: int = 0):
        self.dims = dims
        self.idims = internal_dims(dims)
        self.assembler = Assembler(list(VQA_MODULE_NAMES), op_code=VQA_OP_CODE)
        self.engine = Engine(self.idims, self.assembler, device=device)
        self._feat_c = None
        self._slabs = []           # weak references to the resident input slabs (feature_slab)

    def load_weights(self, weights: Dict[str, object]):
        shapes = vqa_variable_shapes(self.dims)
        missing = set(shapes) - set(weights)
        if missing:
            raise KeyError('missing variables: %s' % sorted(missing)[:3])
        padded = {}
        for name, shape in shapes.items():
            w = weights[name]
            w = w.detach().cpu().numpy() if hasattr(w, 'detach') else np.asarray(w)
            if tuple(w.shape) != tuple(shape):
                raise ValueError('shape mismatch for %s: %s vs %s' % (name, w.shape, shape))
            padded[name] = pad_variable(name, w, self.dims, self.idims)
        self.engine.load_weights(padded)

    def feature_slab(self, n: Optional[int] = None):
        """A resident input slab [n, H, W, Dp] in the layout the kernels read: the D image channels, the two
        coordinate channels of add_spatial_coordinate_map (models_vqa/nmn3_modules.py:11-31) and zero padding
        to Dp.  The coordinates do not depend on the image, so they are written ONCE, here; a client writes
        its features into `slab[..., :D]` (any copy; the bytes it moves are the same) and hands the slab to
        forward() as `image_feat_batch` -- no per-pass n2nmn_add_coords, which at 1024 rows re-copied 1.6 GB
        (0.75 ms of a 16.4 ms pass).  The slab stays valid for the life of the engine."""
        import torch
        e, d, di = self.engine, self.dims, self.idims
        n = di.N if n is None else int(n)
        slab = torch.empty((n, di.H, di.W, di.D), dtype=torch.float32, device=e.device)
        zero = torch.zeros((1, d.H, d.W, d.D), dtype=torch.float32, device=e.device)
        one = torch.empty((1, di.H, di.W, di.D), dtype=torch.float32, device=e.device)
        _lib.check(e._lib.n2nmn_add_coords(e._ctx, zero.data_ptr(), 1, d.D, one.data_ptr(), e.stream()))
        slab.copy_(one.expand_as(slab))
        # remembered by a WEAK reference and recognised by storage identity (features_with_coords): once the caller
        # drops the slab the entry dies with it, so an unrelated tensor that the allocator later places at the same
        # address is never mistaken for one that already carries the coordinate channels
        import weakref
        self._slabs = [r for r in self._slabs if r() is not None]
        self._slabs.append(weakref.ref(slab))
        return slab

    def _slab_of(self, t):
        """the live feature_slab() tensor `t` is a row-aligned contiguous view of, or None"""
        sp = t.untyped_storage().data_ptr()
        for r in self._slabs:
            base = r()
            if base is None or base.untyped_storage().data_ptr() != sp:
                continue
            row = base.shape[1] * base.shape[2] * base.shape[3]
            if t.is_contiguous() and tuple(t.shape[1:]) == tuple(base.shape[1:]) and t.storage_offset() % row == 0 and \
                    t.storage_offset() // row + t.shape[0] <= base.shape[0]:
                return base
        return None

    def features_with_coords(self, image_feat):
        """[N,H,W,D] image features -> [N,H,W,Dp] with the coordinate map appended (on the GPU); a slab
        of `feature_slab` already has it and is returned as it is."""
        import torch
        e, d, di = self.engine, self.dims, self.idims
        if hasattr(image_feat, 'data_ptr') and image_feat.dim() == 4 and image_feat.shape[-1] == di.D != d.D:
            if self._slab_of(image_feat) is None:
                raise ValueError('a [N, H, W, %d] tensor must be a slab of feature_slab() that is still alive (or whole '
                                 'rows of one); raw image features are [N, H, W, %d]' % (di.D, d.D))
            return image_feat
        feat = e._dev(image_feat, torch.float32)
        n = feat.shape[0]
        if self._feat_c is None or self._feat_c.shape[0] != n:
            self._feat_c = torch.empty((n, di.H, di.W, di.D), dtype=torch.float32, device=e.device)
        _lib.check(e._lib.n2nmn_add_coords(e._ctx, feat.data_ptr(), n, d.D, self._feat_c.data_ptr(),
                                           e.stream()))
        return self._feat_c

    def add_question_prior(self, scores):
        """scores += question_prior_net(encoder states of the last phase 1)  (models_vqa/nmn3_model.py:106-114,
        question_prior_net.py:10-28), in place; returns scores"""
        e = self.engine
        _lib.check(e._lib.n2nmn_question_prior_add(e._ctx, scores.shape[0], scores.data_ptr(), e.stream()))
        return scores

    def forward(self, batch, use_gt_layout: bool = False, gt_layout=None, forced_tokens=None,
                use_qpn: bool = True, host_assemble: bool = False, fetch: bool = True,
                eos_retire: bool = False):
        """phase 1 -> phase 2 (+ question prior).  Returns (scores device tensor [N, num_choices], tokens,
        validity) -- scores = scores_nmn + scores_qpn (models_vqa/nmn3_model.py:106-114); the eval
        script's `scores[:, 0] = -1e10` is the caller's.

        Layouts the decoder chooses (or a DEVICE gt_layout): phase 2 runs straight from the device
        tokens -- n2nmn_execute_tokens assembles and level-schedules the program on the device -- so
        nothing synchronises between the phases; with fetch=False tokens / validity come back as
        device tensors (no synchronisation at all).  host_assemble=True keeps the reference's flow
        (token fetch, C++ Assembler, host level scheduler: exp_vqa/eval_vqa2.py:118-131).

        With use_gt_layout and a HOST gt_layout (numpy, as the reference's data reader delivers it,
        util/vqa_train/data_reader.py) the predicted tokens ARE the ground-truth layout
        (models_vqa/nmn3_netgen_att.py: teacher forcing), so the program is assembled from the host
        copy up front and the call has no host synchronisation either.

        eos_retire (teacher-forced passes of >= 128 rows in a throughput mode): N2NMN_S2S_EOS_RETIRE -- rows
        leave the decoder at their layout's first <eos> (include/n2nmn.h); scores / tokens / validity as the
        full decoder's."""
        e = self.engine
        known = use_gt_layout and isinstance(gt_layout, np.ndarray) and forced_tokens is None
        if known:
            tokens = np.ascontiguousarray(gt_layout, np.int32)
            packed, validity = self.assembler.assemble_packed(tokens)
            gt_dev = e.upload_i32(tokens)
        retire = bool(eos_retire) and use_gt_layout and forced_tokens is None
        s2s = e.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], self.dims.T_decoder,
                        use_gt_layout, gt_dev if known else gt_layout, None, forced_tokens,
                        eos_retire=retire, gt_len_host=e.layout_lengths(tokens) if (retire and known) else None)
        feat_c = self.features_with_coords(batch['image_feat_batch'])
        if known or host_assemble:
            if not known:
                tokens = s2s['predicted_tokens'].cpu().numpy()
                packed, validity = self.assembler.assemble_packed(tokens)
            scores = e.execute(packed, feat_c, s2s['word_vecs'])
        else:
            scores, validity = e.execute_tokens(s2s['predicted_tokens'], feat_c, s2s['word_vecs'])
            tokens = s2s['predicted_tokens']
        if use_qpn and self.dims.qpn_hidden > 0:
            self.add_question_prior(scores)
        if not (known or host_assemble) and fetch:
            tokens = tokens.cpu().numpy()
            validity = validity.cpu().numpy().astype(bool)
        return scores, tokens, validity


def unpad_variable(name: str, g: np.ndarray, d: VQADims, di: Dims) -> np.ndarray:
    """Inverse of pad_variable for a gradient / weight in the GPU context's padded shape."""
    L, Lp = d.lstm_dim, di.lstm_dim
    Dc = d.D + 2
    g = np.asarray(g)

    def gate_cols(x):
        return np.concatenate([x[..., k * Lp:k * Lp + L] for k in range(4)], axis=-1)

    if 'basic_lstm_cell/weights' in name:
        x = gate_cols(g)
        if 'cell_0' in name:
            E = g.shape[0] - Lp
            return x[:E + L]
        return np.concatenate([x[:L], x[Lp:Lp + L]], axis=0)
    if 'basic_lstm_cell/biases' in name:
        return gate_cols(g)
    if name.endswith('encoder_h_transform/weights') or name.endswith('att_prediction/weights'):
        return g[:L, :L]
    if name.endswith('encoder_h_transform/biases') or name.endswith('att_prediction/biases') \
            or name.endswith('att_prediction/v'):
        return g[:L]
    if name.endswith('token_prediction/weights') or name.endswith('question_prior_net/fc1/weights'):
        return np.concatenate([g[:L], g[Lp:Lp + L]], axis=0)
    if name.endswith('conv_image/weights') or name.endswith('fc_att/weights'):
        return g[:Dc]
    return g


class VQATrainer(Trainer):
    """One iteration of exp_vqa/train_vqa_gt_layout.py:148-190 on the GPU: models_vqa forward with
    encoder / decoder / question-prior dropout, total = mean(-log_seq_prob) + mean(CE), every
    gradient, Adam without clipping (:119-123) and without weight decay (:42)."""

    def __init__(self, vqa: VQAEngine, lr: float = 1e-3, weight_decay: float = 0.0, dist=None,
                 rccl=None, encoder_dropout: bool = True, decoder_dropout: bool = True,
                 qpn_dropout: bool = True, keep_prob: float = 0.5):
        super().__init__(vqa.engine, weight_decay=weight_decay, lr=lr, max_grad_l2_norm=0.0,
                         dist=dist, rccl=rccl)
        self.vqa = vqa
        self.dropout = dict(enc0=encoder_dropout, dec0=decoder_dropout, qpn_h=qpn_dropout,
                            qpn_fc1=qpn_dropout and vqa.dims.qpn_hidden > 0)
        self.keep_prob = float(keep_prob)
        self.masks = None            # dict of keep masks for the next step (tests); None: drawn
        self.seed = 0                # stream of n2nmn_dropout_multipliers; the offset advances
        self._drawn = 0
        self._mult = None
        self._reuse = None

    def _multipliers(self, T, N, Td):
        """Reference-shaped {0, 1} keep masks -> multipliers (0 or 1 / keep_prob) in the context's
        padded layout (padded hidden units are zero anyway; their multiplier is 0)."""
        import torch
        d, di, dev = self.vqa.dims, self.vqa.idims, self.engine.device
        L, Lp = d.lstm_dim, di.lstm_dim
        shapes = dict(enc0=(T, N, L), dec0=(Td, N, L), qpn_h=(N, d.num_layers * L),
                      qpn_fc1=(N, d.qpn_hidden))
        out = {}
        for key, on in self.dropout.items():
            if not on:
                continue
            if self.masks is not None:
                keep = torch.as_tensor(np.asarray(self.masks[key], np.float32), device=dev)
                if tuple(keep.shape) != shapes[key]:
                    raise ValueError('dropout mask %s: shape %s, expected %s' %
                                     (key, tuple(keep.shape), shapes[key]))
                m = keep / self.keep_prob
            else:
                m = torch.empty(shapes[key], dtype=torch.float32, device=dev)
                _lib.check(self._lib.n2nmn_dropout_multipliers(
                    m.data_ptr(), m.numel(), self.keep_prob, self.seed, self._drawn,
                    self.engine.stream()))
                self._drawn += m.numel()
            if key in ('enc0', 'dec0'):
                p = torch.zeros(shapes[key][:2] + (Lp,), device=dev)
                p[..., :L] = m
            elif key == 'qpn_h':
                p = torch.zeros((N, 2 * Lp), device=dev)
                p[:, :L] = m[:, :L]
                p[:, Lp:Lp + L] = m[:, L:]
            else:
                p = m
            out[key] = p.contiguous()
        return out

    def _io(self, batch, gt_layout, objective: int = 0):
        b = dict(batch)
        b['image_feat_batch'] = self.vqa.features_with_coords(batch['image_feat_batch'])
        io, packed, validity = super()._io(b, gt_layout, objective)
        if self._reuse is None:
            self._mult = self._multipliers(io.T_enc, io.N, io.T_dec)
        else:                        # step_rl: the masks the layout was sampled under
            self._mult, self._reuse = self._reuse, None
        for key, field in (('enc0', 'drop_enc0'), ('dec0', 'drop_dec0'), ('qpn_h', 'drop_qpn_h'),
                           ('qpn_fc1', 'drop_qpn_fc1')):
            if key in self._mult:
                setattr(io, field, self._mult[key].data_ptr())
        return io, packed, validity

    def step_rl(self, batch, sample_uniforms, lr=1e-4, update: bool = True):
        """One iteration of exp_vqa/train_vqa_rl_gt_layout.py:150-196: the layout is sampled from the
        network WITH this step's dropout masks (the reference samples inside the training graph),
        fetched and assembled on the host, then the policy-gradient loss (:106-126, every layout
        counted as valid like :112) is differentiated under the same masks."""
        import torch
        e = self.engine
        seq = torch.as_tensor(batch['input_seq_batch'])
        T, N = seq.shape
        mult = self._multipliers(T, N, self.vqa.dims.T_decoder)
        s2s = e.seq2seq(batch['input_seq_batch'], batch['seq_length_batch'], self.vqa.dims.T_decoder,
                        sample_uniforms=sample_uniforms,
                        dropout=(mult.get('enc0'), mult.get('dec0')))
        tokens = s2s['predicted_tokens'].cpu().numpy()
        self._reuse = mult
        scale = self.forward_backward(batch, tokens, objective=1)
        if update:
            saved = self.hyper['lr']
            self.hyper['lr'] = saved if lr is None else float(lr)
            self.apply(scale)
            self.hyper['lr'] = saved
        validity = self._keep[6].cpu().numpy().astype(bool) if self._keep[6] is not None else None
        return self.losses, tokens, validity

    def gradients_reference_shaped(self):
        """name -> numpy gradient in the reference's (unpadded) variable shape."""
        d, di = self.vqa.dims, self.vqa.idims
        return {k: unpad_variable(k, v.detach().cpu().numpy(), d, di)
                for k, v in self.gradients().items()}

    def weights_reference_shaped(self):
        d, di = self.vqa.dims, self.vqa.idims
        return {k: unpad_variable(k, v.cpu().numpy(), d, di) for k, v in self.get_weights().items()}
