"""TEST DOUBLE (tests/ only): an object with the slice of n2nmn_amd.engine.Engine's interface that
NMN3Model / AttentionSeq2Seq / Modules use, computing with the CPU oracle instead of the HIP library.

It exists for ONE purpose: tests/test_reference_driver_source.py executes the reference's own
exp_clevr/eval_clevr.py, unmodified, against the drop-in's Python face on a box without a GPU (the
reference checkout exists only here, the GPU only on the gpurun box), to prove that every name,
argument and attribute the driver uses binds.  The product never imports this file; on the GPU the
same loop is tests/test_gpu_end2end.py::test_reference_shaped_session_loop."""
import ctypes as C

import numpy as np
import torch

from oracle import n2nmn_oracle as O
from n2nmn_amd import _lib
from n2nmn_amd.nmn3_assembler import OP_CODE
from n2nmn_amd.spec import Dims

_OP_NAME = {v: k for k, v in OP_CODE.items()}


class OracleEngine:
    def __init__(self, dims: Dims, assembler, device=0, _parent=None):
        self.dims, self.assembler = dims, assembler
        self.device = torch.device('cpu')
        self._parent = None
        self.weights = None
        self.calls = dict(seq2seq=0, execute=0)

    def load_weights(self, weights, strict=True):
        self.weights = {k: np.asarray(v, np.float64) for k, v in weights.items()}

    def variable_names(self):
        from n2nmn_amd.spec import variable_shapes
        return dict(variable_shapes(self.dims))

    def get_weights(self):
        return {k: np.asarray(v, np.float32) for k, v in self.weights.items()}

    def seq2seq(self, input_seq, seq_len, T_dec=None, use_gt_layout=False, gt_layout=None,
                sample_uniforms=None, forced_tokens=None, debug=False, **kw):
        self.calls['seq2seq'] += 1
        seq, lens = np.asarray(input_seq, np.int32), np.asarray(seq_len, np.int32)
        a = self.assembler
        enc = O.encoder_forward(self.weights, seq, lens, np.float64)
        dec = O.decoder_forward(self.weights, enc, a.P, a.W, a.b, T_dec or self.dims.T_decoder, np.float64,
                                use_gt_layout=use_gt_layout, gt_layout=gt_layout,
                                sample_uniforms=sample_uniforms, forced_tokens=forced_tokens)
        out = {k: torch.as_tensor(np.asarray(dec[k])) for k in
               ('predicted_tokens', 'token_probs', 'neg_entropy', 'word_vecs')}
        out['atts'] = torch.as_tensor(dec['atts'][..., 0])
        out['log_seq_prob'] = torch.as_tensor(np.sum(np.log(dec['token_probs']), axis=0))
        out['predicted_tokens'] = out['predicted_tokens'].to(torch.int32)
        return out

    def execute(self, packed, image_feat, word_vecs, reuse_buffers=True):
        self.calls['execute'] += 1
        n = packed.num_nodes
        nodes = (_lib.Node * max(n, 1))()
        _lib.check(_lib.lib().n2nmn_program_get_nodes(packed.handle, nodes, n))
        feat = np.asarray(image_feat, np.float64)
        wv = np.asarray(word_vecs, np.float64)
        scores = np.zeros((packed.num_rows, self.dims.num_choices))

        def expr(i):
            nd = nodes[i]
            e = dict(module=_OP_NAME[nd.op], time_idx=nd.time_idx, batch_idx=nd.batch_idx)
            if nd.in0 >= 0:
                e['input_0'] = expr(nd.in0)
            if nd.in1 >= 0:
                e['input_1'] = expr(nd.in1)
            return e
        for i in range(n):
            if nodes[i].out_row >= 0:
                scores[nodes[i].out_row] = O.eval_expr(self.weights, expr(i), feat, wv,
                                                       self.dims.num_choices, np.float64)
        return torch.as_tensor(scores)


# ---- the training step behind n2nmn_amd.runtime_train.TrainStep (exp_clevr/train_clevr*_gt_layout.py) -------
class OracleTrainer:
    """n2nmn_amd.train.Trainer's interface as runtime_train.TrainStep uses it, computing with the fp64 autograd
    oracle (oracle/n2nmn_oracle_grad.py): forward_backward -> losses / scores / gradients, apply -> per-tensor
    clip-by-norm + Adam on the OracleEngine's weights."""

    made = []            # every trainer built (tests look at the hyper-parameters the script's graph produced)

    def __init__(self, engine, weight_decay=5e-6, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                 max_grad_l2_norm=10.0, dist=None, rccl=None):
        from oracle import n2nmn_oracle_grad as G
        self.G = G
        self.engine = engine
        self.weight_decay = weight_decay
        self.hyper = dict(lr=lr, beta1=beta1, beta2=beta2, eps=eps, max_grad_l2_norm=max_grad_l2_norm)
        self.rl = dict(invalid_expr_loss=0.5, lambda_entropy=0.005, baseline_decay=0.99)
        self._baseline = self.rl['invalid_expr_loss']
        self.iteration = 0
        self.m = self.v = None
        self.grads = None
        self.losses = np.zeros(8, np.float32)
        self.scores = None
        self.last_validity = None
        self.history = []          # (objective, losses dict) per step
        OracleTrainer.made.append(self)

    def set_baseline(self, value):
        self._baseline = float(value)

    def get_baseline(self):
        return float(self._baseline)

    def _token_validity(self, tokens, T_dec):
        """[T_dec, N, V] masks of the automaton along the given token sequences (nmn3_netgen_att.py:8-15,200-203)"""
        a = self.engine.assembler
        Td, N = tokens.shape
        X = np.tile(np.array([[0, 0, T_dec]], np.int64), (N, 1))
        out = np.zeros((Td, N, a.P.shape[0]), bool)
        for t in range(Td):
            out[t] = O.valid_tokens(X, a.W, a.b)
            X = X + a.P[tokens[t]]
        return out

    def forward_backward(self, batch, gt_layout, reduce=True, objective=0):
        G, e = self.G, self.engine
        tokens = np.asarray(gt_layout, np.int32)
        names = list(e.assembler.module_names)
        b = {k: np.asarray(v) for k, v in batch.items()}
        if objective == 0:
            L, g, ex = G.loss_and_grads(e.weights, names, b, tokens.shape[0], e.dims.num_choices, tokens,
                                        weight_decay=self.weight_decay)
            self.losses[:4] = [L['avg_sample_loss'], L['seq_likelihood_loss'], L['l2_reg'], L['total_loss']]
            self.last_validity = np.ones(tokens.shape[1], bool)
        else:
            tv = self._token_validity(tokens, tokens.shape[0])
            L, g, ex = G.loss_and_grads_rl(e.weights, names, b, tokens.shape[0], e.dims.num_choices, tokens, tv,
                                           self._baseline, self.rl['invalid_expr_loss'], self.rl['lambda_entropy'],
                                           self.weight_decay, self.rl['baseline_decay'])
            self.losses[:5] = [L['avg_sample_loss'], L['policy_gradient_loss'], L['l2_reg'], L['total_loss'],
                               L['entropy_reg']]
            self._baseline = L['new_baseline']
            self.last_validity = ex['validity']
        self.grads, self.scores = g, ex['scores']
        self.history.append((objective, dict(L)))
        return 1.0

    def apply(self, scale=1.0):
        """per-tensor tf.clip_by_norm, then tf.train.AdamOptimizer's update (TF 1.0.0: lr_t = lr sqrt(1 - b2^t) /
        (1 - b1^t); w -= lr_t m / (sqrt(v) + eps)) -- in torch fp64 (oracle_grad.adam_step is the numpy statement
        of the same step; tests/test_reference_train_driver_source.py checks one against the other)"""
        e, h = self.engine, self.hyper
        self.iteration += 1
        t = self.iteration
        if self.m is None:
            self.m = {k: torch.zeros(v.shape, dtype=torch.float64) for k, v in e.weights.items()}
            self.v = {k: torch.zeros(v.shape, dtype=torch.float64) for k, v in e.weights.items()}
        clip = h['max_grad_l2_norm'] if h['max_grad_l2_norm'] and h['max_grad_l2_norm'] > 0 else None
        lr_t = h['lr'] * np.sqrt(1.0 - h['beta2'] ** t) / (1.0 - h['beta1'] ** t)
        new = {}
        for k, w in e.weights.items():
            g = torch.as_tensor(np.asarray(self.grads[k], np.float64)) * scale
            if clip is not None:
                g = g * (clip / max(float(torch.linalg.vector_norm(g)), clip))
            self.m[k].mul_(h['beta1']).add_(g, alpha=1.0 - h['beta1'])
            self.v[k].mul_(h['beta2']).addcmul_(g, g, value=1.0 - h['beta2'])
            new[k] = (torch.as_tensor(np.asarray(w, np.float64)) - lr_t * self.m[k] / (self.v[k].sqrt() + h['eps'])).numpy()
        e.weights = new


# ---- models_vqa: the double behind n2nmn_amd.models_vqa.NMN3Model (exp_vqa/eval_vqa2.py) -------------------
class _OracleVQASeq2Seq:
    """the slice of n2nmn_amd.engine.Engine the models_vqa face uses: seq2seq / execute, with the VQA oracle"""

    def __init__(self, owner):
        self.o = owner
        self.device = torch.device('cpu')

    def seq2seq(self, input_seq, seq_len, T_dec=None, use_gt_layout=False, gt_layout=None, sample_uniforms=None,
                forced_tokens=None, dropout=None, **kw):
        """dropout: (layer-0 output multipliers of the encoder, of the decoder) as VQATrainer._multipliers makes them
        (0 or 1 / keep_prob): the training face's phase 1"""
        o = self.o
        o.calls['seq2seq'] += 1
        seq, lens = np.asarray(input_seq, np.int32), np.asarray(seq_len, np.int32)
        P, W, b = O.build_validity_mats(list(O.VQA_MODULE_NAMES))
        keep = [None if m is None else (np.asarray(m) > 0).astype(np.float64) for m in (dropout or (None, None))]
        o.enc = O.encoder_forward(o.weights, seq, lens, np.float64, drop0=keep[0])
        dec = O.decoder_forward(o.weights, o.enc, P, W, b, T_dec or o.dims.T_decoder, np.float64,
                                use_gt_layout=use_gt_layout, gt_layout=None if gt_layout is None else np.asarray(gt_layout),
                                sample_uniforms=None if sample_uniforms is None else np.asarray(sample_uniforms),
                                forced_tokens=forced_tokens, drop0=keep[1])
        out = {k: torch.as_tensor(np.asarray(dec[k])) for k in
               ('predicted_tokens', 'token_probs', 'neg_entropy', 'word_vecs')}
        out['atts'] = torch.as_tensor(dec['atts'][..., 0])
        out['log_seq_prob'] = torch.as_tensor(np.sum(np.log(dec['token_probs']), axis=0))
        out['predicted_tokens'] = out['predicted_tokens'].to(torch.int32)
        return out

    def execute(self, packed, feat_c, word_vecs, reuse_buffers=True):
        o = self.o
        o.calls['execute'] += 1
        n = packed.num_nodes
        nodes = (_lib.Node * max(n, 1))()
        _lib.check(_lib.lib().n2nmn_program_get_nodes(packed.handle, nodes, n))
        feat = np.asarray(feat_c, np.float64)
        wv = np.asarray(word_vecs, np.float64)
        names = {1: '_Find', 3: '_Transform', 5: '_And', 13: '_Describe'}       # n2nmn_amd.vqa.VQA_OP_CODE

        def expr(i):
            nd = nodes[i]
            e = dict(module=names[nd.op], time_idx=nd.time_idx, batch_idx=nd.batch_idx)
            if nd.in0 >= 0:
                e['input_0'] = expr(nd.in0)
            if nd.in1 >= 0:
                e['input_1'] = expr(nd.in1)
            return e
        scores = np.zeros((packed.num_rows, o.dims.num_choices))
        for i in range(n):
            if nodes[i].out_row >= 0:
                scores[nodes[i].out_row] = O.eval_expr_vqa(o.weights, expr(i), feat, wv, o.dims.num_choices,
                                                           np.float64)
        return torch.as_tensor(scores)


class OracleVQAEngine:
    """n2nmn_amd.vqa.VQAEngine's interface as the models_vqa face uses it"""

    def __init__(self, dims, device=0):
        from n2nmn_amd.nmn3_assembler import Assembler
        from n2nmn_amd.vqa import VQA_MODULE_NAMES, VQA_OP_CODE
        assert VQA_OP_CODE == {'_Find': 1, '_Transform': 3, '_And': 5, '_Describe': 13}
        self.assembler = Assembler(list(VQA_MODULE_NAMES), op_code=VQA_OP_CODE)
        self.dims = dims
        self.engine = _OracleVQASeq2Seq(self)
        self.weights = None
        self.enc = None
        self.calls = dict(seq2seq=0, execute=0)

    def load_weights(self, weights):
        self.weights = {k: np.asarray(v, np.float64) for k, v in weights.items()}

    def weights_reference_shaped(self):
        return dict(self.weights)

    def features_with_coords(self, image_feat):
        return O.add_spatial_coordinate_map(np.asarray(image_feat, np.float64))

    def add_question_prior(self, scores):
        return scores + torch.as_tensor(O.question_prior_net(self.weights, self.enc['states']))


class OracleVQATrainer(OracleTrainer):
    """n2nmn_amd.vqa.VQATrainer's interface as the models_vqa face and runtime_train.TrainStep use it
    (exp_vqa/train_vqa2_gt_layout.py / train_vqa2_rl_gt_layout.py): dropout masks from the face, fp64 autograd of
    oracle/n2nmn_oracle_grad.py (loss_and_grads_vqa; loss_and_grads_rl(vqa_masks=...)), Adam without clipping unless
    the graph clips."""

    def __init__(self, vqa, lr=1e-3, weight_decay=0.0, dist=None, rccl=None, encoder_dropout=True,
                 decoder_dropout=True, qpn_dropout=True, keep_prob=0.5):
        super().__init__(vqa, weight_decay=weight_decay, lr=lr, max_grad_l2_norm=0.0)
        self.vqa = vqa
        self.dropout = dict(enc0=encoder_dropout, dec0=decoder_dropout, qpn_h=qpn_dropout,
                            qpn_fc1=qpn_dropout and vqa.dims.qpn_hidden > 0)
        assert keep_prob == 0.5
        self.masks = None
        self._reuse = None
        self.mask_history = []       # the keep masks of every step (the recording stores their checksums)

    def _multipliers(self, T, N, Td):
        assert self.masks is not None, 'the face hands the masks of a handle to the trainer'
        return {k: np.asarray(v, np.float64) * 2.0 for k, v in self.masks.items() if self.dropout.get(k)}

    def forward_backward(self, batch, gt_layout, reduce=True, objective=0):
        G, e = self.G, self.engine
        tokens = np.asarray(gt_layout, np.int32)
        mult, self._reuse = self._reuse, None
        assert mult is not None, 'phase 1 of the handle draws the masks (models_vqa.NMN3Model.run_phase1_training)'
        masks = {k: (np.asarray(v) > 0).astype(np.float64) for k, v in mult.items()}
        self.mask_history.append(masks)
        b = {k: np.asarray(v) for k, v in batch.items()}
        C = e.dims.num_choices
        if objective == 0:
            L, g, ex = G.loss_and_grads_vqa(e.weights, b, tokens.shape[0], C, tokens, masks or None,
                                            weight_decay=self.weight_decay)
            self.losses[:4] = [L['avg_sample_loss'], L['seq_likelihood_loss'], L['l2_reg'], L['total_loss']]
            self.last_validity = np.ones(tokens.shape[1], bool)
        else:
            tv = self._token_validity(tokens, tokens.shape[0])
            L, g, ex = G.loss_and_grads_rl(e.weights, list(O.VQA_MODULE_NAMES), b, tokens.shape[0], C, tokens, tv,
                                           self._baseline, self.rl['invalid_expr_loss'], self.rl['lambda_entropy'],
                                           self.weight_decay, self.rl['baseline_decay'], vqa_masks=masks or None)
            self.losses[:5] = [L['avg_sample_loss'], L['policy_gradient_loss'], L['l2_reg'], L['total_loss'],
                               L['entropy_reg']]
            self._baseline = L['new_baseline']
            self.last_validity = ex['validity']
        self.grads, self.scores = g, ex['scores']
        self.history.append((objective, dict(L)))
        return 1.0


# ---- models_shapes: the double behind n2nmn_amd.models_shapes.NMN3ModelAtt (exp_shapes/eval_shapes.py) ------
class OracleShapesEngine:
    """the slice of Engine the models_shapes face uses (seq2seq, execute, fc, set_validity_tables,
    load_weights with ENGINE-side names), computing with oracle/n2nmn_oracle_shapes.py"""

    def __init__(self, dims, assembler, device=0, _parent=None):
        from oracle import n2nmn_oracle_shapes as S
        self.S = S
        self.dims, self.assembler = dims, assembler
        self.device = torch.device('cpu')
        self.weights = None
        self.calls = dict(seq2seq=0, execute=0, fc=0)
        self.tables = None

    def set_validity_tables(self, P, W, b):
        self.tables = (np.asarray(P), np.asarray(W), np.asarray(b))

    def _dev(self, x, dtype):
        return None if x is None else torch.as_tensor(np.asarray(x)).to(dtype)

    def load_weights(self, weights, strict=True):
        # engine-side names -> the models_shapes names the oracle reads
        S, w = self.S, {}
        mv = S._MOD + 'module_variables/'
        for k, v in weights.items():
            v = np.asarray(v, np.float64)
            if k.startswith(mv):
                scope, rest = k[len(mv):].split('/', 1)
                scope = scope.replace('ExistModule', 'AnswerModule')
                if scope in ('FindModule', 'TransformModule', 'AnswerModule'):
                    w[S._MOD + scope + '/' + scope + '/' + rest] = v
            else:
                w[k] = v
        self.weights = w

    def fc(self, A, W, bias=None, relu=False):
        self.calls['fc'] += 1
        out = np.asarray(A, np.float64) @ np.asarray(W, np.float64)
        if bias is not None:
            out = out + np.asarray(bias, np.float64)
        return torch.as_tensor(np.maximum(out, 0) if relu else out)

    def seq2seq(self, input_seq, seq_len, T_dec=None, use_gt_layout=False, gt_layout=None, *a, **kw):
        self.calls['seq2seq'] += 1
        assert self.tables is not None and not any(t.any() for t in self.tables), 'all-valid tables expected'
        S = self.S
        enc = O.encoder_forward(self.weights, np.asarray(input_seq, np.int32), np.asarray(seq_len, np.int32),
                                np.float64)
        dec = S.decoder_forward(self.weights, enc, T_dec, self.assembler.EOS_idx, np.float64, use_gt_layout,
                                gt_layout)
        out = dict(predicted_tokens=torch.as_tensor(dec['predicted_tokens']).to(torch.int32),
                   word_vecs=torch.as_tensor(dec['word_vecs']), atts=torch.as_tensor(dec['atts'][..., 0]))
        return out

    def execute(self, packed, image_feat, word_vecs, reuse_buffers=True):
        self.calls['execute'] += 1
        S = self.S
        n = packed.num_nodes
        nodes = (_lib.Node * max(n, 1))()
        _lib.check(_lib.lib().n2nmn_program_get_nodes(packed.handle, nodes, n))
        names = {1: '_Find', 4: '_Transform', 5: '_And', 7: '_Answer'}
        feat = np.asarray(image_feat, np.float64)
        wv = np.asarray(word_vecs, np.float64)
        mw = S._mw(self.weights)

        def expr(i):
            nd = nodes[i]
            e = dict(module=names[nd.op], time_idx=nd.time_idx, batch_idx=nd.batch_idx)
            if nd.in0 >= 0:
                e['input_0'] = expr(nd.in0)
            if nd.in1 >= 0:
                e['input_1'] = expr(nd.in1)
            return e
        scores = np.zeros((packed.num_rows, self.dims.num_choices))
        for i in range(n):
            if nodes[i].out_row >= 0:
                scores[nodes[i].out_row] = S.eval_expr(mw, expr(i), feat, wv, self.dims.num_choices, np.float64)
        return torch.as_tensor(scores)
