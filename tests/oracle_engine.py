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
