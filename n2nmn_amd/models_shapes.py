"""Python face of the reference's `models_shapes` package for exp_shapes/eval_shapes.py (BASELINE.json
configs[0], the reference's own CPU-runnable plumbing case): the names, constructor arguments and attributes
that script uses (exp_shapes/eval_shapes.py:23-24,73,122-132,138-175), over the HIP engine.

    from models_shapes.nmn3_assembler import Assembler  -> n2nmn_amd.models_shapes.Assembler
    from models_shapes.nmn3_model import NMN3ModelAtt   -> n2nmn_amd.models_shapes.NMN3ModelAtt

    NMN3ModelAtt(image_batch, text_seq_batch, seq_length_batch, T_decoder, num_vocab_txt, embed_dim_txt,
                 num_vocab_nmn, embed_dim_nmn, lstm_dim, num_layers, EOS_idx, encoder_dropout,
                 decoder_dropout, decoder_sampling, num_choices, use_gt_layout=None, gt_layout_batch=None,
                 scope='neural_module_network', reuse=None)             (models_shapes/nmn3_model.py:15-21)
    .predicted_tokens .word_vecs .atts       phase 1 (:29-50)        .compiler .scores      phase 2 (:52-100)

What models_shapes is made of, and where it runs here:
  * shapes_convnet (models_shapes/shapes_convnet.py:8-17): a 10x10 stride-10 VALID convolution + ReLU and a
    1x1 convolution + ReLU = two GEMMs over im2col rows -> n2nmn_fc_forward (the reshape / permute that
    builds the rows is data movement, done on the torch tensor);
  * the layout generator (models_shapes/nmn3_netgen_att.py): the CLEVR encoder / decoder kernels at lstm_dim
    256 WITHOUT a validity automaton (all-zero tables: every token valid at every step) and with the
    reference's <eos> latch (:215-222: every token behind the first <eos> is <eos>) applied to the fetched
    tokens -- the tokens in front of it, which are all the assembler reads, do not depend on it;
  * the modules (models_shapes/nmn3_modules.py:27-144): Find at map_dim 500, Transform with a 3x3 kernel, And,
    Answer = fc([min, mean, max]) = the CLEVR operators' kernels (Answer has ExistModule's arithmetic).
token_probs / neg_entropy / log_seq_prob of models_shapes follow the latch too (probability 1, entropy 0 behind
the <eos>) and are not offered by this inference face.
"""
from __future__ import annotations

import numpy as np

from .engine import Engine, _torch
from .nmn3_assembler import Assembler as _Assembler
from .nmn3_model import Compiler
from .runtime import Fetch, register_model, resolve, to_numpy
from .spec import Dims, variable_shapes

SHAPES_MODULE_NAMES = ('_Find', '_Transform', '_And', '_Answer', '<eos>')   # exp_shapes/data/vocabulary_layout.txt
# module -> C-ABI operator; _Answer = fc([min, mean, max]) is ExistModule's arithmetic (nmn3_modules.py:122-144)
SHAPES_OP_CODE = {'_Find': 1, '_Transform': 4, '_And': 5, '_Filter': 2, '_Answer': 7}
SHAPES_INPUT_NUM = {'_Find': 0, '_Transform': 1, '_And': 2, '_Filter': 1, '_Answer': 1}   # nmn3_assembler.py:9-18
SHAPES_OUTPUT_TYPE = {'_Find': 'att', '_Transform': 'att', '_And': 'att', '_Filter': 'att', '_Answer': 'ans'}
_P = 'neural_module_network/'
_CNN = _P + 'image_feature_cnn/shapes_convnet/'
_MOD = _P + 'layout_execution/'


class Assembler(_Assembler):
    """models_shapes/nmn3_assembler.py: Assembler(module_vocab_file)"""

    def __init__(self, module_vocab_file):
        super().__init__(module_vocab_file, op_code=SHAPES_OP_CODE, input_num=SHAPES_INPUT_NUM,
                         output_type=SHAPES_OUTPUT_TYPE)


class NMN3ModelAtt:
    def __init__(self, image_batch, text_seq_batch, seq_length_batch, T_decoder, num_vocab_txt, embed_dim_txt,
                 num_vocab_nmn, embed_dim_nmn, lstm_dim, num_layers, EOS_idx, encoder_dropout,
                 decoder_dropout, decoder_sampling, num_choices, use_gt_layout=None, gt_layout_batch=None,
                 scope='neural_module_network', reuse=None, engine: Engine = None, device: int = 0,
                 max_batch: int = 256, T_encoder: int = 15, map_dim: int = 500, feat_dim: int = 64,
                 hidden_dim: int = 64, kernel_size: int = 3):
        if encoder_dropout or decoder_dropout or decoder_sampling:
            raise NotImplementedError('the inference face takes no dropout / sampling (exp_shapes/eval_shapes.py:33-35)')
        names = list(SHAPES_MODULE_NAMES)
        if num_vocab_nmn != len(names) or EOS_idx != names.index('<eos>'):
            raise ValueError('models_shapes layout vocabulary is %r' % (names,))
        shp = getattr(image_batch, 'shape', None)
        if shp is None or len(shp) != 4 or any(s is None for s in shp[1:]):
            raise ValueError('image_batch needs a static [N, H_im, W_im, 3] shape')
        self.H_im, self.W_im, self.C_im = int(shp[1]), int(shp[2]), int(shp[3])
        self.stride = 10                                       # shapes_convnet: kernel 10, stride 10, VALID
        if self.H_im % self.stride or self.W_im % self.stride:
            raise ValueError('image size must be a multiple of the 10x10 stride-10 kernel')
        self.assembler = _Assembler(names, op_code=SHAPES_OP_CODE, input_num=SHAPES_INPUT_NUM,
                                    output_type=SHAPES_OUTPUT_TYPE)
        if engine is None:
            dims = Dims(H=self.H_im // self.stride, W=self.W_im // self.stride, D=feat_dim, map_dim=map_dim,
                        embed_dim_txt=embed_dim_txt, embed_dim_nmn=embed_dim_nmn, lstm_dim=lstm_dim,
                        num_layers=num_layers, num_vocab_txt=num_vocab_txt, num_vocab_nmn=num_vocab_nmn,
                        num_choices=num_choices, T_encoder=T_encoder, T_decoder=T_decoder, N=max_batch,
                        kernel_size=kernel_size)
            engine = Engine(dims, self.assembler, device)
            # models_shapes' decoder has no validity automaton: every token is valid at every step
            V = num_vocab_nmn
            engine.set_validity_tables(np.zeros((V, 3), np.int32), np.zeros((3, V, 4), np.int32),
                                       np.zeros((V, 4), np.int32))
        elif not getattr(engine, 'all_tokens_valid', False):
            # (a caller's engine keeps the tables it was given: installing the all-valid tables behind its back
            # would change what its other users decode)
            raise ValueError('models_shapes decodes without a validity automaton: give it an engine whose '
                             'set_validity_tables() installed all-zero tables, or let it build its own')
        self.engine = engine
        self.EOS_idx = EOS_idx
        self.T_decoder = T_decoder
        self.hidden_dim = hidden_dim
        self.image_batch = image_batch
        self._inputs = dict(input_seq=text_seq_batch, seq_len=seq_length_batch, use_gt_layout=use_gt_layout,
                            gt_layout=gt_layout_batch)
        self._cnn = None
        for name in ('predicted_tokens', 'word_vecs', 'atts'):
            setattr(self, name, Fetch(self, name, 1))
        self.image_feat_grid = Fetch(self, 'image_feat_grid', 1)
        self.compiler = Compiler(self.assembler)
        self.scores = Fetch(self, 'scores', 2)
        register_model(self)

    # -- variables -------------------------------------------------------------------------------------
    def load_weights(self, weights):
        """reference-named variables of the models_shapes graph (oracle/n2nmn_oracle_shapes.variable_shapes
        lists them): the convnet's stay with the face, the seq2seq's go to the engine under their own names,
        the modules' under the engine's (<X>Module/<X>Module/... of the ScopedLayer naming ->
        module_variables/<X>Module/...; AnswerModule -> ExistModule); CLEVR variables models_shapes does not
        have are zero."""
        w = {k: (v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)) for k, v in weights.items()}
        self._cnn = {k: np.ascontiguousarray(w[_CNN + k], np.float32)
                     for k in ('conv_1/weights', 'conv_1/biases', 'conv_2/weights', 'conv_2/biases')}
        out = {k: np.zeros(shp, np.float32) for k, shp in variable_shapes(self.engine.dims).items()}
        for k, v in w.items():
            if k.startswith(_MOD):
                name = k[len(_MOD):].split('/', 1)[1].replace('AnswerModule/', 'ExistModule/')
                k2 = _MOD + 'module_variables/' + name
            elif k.startswith(_CNN):
                continue
            else:
                k2 = k
            if k2 in out:
                if tuple(out[k2].shape) != tuple(v.shape):
                    raise ValueError('shape mismatch for %s: %s vs %s' % (k, v.shape, out[k2].shape))
                out[k2] = np.ascontiguousarray(v, np.float32)
        self.engine.load_weights(out)

    # -- eager execution -------------------------------------------------------------------------------
    def convnet(self, images):
        """shapes_convnet: [N, H_im, W_im, 3] (mean-subtracted) -> device tensor [N, H, W, feat_dim]"""
        torch = _torch()
        if self._cnn is None:
            raise RuntimeError('NMN3ModelAtt: weights not loaded')
        e, s = self.engine, self.stride
        x = e._dev(images, torch.float32)
        n, H, W = x.shape[0], self.H_im // s, self.W_im // s
        rows = x.reshape(n, H, s, W, s, self.C_im).permute(0, 1, 3, 2, 4, 5).reshape(n * H * W, s * s * self.C_im)
        k1 = self._cnn['conv_1/weights'].reshape(s * s * self.C_im, -1)
        c1 = e.fc(rows.contiguous(), k1, self._cnn['conv_1/biases'], relu=True)
        k2 = self._cnn['conv_2/weights']
        c2 = e.fc(c1, k2.reshape(k2.shape[2], k2.shape[3]), self._cnn['conv_2/biases'], relu=True)
        return c2.reshape(n, H, W, -1)

    def run_phase1(self, feeds=None):
        feeds = feeds or {}
        seq = resolve(self._inputs['input_seq'], feeds)
        lens = resolve(self._inputs['seq_len'], feeds)
        use_gt = self._inputs['use_gt_layout']
        use_gt = bool(resolve(use_gt, feeds)) if use_gt is not None else False
        gt = self._inputs['gt_layout']
        gt = resolve(gt, feeds) if (gt is not None and use_gt) else None
        out = dict(self.engine.seq2seq(seq, lens, self.T_decoder, use_gt, gt))
        # the <eos> latch of models_shapes/nmn3_netgen_att.py:215-222: behind the first <eos> every token is <eos>
        tok = to_numpy(out['predicted_tokens']).copy()
        seen = np.cumsum(tok == self.EOS_idx, axis=0) > 0
        behind = np.vstack([np.zeros((1, tok.shape[1]), bool), seen[:-1]])
        tok[behind] = self.EOS_idx
        out['predicted_tokens'] = tok
        out['image_feat_grid'] = self.convnet(resolve(self.image_batch, feeds))
        return out

    def run_phase2(self, packed, feat, word_vecs):
        return self.engine.execute(packed, feat, word_vecs)

    def _fetch(self, f, handle):
        if handle.phase1 is None:
            handle.phase1 = self.run_phase1(handle.feeds)
        if f.phase == 1:
            return to_numpy(handle.phase1[f.name])
        if 'scores' not in handle.results:
            packed = resolve(self.compiler.loom_input_tensor, handle.feeds)
            handle.results['scores'] = self.run_phase2(packed, handle.phase1['image_feat_grid'],
                                                       handle.phase1['word_vecs'])
        return to_numpy(handle.results['scores'])


__all__ = ['Assembler', 'NMN3ModelAtt', 'SHAPES_MODULE_NAMES']
