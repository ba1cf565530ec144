"""Model wiring with the reference's constructor and attributes (models_clevr/nmn3_model.py:15-166):

    NMN3Model(image_feat_grid, text_seq_batch, seq_length_batch, T_decoder, num_vocab_txt,
              embed_dim_txt, num_vocab_nmn, embed_dim_nmn, lstm_dim, num_layers, assembler,
              encoder_dropout, decoder_dropout, decoder_sampling, num_choices,
              use_gt_layout=None, gt_layout_batch=None, scope='neural_module_network', reuse=None)

    .predicted_tokens .token_probs .word_vecs .neg_entropy .atts .log_seq_prob   (phase 1)
    .modules .compiler .scores                                                   (phase 2)

`compiler.build_feed_dict(expr_list)` returns `{compiler.loom_input_tensor: <packed program>}`
exactly where TensorFlow-Fold returned serialized Loom inputs (exp_clevr/eval_clevr.py:128); the
second `partial_run` executes the packed program with the HIP module kernels.

Extra keyword `engine=` (or `device=` / `max_batch=` / `T_encoder=` to build one) selects the GPU
context that holds the weights; `model.load_weights(dict)` registers reference-named variables.
"""
from __future__ import annotations

from .engine import Engine
from .nmn3_assembler import PackedLayouts
from .nmn3_modules import Modules
from .nmn3_netgen_att import AttentionSeq2Seq, PHASE1_OUTPUTS
from .runtime import Fetch, Placeholder, register_model, resolve, to_numpy
from .spec import Dims, INVALID_EXPR


class Compiler:
    """The two members of td.Compiler the reference's loop uses (nmn3_model.py:158-159)."""

    def __init__(self, assembler):
        self._assembler = assembler
        self.loom_input_tensor = Placeholder(name='loom_input_tensor')

    def build_feed_dict(self, expr_list):
        if isinstance(expr_list, PackedLayouts):
            packed = expr_list
        else:
            packed = self._assembler.pack_expr_list(expr_list)
        return {self.loom_input_tensor: packed}


class NMN3Model:
    def __init__(self, image_feat_grid, text_seq_batch, seq_length_batch, T_decoder,
                 num_vocab_txt, embed_dim_txt, num_vocab_nmn, embed_dim_nmn, lstm_dim, num_layers,
                 assembler, encoder_dropout, decoder_dropout, decoder_sampling, num_choices,
                 use_gt_layout=None, gt_layout_batch=None, scope='neural_module_network',
                 reuse=None, engine: Engine = None, device: int = 0, max_batch: int = 64,
                 T_encoder: int = 45, map_dim: int = 250, kernel_size: int = 5):
        if engine is None:
            shp = getattr(image_feat_grid, 'shape', None)
            if shp is None or len(shp) != 4 or any(s is None for s in shp[1:]):
                raise ValueError('image_feat_grid needs a static [N,H,W,D] shape (placeholder or '
                                 'tensor) to size the engine')
            dims = Dims(H=int(shp[1]), W=int(shp[2]), D=int(shp[3]), map_dim=map_dim,
                        embed_dim_txt=embed_dim_txt, embed_dim_nmn=embed_dim_nmn,
                        lstm_dim=lstm_dim, num_layers=num_layers, num_vocab_txt=num_vocab_txt,
                        num_vocab_nmn=num_vocab_nmn, num_choices=num_choices,
                        T_encoder=T_encoder, T_decoder=T_decoder, N=max_batch,
                        kernel_size=kernel_size)
            engine = Engine(dims, assembler, device)
        self.engine = engine
        self.image_feat_grid = image_feat_grid
        self.att_seq2seq = AttentionSeq2Seq(
            text_seq_batch, seq_length_batch, T_decoder, num_vocab_txt, embed_dim_txt,
            num_vocab_nmn, embed_dim_nmn, lstm_dim, num_layers, assembler, encoder_dropout,
            decoder_dropout, decoder_sampling, use_gt_layout, gt_layout_batch, engine=engine)
        for name in PHASE1_OUTPUTS:
            setattr(self, name, Fetch(self, name, 1))
        self.modules = Modules(image_feat_grid, None, num_choices, engine=engine)
        self.compiler = Compiler(assembler)
        self.scores = Fetch(self, 'scores', 2)
        # nmn3_model.py:33-34,161-166: entropy_reg = mean(neg_entropy) (phase 1); l2_reg = sum of l2_loss over
        # the weight matrices (a term of the training loss: n2nmn_amd.runtime_train)
        self.entropy_reg = Fetch(self, 'entropy_reg', 1)
        self.l2_reg = Fetch(self, 'l2_reg', 2)
        self._has_weights = False
        register_model(self)

    def load_weights(self, weights):
        self.engine.load_weights(weights)
        self._has_weights = True

    def get_weights(self):
        return self.engine.get_weights()

    def variable_shapes(self):
        """reference variable name -> shape, in graph order (what tf.trainable_variables() lists)"""
        return self.engine.variable_names()

    def initialize_variables(self, seed: int = 0):
        """sess.run(tf.global_variables_initializer()): the reference's declared initial values
        (runtime_train.initial_weights)"""
        from .runtime_train import initial_weights
        self.load_weights(initial_weights(self.variable_shapes(), seed))

    # -- eager execution --------------------------------------------------------------------
    def run_phase1(self, feeds=None, **kw):
        return self.att_seq2seq.run(feeds, **kw)

    def run_phase2(self, packed, image_feat, word_vecs):
        return self.engine.execute(packed, image_feat, word_vecs)

    def _fetch(self, f, handle):
        if handle.phase1 is None:
            handle.phase1 = self.run_phase1(handle.feeds)
        if f.name == 'entropy_reg':
            import numpy as np
            return np.float32(np.mean(to_numpy(handle.phase1['neg_entropy']), dtype=np.float32))
        if f.phase == 1:
            return to_numpy(handle.phase1[f.name])
        if f.name == 'l2_reg':
            if 'l2_reg' not in handle.results:
                import numpy as np
                handle.results['l2_reg'] = np.float32(sum(
                    0.5 * float((to_numpy(v).astype(np.float64) ** 2).sum())
                    for k, v in self.get_weights().items() if k.endswith('/weights')))
            return handle.results['l2_reg']
        if 'scores' not in handle.results:
            packed = resolve(self.compiler.loom_input_tensor, handle.feeds)
            feat = resolve(self.image_feat_grid, handle.feeds)
            handle.results['scores'] = self.run_phase2(packed, feat, handle.phase1['word_vecs'])
        return to_numpy(handle.results['scores'])


__all__ = ['NMN3Model', 'Compiler', 'INVALID_EXPR']
