"""Python face of the reference's `models_vqa` package for the inference driver exp_vqa/eval_vqa2.py:
the names, constructor arguments and attributes that script uses (exp_vqa/eval_vqa2.py:23-25,56,72-91,
103-137), over the HIP engine of n2nmn_amd.vqa.

    from models_vqa.nmn3_assembler import Assembler   -> n2nmn_amd.models_vqa.Assembler
    from models_vqa.nmn3_model import NMN3Model       -> n2nmn_amd.models_vqa.NMN3Model
    from util.vqa_train.data_reader import DataReader -> n2nmn_amd.models_vqa.DataReader

    NMN3Model(image_feat_grid, text_seq_batch, seq_length_batch, T_decoder, num_vocab_txt, embed_dim_txt,
              num_vocab_nmn, embed_dim_nmn, lstm_dim, num_layers, assembler, encoder_dropout,
              decoder_dropout, decoder_sampling, num_choices, use_qpn, qpn_dropout,
              reduce_visfeat_dim=False, new_visfeat_dim=256, use_gt_layout=None, gt_layout_batch=None,
              scope='neural_module_network', reuse=None)                 (models_vqa/nmn3_model.py:15-22)
    .predicted_tokens .token_probs .word_vecs .neg_entropy .atts .log_seq_prob   phase 1 (:40-56)
    .compiler (.loom_input_tensor, .build_feed_dict) .scores                     phase 2 (:58-114)

`scores` = scores_nmn + scores_qpn when use_qpn (:106-114); the script's `scores_val[:, 0] = -1e10`
(eval_vqa2.py:137) is the script's.

Training (round 6): with the dropout switches / decoder_sampling set as exp_vqa/train_vqa2_gt_layout.py:31-40 and
train_vqa2_rl_gt_layout.py set them, the face serves the training drivers through n2nmn_amd.runtime_train: the loss
graph the script builds is matched onto n2nmn_amd.vqa.VQATrainer (`new_trainer`), `.entropy_reg` / `.l2_reg` /
`.log_seq_prob` are fetch handles, and ONE set of dropout keep masks is drawn per partial_run handle (host generator,
`dropout_seed`) -- phase 1 (the sampled layout, entropy_reg) and the training step see the same masks, as TensorFlow
evaluates a dropout op once per handle.
"""
from __future__ import annotations

import numpy as np

from .data_reader import DataReader as _DataReader
from .nmn3_assembler import Assembler as _Assembler
from .nmn3_model import Compiler
from .nmn3_netgen_att import PHASE1_OUTPUTS
from .runtime import Fetch, register_model, resolve, to_numpy
from .vqa import VQA_MODULE_NAMES, VQA_OP_CODE, VQADims, VQAEngine


class Assembler(_Assembler):
    """models_vqa/nmn3_assembler.py: Assembler(module_vocab_file) over the five-token VQA layout vocabulary"""

    def __init__(self, module_vocab_file):
        super().__init__(module_vocab_file, op_code=VQA_OP_CODE)


class DataReader(_DataReader):
    """util/vqa_train/data_reader.py: DataReader(imdb_file, shuffle=True, one_pass=False, prefetch_num=8,
    **data_params)"""

    def __init__(self, imdb_file, shuffle=True, one_pass=False, prefetch_num=8, **kwargs):
        kwargs.pop('variant', None)
        super().__init__(imdb_file, shuffle=shuffle, one_pass=one_pass, prefetch_num=prefetch_num,
                         variant='vqa', **kwargs)


class NMN3Model:
    def __init__(self, image_feat_grid, text_seq_batch, seq_length_batch, T_decoder, num_vocab_txt,
                 embed_dim_txt, num_vocab_nmn, embed_dim_nmn, lstm_dim, num_layers, assembler,
                 encoder_dropout, decoder_dropout, decoder_sampling, num_choices, use_qpn, qpn_dropout,
                 reduce_visfeat_dim=False, new_visfeat_dim=256, use_gt_layout=None, gt_layout_batch=None,
                 scope='neural_module_network', reuse=None, engine: VQAEngine = None, device: int = 0,
                 max_batch: int = 64, T_encoder: int = 26, map_dim: int = 1024, qpn_hidden: int = 500):
        if reduce_visfeat_dim:
            raise NotImplementedError('reduce_visfeat_dim=True (the extra 1x1 convolution of '
                                      'models_vqa/nmn3_model.py:27-34) is not part of the drop-in; '
                                      'exp_vqa/eval_vqa2.py runs with False')
        # dropout / sampling: the TRAINING drivers (train_vqa2_gt_layout.py:31-40, train_vqa2_rl_gt_layout.py:36); a
        # fetch outside a training handle runs without dropout (what the eval scripts build)
        self.dropout = dict(enc0=bool(encoder_dropout), dec0=bool(decoder_dropout), qpn=bool(qpn_dropout))
        self.decoder_sampling = bool(decoder_sampling)
        self.dropout_seed, self.sample_seed = 0, 0
        self._drop_gen = self._sample_gen = None
        self.dropout_masks = None            # tests: {'enc0','dec0','qpn_h','qpn_fc1'} keep masks for the NEXT handle
        self.last_masks = None
        if list(assembler.module_names) != list(VQA_MODULE_NAMES):
            raise ValueError('assembler vocabulary %r is not the models_vqa layout vocabulary %r' %
                             (assembler.module_names, list(VQA_MODULE_NAMES)))
        if engine is None:
            shp = getattr(image_feat_grid, 'shape', None)
            if shp is None or len(shp) != 4 or any(s is None for s in shp[1:]):
                raise ValueError('image_feat_grid needs a static [N,H,W,D] shape (placeholder or tensor) '
                                 'to size the engine')
            dims = VQADims(H=int(shp[1]), W=int(shp[2]), D=int(shp[3]), map_dim=map_dim,
                           embed_dim_txt=embed_dim_txt, embed_dim_nmn=embed_dim_nmn, lstm_dim=lstm_dim,
                           num_layers=num_layers, num_vocab_txt=num_vocab_txt, num_vocab_nmn=num_vocab_nmn,
                           num_choices=num_choices, T_encoder=T_encoder, T_decoder=T_decoder, N=max_batch,
                           qpn_hidden=qpn_hidden if use_qpn else 0)
            engine = VQAEngine(dims, device=device)
        d = engine.dims
        want = dict(num_vocab_txt=num_vocab_txt, embed_dim_txt=embed_dim_txt, num_vocab_nmn=num_vocab_nmn,
                    embed_dim_nmn=embed_dim_nmn, lstm_dim=lstm_dim, num_layers=num_layers,
                    num_choices=num_choices)
        for k, v in want.items():
            if getattr(d, k) != v:
                raise ValueError('%s=%r differs from the engine dims (%r)' % (k, v, getattr(d, k)))
        if T_decoder > d.T_decoder:
            raise ValueError('T_decoder exceeds the engine capacity')
        self.vqa = engine
        self.engine = engine.engine
        self.assembler = assembler
        self.T_decoder = T_decoder
        self.use_qpn = bool(use_qpn)
        self.qpn_dropout = qpn_dropout
        self.reduce_visfeat_dim = reduce_visfeat_dim
        self.image_feat_grid = image_feat_grid
        self._inputs = dict(input_seq=text_seq_batch, seq_len=seq_length_batch, use_gt_layout=use_gt_layout,
                            gt_layout=gt_layout_batch)
        for name in PHASE1_OUTPUTS:
            setattr(self, name, Fetch(self, name, 1))
        self.compiler = Compiler(assembler)
        self.scores = Fetch(self, 'scores', 2)
        self.entropy_reg = Fetch(self, 'entropy_reg', 1)       # models_vqa/nmn3_model.py:36-37
        self.l2_reg = Fetch(self, 'l2_reg', 2)                 # :116-120
        self._weights_ref = None
        register_model(self)

    def load_weights(self, weights):
        """reference-named, reference-shaped variables (n2nmn_amd.vqa.vqa_variable_shapes)"""
        self.vqa.load_weights(weights)

    def variable_shapes(self):
        from .vqa import vqa_variable_shapes
        return vqa_variable_shapes(self.vqa.dims)

    def get_weights(self):
        """reference-named, reference-shaped (unpadded) variables as they are on the device now"""
        if hasattr(self.vqa, 'weights_reference_shaped'):
            return self.vqa.weights_reference_shaped()
        from .vqa import unpad_variable
        d, di = self.vqa.dims, self.vqa.idims
        return {k: unpad_variable(k, to_numpy(v), d, di) for k, v in self.engine.get_weights().items()}

    def initialize_variables(self, seed: int = 0):
        from .runtime_train import initial_weights
        self.load_weights(initial_weights(self.variable_shapes(), seed))

    # -- training (n2nmn_amd.runtime_train) -------------------------------------------------------------
    def new_trainer(self, plan, op):
        """the Trainer behind a fetched `train_step`: VQATrainer with the graph's constants"""
        from . import vqa as _vqa
        h = op.optimizer.hyper
        tr = _vqa.VQATrainer(self.vqa, lr=h['lr'], weight_decay=plan.weight_decay,
                             encoder_dropout=self.dropout['enc0'], decoder_dropout=self.dropout['dec0'],
                             qpn_dropout=self.dropout['qpn'] and self.use_qpn)
        tr.hyper.update(beta1=h['beta1'], beta2=h['beta2'], eps=h['eps'],
                        max_grad_l2_norm=float(op.clip_norm) if op.clip_norm is not None else 0.0)
        return tr

    def _draw_masks(self, T, N):
        """{0, 1} keep masks of one handle, reference-shaped (VQATrainer._multipliers pads them)"""
        d = self.vqa.dims
        if self.dropout_masks is not None:
            masks, self.dropout_masks = self.dropout_masks, None
            return masks
        if self._drop_gen is None:
            self._drop_gen = np.random.default_rng(self.dropout_seed)
        g = self._drop_gen
        shapes = dict(enc0=(T, N, d.lstm_dim), dec0=(self.T_decoder, N, d.lstm_dim),
                      qpn_h=(N, d.num_layers * d.lstm_dim), qpn_fc1=(N, d.qpn_hidden))
        on = dict(enc0=self.dropout['enc0'], dec0=self.dropout['dec0'], qpn_h=self.dropout['qpn'] and self.use_qpn,
                  qpn_fc1=self.dropout['qpn'] and self.use_qpn and d.qpn_hidden > 0)
        return {k: (g.random(shapes[k]) < 0.5).astype(np.float32) for k in shapes if on[k]}

    def run_phase1_training(self, handle):
        """phase 1 of a handle that will fetch `train_step`: this handle's dropout masks are drawn here, the
        decoder runs (and samples) under them, and the trainer is told to differentiate under the same ones"""
        feeds = handle.feeds
        step = handle.train_step()
        tr = step.ensure_trainer()
        seq = resolve(self._inputs['input_seq'], feeds)
        lens = resolve(self._inputs['seq_len'], feeds)
        T, N = np.asarray(to_numpy(seq)).shape
        masks = self._draw_masks(T, N)
        self.last_masks = masks
        tr.masks = masks
        mult = tr._multipliers(T, N, self.T_decoder)
        tr.masks = None
        tr._reuse = mult
        use_gt = self._inputs['use_gt_layout']
        use_gt = bool(resolve(use_gt, feeds)) if use_gt is not None else False
        gt = self._inputs['gt_layout']
        gt = resolve(gt, feeds) if (gt is not None and use_gt) else None
        uni = None
        if self.decoder_sampling and not use_gt:
            import torch
            if self._sample_gen is None:
                self._sample_gen = torch.Generator(device='cpu')
                self._sample_gen.manual_seed(self.sample_seed)
            uni = torch.rand((self.T_decoder, N), generator=self._sample_gen).to(self.engine.device)
        return self.engine.seq2seq(seq, lens, self.T_decoder, use_gt, gt, uni,
                                   dropout=(mult.get('enc0'), mult.get('dec0')))

    # -- eager execution ------------------------------------------------------------------------------
    def run_phase1(self, feeds=None):
        feeds = feeds or {}
        seq = resolve(self._inputs['input_seq'], feeds)
        lens = resolve(self._inputs['seq_len'], feeds)
        use_gt = self._inputs['use_gt_layout']
        use_gt = bool(resolve(use_gt, feeds)) if use_gt is not None else False
        gt = self._inputs['gt_layout']
        gt = resolve(gt, feeds) if (gt is not None and use_gt) else None
        return self.engine.seq2seq(seq, lens, self.T_decoder, use_gt, gt)

    def run_phase2(self, packed, image_feat, word_vecs):
        """the packed program on image features WITHOUT the coordinate channels (the reference appends them
        inside Modules, models_vqa/nmn3_modules.py:11-31); + the question prior"""
        feat_c = self.vqa.features_with_coords(image_feat)
        scores = self.engine.execute(packed, feat_c, word_vecs)
        if self.use_qpn and self.vqa.dims.qpn_hidden > 0:
            scores = self.vqa.add_question_prior(scores)
        return scores

    def _fetch(self, f, handle):
        if handle.phase1 is None:
            training = getattr(handle, 'train_const', None) is not None
            handle.phase1 = self.run_phase1_training(handle) if training else self.run_phase1(handle.feeds)
        if f.name == 'entropy_reg':
            return np.float32(np.mean(to_numpy(handle.phase1['neg_entropy']), dtype=np.float32))
        if f.phase == 1:
            return to_numpy(handle.phase1[f.name])
        if f.name == 'l2_reg':
            if 'l2_reg' not in handle.results:
                handle.results['l2_reg'] = np.float32(sum(
                    0.5 * float((np.asarray(to_numpy(v), np.float64) ** 2).sum())
                    for k, v in self.get_weights().items() if k.endswith('weights')))
            return handle.results['l2_reg']
        if 'scores' not in handle.results:
            packed = resolve(self.compiler.loom_input_tensor, handle.feeds)
            feat = resolve(self.image_feat_grid, handle.feeds)
            handle.results['scores'] = self.run_phase2(packed, feat, handle.phase1['word_vecs'])
        return to_numpy(handle.results['scores'])


__all__ = ['Assembler', 'DataReader', 'NMN3Model']
