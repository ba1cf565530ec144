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
(eval_vqa2.py:137) is the script's.  Training (dropout, losses) goes through n2nmn_amd.vqa.VQATrainer; this
face is the inference drop-in, so the dropout switches must be False as in the eval script.
"""
from __future__ import annotations

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
        if encoder_dropout or decoder_dropout or qpn_dropout:
            raise NotImplementedError('the inference face takes no dropout (exp_vqa/eval_vqa2.py:80-85); '
                                      'training with dropout: n2nmn_amd.vqa.VQATrainer')
        if decoder_sampling:
            raise NotImplementedError('decoder_sampling: the policy-gradient path is n2nmn_amd.vqa.VQATrainer.step_rl')
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
        register_model(self)

    def load_weights(self, weights):
        """reference-named, reference-shaped variables (n2nmn_amd.vqa.vqa_variable_shapes)"""
        self.vqa.load_weights(weights)

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
            handle.phase1 = self.run_phase1(handle.feeds)
        if f.phase == 1:
            return to_numpy(handle.phase1[f.name])
        if 'scores' not in handle.results:
            packed = resolve(self.compiler.loom_input_tensor, handle.feeds)
            feat = resolve(self.image_feat_grid, handle.feeds)
            handle.results['scores'] = self.run_phase2(packed, feat, handle.phase1['word_vecs'])
        return to_numpy(handle.results['scores'])


__all__ = ['Assembler', 'DataReader', 'NMN3Model']
