"""Layout generator with the reference's constructor (models_clevr/nmn3_netgen_att.py:46-71):

    AttentionSeq2Seq(input_seq_batch[T,N] i32, seq_length_batch[N] i32, T_decoder, num_vocab_txt,
                     embed_dim_txt, num_vocab_nmn, embed_dim_nmn, lstm_dim, num_layers, assembler,
                     encoder_dropout, decoder_dropout, decoder_sampling, use_gt_layout=None,
                     gt_layout_batch=None, scope='encoder_decoder', reuse=None)

Attributes `.predicted_tokens .token_probs .neg_entropy .atts .word_vecs` (:305-322) are fetch
handles; `run(feeds)` executes phase 1 through the C-ABI (`n2nmn_seq2seq_forward`: 2-layer LSTM
encoder, attentional LSTM decoder under the validity automaton, text-attention word vectors) and
returns device tensors.  Inputs may be placeholders (bound by `feeds`) or tensors.
"""
from __future__ import annotations

from .engine import Engine
from .runtime import Fetch, resolve

PHASE1_OUTPUTS = ('predicted_tokens', 'token_probs', 'neg_entropy', 'atts', 'word_vecs',
                  'log_seq_prob')


class AttentionSeq2Seq:
    def __init__(self, input_seq_batch, seq_length_batch, T_decoder, num_vocab_txt, embed_dim_txt,
                 num_vocab_nmn, embed_dim_nmn, lstm_dim, num_layers, assembler, encoder_dropout,
                 decoder_dropout, decoder_sampling, use_gt_layout=None, gt_layout_batch=None,
                 scope='encoder_decoder', reuse=None, engine: Engine = None, sample_seed: int = 0,
                 dropout_seed: int = 0):
        if engine is None:
            raise ValueError('AttentionSeq2Seq needs engine=Engine(...)')
        d = engine.dims
        want = dict(num_vocab_txt=num_vocab_txt, embed_dim_txt=embed_dim_txt,
                    num_vocab_nmn=num_vocab_nmn, embed_dim_nmn=embed_dim_nmn, lstm_dim=lstm_dim,
                    num_layers=num_layers)
        for k, v in want.items():
            if getattr(d, k) != v:
                raise ValueError('%s=%r differs from the engine dims (%r)' % (k, v, getattr(d, k)))
        if T_decoder > d.T_decoder:
            raise ValueError('T_decoder exceeds the engine capacity')
        # encoder_dropout / decoder_dropout: DropoutWrapper(output_keep_prob=0.5) on the output of every
        # LSTM layer but the last (:17-44) -> multipliers on layer 0's output, handed to the C-ABI as
        # n2nmn_seq2seq_io.drop_enc0 / drop_dec0.  (Inactive in every CLEVR script of the reference,
        # train_clevr_gt_layout.py:31-32, but part of the constructor contract.)
        self.keep_prob = 0.5
        self.dropout_masks = None      # {'enc0': [T_enc, N, L], 'dec0': [T_dec, N, L]} {0,1} keep masks
        #                                for the next run (tests: TF's RNG stream is not reproducible);
        #                                None: drawn by n2nmn_dropout_multipliers (seed, offset)
        self.dropout_seed = dropout_seed
        self._drawn = 0
        self.engine = engine
        self.T_decoder = T_decoder
        self.encoder_num_vocab = num_vocab_txt
        self.encoder_embed_dim = embed_dim_txt
        self.decoder_num_vocab = num_vocab_nmn
        self.decoder_embed_dim = embed_dim_nmn
        self.lstm_dim = lstm_dim
        self.num_layers = num_layers
        self.EOS_token = assembler.EOS_idx
        self.P, self.W, self.b = assembler.P, assembler.W, assembler.b
        self.encoder_dropout = encoder_dropout
        self.decoder_dropout = decoder_dropout
        self.decoder_sampling = decoder_sampling
        self._inputs = dict(input_seq=input_seq_batch, seq_len=seq_length_batch,
                            use_gt_layout=use_gt_layout, gt_layout=gt_layout_batch)
        self._gen = None
        self._seed = sample_seed
        for name in PHASE1_OUTPUTS:
            setattr(self, name, Fetch(self, name, 1))

    def run(self, feeds=None, forced_tokens=None, debug=False):
        """Execute phase 1; returns {name: device tensor}."""
        import torch
        feeds = feeds or {}
        seq = resolve(self._inputs['input_seq'], feeds)
        lens = resolve(self._inputs['seq_len'], feeds)
        use_gt = self._inputs['use_gt_layout']
        use_gt = bool(resolve(use_gt, feeds)) if use_gt is not None else False
        gt = self._inputs['gt_layout']
        gt = resolve(gt, feeds) if (gt is not None and use_gt) else None
        uni = None
        if self.decoder_sampling:
            # stand-in for tf.multinomial's private RNG stream (nmn3_netgen_att.py:216-217)
            # drawn on the host generator (640 floats a batch), so that a seed names the same layouts on every
            # device -- the recorded training-driver runs are replayed on another box
            if self._gen is None:
                self._gen = torch.Generator(device='cpu')
                self._gen.manual_seed(self._seed)
            n = torch.as_tensor(seq).shape[1]
            uni = torch.rand((self.T_decoder, n), generator=self._gen).to(self.engine.device)
        drop = None
        if self.encoder_dropout or self.decoder_dropout:
            T, n = torch.as_tensor(seq).shape
            drop = (self._multipliers('enc0', (T, n, self.lstm_dim)) if self.encoder_dropout else None,
                    self._multipliers('dec0', (self.T_decoder, n, self.lstm_dim))
                    if self.decoder_dropout else None)
        return self.engine.seq2seq(seq, lens, self.T_decoder, use_gt, gt, uni, forced_tokens,
                                   debug=debug, dropout=drop)

    def _multipliers(self, key, shape):
        """0 or 1 / keep_prob per element of layer 0's output: from the masks a test supplied, or
        from the library's counter-based generator (a fresh stretch of its stream per run)."""
        import numpy as np
        import torch
        from . import _lib
        e = self.engine
        if self.dropout_masks is not None:
            keep = torch.as_tensor(np.asarray(self.dropout_masks[key], np.float32), device=e.device)
            if tuple(keep.shape) != tuple(shape):
                raise ValueError('dropout mask %s: shape %s, expected %s' % (key, tuple(keep.shape), shape))
            return (keep / self.keep_prob).contiguous()
        m = torch.empty(shape, dtype=torch.float32, device=e.device)
        _lib.check(e._lib.n2nmn_dropout_multipliers(m.data_ptr(), m.numel(), self.keep_prob,
                                                    self.dropout_seed, self._drawn, e.stream()))
        self._drawn += m.numel()
        return m

    def _fetch(self, f, handle):
        from .runtime import to_numpy
        if handle.phase1 is None:
            handle.phase1 = self.run(handle.feeds)
        return to_numpy(handle.phase1[f.name])
