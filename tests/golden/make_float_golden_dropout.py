#!/usr/bin/env python3
"""Reference-code fixture for the CLEVR layout generator WITH LSTM dropout:
the reference's unmodified models_clevr/nmn3_model.py + nmn3_netgen_att.py run under the eager TF1 /
Fold stand-in (oracle/tf1_stub) with `encoder_dropout=True, decoder_dropout=True`
(models_clevr/nmn3_netgen_att.py:17-44: DropoutWrapper(output_keep_prob=0.5) on LSTM layer 0).  TF
draws its masks from a private RNG stream; here they are INPUTS (float_cases.clevr_dropout_masks), handed
to the graph in the order it asks for them: T_enc encoder steps, then T_dec decoder steps.

Stored (tests/golden/float_golden_dropout.npz): free-running and teacher-forced decoding -- tokens,
token_probs, neg_entropy, log_seq_prob, answer logits, validity.  Weights and inputs are seeded.

    python tests/golden/make_float_golden_dropout.py [--check]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_float_golden as G          # noqa: E402  (sets up the stand-in and the reference imports)
import float_cases as FC               # noqa: E402
import torch                           # noqa: E402

tf = G.tf
OUT = os.path.join(HERE, 'float_golden_dropout.npz')
CASE = 'gt'                            # N = 12, T_enc = 11, T_dec = 10 (every template + two extras)


def build(use_gt, gt=None):
    from models_clevr.nmn3_assembler import Assembler
    from models_clevr.nmn3_model import NMN3Model
    d, batch = FC.clevr_inputs(CASE)
    masks = FC.clevr_dropout_masks(d)
    G.fresh_graph(FC.clevr_weights(), d.N)
    order = [masks['enc0'][t] for t in range(d.T_encoder)] + [masks['dec0'][t] for t in range(d.T_decoder)]
    it = iter(order)

    def next_mask(shape, keep_prob):
        m = next(it)
        assert tuple(shape) == m.shape and keep_prob == 0.5, (shape, m.shape, keep_prob)
        return m
    tf.config.dropout_masks = next_mask
    asm = Assembler(os.path.join(G.REF, 'exp_clevr/data/vocabulary_layout.txt'))
    kw = {}
    if use_gt:
        kw = dict(use_gt_layout=torch.tensor(True), gt_layout_batch=torch.as_tensor(gt))
    model = NMN3Model(G.T64(batch['image_feat_batch']), torch.as_tensor(batch['input_seq_batch']),
                      torch.as_tensor(batch['seq_length_batch']), T_decoder=d.T_decoder,
                      num_vocab_txt=d.num_vocab_txt, embed_dim_txt=d.embed_dim_txt,
                      num_vocab_nmn=d.num_vocab_nmn, embed_dim_nmn=d.embed_dim_nmn,
                      lstm_dim=d.lstm_dim, num_layers=d.num_layers, assembler=asm,
                      encoder_dropout=True, decoder_dropout=True, decoder_sampling=False,
                      num_choices=d.num_choices, **kw)
    assert next(it, None) is None, 'not every dropout mask was consumed'
    tf.config.dropout_masks = None
    tokens = G.n(model.predicted_tokens)
    exprs, validity = asm.assemble(tokens)
    scores = tf.Session().run(model.scores, feed_dict=model.compiler.build_feed_dict(exprs))
    return d, model, np.asarray(validity, bool), scores


def generate():
    out = {}
    d0 = FC.clevr_dims(CASE)
    for key, use_gt in (('greedy', False), ('gt', True)):
        d, model, validity, scores = build(use_gt, FC.gt_layouts(d0) if use_gt else None)
        out[key + '/predicted_tokens'] = G.n(model.predicted_tokens).astype(np.int32)
        out[key + '/token_probs'] = G.n(model.token_probs)
        out[key + '/neg_entropy'] = G.n(model.neg_entropy)
        out[key + '/log_seq_prob'] = G.n(model.log_seq_prob)
        out[key + '/scores'] = G.n(scores)
        out[key + '/validity'] = validity
        print('%s: %d valid layouts of %d' % (key, int(validity.sum()), d.N), flush=True)
    # the masks really change the result: the same graph without dropout
    dd, batch, asm, model, exprs, validity, scores = G.build_clevr(CASE, use_gt=True, gt=FC.gt_layouts(d0))
    diff = float(np.max(np.abs(G.n(scores) - out['gt/scores'])))
    diff_lsp = float(np.max(np.abs(G.n(model.log_seq_prob) - out['gt/log_seq_prob'])))
    print('without dropout: logits move by %.3e, log_seq_prob by %.3e' % (diff, diff_lsp))
    assert diff > 5e-4 and diff_lsp > 5e-4, (diff, diff_lsp)      # several times the 1e-4 parity bar
    out['gt/scores_without_dropout_maxdiff'] = np.float64(diff)
    out['gt/log_seq_prob_without_dropout_maxdiff'] = np.float64(diff_lsp)
    return out


def main():
    out = generate()
    if '--check' in sys.argv:
        old = np.load(OUT)
        worst = max(float(np.max(np.abs(np.asarray(old[k], np.float64) - np.asarray(out[k], np.float64))))
                    for k in out)
        print('committed fixture vs regenerated: max |diff| = %.3e' % worst)
        assert sorted(old.files) == sorted(out) and worst <= 1e-12
        return
    np.savez_compressed(OUT, **out)
    print('wrote %s (%.1f KB)' % (OUT, os.path.getsize(OUT) / 1024))


if __name__ == '__main__':
    main()
