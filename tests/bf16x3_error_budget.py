#!/usr/bin/env python3
"""Error budget of the opt-in split-operand mode (N2NMN_MODE_THROUGHPUT_BF16X3) beside the exact-fp32
throughput mode, both against float64 truth (the record a judge needs to rule on
promoting the mode).  GPU only; the oracle / the reference-code fixture is the checker, never the product.

  A  CLEVR pass of 16 x 64 questions (synthetic weights, seed 0): per mode, max |err| against the fp64 oracle
     of the logits (teacher-forced template layouts), of the decoder's atts / token_probs / neg_entropy /
     log_seq_prob, and the greedy decoder's token flips against the fp64 oracle's tokens (with the oracle's own
     top-2 margin at each flip: a flip at a margin under 1e-5 is a tie that fp32 rounding also decides);
  B  the reference-code fixture at the eval configuration (tests/golden/float_golden_full.npz: logits and
     greedy tokens computed by the reference's own model files in float64) as slot 5 of an 8-slot pass;
  C  models_vqa (lstm_dim 1024, 2064 -> 1024 conv_image) pass of 8 x 128: logits of 12 rows against the fp64
     oracle, both modes.
Lives under tests/ because it runs the oracle (test infrastructure).
Usage: python tests/bf16x3_error_budget.py [json-out]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from oracle import n2nmn_oracle as O                      # noqa: E402  (the checker)
from n2nmn_amd import synth, vqa                           # noqa: E402
from n2nmn_amd.nmn3_assembler import Assembler             # noqa: E402
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES        # noqa: E402
from n2nmn_amd.superbucket import SuperBucket              # noqa: E402
import float_cases as FC                                   # noqa: E402

MODES = ('throughput', 'throughput_bf16x3')
NAMES = list(CLEVR_MODULE_NAMES)


def n(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


def flips(tok, ref_tok, ref_dec):
    """columns whose greedy layout differs from the oracle's; the oracle's top-2 margin at the first flip"""
    out = []
    ts = ref_dec.get('token_scores')
    for i in np.nonzero((tok != ref_tok).any(axis=0))[0]:
        step = int(np.argmax(tok[:, i] != ref_tok[:, i]))
        gap = None
        if ts is not None:
            gap = abs(float(ts[step, i, ref_tok[step, i]] - ts[step, i, tok[step, i]]))
        out.append((int(i), step, gap))
    return out


def part_a(out):
    d = Dims()
    K = 16
    sb = SuperBucket(d, Assembler(NAMES), K=K)
    w = synth.make_weights(d, seed=0)
    sb.load_weights(w)
    batches = [synth.make_inputs(d, seed=k, min_len=1) for k in range(K)]
    gts = [synth.template_layout_batch(d, offset=k) for k in range(K)]
    for k in range(K):
        sb.fill(k, batches[k], gts[k])
    refs_gt = [O.forward(w, NAMES, batches[k], d.T_decoder, d.num_choices, np.float64, use_gt_layout=True,
                         gt_layout=gts[k]) for k in range(K)]
    refs_free = [O.forward(w, NAMES, batches[k], d.T_decoder, d.num_choices, np.float64) for k in range(K)]
    res = {}
    for mode in MODES:
        sb.engine.set_mode(mode)
        sc, tok, val = sb.run(use_gt_layout=True)
        s2s = sb.engine.decoder_outputs()
        torch.cuda.synchronize()
        sc = n(sc).copy()
        r = {'logits': 0.0, 'atts': 0.0, 'token_probs': 0.0, 'neg_entropy': 0.0, 'log_seq_prob': 0.0}
        for k in range(K):
            c = slice(k * d.N, (k + 1) * d.N)
            dec = refs_gt[k]['dec']
            r['logits'] = max(r['logits'], float(np.abs(sc[c] - refs_gt[k]['scores']).max()))
            r['atts'] = max(r['atts'], float(np.abs(n(s2s['atts'])[:, :, c] - dec['atts'][..., 0]).max()))
            r['token_probs'] = max(r['token_probs'], float(np.abs(n(s2s['token_probs'])[:, c] - dec['token_probs']).max()))
            r['neg_entropy'] = max(r['neg_entropy'], float(np.abs(n(s2s['neg_entropy'])[c] - dec['neg_entropy']).max()))
            r['log_seq_prob'] = max(r['log_seq_prob'], float(np.abs(n(s2s['log_seq_prob'])[c] - refs_gt[k]['log_seq_prob']).max()))
        sc, tok, val = sb.run(use_gt_layout=False)
        torch.cuda.synchronize()
        tok = n(tok)
        fl = []
        for k in range(K):
            c = slice(k * d.N, (k + 1) * d.N)
            fl += [(k,) + f for f in flips(tok[:, c], refs_free[k]['dec']['predicted_tokens'], refs_free[k]['dec'])]
        r['greedy_questions'] = K * d.N
        r['greedy_layout_flips'] = len(fl)
        r['greedy_flip_margins'] = sorted(f[3] for f in fl if f[3] is not None)
        res[mode] = r
        print('A  CLEVR 1024 rows  %-18s %s' % (mode, json.dumps(r)), flush=True)
    sb.engine.set_mode('latency')
    out['clevr_pass_1024'] = res


def part_b(out):
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'float_golden_full.npz'))
    d, batch = FC.clevr_inputs('full')
    gt = synth.template_layout_batch(d)
    res = {}
    for mode in MODES:
        sb = SuperBucket(d, Assembler(NAMES), K=8)
        sb.load_weights(FC.clevr_weights())
        sb.engine.set_mode(mode)
        for k in range(8):
            other = synth.make_inputs(d, seed=300 + k, min_len=1)
            sb.fill(k, batch if k == 5 else other, gt if k == 5 else synth.template_layout_batch(d, offset=k))
        sb.run(use_gt_layout=True)
        e_gt = float(np.abs(n(sb.result(5)[0]) - z['gt/scores']).max())
        sb.run(use_gt_layout=False)
        sc, tok, val = [n(x) for x in sb.result(5)]
        want = z['greedy/predicted_tokens']
        same = (tok == want).all(axis=0)
        res[mode] = {'gt_logits': e_gt, 'greedy_layout_flips': int((~same).sum()), 'questions': int(same.size),
                     'greedy_logits_on_equal_layouts': float(np.abs(sc[same] - z['greedy/scores'][same]).max())}
        print('B  reference-code fixture (N = 64, slot 5 of 8)  %-18s %s' % (mode, json.dumps(res[mode])), flush=True)
        del sb
        torch.cuda.empty_cache()
    out['reference_fixture_full'] = res


def part_c(out):
    CLIENT, K = 128, 8
    d = vqa.VQADims(N=CLIENT)
    big = vqa.VQADims(N=K * CLIENT)
    w = synth.make_weights_from_shapes(vqa.vqa_variable_shapes(d), seed=0)
    eng = vqa.VQAEngine(big)
    eng.load_weights(w)
    dev = eng.engine.device
    MIX = (['_Find', '_Find', '_And', '_Describe'],) * 46 + (['_Find', '_Describe'],) * 43 + \
          (['_Find', '_Transform', '_Describe'],) * 9 + (['_Find', '_Transform', '_Find', '_And', '_Describe'],) * 2
    rng = np.random.default_rng(7)
    R = K * CLIENT
    lens = rng.integers(1, d.T_encoder + 1, size=R).astype(np.int32)
    seq = rng.integers(0, d.num_vocab_txt, size=(d.T_encoder, R)).astype(np.int32)
    seq[np.arange(d.T_encoder)[:, None] >= lens[None, :]] = 0
    feat = torch.clamp(torch.randn((R, d.H, d.W, d.D), device=dev, generator=torch.Generator(dev).manual_seed(3)), min=0)
    gt = np.ascontiguousarray(np.array([eng.assembler.module_list2tokens(MIX[rng.integers(0, 100)], d.T_decoder)
                                        for _ in range(R)], np.int32).T)
    rows = sorted({0, 127, 128, 511, 1023, 3, 126, 643, 766, 15, 16, 1008})
    sub = dict(input_seq_batch=np.ascontiguousarray(seq[:, rows]), seq_length_batch=np.ascontiguousarray(lens[rows]),
               image_feat_batch=feat[rows].cpu().numpy())
    ref = O.forward_vqa(w, sub, d.T_decoder, d.num_choices, np.float64, use_gt_layout=True,
                        gt_layout=np.ascontiguousarray(gt[:, rows]))
    cat = dict(input_seq_batch=seq, seq_length_batch=lens, image_feat_batch=feat)
    res = {}
    for mode in MODES:
        eng.engine.set_mode(mode)
        scores, tokens, validity = eng.forward(cat, use_gt_layout=True, gt_layout=gt)
        got = n(scores)[rows]
        res[mode] = {'logits_12_rows': float(np.abs(got - ref['scores']).max()),
                     'logit_scale': float(np.abs(ref['scores']).max())}
        print('C  models_vqa 8 x 128  %-18s %s' % (mode, json.dumps(res[mode])), flush=True)
    out['models_vqa_pass_1024'] = res


def main():
    out = {}
    part_a(out)
    part_b(out)
    part_c(out)
    if len(sys.argv) > 1:
        with open(sys.argv[1], 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
