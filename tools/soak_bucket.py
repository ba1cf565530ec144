#!/usr/bin/env python3
"""Soak of the super-bucket vs single-batch cross-check (tests/test_gpu_stress_bucket.py's body with
fresh random shapes every trial, the eos_retire option on in half of them): hunts rare mismatches.  Usage: soak_bucket.py [trials] (GPU only)"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(trials):
    import torch
    from n2nmn_amd import synth
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
    from n2nmn_amd.superbucket import SuperBucket
    asm = Assembler(list(CLEVR_MODULE_NAMES))
    rng = np.random.default_rng(7)
    engines = {}
    bad = 0
    for trial in range(trials):
        Nb = int(rng.choice([16, 40, 64]))
        K = int(rng.choice([2, 4, 8, 16]))
        mode = str(rng.choice(['throughput', 'latency']))
        use_gt = bool(rng.integers(0, 2))
        retire = bool(rng.integers(0, 2))     # N2NMN_S2S_EOS_RETIRE (ignored where its preconditions fail)
        key = (Nb, K)
        if key not in engines:
            d = Dims(N=Nb)
            w = synth.make_weights(d, seed=1)
            one = Engine(d, asm); one.load_weights(w)
            sb = SuperBucket(d, asm, K); sb.load_weights(w)
            engines[key] = (d, one, sb)
        d, one, sb = engines[key]
        sb.engine.set_mode(mode)
        batches, gts = [], []
        for k in range(K):
            b = synth.make_inputs(d, seed=100000 + 50 * trial + k, min_len=1)
            batches.append(b)
            gts.append(synth.template_layout_batch(d, offset=int(rng.integers(0, 10))))
            sb.fill(k, b, gts[-1] if use_gt else None)
        sb.run(use_gt_layout=use_gt, eos_retire=retire)
        for k in range(K):
            s1, t1, v1 = one.forward(batches[k], use_gt_layout=use_gt, gt_layout=gts[k] if use_gt else None)
            s2, t2, v2 = sb.result(k)
            s1 = torch.as_tensor(s1).cpu().numpy(); s2 = s2.cpu().numpy(); t2 = t2.cpu().numpy()
            tok_same = np.array_equal(np.asarray(t1), t2)
            same_cols = (np.asarray(t1) == t2).all(axis=0)
            err = float(np.abs(s1 - s2)[same_cols].max()) if same_cols.any() else 0.0
            if err > 2e-5 or not tok_same:
                bad += 1
                print('trial %d slot %d Nb=%d K=%d %s gt=%s retire=%s: tokens equal %s (%d of %d columns differ), '
                      'max |dlogit| on equal-token columns %.3e' %
                      (trial, k, Nb, K, mode, use_gt, retire, tok_same, int((~same_cols).sum()), Nb, err), flush=True)
    print('soak: %d trials, %d mismatching slots' % (trials, bad))


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 150)
