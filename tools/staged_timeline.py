#!/usr/bin/env python3
"""Phase times (shader clocks of thread 0, n2nmn_debug_walk_timeline) of the staged walker's jobs in ONE
pass of 16 client batches on the template mix: per Transform / FindSameProperty job of walk_heavy_kernel
and per question of walk_light_kernel.  Usage: python tools/staged_timeline.py"""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from n2nmn_amd import synth, _lib
from n2nmn_amd.nmn3_assembler import Assembler
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
from n2nmn_amd.superbucket import SuperBucket


def main():
    K = 16
    d = Dims()
    names = list(CLEVR_MODULE_NAMES)
    asm = Assembler(names)
    sb = SuperBucket(d, asm, K=K)
    sb.load_weights(synth.make_weights(d, seed=0))
    sb.engine.set_mode('throughput')
    for k in range(K):
        sb.fill(k, synth.make_inputs(d, seed=k), synth.template_layout_batch(d, offset=k))
    for _ in range(3):
        sb.run(use_gt_layout=True)
    torch.cuda.synchronize()
    Q = K * d.N
    tl = torch.zeros((Q, 32, 4), dtype=torch.int64, device='cuda')
    eng = sb.engine
    _lib.check(eng._lib.n2nmn_debug_walk_timeline(eng._ctx, tl.data_ptr()))
    sb.run(use_gt_layout=True)
    torch.cuda.synchronize()
    _lib.check(eng._lib.n2nmn_debug_walk_timeline(eng._ctx, None))
    t = tl.cpu().numpy()
    toks = sb.gt_layout.cpu().numpy()
    idx = {n: i for i, n in enumerate(names)}

    def report(name, rows, labels):
        rows = np.array(rows, np.float64)
        if not len(rows):
            return
        print('%-28s n=%4d' % (name, len(rows)), end='')
        for j, lab in enumerate(labels):
            print('  %s med %6.0f p90 %6.0f' % (lab, np.median(rows[:, j]), np.percentile(rows[:, j], 90)), end='')
        print()

    for op, labels in (('_Transform', ('padded map', 'mfma', 'fold+write', 'total (from operands in LDS)')),
                       ('_FindSameProperty', ('subtree+pool', 'fc_att', 'epilogue', 'total'))):
        rows = []
        for q in range(Q):
            for tt in np.nonzero(toks[:, q] == idx[op])[0]:
                a, b, c, e = t[q, tt]
                if a and e:
                    rows.append((b - a, c - b, e - c, e - a))
        report('heavy ' + op, rows, labels)
    rows = []
    for q in range(Q):
        for tt in np.nonzero(toks[:, q] == idx['_Transform'])[0]:
            a, b, c, e = t[q, tt]
            x0, x1, x2, x3 = t[q, tt + 16]
            if a and e and x0:
                rows.append((x0 - b, x1 - x0, x2 - x1, x3 - x2, c - x3))
    report('Transform matrix phase', rows, ('B frags', 'tile 0', 'tile 1', 'folds', 'barrier'))
    starts = t[:, :, 0][t[:, :, 0] > 0]
    heavy = [(q, tt) for q in range(Q) for tt in range(31) if t[q, tt, 0] and t[q, tt, 3]]
    if heavy:
        t0 = min(t[q, tt, 0] for q, tt in heavy)
        t1 = max(t[q, tt, 3] for q, tt in heavy)
        print('walk_heavy: first job start -> last job end %d clocks (%d jobs)' % (t1 - t0, len(heavy)))
    rows, kinds = [], {}
    for q in range(Q):
        a, b, c, e = t[q, 31]
        if a and e:
            n_nodes = int((toks[:, q] != idx['<eos>']).sum())
            root = names[toks[n_nodes - 1, q]]
            kinds.setdefault(root, []).append((b - a, (c - b) if c else 0, (e - c) if c else e - b, e - a))
    for root, r in sorted(kinds.items()):
        report('light root ' + root, r, ('prog+tree', 'reduce', 'fc_out/pool', 'total'))
    la = [(t[q, 31, 0], t[q, 31, 3]) for q in range(Q) if t[q, 31, 0] and t[q, 31, 3]]
    if la:
        print('walk_light: first start -> last end %d clocks (%d questions)' %
              (max(x[1] for x in la) - min(x[0] for x in la), len(la)))


if __name__ == '__main__':
    main()
