#!/usr/bin/env python3
"""Per-launch time of every kernel of the walker family in ONE pass of 16 client batches (1024 questions),
template mix: each launch of the last pass replayed back to back inside one HIP event pair
(n2nmn_debug_walk_replay).  Usage: python tools/walk_stage_bench.py [templates|clevr_like] [devlayouts]
(devlayouts: the layouts are handed over as device tensors -- no host copy, so no nesting bound: adaptive
level count and a fall-back walker launch)"""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from n2nmn_amd import synth
from n2nmn_amd.nmn3_assembler import Assembler
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
from n2nmn_amd.superbucket import SuperBucket


def main():
    K = 16
    mix = sys.argv[1] if len(sys.argv) > 1 else 'templates'
    d = Dims()
    asm = Assembler(list(CLEVR_MODULE_NAMES))
    sb = SuperBucket(d, asm, K=K)
    sb.load_weights(synth.make_weights(d, seed=0))
    sb.engine.set_mode('throughput')
    for k in range(K):
        gt = synth.template_layout_batch(d, offset=k) if mix == 'templates' else synth.clevr_like_layout_batch(d, seed=k)
        sb.fill(k, synth.make_inputs(d, seed=k), torch.as_tensor(gt).cuda() if 'devlayouts' in sys.argv[2:] else gt)
    for _ in range(4):
        sb.run(use_gt_layout=True)
    torch.cuda.synchronize()
    eng = sb.engine
    names = {0: 'walker (heavy + fspepi per level, light, fall-back)', 5: '  walk_heavy (level 0: FSP stage A + Transform halves)',
             6: '  walk_fspepi (level 0: FSP stage B)', 7: '  walk_light', 8: '  walk_kernel (fall-back list)',
             3: 'walk_find', 1: 'walk_pool', 2: 'walk_fcatt + walk_heads', 4: 'walk_tmap'}
    tot = 0.0
    for which in (4, 3, 0, 5, 6, 7, 8, 1, 2):
        us = min(eng.walk_replay_us(which, 50) for _ in range(3))
        print('%-56s %8.2f us' % (names[which], us), flush=True)
        if which in (3, 0, 1):
            tot += us
    print('find + walker + pool = %.2f us (308 MB at 8 TB/s x 0.40 = 96.2 us)' % tot)


if __name__ == '__main__':
    main()
