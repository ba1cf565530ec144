#!/usr/bin/env python3
"""Roofline sweep of the attention-module kernels (pool = softmax attention pooling over the
[H*W, D] feature map: Describe / SameProperty / FindSameProperty) as a
function of the number of module instances per launch.  The end-to-end batch (64 questions) only
gives ~15 pooling jobs per launch; this shows what the kernel does when a launch carries enough work.
Algorithmic bytes per Describe instance: 307 200 B features + logits + partial fc rows."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from n2nmn_amd import synth
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
    NIMG = 1024                      # 1024 x 307 KB = 315 MB of features: larger than L2 + MALL
    d = Dims(N=NIMG, T_decoder=4)
    eng = Engine(d, Assembler(list(CLEVR_MODULE_NAMES)))
    eng.load_weights(synth.make_weights(d, seed=0))
    dev = eng.device
    g = torch.Generator(device='cpu').manual_seed(0)
    feat = torch.relu(torch.randn((NIMG, d.H, d.W, d.D), generator=g)).to(dev)
    wv = torch.randn((d.T_decoder, NIMG, d.embed_dim_txt), generator=g).to(dev)
    rows = []
    for op in ('_Describe', '_SameProperty'):
        for nb in (16, 64, 256, 1024):
            att0 = torch.randn((nb, d.H, d.W, 1), generator=g).to(dev)
            att1 = torch.randn((nb, d.H, d.W, 1), generator=g).to(dev)
            ins = [att0] if op == '_Describe' else [att0, att1]
            t_idx = np.zeros(nb, np.int32)
            b_idx = (np.arange(nb) * 7 % NIMG).astype(np.int32)       # distinct images
            for _ in range(3):
                eng.module_forward(op, ins, t_idx, b_idx, feat, wv)
            torch.cuda.synchronize()
            reps = 20
            eng.profile_begin()
            for r in range(reps):
                bi = ((np.arange(nb) * 7 + 131 * r) % NIMG).astype(np.int32)
                eng.module_forward(op, ins, t_idx, bi, feat, wv)
            fams = {f['name']: f for f in eng.profile_end()}
            f = fams['pool']
            us = 1e3 * f['total_ms'] / f['launches']
            gbs = f['bytes'] / f['launches'] / (us * 1e-6) / 1e9
            rows.append(dict(op=op, instances=nb, pool_us=round(us, 2),
                             algorithmic_MB=round(f['bytes'] / f['launches'] / 1e6, 2),
                             GBps=round(gbs, 1), frac_of_8TBps=round(gbs / 8000.0, 4)))
            print(rows[-1], flush=True)
    print(json.dumps(rows))


if __name__ == '__main__':
    main()
