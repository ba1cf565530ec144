#!/usr/bin/env python3
"""PCIe-inclusive throughput: the bench's super-bucket pipeline (2 buckets x 8 batches of 64) with
every batch's inputs (19.7 MB of features + text) copied from PINNED HOST memory into its slot on a
copy stream, under the other bucket's compute.  bench.py's `value` has the inputs resident in HBM;
this is the rate when they are not."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(passes=40, K=8):
    import torch
    from n2nmn_amd import synth
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
    from n2nmn_amd.superbucket import SuperBucket
    d = Dims()
    asm = Assembler(list(CLEVR_MODULE_NAMES))
    w = synth.make_weights(d, seed=0)
    buckets = [SuperBucket(d, asm, K=K) for _ in range(2)]
    for b in buckets:
        b.load_weights(w)
        b.engine.set_mode('throughput')
    dev = buckets[0].engine.device
    host = []
    for i in range(2 * K):
        b = synth.make_inputs(d, seed=100 + i)
        host.append({k: torch.as_tensor(np.ascontiguousarray(v)).pin_memory()
                     for k, v in b.items() if k in ('input_seq_batch', 'seq_length_batch', 'image_feat_batch')})
    gt = torch.as_tensor(synth.template_layout_batch(d)).to(dev)
    for b in buckets:
        for k in range(K):
            b.slot(k)['gt_layout_batch'].copy_(gt)
    copy = torch.cuda.Stream(device=dev)
    comp = [torch.cuda.Stream(device=dev) for _ in buckets]
    copied = [torch.cuda.Event() for _ in buckets]
    released = [torch.cuda.Event() for _ in buckets]

    def one_pass(i, first):
        s = i & 1
        bk = buckets[s]
        with torch.cuda.stream(copy):
            if not first:
                copy.wait_event(released[s])
            for k in range(K):
                slot = bk.slot(k)
                for key, t in host[s * K + k].items():
                    slot[key].copy_(t, non_blocking=True)
            copied[s].record(copy)
        with torch.cuda.stream(comp[s]):
            comp[s].wait_event(copied[s])
            bk.run(use_gt_layout=True)
            released[s].record(comp[s])

    for i in range(4):
        one_pass(i, i < 2)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(passes):
        one_pass(i, False)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    nq = passes * K * d.N
    mb = passes * K * (d.N * d.H * d.W * d.D * 4) / 1e6
    print('host-fed: %.1f questions/s, %.3f ms per batch of 64, H2D %.1f GB/s (features only)'
          % (nq / dt, 1e3 * dt / (passes * K), mb / 1e3 / dt))


if __name__ == '__main__':
    main()
