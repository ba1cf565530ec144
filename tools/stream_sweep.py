#!/usr/bin/env python3
"""Throughput of S streams x K super-bucketed batches (questions/s), pre-spawned worker threads.
Usage: python tools/stream_sweep.py "S,K" ...   (env GPU_MAX_HW_QUEUES is honoured)"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from n2nmn_amd import synth
from n2nmn_amd.nmn3_assembler import Assembler
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
from n2nmn_amd.superbucket import SuperBucket


def measure(S, K, passes=30):
    d = Dims()
    asm = Assembler(list(CLEVR_MODULE_NAMES))
    root = SuperBucket(d, asm, K)
    root.load_weights(synth.make_weights(d, seed=0))
    sbs = [root] + [SuperBucket(d, asm, K, engine=root.engine.fork()) for _ in range(S - 1)]
    for s, sb in enumerate(sbs):
        sb.engine.set_mode('throughput' if S > 1 else 'latency')
        for k in range(K):
            sb.fill(k, synth.make_inputs(d, seed=s * 16 + k), synth.template_layout_batch(d, offset=k))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(S)]
    start, done = threading.Barrier(S + 1), threading.Barrier(S + 1)
    stop = []

    def worker(i):
        torch.cuda.set_device(0)
        while True:
            start.wait()
            if stop:
                return
            with torch.cuda.stream(streams[i]):
                for _ in range(passes):
                    sbs[i].run(use_gt_layout=True)
                streams[i].synchronize()
            done.wait()

    th = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(S)]
    for t in th:
        t.start()
    res = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        start.wait(); done.wait()
        dt = time.perf_counter() - t0
        res.append(S * K * d.N * passes / dt)
    stop.append(1)
    start.wait()
    return res


if __name__ == '__main__':
    for arg in sys.argv[1:]:
        S, K = [int(x) for x in arg.split(',')]
        r = measure(S, K)
        print('streams %d x inflight %d (queues %s): %s q/s' %
              (S, K, os.environ.get('GPU_MAX_HW_QUEUES', 'default'), ['%.0f' % x for x in r]), flush=True)
