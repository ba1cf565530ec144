#!/usr/bin/env python3
"""Training step (forward + backward + Adam) at 64 / 128 / 256 questions per step, recurrent-step mode
latency vs throughput.  The reference trains at 64 (train_clevr_gt_layout.py:35); larger batches are
a different optimisation problem and are measured only to show where the step's time goes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from n2nmn_amd import synth  # noqa: E402
from n2nmn_amd.engine import Engine  # noqa: E402
from n2nmn_amd.nmn3_assembler import Assembler  # noqa: E402
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES  # noqa: E402
from n2nmn_amd.train import Trainer  # noqa: E402

for N in (64, 128, 256, 512):
    for mode in ('latency', 'throughput'):
        d = Dims(N=N, T_decoder=10)
        eng = Engine(d, Assembler(list(CLEVR_MODULE_NAMES)))
        eng.load_weights(synth.make_weights(d, seed=0))
        eng.set_mode(mode)
        tr = Trainer(eng)
        batches = [{k: torch.as_tensor(v).cuda() for k, v in synth.make_inputs(d, seed=i).items()}
                   for i in range(3)]
        gts = [synth.template_layout_batch(d, offset=i) for i in range(3)]
        for i in range(5):
            tr.step(batches[i % 3], gts[i % 3])
        torch.cuda.synchronize()
        n = 40
        t0 = time.perf_counter()
        for i in range(n):
            tr.step(batches[i % 3], gts[i % 3])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print('N=%4d %-10s %7.3f ms per step  %8.0f questions/s' % (N, mode, 1e3 * dt, N / dt), flush=True)
        del tr, eng
        torch.cuda.empty_cache()
