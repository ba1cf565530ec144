#!/usr/bin/env python3
"""Host-side time of each call of one training step (no synchronisation inside the loop): a call
that takes about as long as the GPU work queued before it is blocking on the device."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from n2nmn_amd import _lib, synth  # noqa: E402
from n2nmn_amd.engine import Engine  # noqa: E402
from n2nmn_amd.nmn3_assembler import Assembler  # noqa: E402
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES  # noqa: E402
from n2nmn_amd.train import Trainer  # noqa: E402

d = Dims(N=64, T_decoder=10)
eng = Engine(d, Assembler(list(CLEVR_MODULE_NAMES)))
eng.load_weights(synth.make_weights(d, seed=0))
tr = Trainer(eng)
batches = [{k: torch.as_tensor(v).cuda() for k, v in synth.make_inputs(d, seed=i).items()}
           for i in range(4)]
gts = [synth.template_layout_batch(d, offset=i) for i in range(4)]
for i in range(5):
    tr.step(batches[i % 4], gts[i % 4])
torch.cuda.synchronize()
names = ['_io (H2D gt + assemble)', 'train_forward', 'train_backward(0)', 'train_backward(1)',
         'adam_step']
acc = [0.0] * len(names)
n = 40
t_all = time.perf_counter()
for i in range(n):
    b, gt = batches[i % 4], gts[i % 4]
    t0 = time.perf_counter()
    io, packed, _ = tr._io(b, gt)
    s = eng.stream()
    t1 = time.perf_counter()
    _lib.check(tr._lib.n2nmn_train_forward(tr._ctx, C.byref(io), packed.handle, s))
    t2 = time.perf_counter()
    _lib.check(tr._lib.n2nmn_train_backward(tr._ctx, C.byref(io), packed.handle, 0, s))
    t3 = time.perf_counter()
    _lib.check(tr._lib.n2nmn_train_backward(tr._ctx, C.byref(io), packed.handle, 1, s))
    t4 = time.perf_counter()
    tr.apply(1.0)
    t5 = time.perf_counter()
    for j, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        acc[j] += v
t_host = time.perf_counter() - t_all
torch.cuda.synchronize()
t_tot = time.perf_counter() - t_all
for nm, v in zip(names, acc):
    print('%-28s %8.1f us' % (nm, 1e6 * v / n))
print('host loop %.1f us/step, wall incl. final sync %.1f us/step' % (1e6 * t_host / n, 1e6 * t_tot / n))
