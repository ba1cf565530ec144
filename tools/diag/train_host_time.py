"""Is the training step host-bound?  Per-step host enqueue time (perf_counter around Trainer.step, no sync) against the
step's wall time with one synchronise at the end; the same with the host pieces timed one by one.
    python tools/diag/train_host_time.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import torch
from n2nmn_amd import synth, _lib
from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
from n2nmn_amd.nmn3_assembler import Assembler
from n2nmn_amd.engine import Engine
from n2nmn_amd.train import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
knobs = dict(kv.split('=', 1) for kv in sys.argv[2:])      # e.g. train_chunks=25 train_bg_wgs=512
d = Dims(N=64, T_decoder=10)
eng = Engine(d, Assembler(list(CLEVR_MODULE_NAMES)), device=0)
eng.load_weights(synth.make_weights(d, seed=0))
for k, v in knobs.items():
    eng.debug_set(k, v)
tr = Trainer(eng)
dev = eng.device
batches = [{k: torch.as_tensor(v).to(dev) for k, v in synth.make_inputs(d, seed=i).items()} for i in range(4)]
gts = [synth.template_layout_batch(d, offset=i) for i in range(4)]
for i in range(20):
    tr.step(batches[i % 4], gts[i % 4])
torch.cuda.synchronize(dev)
host = []
t0 = time.perf_counter()
for i in range(steps):
    a = time.perf_counter()
    tr.step(batches[i % 4], gts[i % 4])
    host.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize(dev)
t2 = time.perf_counter()
host = np.array(host) * 1e3
if knobs:
    print('knobs', knobs, 'wall %.4f ms/step' % ((t2 - t0) / steps * 1e3))
    sys.exit(0)
print('steps %d: wall %.3f ms/step; host loop %.3f ms/step (median call %.3f, p10 %.3f, p90 %.3f); drain after the loop %.3f ms'
      % (steps, (t2 - t0) / steps * 1e3, (t1 - t0) / steps * 1e3, np.median(host), np.percentile(host, 10),
         np.percentile(host, 90), (t2 - t1) * 1e3))
# the pieces, host only
import ctypes as C
def tm(f, n=200):
    a = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - a) / n * 1e6
print('_io (device-resident batch, host layout -> assemble + upload): %.1f us' % tm(lambda: tr._io(batches[0], gts[0])))
gt_host = np.ascontiguousarray(gts[0], np.int32)
print('assemble_packed alone: %.1f us' % tm(lambda: eng.assembler.assemble_packed(gt_host)))
print('upload_i32 alone: %.1f us' % tm(lambda: eng.upload_i32(gt_host)))
torch.cuda.synchronize(dev)
# one step, synchronised before and after: the latency of a step the host cannot run ahead of
lat = []
for i in range(50):
    torch.cuda.synchronize(dev)
    a = time.perf_counter()
    tr.step(batches[i % 4], gts[i % 4])
    b = time.perf_counter()
    torch.cuda.synchronize(dev)
    lat.append((b - a, time.perf_counter() - a))
lat = np.array(lat) * 1e3
print('synchronised steps: host enqueue %.3f ms, enqueue + drain %.3f ms' % (np.median(lat[:, 0]), np.median(lat[:, 1])))
