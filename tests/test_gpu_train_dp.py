"""Data-parallel training on the HIP path with world size 2.  The gpurun box has ONE GPU, so the two
ranks share cuda:0 and exchange gradients over gloo (RCCL refuses two ranks on one device); the
protocol under test is the product's: n2nmn_amd.train.Trainer with a process group -- phase-0
backward, all-reduce of the late bucket, phase-1 backward, all-reduce of the early bucket, 1/world
scale inside the Adam kernel.  Checks: replicas stay identical, and two DP steps equal two steps of
ONE process on the concatenated (global) batch."""
import json
import os
import re
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch
    from n2nmn_amd.dp import DataParallel
    from n2nmn_amd import synth
    from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.train import Trainer

    torch.cuda.set_device(0)
    dp = DataParallel(backend='gloo')
    NB = 16
    d = Dims(N=NB, T_decoder=10)
    names = list(CLEVR_MODULE_NAMES)
    w = synth.make_weights(d, seed=0)

    def shard(rank, step):
        b = synth.make_inputs(d, seed=rank * 1000 + step, n=NB, min_len=1)
        return b, synth.template_layout_batch(d, n=NB, offset=rank + step)

    eng = Engine(d, Assembler(names))
    eng.load_weights(w)
    tr = Trainer(eng, dist=dp._dist)
    tg = None
    if dp.rank == 0:                       # one process, global batch = both shards
        dg = Dims(N=2 * NB, T_decoder=10)
        eg = Engine(dg, Assembler(names))
        eg.load_weights(w)
        tg = Trainer(eg)
    err = 0.0
    for step in (1, 2):
        b, gt = shard(dp.rank, step)
        scale = tr.forward_backward(b, gt)           # includes the two bucketed all-reduces
        torch.cuda.synchronize()
        if tg is not None:
            parts = [shard(r, step) for r in range(dp.world)]
            big = {k: np.concatenate([p[0][k] for p in parts],
                                     axis=1 if k == 'input_seq_batch' else 0) for k in parts[0][0]}
            tg.forward_backward(big, np.concatenate([p[1] for p in parts], axis=1), reduce=False)
            torch.cuda.synchronize()
            for name, (off, n, shape) in tr.layout.items():
                got = tr.grads[off:off + n].double().cpu().numpy() * scale
                ref = tg.grads[off:off + n].double().cpu().numpy()
                err = max(err, float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-7)))
            tg.apply(1.0)
        tr.apply(scale)
    torch.cuda.synchronize()
    got = {k: v.cpu().numpy() for k, v in tr.get_weights().items()}
    digest = float(sum(np.abs(x.astype(np.float64)).sum() for x in got.values()))
    print('RESULT ' + json.dumps(dict(rank=dp.rank, digest=digest, err=err)), flush=True)
    dp.close()
''') % ROOT


def test_two_ranks_one_gpu_gloo(tmp_path):
    script = tmp_path / 'dp_train_worker.py'
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, OMP_NUM_THREADS='4'))
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    res = sorted((json.loads(m) for m in re.findall(r'RESULT (\{.*?\})', p.stdout)),
                 key=lambda r: r['rank'])
    assert [r['rank'] for r in res] == [0, 1]
    assert res[0]['digest'] == res[1]['digest']          # replicas bit-identical after 2 steps
    # averaged shard gradients == gradient of the global batch (the losses are batch means), per
    # variable relative to its scale, at both steps (the second one on the updated weights; the two
    # trajectories only differ by fp32 summation order, amplified by Adam's lr-sized first step)
    assert res[0]['err'] < 5e-3, res


def test_c_abi_rccl_communicator_world_1():
    import numpy as np
    from n2nmn_amd import synth
    from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
    """n2nmn_comm_* / n2nmn_allreduce_grads (include/n2nmn.h 6b) on a 1-rank RCCL communicator: the
    code path of the multi-GPU step (library-owned side stream, event fork / join, two buckets) gives
    the gradients and the update of the single-GPU step."""
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.train import Trainer, RcclBuckets
    d = Dims(T_decoder=10)
    eng = Engine(d, Assembler(list(CLEVR_MODULE_NAMES)))
    w = synth.make_weights(d, seed=0)
    batch = synth.make_inputs(d, seed=11)
    gt = synth.template_layout_batch(d, offset=2)
    out = {}
    for use in (False, True):
        eng.load_weights(w)
        tr = Trainer(eng, rccl=use)
        assert isinstance(tr.buckets, RcclBuckets) == use
        scale = tr.forward_backward(batch, gt)           # reduce=True: both buckets issued
        assert scale == 1.0
        g = tr.grads.detach().cpu().numpy().copy()
        tr.apply(scale)
        w1 = {k: v.cpu().numpy() for k, v in tr.get_weights().items()}
        out[use] = (g, w1, tr.losses.cpu().numpy().copy())
        if use:
            assert tr.buckets.world == 1
            tr.buckets.close()
    # the backward pass accumulates several gradients with float atomics, so two runs of the SAME
    # step agree to round-off (~1e-7 relative to the largest gradient), not bit for bit
    def close(a, b, what):
        tol = 2e-6 * max(float(np.abs(b).max()), 1e-6)
        assert float(np.abs(a - b).max()) <= tol, what
    close(out[True][0], out[False][0], 'gradients')
    close(out[True][2], out[False][2], 'losses')
    for k in out[True][1]:
        # the first Adam step moves a weight by lr * g / (|g| + eps): round-off in a near-zero g
        # is amplified up to the step size (lr = 1e-3), so the bulk must agree and no element may
        # be further apart than two steps
        diff = np.abs(out[True][1][k] - out[False][1][k])
        assert float(diff.max()) <= 2.001e-3 and float(diff.mean()) <= 2e-6, k
