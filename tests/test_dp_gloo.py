"""N > 1 path on CPU: two processes over gloo run the forward path's data-parallel protocol
(n2nmn_amd/dp.py, used by bench.py): per-rank question shards, no data collective, barrier-bracketed
timing with the max over ranks."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import json, os, sys, time
    sys.path.insert(0, %r)
    import numpy as np
    from n2nmn_amd.dp import DataParallel
    from n2nmn_amd import synth
    from n2nmn_amd.spec import Dims
    dp = DataParallel(backend='gloo')
    d = Dims(N=4, T_encoder=6)
    # every rank builds ITS OWN batches from its shard seeds
    seeds = [dp.batch_seed(i) for i in range(3)]
    sums = [int(synth.make_inputs(d, seed=s)['input_seq_batch'].sum()) for s in seeds]
    steps = 5
    def run():
        for _ in range(steps):
            time.sleep(0.01 * (1 + dp.rank))      # rank 1 is slower: max-over-ranks must see it
    elapsed = dp.timed(run, sync=lambda: None)
    out = dict(rank=dp.rank, world=dp.world, seeds=seeds, sums=sums, elapsed=elapsed,
               qps=dp.throughput(steps * d.N, elapsed))
    print('RESULT ' + json.dumps(out), flush=True)
    dp.close()
''') % ROOT


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS='1')
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    import json
    import re
    # the two ranks share stdout; their lines can interleave, so pull out the JSON objects by regex
    res = [json.loads(m) for m in re.findall(r'RESULT (\{.*?\})', p.stdout)]
    assert sorted(r['rank'] for r in res) == [0, 1] and all(r['world'] == 2 for r in res)
    r0, r1 = sorted(res, key=lambda r: r['rank'])
    assert r0['seeds'] != r1['seeds'] and r0['sums'] != r1['sums']        # disjoint shards
    # both ranks report the SAME (max) elapsed time, dominated by the slow rank (5 * 20 ms)
    assert abs(r0['elapsed'] - r1['elapsed']) < 1e-9
    assert r0['elapsed'] >= 0.1
    # whole-job throughput counts the questions of all ranks
    assert abs(r0['qps'] - 2 * 5 * 4 / r0['elapsed']) < 1e-6
