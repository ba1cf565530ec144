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
               qps=dp.throughput(steps * d.N, elapsed), group=dp.group_size(),
               fastest=dp.last_min_elapsed)
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
    # what bench.py's forward line reports for N > 1: the ranks the GROUP holds (n_gpus, not --gpus) and
    # the fastest rank's time next to the slowest (per_rank_value_min_max)
    assert r0['group'] == 2 and r1['group'] == 2
    assert abs(r0['fastest'] - r1['fastest']) < 1e-9 and 0.05 <= r0['fastest'] < r0['elapsed']


TRAIN_WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch
    from n2nmn_amd.dp import DataParallel
    from n2nmn_amd.train import GradBuckets
    from n2nmn_amd import synth
    from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES, variable_shapes
    from oracle import n2nmn_oracle_grad as G

    dp = DataParallel(backend='gloo')
    d = Dims(H=4, W=5, D=32, map_dim=18, embed_dim_txt=12, embed_dim_nmn=12, lstm_dim=16,
             num_vocab_txt=11, num_choices=7, T_encoder=6, T_decoder=8, N=6)
    names = list(CLEVR_MODULE_NAMES)
    WD = 5e-3
    shapes = variable_shapes(d)
    order = list(shapes)                               # registration order == flat layout order
    sizes = [int(np.prod(shapes[k])) for k in order]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    split = int(offs[[i for i, k in enumerate(order) if '/encoder/' not in k][0]])
    w = synth.make_weights(d, seed=1, dtype=np.float64)   # identical replicas
    m = {k: np.zeros_like(v) for k, v in w.items()}
    v = {k: np.zeros_like(x) for k, x in w.items()}
    import torch.distributed as dist
    for step in (1, 2):
        batch = synth.make_inputs(d, seed=dp.batch_seed(step), n=d.N, min_len=1)   # this rank's shard
        gt = synth.template_layout_batch(d, n=d.N, offset=dp.rank + step)
        _, grads, _ = G.loss_and_grads(w, names, batch, d.T_decoder, d.num_choices, gt, WD)
        flat = torch.zeros(int(offs[-1]), dtype=torch.float64)
        for i, k in enumerate(order):
            flat[offs[i]:offs[i + 1]] = torch.as_tensor(grads[k].reshape(-1))
        buckets = GradBuckets(flat, split, dist)
        buckets.reduce_late()                          # decoder + modules first
        buckets.reduce_early()                         # encoder last
        scale = buckets.wait()
        avg = {k: (flat[offs[i]:offs[i + 1]].numpy() * scale).reshape(shapes[k])
               for i, k in enumerate(order)}
        # what a single process would compute on the concatenation of both shards
        if dp.rank == 0:
            bs, gts = [], []
            for r in range(dp.world):
                bs.append(synth.make_inputs(d, seed=r * 1000 + step, n=d.N, min_len=1))
                gts.append(synth.template_layout_batch(d, n=d.N, offset=r + step))
            big = {k: np.concatenate([b[k] for b in bs], axis=1 if k == 'input_seq_batch' else 0)
                   for k in bs[0]}
            _, gref, _ = G.loss_and_grads(w, names, big, d.T_decoder, d.num_choices,
                                          np.concatenate(gts, axis=1), WD)
            err = max(float(np.abs(avg[k] - gref[k]).max()) for k in order)
        else:
            err = 0.0
        w, m, v = G.adam_step(w, avg, m, v, step)
    digest = float(sum(np.abs(x).sum() for x in w.values()))
    print('RESULT ' + json.dumps(dict(rank=dp.rank, split=split, total=int(offs[-1]), err=err,
                                      digest=digest)), flush=True)
    dp.close()
''') % ROOT


def test_two_ranks_gloo_training_protocol(tmp_path):
    """world size 2 on gloo: per-rank shard gradients (oracle), the two-bucket all-reduce of
    n2nmn_amd.train.GradBuckets and the 1/world scale reproduce the gradient of the global batch
    (the reference's losses are batch means), and both replicas stay bit-identical after Adam."""
    script = tmp_path / 'train_worker.py'
    script.write_text(TRAIN_WORKER)
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS='1')
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    import re
    res = sorted((json.loads(m) for m in re.findall(r'RESULT (\{.*?\})', p.stdout)),
                 key=lambda r: r['rank'])
    assert [r['rank'] for r in res] == [0, 1]
    assert 0 < res[0]['split'] < res[0]['total']
    assert res[0]['err'] < 1e-12                       # DP average == global-batch gradient
    assert res[0]['digest'] == res[1]['digest']        # replicas in lock step
