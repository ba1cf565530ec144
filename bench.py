#!/usr/bin/env python3
"""Benchmark of the N2NMN CLEVR forward hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one pass of the hot path over one batch of 64 questions (BASELINE.json configs[1]:
CLEVR forward, fixed ground-truth layouts, 10x15x512 synthetic pool5 features): phase 1 (LSTM
encoder + teacher-forced attentional decoder), C++ assemble/pack of the layout, phase 2 (module
network) -> answer logits in HBM.  With ground-truth layouts the predicted tokens ARE the host's
gt_layout (models_clevr/nmn3_netgen_att.py:236-238), so the program is assembled from the host copy
while phase 1 runs and the step never synchronises; --fetch-tokens (and config 3, where the decoder
chooses the layout) fetches predicted_tokens to the host between the phases as
exp_clevr/eval_clevr.py:111-125 does.  Inputs are resident in HBM before the timed region.
Multi-GPU: the path shards by question with no data-path collective (weak scaling: every rank runs
its own batch-64 stream); timing = barrier + synchronize on both sides, max over ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, shared
# with the framework's own streams).  The batches in flight only overlap when each stream has a
# queue of its own, and four step kernels (8 waves per workgroup, 2 per SIMD) exactly fill the 8
# wave slots of a SIMD: 4 streams on >= 6 queues measured 142-143 k questions/s, 4 streams on the
# default 4 queues 108 k, 6 streams on 4 queues 128 k, 5 or more truly concurrent streams 85-89 k
# (the fifth kernel's workgroups wait for wave slots).  Must be set before the runtime starts.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3    # MI355X_MICROARCH.md: fp32-input MFMA dense peak
MFMA_FAMILIES = ('lstm_step', 'gemm_pk', 'lstm_bwd_step', 'gemm_tn')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', type=int, default=2, choices=(2, 3, 4, 5),
                    help='2: fixed gt layouts (metric config); 3: greedy decoder layouts; '
                         '4: training step (forward + backward + RCCL all-reduce + Adam); '
                         '5: models_vqa forward (14x14x2048 feats, batch 128)')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--streams', type=int, default=4,
                    help='independent batches in flight per GPU (one host thread + HIP stream + '
                         'forked context each; weights shared)')
    ap.add_argument('--fetch-tokens', action='store_true',
                    help='config 2: fetch predicted_tokens from the device before assembling (as '
                         'eval_clevr.py does) instead of assembling from the host copy of gt_layout')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true')
    return ap.parse_args()


PMC_FILE = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
PMC_FILE_TRAIN = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic_train.json')
PMC_KERNEL = {'lstm_step': 'lstm_step_kernel<4, 0>', 'dec_attn': 'dec_attn_kernel<256>',
              'gemm_pk': 'gemm_pk_kernel', 'att_ops': 'att_ops_kernel', 'pool': 'pool_kernel',
              'textmap': 'textmap_kernel', 'heads': 'heads_kernel', 'word_vecs': 'word_vecs_kernel',
              'lstm_bwd_step': 'lstm_bwd_step_kernel', 'gemm_tn': 'gemm_tn_kernel',
              'optimiser': 'adam_kernel'}


def pmc_traffic(family, path=None):
    """HBM bytes per launch of the kernel behind a profiler family, from the committed rocprofv3
    PMC passes (tools/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE runs of this same bench
    command, gfx950 x2 read correction).  PMC counters cannot be read inside the timed run."""
    try:
        data = json.load(open(path or PMC_FILE))['kernels']
        for prefix, kname in PMC_KERNEL.items():
            hit = [k for k in data if k.startswith(kname)]       # (template arguments vary)
            if family.startswith(prefix) and hit:
                return data[hit[0]]['hbm_bytes_per_launch'], \
                    'profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)' % \
                    os.path.basename(path or PMC_FILE)
    except Exception:
        pass
    return None, None


def cpu_baseline(d, w, batch, gt, names, use_gt):
    """fp32 numpy oracle (the CPU restatement of the reference; the reference's TF1 path cannot
    run here) on a bounded sample: whole batches until ~10 s have passed (max 3)."""
    import numpy as np
    from oracle import n2nmn_oracle as O
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    t0 = time.perf_counter()
    nb = 0
    while nb < 3 and (nb == 0 or time.perf_counter() - t0 < 10.0):
        O.forward(w, names, batch, d.T_decoder, d.num_choices, np.float32,
                  use_gt_layout=use_gt, gt_layout=gt)
        nb += 1
    dt = time.perf_counter() - t0
    return {'value': round(nb * d.N / dt, 2), 'unit': 'questions/sec', 'cores': int(threads),
            'kind': 'port',
            'sample': '%d batch(es) of %d questions, same synthetic inputs, fp32 numpy oracle '
                      '(oracle/n2nmn_oracle.py; BLAS threads=%d of %d host cores); the reference '
                      'TF1/Fold CPU path is not runnable here' % (nb, d.N, threads,
                                                                   os.cpu_count() or 0)}


def kernel_rows(fams, ksteps):
    rows = []
    for f in fams:
        if f['launches'] == 0:
            continue
        bound = 'mfma' if f['name'].startswith(MFMA_FAMILIES) else 'hbm'
        avg_s = f['total_ms'] * 1e-3 / f['launches']
        if bound == 'mfma':
            ach = f['flops'] / f['launches'] / avg_s / 1e12
            peak, unit = MFMA_F32_PEAK_TF, 'TFLOP/s'
        else:
            ach = f['bytes'] / f['launches'] / avg_s / 1e9
            peak, unit = HBM_PEAK_GBS, 'GB/s'
        rows.append({'kernel': f['name'], 'bound': bound, 'launches_per_step':
                     round(f['launches'] / ksteps, 2), 'avg_us': round(avg_s * 1e6, 3),
                     'us_per_step': round(f['total_ms'] * 1e3 / ksteps, 2),
                     'achieved': round(ach, 3), 'peak': peak, 'unit': unit,
                     'frac': round(ach / peak, 4)})
    rows.sort(key=lambda r: -r['us_per_step'])
    return rows


def bench_train(args, dp, local_rank):
    """BASELINE.json configs[3]: exp_clevr/train_clevr_gt_layout.py loop body -- forward + backward
    + gradient all-reduce (RCCL, two buckets, the late one overlapping the encoder's backward) +
    per-tensor clip + Adam -- batch 64 per GPU, T_dec = 10, gt layouts.  One step = one iteration."""
    import numpy as np
    import torch
    from n2nmn_amd import synth
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES
    from n2nmn_amd.train import Trainer

    rank, world = dp.rank, dp.world
    d = Dims(N=args.batch, T_decoder=10)
    names = list(CLEVR_MODULE_NAMES)
    eng = Engine(d, Assembler(names), device=local_rank)
    w = synth.make_weights(d, seed=0)          # identical replicas on every rank
    eng.load_weights(w)
    dev = eng.device
    tr = Trainer(eng, dist=dp._dist)
    n_batches = 4
    batches, gts = [], []
    for i in range(n_batches):
        b = synth.make_inputs(d, seed=dp.batch_seed(i))
        batches.append({k: torch.as_tensor(v).to(dev) for k, v in b.items()})
        gts.append(synth.template_layout_batch(d, offset=i))      # host: assembled per step

    def run_steps(first, count):
        for i in range(first, first + count):
            tr.step(batches[i % n_batches], gts[i % n_batches])

    run_steps(0, args.warmup)
    elapsed = dp.timed(lambda: run_steps(args.warmup, args.steps),
                       sync=lambda: torch.cuda.synchronize(dev))
    out = None
    if rank == 0:
        out = {
            'metric': 'questions/sec (training step: forward + backward + all-reduce + Adam) on '
                      'CLEVR 10x15x512 feats, batch 64 per GPU',
            'value': round(dp.throughput(d.N * args.steps, elapsed), 1), 'unit': 'questions/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE.json configs[3]: CLEVR train_clevr_gt_layout.py step, '
                                   'gt layouts (10-template mix), batch %d per GPU, T_enc=45, '
                                   'T_dec=10, weight_decay 5e-6, clip 10, Adam' % d.N,
                       'global_batch': world * d.N,
                       'parallelism': 'dp%d: flat fp32 gradient (%d floats), 2 RCCL all-reduce '
                                      'buckets per step' % (world, tr.numel)},
            'final_total_loss': float(tr.losses[3].item()),
        }
    if rank == 0 and not args.no_profile and world == 1:
        ksteps = min(args.steps, 20)
        eng.profile_begin()
        run_steps(0, ksteps)
        rows = kernel_rows(eng.profile_end(), ksteps)
        dom = rows[0]
        out['roofline'] = {'kernel': dom['kernel'], 'bound': dom['bound'],
                           'achieved': dom['achieved'], 'peak': dom['peak'], 'unit': dom['unit'],
                           'frac': dom['frac'], 'traffic': pmc_traffic(dom['kernel'], PMC_FILE_TRAIN)[0],
                           'traffic_source': pmc_traffic(dom['kernel'], PMC_FILE_TRAIN)[1],
                           'avg_us': dom['avg_us'],
                           'measured': 'hipEvent pairs around each launch, separate pass of %d '
                                       'steps' % ksteps}
        out['kernels'] = rows
        out['gpu_us_per_step'] = round(sum(r['us_per_step'] for r in rows), 1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import n2nmn_oracle_grad as G
        b0 = synth.make_inputs(d, seed=0)
        t0 = time.perf_counter()
        G.loss_and_grads(w, names, b0, d.T_decoder, d.num_choices,
                         synth.template_layout_batch(d), 5e-6)
        dt = time.perf_counter() - t0
        out['cpu_baseline'] = {'value': round(d.N / dt, 2), 'unit': 'questions/sec',
                               'cores': int(torch.get_num_threads()), 'kind': 'port',
                               'sample': '1 batch of %d questions, forward+backward of the torch-'
                                         'autograd fp64 oracle (oracle/n2nmn_oracle_grad.py), no '
                                         'optimiser step; the reference TF1/Fold path is not '
                                         'runnable here' % d.N}
    if rank == 0:
        print(json.dumps(out), flush=True)
    dp.close()


def bench_vqa(args, dp, local_rank):
    """BASELINE.json configs[4]: models_vqa forward (exp_vqa/eval_vqa2.py:103-137) -- seq2seq with
    the 17742-word vocabulary and lstm_dim 1000, coordinate map, the 4-module network at map_dim
    1024 on 14x14x2048 features, question prior net -- batch 128 per GPU, ground-truth layouts from
    the v2 validation histogram."""
    import numpy as np
    import torch
    from n2nmn_amd import synth, vqa
    rank, world = dp.rank, dp.world
    d = vqa.VQADims(N=128 if args.batch == 64 else args.batch)
    eng = vqa.VQAEngine(d, device=local_rank)
    w = synth.make_weights_from_shapes(vqa.vqa_variable_shapes(d), seed=0)
    eng.load_weights(w)
    dev = eng.engine.device
    mix = (['_Find', '_Find', '_And', '_Describe'],) * 46 + (['_Find', '_Describe'],) * 43 + \
          (['_Find', '_Transform', '_Describe'],) * 9 + (['_Find', '_Transform', '_Find', '_And', '_Describe'],) * 2
    rng = np.random.default_rng(dp.batch_seed(0))
    batches, gts = [], []
    for i in range(3):
        lens = rng.integers(3, d.T_encoder + 1, size=d.N).astype(np.int32)
        seq = rng.integers(0, d.num_vocab_txt, size=(d.T_encoder, d.N)).astype(np.int32)
        seq[np.arange(d.T_encoder)[:, None] >= lens[None, :]] = 0
        feat = torch.relu(torch.randn((d.N, d.H, d.W, d.D), generator=torch.Generator().manual_seed(i)))
        batches.append(dict(input_seq_batch=torch.as_tensor(seq).to(dev),
                            seq_length_batch=torch.as_tensor(lens).to(dev),
                            image_feat_batch=feat.to(dev)))
        order = rng.permutation(100)
        gts.append(torch.as_tensor(np.array(
            [eng.assembler.module_list2tokens(mix[order[n % 100]], d.T_decoder) for n in range(d.N)],
            np.int32).T).to(dev))

    def run_steps(first, count):
        for i in range(first, first + count):
            eng.forward(batches[i % 3], use_gt_layout=True, gt_layout=gts[i % 3])

    run_steps(0, args.warmup)
    elapsed = dp.timed(lambda: run_steps(args.warmup, args.steps),
                       sync=lambda: torch.cuda.synchronize(dev))
    out = None
    if rank == 0:
        out = {'metric': 'questions/sec (forward) on VQAv2 14x14x2048 feats, batch %d per GPU' % d.N,
               'value': round(dp.throughput(d.N * args.steps, elapsed), 1), 'unit': 'questions/sec',
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': round(1e3 * elapsed / args.steps, 4), 'higher_is_better': True,
               'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': 'BASELINE.json configs[4]: models_vqa forward, gt layouts (v2 val '
                                      'histogram), batch %d per GPU, T_enc=26, T_dec=13, single stream'
                                      % d.N, 'global_batch': world * d.N,
                          'parallelism': 'dp%d (question-sharded)' % world}}
        if not args.no_profile:
            ksteps = min(args.steps, 10)
            eng.engine.profile_begin()
            run_steps(0, ksteps)
            rows = kernel_rows(eng.engine.profile_end(), ksteps)
            out['kernels'] = rows
            out['gpu_us_per_step'] = round(sum(r['us_per_step'] for r in rows), 1)
        print(json.dumps(out), flush=True)
    dp.close()


def main():
    args = parse()
    import numpy as np
    import torch

    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    from n2nmn_amd.dp import DataParallel
    dp = DataParallel(backend='nccl', device=torch.device('cuda', local_rank))
    rank, world = dp.rank, dp.world
    if args.config == 4:
        return bench_train(args, dp, local_rank)
    if args.config == 5:
        return bench_vqa(args, dp, local_rank)

    from n2nmn_amd import synth
    from n2nmn_amd.engine import Engine
    from n2nmn_amd.nmn3_assembler import Assembler
    from n2nmn_amd.spec import Dims, CLEVR_MODULE_NAMES

    d = Dims(N=args.batch)
    names = list(CLEVR_MODULE_NAMES)
    asm = Assembler(names)
    eng = Engine(d, asm, device=local_rank)
    w = synth.make_weights(d, seed=0)
    eng.load_weights(w)
    dev = eng.device
    # every rank streams its own questions (weak scaling); a few distinct batches are cycled so
    # the feature maps of a step are not the ones the previous step left in the caches
    n_batches = 4
    batches, gts = [], []
    for i in range(n_batches):
        b = synth.make_inputs(d, seed=dp.batch_seed(i))
        batches.append({k: torch.as_tensor(v).to(dev) for k, v in b.items()})
        gt_host = synth.template_layout_batch(d, offset=i)
        # gt layouts arrive as host arrays (the reference's data reader, data_reader.py:74-82): the
        # engine assembles the program from them while phase 1 runs -- no token fetch, no sync
        gts.append(torch.as_tensor(gt_host).to(dev) if args.fetch_tokens else gt_host)
    use_gt = args.config == 2

    S = max(1, args.streams)
    engines = [eng] + [eng.fork() for _ in range(S - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [None]

    def set_mode(mode):
        for e in engines:
            e.set_mode(mode)

    set_mode('throughput' if S > 1 else 'latency')

    def step(i, e=eng):
        b = batches[i % n_batches]
        return e.forward(b, use_gt_layout=use_gt, gt_layout=gts[i % n_batches] if use_gt else None)

    def run_steps(first, count):
        """`count` steps starting at global index `first`, spread round-robin over S workers."""
        if S == 1:
            for i in range(first, first + count):
                step(i)
            return
        import threading
        errs = []

        def worker(k):
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(streams[k]):
                    for i in range(first + k, first + count, S):
                        step(i, engines[k])
                    streams[k].synchronize()
            except Exception as ex:   # surface worker failures instead of hanging
                errs.append(ex)

        th = [threading.Thread(target=worker, args=(k,)) for k in range(S)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]

    run_steps(0, args.warmup)
    # barrier + synchronize on both sides of EXACTLY `steps` steps; max over ranks
    elapsed = dp.timed(lambda: run_steps(args.warmup, args.steps),
                       sync=lambda: torch.cuda.synchronize(dev))

    out = None
    if rank == 0:
        qps = dp.throughput(d.N * args.steps, elapsed)
        out = {
            'metric': 'questions/sec (forward) on CLEVR 10x15x512 feats, batch 64',
            'value': round(qps, 1), 'unit': 'questions/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE.json configs[%d]: CLEVR forward, %s, batch %d per GPU, '
                                   '10x15x512 synthetic pool5, T_enc=45, T_dec=20' %
                                   (args.config - 1,
                                    'fixed ground-truth layouts (10-template mix, teacher-forced '
                                    'decoder)' if use_gt else 'layouts sampled by the greedy seq2seq '
                                    'decoder', d.N),
                       'global_batch': world * d.N, 'parallelism': 'dp%d (question-sharded, no '
                       'data-path collective)' % world, 'streams_per_gpu': S,
                       'hw_queues': os.environ.get('GPU_MAX_HW_QUEUES'),
                       'lstm_tile_mode': 'throughput (32x32)' if S > 1 else 'latency (64x16)',
                       'host_sync': 'predicted_tokens D2H between phase 1 and phase 2'
                       if (not use_gt or args.fetch_tokens) else
                       'none: with gt layouts the program is assembled from the host copy of '
                       'gt_layout (= the predicted tokens, nmn3_netgen_att.py:236-238)'},
        }

    # ---- the same workload with ONE batch in flight (latency-oriented number)
    if rank == 0 and S > 1:
        S_saved, S = S, 1
        set_mode('latency')
        n1 = min(args.steps, 100)
        run_steps(0, 10)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run_steps(10, n1)
        torch.cuda.synchronize(dev)
        e1 = time.perf_counter() - t0
        S = S_saved
        out['single_stream'] = {'value': round(d.N * n1 / e1, 1), 'unit': 'questions/sec',
                                'ms_per_step': round(1e3 * e1 / n1, 4), 'steps': n1}

    # ---- per-kernel roofline: HIP events around every launch, separate pass of the same steps
    if rank == 0 and not args.no_profile:
        ksteps = min(args.steps, 50)
        eng.profile_begin()
        for i in range(ksteps):
            step(i)
        rows = kernel_rows(eng.profile_end(), ksteps)
        dom = rows[0]
        traffic, traffic_src = pmc_traffic(dom['kernel'])
        out['roofline'] = {'kernel': dom['kernel'], 'bound': dom['bound'],
                           'achieved': dom['achieved'], 'peak': dom['peak'], 'unit': dom['unit'],
                           'frac': dom['frac'], 'traffic': traffic, 'traffic_source': traffic_src,
                           'avg_us': dom['avg_us'],
                           'measured': 'hipEvent pairs around each launch on the launch stream, '
                                       'separate pass of %d steps right after the timed region; an '
                                       'event pair includes ~1.5 us of marker / dispatch latency: '
                                       'rocprofv3 --kernel-trace reports 8.6-9.1 us for this kernel '
                                       '(profiles/r01_h_kernel_stats.txt, r01_lstm_microbench.txt)'
                                       % ksteps}
        out['kernels'] = rows
        out['gpu_us_per_step'] = round(sum(r['us_per_step'] for r in rows), 1)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        b0 = synth.make_inputs(d, seed=0)
        gt0 = synth.template_layout_batch(d) if use_gt else None
        out['cpu_baseline'] = cpu_baseline(d, w, b0, gt0, names, use_gt)

    if rank == 0:
        print(json.dumps(out), flush=True)
    dp.close()


if __name__ == '__main__':
    main()
