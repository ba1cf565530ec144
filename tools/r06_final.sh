# round-6 closing call: the GPU suite, then the bench lines that profiles/r06_bench*.json hold
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06final; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_suite.log
cat $O/gpu_suite.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err
timeout 600 python bench.py --steps 20 --warmup 5 --lstm-mode throughput_bf16x3 --no-cpu-baseline > $O/bench_steps20_bf16x3.json 2> $O/bench_bf16x3.err
for f in bench bench_steps20 bench_torchrun1 bench_steps20_bf16x3; do python - <<PY
import json
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
    print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('roofline_attention',{}).get('byte_weighted',{}).get('frac'), (d.get('cpu_baseline') or {}).get('value'))
except Exception as e:
    print('$f', 'FAILED', e)
PY
done
