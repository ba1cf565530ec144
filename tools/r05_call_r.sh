# round 5: models_vqa resident input slab (coordinate channels written once) -- tests + config 5 numbers
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r05_r
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_vqa.py tests/test_gpu_vqa_bench_geometry.py tests/test_gpu_device_sched.py tests/test_gpu_eval_driver_trace.py -q -m gpu --timeout 300 2>&1 | tail -3
timeout 400 python bench.py --config 5 --steps 24 --warmup 3 --no-cpu-baseline > $O/vqa_bench.json 2>/dev/null
python - <<'P'
import json
o=json.loads(open('/root/repo/gpurun_out/r05_r/vqa_bench.json').read().strip().splitlines()[-1])
print(o['value'], o['ms_per_step'], {k:(v.get('value'),v.get('ms_per_step')) for k,v in o.items() if isinstance(v,dict) and 'value' in v})
for k in o.get('passes',{}).get('kernels',[])[:6]: print(k['kernel'], k['us_per_step'], k['frac'])
P
