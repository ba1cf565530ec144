# round 5, call n: conv_image inside the walker call (default for passes) -- tests + bench line
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r05_n
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu --timeout 300 -x > $O/pytest_all.log 2>&1
grep -n "passed\|failed" $O/pytest_all.log | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.err
python - <<'P'
import json
o=json.loads(open('/root/repo/gpurun_out/r05_n/bench.json').read().strip().splitlines()[-1])
print(o['value'], o['roofline']['frac'], o['roofline']['kernel'])
print(o['roofline_attention']['byte_weighted'])
for k in o['roofline_attention']['kernels']: print(k['kernel'][:40], k.get('avg_us'), k.get('event_pair_us'), k.get('frac'))
print({k: (v.get('value') if isinstance(v, dict) else v) for k, v in o.items() if k in ('bf16x3','eos_retire','config4','config5','single_batch')})
print('config3', o['config3']['single_batch'], o['config3']['super_bucket'])
for k in o.get('kernels', [])[:8]: print(k['kernel'], k['avg_us'], k['frac'], k.get('launches_per_step'))
P
