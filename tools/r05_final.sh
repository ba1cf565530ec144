# round 5, final validation box: full GPU suite, smoke, the driver-shaped bench line
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
T=r05w
cd $R
timeout 1200 python -m pytest tests -q -m gpu --timeout 300 > $O/${T}_pytest_all.log 2>&1
grep -n "passed\|failed" $O/${T}_pytest_all.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/${T}_bench_steps20.json 2> $O/${T}_bench_steps20.err
python - <<'P'
import json
o=json.loads(open('/root/repo/gpurun_out/r05w_bench_steps20.json').read().strip().splitlines()[-1])
print(o['value'], o['ms_per_step'], o['value_min_max'], o['config']['streams_per_gpu'], 'attn', o['roofline_attention']['byte_weighted']['frac'], 'eos', o['eos_retire']['value'], o['eos_retire']['mixes']['clevr_like']['value'], o['eos_retire']['config3_passes']['value'], o['eos_retire']['config3_passes']['tokens_equal'], 'bf', o['bf16x3']['value'], 'c3', o['config3']['super_bucket']['value'], 'c5', o['config5']['passes']['value'], 'parity', o['parity_check']['max_abs_logit_err'])
P
