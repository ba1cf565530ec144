# round 5, call j: full GPU suite, bench line, rocprofv3 kernel stats (cold) after the stage regrouping + nesting bound
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r05_j
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu --timeout 300 > $O/pytest_all.log 2>&1
tail -5 $O/pytest_all.log
timeout 150 python tools/walk_stage_bench.py templates 2>&1 | grep -v amdgpu.ids | tee $O/stage.log
timeout 150 python tools/walk_stage_bench.py clevr_like 2>&1 | grep -v amdgpu.ids | tee $O/stage_clevr_like.log
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/tr16 -- python $GRAFT_REPO_ROOT/bench.py --plain --streams 1 --inflight 16 --steps 12 --warmup 2 > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(ls $O/tr16/*/*.db | head -1) > $O/kernel_stats_1x16.txt; rm -rf $O/tr16)
grep -i "walk" $O/kernel_stats_1x16.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.err
python - <<'P'
import json
o=json.loads(open('/root/repo/gpurun_out/r05_j/bench.json').read().strip().splitlines()[-1])
print(o['value'], o['roofline'])
print(o['roofline_attention']['byte_weighted'])
for k in o['roofline_attention']['kernels']: print(k['kernel'][:40], k.get('avg_us'), k.get('event_pair_us'), k.get('frac'))
print({k: (v.get('value') if isinstance(v, dict) else v) for k, v in o.items() if k in ('bf16x3','eos_retire','config4','config5','single_batch')})
P
