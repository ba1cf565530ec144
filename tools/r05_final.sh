# round 5, final validation box: full GPU suite, smoke, the driver-shaped bench lines, config 5 trace
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
T=r05y
cd $R
timeout 1200 python -m pytest tests -q -m gpu --timeout 300 > $O/${T}_pytest_all.log 2>&1
grep -n "passed\|failed" $O/${T}_pytest_all.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
/usr/bin/time -v timeout 900 python bench.py --steps 20 --warmup 5 > $O/${T}_bench_steps20.json 2> $O/${T}_bench_steps20.err
grep "Elapsed (wall" $O/${T}_bench_steps20.err
timeout 400 python bench.py --config 5 --steps 24 --warmup 3 --no-cpu-baseline > $O/${T}_vqa_bench.json 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${T}_tr -- python $R/bench.py --config 5 --steps 24 --warmup 2 --no-profile > /dev/null 2>&1; python $R/tools/rocprof_summary.py $(ls $O/${T}_tr/*/*.db | head -1) > $O/${T}_config5_kernel_stats.txt; rm -rf $O/${T}_tr)
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/${T}_tr -- python $R/bench.py --plain --config 3 --streams 1 --inflight 16 --steps 12 --warmup 2 --eos-retire > /dev/null 2>&1; python $R/tools/rocprof_summary.py $(ls $O/${T}_tr/*/*.db | head -1) > $O/${T}_config3_eos_retire_kernel_stats.txt; rm -rf $O/${T}_tr)
head -14 $O/${T}_config3_eos_retire_kernel_stats.txt
