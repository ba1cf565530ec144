# round 5, measurement call b: walker stage times (two find16 variants), Transform timeline, rocprofv3 kernel
# stats of a single-stream pass (cold per-kernel durations), the walker / eos_retire tests, one bench line
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r05_b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 150 python tools/walk_stage_bench.py > $O/stage_f16.log 2>&1
N2NMN_WALK_FIND16=2 timeout 120 python tools/walk_stage_bench.py > $O/stage_f16_nopf.log 2>&1
timeout 120 python tools/staged_timeline.py > $O/timeline.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/tr16 -- python $GRAFT_REPO_ROOT/bench.py --plain --streams 1 --inflight 16 --steps 12 --warmup 2 > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(ls $O/tr16/*/*.db | head -1) > $O/kernel_stats_1x16.txt; rm -rf $O/tr16)
timeout 400 python -m pytest tests/test_gpu_walker.py tests/test_gpu_eos_retire.py tests/test_gpu_superbucket.py tests/test_gpu_kernels.py -q -m gpu --timeout 240 > $O/pytest.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest.log; tail -c 400 $O/bench.err
