# round 5, measurement call c: walker stage times (find variants), timeline, rocprofv3 kernel stats (cold), tests
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r05_c
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 150 python tools/walk_stage_bench.py > $O/stage_f1.log 2>&1
N2NMN_WALK_FIND16=3 timeout 120 python tools/walk_stage_bench.py > $O/stage_f3.log 2>&1
timeout 120 python tools/staged_timeline.py > $O/timeline.log 2>&1
for F in 1 3; do
(cd /tmp && export TMPDIR=/tmp && N2NMN_WALK_FIND16=$F timeout 200 rocprofv3 --kernel-trace --stats -d $O/tr16 -- python $GRAFT_REPO_ROOT/bench.py --plain --streams 1 --inflight 16 --steps 12 --warmup 2 > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(ls $O/tr16/*/*.db | head -1) > $O/kernel_stats_1x16_f$F.txt; rm -rf $O/tr16)
done
timeout 400 python -m pytest tests/test_gpu_walker.py tests/test_gpu_eos_retire.py tests/test_gpu_superbucket.py tests/test_gpu_kernels.py -q -m gpu --timeout 240 > $O/pytest.log 2>&1
N2NMN_WALK_FIND16=3 timeout 300 python -m pytest tests/test_gpu_walker.py tests/test_gpu_bench_config.py -q -m gpu --timeout 240 > $O/pytest_f3.log 2>&1
tail -3 $O/pytest.log $O/pytest_f3.log
