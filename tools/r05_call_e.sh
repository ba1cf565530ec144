# round 5, measurement call e: single-launch staged walker
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r05_e
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_walker.py -q -m gpu --timeout 120 -x > $O/pytest_walker.log 2>&1
tail -5 $O/pytest_walker.log
timeout 150 python tools/walk_stage_bench.py > $O/stage.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/tr16 -- python $GRAFT_REPO_ROOT/bench.py --plain --streams 1 --inflight 16 --steps 12 --warmup 2 > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(ls $O/tr16/*/*.db | head -1) > $O/kernel_stats_1x16.txt; rm -rf $O/tr16)
timeout 400 python -m pytest tests/test_gpu_eos_retire.py tests/test_gpu_superbucket.py tests/test_gpu_bench_config.py -q -m gpu --timeout 240 > $O/pytest.log 2>&1
tail -3 $O/pytest.log
