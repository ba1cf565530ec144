# round 5, call g: replay of the three reference drivers, walker stage times, rocprofv3 kernel stats (cold)
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r05_g
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_eval_driver_trace.py tests/test_gpu_shapes.py tests/test_gpu_vqa.py -q -m gpu --timeout 300 > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 150 python tools/walk_stage_bench.py > $O/stage.log 2>&1
timeout 150 python tools/walk_stage_bench.py clevr_like > $O/stage_clevr_like.log 2>&1
timeout 120 python tools/staged_timeline.py > $O/timeline.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/tr16 -- python $GRAFT_REPO_ROOT/bench.py --plain --streams 1 --inflight 16 --steps 12 --warmup 2 > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(ls $O/tr16/*/*.db | head -1) > $O/kernel_stats_1x16.txt; rm -rf $O/tr16)
cat $O/stage.log
