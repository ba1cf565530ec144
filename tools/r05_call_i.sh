# round 5, call i: nesting bound from host layouts (no fall-back launch); Transform halves in the stage-A launch
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r05_i
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_walker.py tests/test_gpu_eos_retire.py tests/test_gpu_superbucket.py tests/test_gpu_stress_bucket.py tests/test_gpu_bench_config.py -q -m gpu --timeout 120 -x > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for TR in 0 1 2; do
N2NMN_WALK_TR_STAGE=$TR timeout 600 python -m pytest tests/test_gpu_walker.py -q -m gpu --timeout 120 -x > $O/pytest_tr$TR.log 2>&1; tail -1 $O/pytest_tr$TR.log
for M in templates clevr_like; do
echo "== tr_stage $TR $M"; N2NMN_WALK_TR_STAGE=$TR timeout 150 python tools/walk_stage_bench.py $M 2>&1 | grep -v amdgpu.ids | tee $O/stage_tr${TR}_$M.log
done; done
echo "== device layouts (no bound), tr 0"; timeout 150 python tools/walk_stage_bench.py templates devlayouts 2>&1 | grep -v amdgpu.ids | tee $O/stage_dev.log
for TR in 0 2; do
(cd /tmp && export TMPDIR=/tmp && N2NMN_WALK_TR_STAGE=$TR timeout 200 rocprofv3 --kernel-trace --stats -d $O/tr16 -- python $GRAFT_REPO_ROOT/bench.py --plain --streams 1 --inflight 16 --steps 12 --warmup 2 > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(ls $O/tr16/*/*.db | head -1) > $O/kernel_stats_1x16_tr$TR.txt; rm -rf $O/tr16)
grep -i "walk" $O/kernel_stats_1x16_tr$TR.txt
done
