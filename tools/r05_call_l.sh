# round 5, call l: the bf16x3 mode's error budget + the mode's tests after the variant clean-up
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r05_l
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python tests/bf16x3_error_budget.py $O/bf16x3_error_budget.json 2>&1 | grep -v amdgpu.ids | tee $O/bf16x3_error_budget.log
timeout 600 python -m pytest tests/test_gpu_lstm_tile3.py tests/test_gpu_stress_bucket.py -q -m gpu --timeout 200 2>&1 | tail -3
timeout 200 python tools/lstm_tile3_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/tile3_bench_final.log
