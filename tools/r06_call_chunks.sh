set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06chunks; mkdir -p $O
cd $R
for k in train_chunks=33 train_chunks=25 train_chunks=20 train_chunks=40 train_chunks=50,20 train_chunks=33 train_bg_wgs=512 train_bg_wgs=1024 train_bg_wgs=0 train_chunks=25; do
  timeout 200 python tools/diag/train_host_time.py 300 $k 2>&1 | grep knobs | tee -a $O/chunks.txt
done
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_kernels.py tests/test_gpu_train_rl.py tests/test_gpu_vqa_train.py tests/test_gpu_train_dp.py tests/test_gpu_reference_fixture.py -x -q 2>&1 | grep "passed\|failed\|rror" | tail -4
