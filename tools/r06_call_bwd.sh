# the reverse-time step with two operand sets in flight: gradients, then the step's time and the kernel's
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06bwd; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_kernels.py tests/test_gpu_train_dp.py tests/test_gpu_train_rl.py tests/test_gpu_vqa_train.py tests/test_gpu_train_driver_trace.py -x -q 2>&1 | tail -8 > $O/train_tests.log
cat $O/train_tests.log
timeout 200 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline > $O/train_bench.json 2> $O/train_bench.err
python -c "
import json
d=json.loads(open('$O/train_bench.json').read().strip().splitlines()[-1])
print('ms_per_step', d.get('ms_per_step'), d.get('value'))
for k in d.get('kernels',[])[:6]: print(k['kernel'], k['avg_us'], k['us_per_step'], k.get('frac'))
"
timeout 200 python tools/diag/train_host_time.py 200 > $O/host_time.txt 2>&1; cat $O/host_time.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr -- python $R/bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline --no-profile > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(ls $O/tr/*/*.db | head -1) > $O/train_kernel_stats.txt; head -12 $O/train_kernel_stats.txt; rm -rf $O/tr
