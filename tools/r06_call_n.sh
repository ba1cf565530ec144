set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06n; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_kernels.py tests/test_gpu_train_dp.py tests/test_gpu_train_rl.py tests/test_gpu_vqa_train.py tests/test_gpu_reference_fixture.py tests/test_gpu_train_driver_trace.py tests/test_gpu_eos_retire.py -x -q 2>&1 | tail -8 > $O/train_tests.log
cat $O/train_tests.log
timeout 200 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline > $O/train_bench_ks.json 2> $O/train_bench_ks.err
python -c "
import json,sys
d=json.loads(open('$O/train_bench_ks.json').read().strip().splitlines()[-1])
print('ms_per_step', d.get('ms_per_step'), d.get('value'))
for k in d.get('kernels',[])[:12]: print(k)
"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr -- python $R/bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline --no-profile > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(ls $O/tr/*/*.db | head -1) > $O/train_kernel_stats.txt; head -25 $O/train_kernel_stats.txt; rm -rf $O/tr
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $O/cal -- $R/tools/microbench/mfma_peak > $O/mfma_peak.txt 2>&1
python $R/tools/pmc_mfma.py $(ls $O/cal/*/*.db | head -1) $(ls $O/cal/*/*.db | head -1) > $O/mfma_peak_pmc.txt 2>&1; cat $O/mfma_peak_pmc.txt; tail -5 $O/mfma_peak.txt; rm -rf $O/cal
