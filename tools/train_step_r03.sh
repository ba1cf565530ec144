#!/bin/bash
# Round-3 training-step measurement: parity tests of the training path, then the step time with the
# side-stream overlap off / unbounded / bounded to 256..768 resident workgroups, then one step kernel by kernel (rocprofv3 --kernel-trace).
set -u
mkdir -p gpurun_out
python -m pytest tests/test_gpu_train.py tests/test_gpu_train_kernels.py tests/test_gpu_train_rl.py \
    tests/test_gpu_train_dp.py tests/test_gpu_vqa_train.py -x -q > gpurun_out/train_tests.log 2>&1
echo "tests exit $?" >> gpurun_out/train_tests.log
tail -3 gpurun_out/train_tests.log
run() {   # label, environment assignments
  local label=$1; shift
  env N2NMN_NOP=1 "$@" python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline --no-profile \
      2> gpurun_out/tb_$label.err | tail -1 > gpurun_out/tb_$label.json
  python - <<PY
import json
d = json.load(open('gpurun_out/tb_$label.json'))
print('$label', d['ms_per_step'], 'ms', d['value'], d['unit'])
PY
}
run default
run sched0 N2NMN_TRAIN_SCHEDULE=0 N2NMN_TRAIN_CHUNKS=0
run default_again
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p4
N2NMN_TRAIN_SCHEDULE=${TRACE_SCHEDULE:-1} rocprofv3 --kernel-trace -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/bench.py --config 4 --steps 20 --warmup 3 \
    --no-cpu-baseline --no-profile > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
db=$(find /tmp/p4 -name '*.db' | head -1)
python tools/trace_step.py $db grad_sqnorm > gpurun_out/train_step_trace.txt
python tools/rocprof_summary.py $db > gpurun_out/train_kernel_stats.txt 2>&1
tail -1 gpurun_out/train_step_trace.txt
head -12 gpurun_out/train_kernel_stats.txt
