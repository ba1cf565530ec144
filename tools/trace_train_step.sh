cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p4
rocprofv3 --kernel-trace -d /tmp/p4 -o t -- python $GRAFT_REPO_ROOT/bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline --no-profile > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
db=$(find /tmp/p4 -name '*.db' | head -1)
python tools/trace_step.py $db grad_sqnorm > gpurun_out/train_step_trace2.txt
head -40 gpurun_out/train_step_trace2.txt
tail -3 gpurun_out/train_step_trace2.txt
