#!/bin/bash
# greedy-decoder passes (bench.py --config 3): kernel summary and one pass in start order
O=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/c3tr
rocprofv3 --kernel-trace --stats -d $O/c3tr -- python /root/repo/bench.py --config 3 --plain --streams 1 --inflight 16 --steps 8 --warmup 2 > /dev/null 2>&1
DB=$(ls $O/c3tr/*/*.db | head -1)
python /root/repo/tools/rocprof_summary.py $DB > $O/c3_kernel_stats.txt
python /root/repo/tools/trace_step.py $DB enc_prepare > $O/c3_pass_trace.txt
rm -rf $O/c3tr
head -24 $O/c3_kernel_stats.txt
cd /root/repo
python bench.py --config 3 --plain --steps 20 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config3 2x16', d['value'], d['ms_per_step'])"
python bench.py --config 3 --plain --streams 1 --steps 20 --warmup 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config3 1x16', d['value'], d['ms_per_step'])"
