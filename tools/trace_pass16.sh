cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
rocprofv3 --kernel-trace --stats -d $O/tr1 -- python /root/repo/bench.py --plain --streams 1 --inflight 16 --steps 8 --warmup 2 > /dev/null 2>&1
DB=$(ls $O/tr1/*/*.db | head -1)
python /root/repo/tools/rocprof_summary.py $DB > $O/r03_b_kernel_stats_1x16.txt
python /root/repo/tools/lstm_step_trace.py $DB 4 > $O/r03_b_pass_trace_1x16.txt
rm -rf $O/tr1
head -16 $O/r03_b_kernel_stats_1x16.txt
awk 'NR<=3 || NR%3==0' $O/r03_b_pass_trace_1x16.txt | head -40
cd /root/repo
python bench.py --streams 2 --inflight 16 --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read())
print(o['value'], o['ms_per_step'], o['parity_check']['max_abs_logit_err'])
print(o['roofline'])
for r in o['kernels']: print(r['kernel'][:40], r['avg_us'], r['frac'], r['achieved'])
"
