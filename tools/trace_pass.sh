cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
rocprofv3 --kernel-trace --stats -d $O/tr1 -- python /root/repo/bench.py --plain --streams 1 --inflight 8 --steps 12 --warmup 3 > /dev/null 2>&1
DB=$(ls $O/tr1/*/*.db | head -1)
python /root/repo/tools/rocprof_summary.py $DB > $O/r03_a_kernel_stats_1x8.txt
python /root/repo/tools/lstm_step_trace.py $DB 5 > $O/r03_a_pass_trace_1x8.txt
rm -rf $O/tr1
head -14 $O/r03_a_kernel_stats_1x8.txt
cat $O/r03_a_pass_trace_1x8.txt | head -80
