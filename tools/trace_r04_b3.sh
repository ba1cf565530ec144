# kernel trace of one stream x 16 client batches in the opt-in bf16x3 mode -> gpurun_out/$1_*
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
TAG=${1:-r04_b3}
rocprofv3 --kernel-trace --stats -d $O/tr_$TAG -- python /root/repo/bench.py --plain --streams 1 --inflight 16 --steps 8 --warmup 2 --lstm-mode throughput_bf16x3 > /dev/null 2>&1
DB=$(ls $O/tr_$TAG/*/*.db | head -1)
python /root/repo/tools/rocprof_summary.py $DB > $O/${TAG}_kernel_stats_1x16.txt
python /root/repo/tools/lstm_step_trace.py $DB 4 > $O/${TAG}_pass_trace_1x16.txt 2>/dev/null
rm -rf $O/tr_$TAG
head -12 $O/${TAG}_kernel_stats_1x16.txt
cd /root/repo
