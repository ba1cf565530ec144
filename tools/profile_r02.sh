# Round-2 measurement run (one gpurun call): PMC traffic passes first (bench.py quotes the committed
# profiles/r02_pmc_traffic.json), then the default bench line, the driver-shaped run, and the two
# rocprofv3 kernel traces.  Summaries land in gpurun_out/r02_q_*; copy them into profiles/.
set -x
cd /root/repo
O=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --plain --streams 1 --inflight 8 --steps 192"
rocprofv3 --pmc FETCH_SIZE -d $O/r02_q_fetch -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/r02_q_write -- $CMD > /dev/null 2>&1
python /root/repo/tools/pmc_traffic.py $(ls $O/r02_q_fetch/*/*.db | head -1) $(ls $O/r02_q_write/*/*.db | head -1) $O/r02_q_pmc_traffic.json
cp $O/r02_q_pmc_traffic.json /root/repo/profiles/r02_pmc_traffic.json
rocprofv3 --kernel-trace --stats -d $O/r02_q_trace8 -- $CMD > /dev/null 2>&1
python /root/repo/tools/rocprof_summary.py $(ls $O/r02_q_trace8/*/*.db | head -1) > $O/r02_q_kernel_stats_inflight8.txt
CMD1="python /root/repo/bench.py --plain --streams 1 --inflight 1 --steps 200"
rocprofv3 --kernel-trace --stats -d $O/r02_q_trace1 -- $CMD1 > /dev/null 2>&1
python /root/repo/tools/rocprof_summary.py $(ls $O/r02_q_trace1/*/*.db | head -1) > $O/r02_q_kernel_stats_single_batch.txt
rm -rf $O/r02_q_fetch $O/r02_q_write $O/r02_q_trace8 $O/r02_q_trace1
cd /root/repo
python bench.py > $O/r02_q_bench.json 2> $O/r02_q_bench.err
python bench.py --steps 20 --warmup 5 > $O/r02_q_bench_steps20.json 2>/dev/null
tail -c 300 $O/r02_q_bench.err
head -12 $O/r02_q_kernel_stats_inflight8.txt
head -10 $O/r02_q_kernel_stats_single_batch.txt
