# Round-3 measurement run (one gpurun call): PMC traffic passes first (bench.py quotes the committed
# profiles/r03_pmc_traffic.json), the rocprofv3 kernel traces of one stream x 16 batches and of the
# default 2 x 16, then the default bench line and the driver-shaped run.  Summaries land in
# gpurun_out/r03_*; copy them into profiles/.
set -x
cd /root/repo
O=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --plain --streams 1 --inflight 16 --steps 12 --warmup 2"
rocprofv3 --pmc FETCH_SIZE -d $O/r03_fetch -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/r03_write -- $CMD > /dev/null 2>&1
python /root/repo/tools/pmc_traffic.py $(ls $O/r03_fetch/*/*.db | head -1) $(ls $O/r03_write/*/*.db | head -1) $O/r03_pmc_traffic.json
cp $O/r03_pmc_traffic.json /root/repo/profiles/r03_pmc_traffic.json
rocprofv3 --kernel-trace --stats -d $O/r03_tr16 -- $CMD > /dev/null 2>&1
DB=$(ls $O/r03_tr16/*/*.db | head -1)
python /root/repo/tools/rocprof_summary.py $DB > $O/r03_kernel_stats_1x16.txt
python /root/repo/tools/lstm_step_trace.py $DB 5 > $O/r03_pass_trace_1x16.txt
CMD2="python /root/repo/bench.py --plain --streams 2 --inflight 16 --steps 20 --warmup 4"
rocprofv3 --kernel-trace --stats -d $O/r03_tr2x16 -- $CMD2 > /dev/null 2>&1
python /root/repo/tools/rocprof_summary.py $(ls $O/r03_tr2x16/*/*.db | head -1) > $O/r03_kernel_stats_2x16.txt
CMD1="python /root/repo/bench.py --plain --streams 1 --inflight 1 --steps 100"
rocprofv3 --kernel-trace --stats -d $O/r03_tr1 -- $CMD1 > /dev/null 2>&1
python /root/repo/tools/rocprof_summary.py $(ls $O/r03_tr1/*/*.db | head -1) > $O/r03_kernel_stats_single_batch.txt
# training step (BASELINE configs[3]): kernel summary and one step in start order with streams
CMD4="python /root/repo/bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline --no-profile"
rocprofv3 --kernel-trace --stats -d $O/r03_tr4 -- $CMD4 > /dev/null 2>&1
DB4=$(ls $O/r03_tr4/*/*.db | head -1)
python /root/repo/tools/rocprof_summary.py $DB4 > $O/r03_train_kernel_stats.txt
python /root/repo/tools/trace_step.py $DB4 grad_sqnorm > $O/r03_train_step_trace.txt
rm -rf $O/r03_fetch $O/r03_write $O/r03_tr16 $O/r03_tr2x16 $O/r03_tr1 $O/r03_tr4
cd /root/repo
python bench.py --config 4 --steps 100 --warmup 10 > $O/r03_train_bench.json 2>/dev/null
python bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline > $O/r03_vqa_bench.json 2>/dev/null
# the driver's multi-GPU launch form, one rank (full line incl. config4 / config5)
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03_bench_torchrun1.json 2> $O/r03_bench_torchrun1.err
tail -c 400 $O/r03_bench_torchrun1.json
python bench.py > $O/r03_bench.json 2> $O/r03_bench.err
python bench.py --steps 20 --warmup 5 > $O/r03_bench_steps20.json 2>/dev/null
tail -c 300 $O/r03_bench.err
head -14 $O/r03_kernel_stats_1x16.txt
head -8 $O/r03_kernel_stats_single_batch.txt
cat $O/r03_pmc_traffic.json | head -60
