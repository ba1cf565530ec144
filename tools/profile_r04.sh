# Round-4 measurement run (one gpurun call): PMC traffic passes (bench.py quotes the committed
# profiles/r04_pmc_traffic.json), rocprofv3 kernel traces of one stream x 16 batches (fp32 and the opt-in
# bf16x3 mode), of the default 2 x 16, of one batch in flight, of the training step and of config 5, then
# the driver-shaped bench runs.  Summaries land in gpurun_out/r04z_*; copy them into profiles/.
set -x
O=/root/repo/gpurun_out
T=r04z
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --plain --streams 1 --inflight 16 --steps 12 --warmup 2"
rocprofv3 --pmc FETCH_SIZE -d $O/${T}_fetch -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/${T}_write -- $CMD > /dev/null 2>&1
python /root/repo/tools/pmc_traffic.py $(ls $O/${T}_fetch/*/*.db | head -1) $(ls $O/${T}_write/*/*.db | head -1) $O/${T}_pmc_traffic.json > $O/${T}_pmc_traffic.txt
rocprofv3 --kernel-trace --stats -d $O/${T}_tr16 -- $CMD > /dev/null 2>&1
DB=$(ls $O/${T}_tr16/*/*.db | head -1)
python /root/repo/tools/rocprof_summary.py $DB > $O/${T}_kernel_stats_1x16.txt
python /root/repo/tools/lstm_step_trace.py $DB 5 > $O/${T}_pass_trace_1x16.txt
rocprofv3 --kernel-trace --stats -d $O/${T}_trb3 -- $CMD --lstm-mode throughput_bf16x3 > /dev/null 2>&1
python /root/repo/tools/rocprof_summary.py $(ls $O/${T}_trb3/*/*.db | head -1) > $O/${T}_bf16x3_kernel_stats_1x16.txt
CMD2="python /root/repo/bench.py --plain --streams 2 --inflight 16 --steps 20 --warmup 4"
rocprofv3 --kernel-trace --stats -d $O/${T}_tr2x16 -- $CMD2 > /dev/null 2>&1
python /root/repo/tools/rocprof_summary.py $(ls $O/${T}_tr2x16/*/*.db | head -1) > $O/${T}_kernel_stats_2x16.txt
CMD1="python /root/repo/bench.py --plain --streams 1 --inflight 1 --steps 100"
rocprofv3 --kernel-trace --stats -d $O/${T}_tr1 -- $CMD1 > /dev/null 2>&1
python /root/repo/tools/rocprof_summary.py $(ls $O/${T}_tr1/*/*.db | head -1) > $O/${T}_kernel_stats_single_batch.txt
CMD4="python /root/repo/bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline --no-profile"
rocprofv3 --kernel-trace --stats -d $O/${T}_tr4 -- $CMD4 > /dev/null 2>&1
DB4=$(ls $O/${T}_tr4/*/*.db | head -1)
python /root/repo/tools/rocprof_summary.py $DB4 > $O/${T}_train_kernel_stats.txt
python /root/repo/tools/trace_step.py $DB4 grad_sqnorm > $O/${T}_train_step_trace.txt
CMD5="python /root/repo/bench.py --config 5 --steps 24 --warmup 2 --no-profile"
rocprofv3 --kernel-trace --stats -d $O/${T}_tr5 -- $CMD5 > /dev/null 2>&1
python /root/repo/tools/rocprof_summary.py $(ls $O/${T}_tr5/*/*.db | head -1) > $O/${T}_config5_kernel_stats.txt
rm -rf $O/${T}_fetch $O/${T}_write $O/${T}_tr16 $O/${T}_trb3 $O/${T}_tr2x16 $O/${T}_tr1 $O/${T}_tr4 $O/${T}_tr5
cd /root/repo
cp $O/${T}_pmc_traffic.json /root/repo/profiles/r04_pmc_traffic.json    # (so that the bench runs below quote it)
python bench.py --config 4 --steps 100 --warmup 10 > $O/${T}_train_bench.json 2>/dev/null
python bench.py --config 5 --steps 24 --warmup 3 --no-cpu-baseline > $O/${T}_vqa_bench.json 2>/dev/null
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_torchrun1.json 2> $O/${T}_bench_torchrun1.err
python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
python bench.py --steps 20 --warmup 5 > $O/${T}_bench_steps20.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --lstm-mode throughput_bf16x3 > $O/${T}_bench_steps20_bf16x3.json 2>/dev/null
tail -c 300 $O/${T}_bench.err
ls -la $O | grep ${T}
