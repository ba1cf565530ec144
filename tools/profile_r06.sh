# Round-6 measurement run (one gpurun call): PMC traffic passes (bench.py quotes the committed
# profiles/r06_pmc_traffic.json), rocprofv3 kernel traces of one stream x 16 batches (fp32, with eos_retire on
# both layout mixes, the opt-in bf16x3 mode), of the default 2 x 16, of one batch in flight, of the training step
# and of config 5, the walker stage replays, then the driver-shaped bench runs.  Summaries land in
# gpurun_out/r06z_*; copy them into profiles/.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
T=r06z
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --plain --streams 1 --inflight 16 --steps 12 --warmup 2"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/${T}_fetch -- $CMD > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/${T}_write -- $CMD > /dev/null 2>&1
python $R/tools/pmc_traffic.py $(ls $O/${T}_fetch/*/*.db | head -1) $(ls $O/${T}_write/*/*.db | head -1) $O/${T}_pmc_traffic.json > $O/${T}_pmc_traffic.txt
trace() {   # name, command...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/${T}_tr -- "$@" > /dev/null 2>&1
  DB=$(ls $O/${T}_tr/*/*.db | head -1)
  python $R/tools/rocprof_summary.py $DB > $O/${T}_$name.txt
}
trace kernel_stats_1x16 $CMD
python $R/tools/lstm_step_trace.py $DB 5 > $O/${T}_pass_trace_1x16.txt
rm -rf $O/${T}_tr
trace eos_retire_kernel_stats_1x16 $CMD --eos-retire; rm -rf $O/${T}_tr
trace eos_retire_clevr_like_kernel_stats_1x16 $CMD --eos-retire --layouts clevr_like; rm -rf $O/${T}_tr
trace clevr_like_kernel_stats_1x16 $CMD --layouts clevr_like; rm -rf $O/${T}_tr
trace bf16x3_kernel_stats_1x16 $CMD --lstm-mode throughput_bf16x3; rm -rf $O/${T}_tr
trace kernel_stats_2x16 python $R/bench.py --plain --streams 2 --inflight 16 --steps 20 --warmup 4; rm -rf $O/${T}_tr
trace kernel_stats_single_batch python $R/bench.py --plain --streams 1 --inflight 1 --steps 100; rm -rf $O/${T}_tr
trace train_kernel_stats python $R/bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline --no-profile
python $R/tools/trace_step.py $DB grad_sqnorm > $O/${T}_train_step_trace.txt
rm -rf $O/${T}_tr
trace config5_kernel_stats python $R/bench.py --config 5 --steps 24 --warmup 2 --no-profile; rm -rf $O/${T}_tr
rm -rf $O/${T}_fetch $O/${T}_write
# hardware MFMA counters: busy cycles and MOPS in separate passes, forward pass and training step
TRN="python $R/bench.py --config 4 --steps 12 --warmup 3 --no-cpu-baseline --no-profile"
for what in fwd trn; do
  if [ $what = fwd ]; then C2="$CMD"; else C2="$TRN"; fi
  timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $O/${T}_pb -- $C2 > /dev/null 2>&1
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $O/${T}_pm -- $C2 > /dev/null 2>&1
  python $R/tools/pmc_mfma.py $(ls $O/${T}_pb/*/*.db | head -1) $(ls $O/${T}_pm/*/*.db | head -1) $O/${T}_pmc_mfma_$what.json > $O/${T}_pmc_mfma_$what.txt 2>&1
  rm -rf $O/${T}_pb $O/${T}_pm
done
trace config3_kernel_stats python $R/bench.py --config 3 --plain --streams 1 --inflight 16 --steps 12 --warmup 2; rm -rf $O/${T}_tr
cd $R
cp $O/${T}_pmc_mfma_fwd.json $R/profiles/r06_pmc_mfma_fwd.json; cp $O/${T}_pmc_mfma_trn.json $R/profiles/r06_pmc_mfma_trn.json
cp $O/${T}_pmc_traffic.json $R/profiles/r06_pmc_traffic.json    # (so that the bench runs below quote it)
timeout 150 python tools/walk_stage_bench.py templates 2>&1 | grep -v amdgpu.ids > $O/${T}_walk_stage_templates.txt
timeout 150 python tools/walk_stage_bench.py clevr_like 2>&1 | grep -v amdgpu.ids > $O/${T}_walk_stage_clevr_like.txt
timeout 200 python bench.py --config 4 --steps 100 --warmup 10 > $O/${T}_train_bench.json 2>/dev/null
timeout 300 python bench.py --config 5 --steps 24 --warmup 3 --no-cpu-baseline > $O/${T}_vqa_bench.json 2>/dev/null
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench_torchrun1.json 2> $O/${T}_bench_torchrun1.err
timeout 300 python bench.py --steps 20 --warmup 5 --lstm-mode throughput_bf16x3 --no-cpu-baseline > $O/${T}_bench_steps20_bf16x3.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $O/${T}_bench_steps20.json 2> $O/${T}_bench_steps20.err
timeout 900 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err
ls -la $O | grep $T
