# MFMA hardware counters of the forward pass (1 stream x 16 batches) and of the training step: two --pmc passes each
# (busy cycles; MOPS), no trace domains beside the counters.  Summaries: gpurun_out/$T_pmc_mfma*.{json,txt}
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
T=${1:-r06}
cd /tmp && export TMPDIR=/tmp
FWD="python $R/bench.py --plain --streams 1 --inflight 16 --steps 12 --warmup 2"
TRN="python $R/bench.py --config 4 --steps 12 --warmup 3 --no-cpu-baseline --no-profile"
for what in fwd trn; do
  if [ $what = fwd ]; then CMD=$FWD; else CMD=$TRN; fi
  timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $O/${T}_pb_$what -- $CMD > /dev/null 2> $O/${T}_pb_$what.err
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $O/${T}_pm_$what -- $CMD > /dev/null 2> $O/${T}_pm_$what.err
  python $R/tools/pmc_mfma.py $(ls $O/${T}_pb_$what/*/*.db | head -1) $(ls $O/${T}_pm_$what/*/*.db | head -1) $O/${T}_pmc_mfma_$what.json > $O/${T}_pmc_mfma_$what.txt 2>&1
  tail -3 $O/${T}_pb_$what.err $O/${T}_pm_$what.err
  rm -rf $O/${T}_pb_$what $O/${T}_pm_$what
done
cat $O/${T}_pmc_mfma_fwd.txt $O/${T}_pmc_mfma_trn.txt
