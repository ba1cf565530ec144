cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
for cfg in "512 24" "512 34" "256 24" "256 34"; do
  set -- $cfg
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $O/pmc_$1_$2 -- python /root/repo/tools/lstm_tile_probe.py $1 $2 50 > /dev/null 2>&1
  echo "== N=$1 variant $2"
  python /root/repo/tools/pmc_sq.py $(ls $O/pmc_$1_$2/*/*.db | head -1) | grep lstm
  rm -rf $O/pmc_$1_$2
done
