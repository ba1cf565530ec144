#!/bin/bash
# models_vqa training step (bench.py --config 6) under the training-schedule switches
run() {
  local label=$1; shift
  env N2NMN_NOP=1 "$@" python bench.py --config 6 --steps 30 --warmup 4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/vt_$label.json
  python -c "
import json; d=json.load(open('gpurun_out/vt_$label.json')); print('$label', d['ms_per_step'], 'ms', d['value'])"
}
run default
run sched0 N2NMN_TRAIN_SCHEDULE=0
run chunks0 N2NMN_TRAIN_CHUNKS=0
run bg0 N2NMN_TRAIN_BG_WGS=0
run dma_all N2NMN_GEMM_DMA_MIN_TILES=0
run dma_off N2NMN_GEMM_DMA=0
run sched0_bg0 N2NMN_TRAIN_SCHEDULE=0 N2NMN_TRAIN_BG_WGS=0
run default_again
