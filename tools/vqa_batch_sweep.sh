#!/bin/bash
# models_vqa forward (bench.py --config 5) at 128 .. 1024 questions per launch, both recurrent-step modes
for b in 128 256 512 1024; do
  for m in latency throughput; do
    N2NMN_VQA_MODE=$m python bench.py --config 5 --batch $b --steps 20 --warmup 3 --no-cpu-baseline \
        2> gpurun_out/vq_${b}_$m.err | tail -1 > gpurun_out/vq_${b}_$m.json
    python - <<PY
import json
try:
    d = json.load(open('gpurun_out/vq_${b}_$m.json'))
    print('batch $b $m', d['ms_per_step'], 'ms', d['value'], 'q/s  gpu_us', d.get('gpu_us_per_step'))
    if '$b' in ('128', '512') :
        for r in d.get('kernels', [])[:9]:
            print('     %-44s %9.1f us  x%5.1f' % (r['kernel'][:44], r['us_per_step'], r.get('launches_per_step', 0)))
except Exception as e:
    print('batch $b $m failed', e)
    import subprocess; print(subprocess.run(['tail', '-5', 'gpurun_out/vq_${b}_$m.err'], capture_output=True, text=True).stdout)
PY
  done
done
