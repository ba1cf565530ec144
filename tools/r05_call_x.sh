cd $GRAFT_REPO_ROOT
timeout 40 python tools/diag/three_stream_repro.py throughput_bf16x3 120 2 2>&1 | grep -v amdgpu.ids | tail -1
timeout 40 python tools/diag/three_stream_repro.py throughput_bf16x3 80 3 2>&1 | grep -v amdgpu.ids | tail -1
timeout 100 python bench.py --plain --lstm-mode throughput_bf16x3 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('bf16x3 value', o['value'], o['ms_per_step'])"
