# reverse-time step: operand sets in flight (UN chunks per set, one or two sets), whole training step each
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06bwd9; mkdir -p $O
cd $R
run() {
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o n2nmn_amd/lib/libn2nmn_hip.so n2nmn_amd/lib/obj/*.o || exit 1
  for rep in 1 2; do
    timeout 200 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline --no-profile > $O/tb_$1_$rep.json 2> $O/tb.err
    python -c "
import json
d=json.loads(open('$O/tb_$1_$rep.json').read().strip().splitlines()[-1])
print('RESULT $1 rep $rep: ms_per_step', d.get('ms_per_step'))
" | tee -a $O/sweep.txt
  done
}
cp tools/diag/csrc/kernels_train_head.hip.tmp /tmp/kernels_train_head.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I n2nmn_amd/csrc -I include -c /tmp/kernels_train_head.hip -o n2nmn_amd/lib/obj/kernels_train.hip.o || exit 1
run head
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I include -c n2nmn_amd/csrc/kernels_train.hip -o n2nmn_amd/lib/obj/kernels_train.hip.o || exit 1
run ring_nact_late
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_kernels.py tests/test_gpu_train_rl.py tests/test_gpu_vqa_train.py -x -q 2>&1 | tail -4
