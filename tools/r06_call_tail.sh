# (the "head" leg needs a copy of the previous commit's source at tools/diag/csrc/*_head.*.tmp:
#  git show HEAD~1:n2nmn_amd/csrc/<file> > tools/diag/csrc/<file>_head.tmp -- not kept in the tree)
# training step: the last chunk's x-table path on the caller's stream beside the side stream's weight gradients
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06tail; mkdir -p $O
cd $R
run() {
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o n2nmn_amd/lib/libn2nmn_hip.so n2nmn_amd/lib/obj/*.o || exit 1
  for rep in 1 2 3; do
    timeout 200 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline --no-profile > $O/tb_$1_$rep.json 2> $O/tb.err
    python -c "
import json
d=json.loads(open('$O/tb_$1_$rep.json').read().strip().splitlines()[-1])
print('RESULT $1 rep $rep: ms_per_step', d.get('ms_per_step'))
" | tee -a $O/sweep.txt
  done
}
cp n2nmn_amd/lib/obj/capi_train.cpp.o /tmp/new.o
cp tools/diag/csrc/capi_train_head.cpp.tmp /tmp/capi_train_head.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wall -Wno-unused-function -I n2nmn_amd/csrc -I include -c /tmp/capi_train_head.cpp -o n2nmn_amd/lib/obj/capi_train.cpp.o || exit 1
run head
cp /tmp/new.o n2nmn_amd/lib/obj/capi_train.cpp.o
run tail_on_main
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_kernels.py tests/test_gpu_train_rl.py tests/test_gpu_vqa_train.py tests/test_gpu_train_dp.py tests/test_gpu_reference_fixture.py -x -q 2>&1 | tail -4
