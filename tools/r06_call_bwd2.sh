# (the "head" leg needs a copy of the previous commit's source at tools/diag/csrc/*_head.*.tmp -- not kept in the tree)
# reverse-time step: operand sets in flight (UN chunks per set, one or two sets), whole training step each
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06fwd1; mkdir -p $O
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
for v in 0 4 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DFWS_NS=$v -I include -c n2nmn_amd/csrc/kernels_seq2seq.hip -o n2nmn_amd/lib/obj/kernels_seq2seq.hip.o || exit 1
  run fwd_ns$v
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $O/fb_$v.json 2> $O/fb.err
  python -c "
import json
d=json.loads(open('$O/fb_$v.json').read().strip().splitlines()[-1])
print('RESULT fwd_ns$v forward value', d.get('value'), 'single_batch', (d.get('single_batch') or {}).get('ms_per_step'), (d.get('single_batch') or {}).get('value'))
"
done
