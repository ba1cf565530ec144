# round 5, call m2: conv_image GEMM inside the walker call, behind the text maps and right before walk_find
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r05_m
mkdir -p $O
cd $GRAFT_REPO_ROOT
for V in 0 1 2 3 0 1 2 3; do
echo "== CONV_LATE=$V"; N2NMN_CONV_LATE=$V timeout 200 python bench.py --plain --steps 40 --warmup 8 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('value', o['value'], o['ms_per_step'], o['value_min_max'])"
done
for V in 0 1 2 3; do
(cd /tmp && export TMPDIR=/tmp && N2NMN_CONV_LATE=$V timeout 200 rocprofv3 --kernel-trace --stats -d $O/tr16 -- python $GRAFT_REPO_ROOT/bench.py --plain --streams 1 --inflight 16 --steps 12 --warmup 2 > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(ls $O/tr16/*/*.db | head -1) > $O/kernel_stats_1x16_late$V.txt; rm -rf $O/tr16)
echo "== CONV_LATE=$V"; grep -i "walk_\|gemm" $O/kernel_stats_1x16_late$V.txt
done
N2NMN_CONV_LATE=3 timeout 300 python -m pytest tests/test_gpu_bench_config.py tests/test_gpu_eos_retire.py -q -m gpu --timeout 200 -x 2>&1 | tail -2
