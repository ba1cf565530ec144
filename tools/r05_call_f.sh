# round 5, call f: the whole GPU suite on the current tree, training-step sweep (side-stream CU mask), bench line
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r05_f
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu --timeout 300 > $O/pytest_all.log 2>&1
tail -5 $O/pytest_all.log
for CUS in 0 32 64 128 192; do
  N2NMN_TRAIN_SIDE_CUS=$CUS timeout 120 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('side_cus=$CUS', o['value'], o['ms_per_step'])" >> $O/train_sweep.log 2>&1
done
for CH in 33 40,10 50,20 25; do
  N2NMN_TRAIN_CHUNKS=$CH timeout 120 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read()); print('chunks=$CH', o['value'], o['ms_per_step'])" >> $O/train_sweep.log 2>&1
done
cat $O/train_sweep.log
timeout 500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.err
