# round 5, validation: full GPU suite + smoke
set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r05_q
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu --timeout 300 > $O/pytest_all.log 2>&1
grep -n "passed\|failed" $O/pytest_all.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
