set -x
cd /root/repo
python bench.py > gpurun_out/r02_p_bench.json 2> gpurun_out/r02_p_bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_p_bench_driver_like.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
CMD="python /root/repo/bench.py --plain --streams 1 --inflight 8 --steps 192"
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02_p_trace -- $CMD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d /root/repo/gpurun_out/r02_p_fetch -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /root/repo/gpurun_out/r02_p_write -- $CMD > /dev/null 2>&1
CMD1="python /root/repo/bench.py --plain --streams 1 --inflight 1 --steps 200"
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02_p_trace1 -- $CMD1 > /dev/null 2>&1
cd /root/repo
ls gpurun_out/r02_p_*/*/ | head -20
