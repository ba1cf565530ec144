# kernel trace of BASELINE configs[4] (models_vqa): single batches of 128 and passes of 8 x 128, layouts as
# host arrays and as device tokens -> gpurun_out/$1_config5_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out
TAG=${1:-r04}
rocprofv3 --kernel-trace --stats -d $O/tr5_$TAG -- python /root/repo/bench.py --config 5 --steps 24 --warmup 2 --no-profile > $O/${TAG}_config5_bench.json 2>/dev/null
DB=$(ls $O/tr5_$TAG/*/*.db | head -1)
python /root/repo/tools/rocprof_summary.py $DB > $O/${TAG}_config5_kernel_stats.txt
rm -rf $O/tr5_$TAG
head -40 $O/${TAG}_config5_kernel_stats.txt
cd /root/repo
