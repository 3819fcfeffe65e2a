# tools/profile_index.sh -- on the GPU box: rocprofv3 kernel stats of the device record indexer
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/profile_index
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/perf_index.py ${1:-4000000} > $O/run.log 2>/dev/null
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
tail -1 $O/run.log
head -14 $O/kernel_stats.csv | cut -c1-150
