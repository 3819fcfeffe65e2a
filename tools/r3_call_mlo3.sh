cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r3m
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cri_stats -- python $R/tools/perf_ml.py 2000000 cri 3 > $O/perf_cri_prof.log 2>&1
find $O/cri_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/cri_kernel_stats.csv
rm -rf $O/cri_stats
head -14 $O/cri_kernel_stats.csv | cut -c1-130
