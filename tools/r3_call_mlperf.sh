cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r3f
mkdir -p $O
LINES=${ML_LINES:-2000000}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ml_stats -- python $R/tools/perf_ml.py $LINES java 8 > $O/perf_ml_prof.log 2>&1
find $O/ml_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/ml_kernel_stats.csv
rm -rf $O/ml_stats
python $R/tools/perf_ml.py $LINES java 8 > $O/perf_ml.log 2>&1
cat $O/perf_ml.log
head -30 $O/ml_kernel_stats.csv | cut -c1-150
