cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py > $R/gpurun_out/bench_r1.json 2> $R/gpurun_out/bench_r1.err
tail -1 $R/gpurun_out/bench_r1.json | cut -c1-600
rm -rf $R/gpurun_out/q_*
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/q_trace -- python $R/bench.py --no-cpu > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/q_fetch -- python $R/bench.py --no-cpu --steps 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/q_write -- python $R/bench.py --no-cpu --steps 2 > /dev/null 2>&1
find $R/gpurun_out -name "*.csv" -newer $R/gpurun_out/bench_r1.json | head -20
