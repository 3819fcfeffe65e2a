# final measurement of round 3: kernel stats + FETCH / WRITE passes of the bench command, then the full GPU suite, smoke() and the default bench line
cd /root/repo
SKIP_CAL=1 bash tools/profile_bench.sh > gpurun_out/profile_run.log 2>&1
tail -25 gpurun_out/profile_run.log | cut -c1-400
O=/root/repo/gpurun_out/r3k
rm -rf $O; mkdir -p $O
cd /root/repo
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
