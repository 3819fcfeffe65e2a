cd /root/repo
mkdir -p gpurun_out/r3m
timeout 1200 python -m pytest tests/test_multiline_gpu.py -m gpu -x -q -k "large" > gpurun_out/r3m/pytest_mlo2.log 2>&1
tail -15 gpurun_out/r3m/pytest_mlo2.log
python tools/perf_ml.py 2000000 cri 5 2>&1 | tail -4
