cd /root/repo
mkdir -p gpurun_out/r3m
timeout 1200 python -m pytest tests/test_multiline_gpu.py -m gpu -x -q  > gpurun_out/r3m/pytest_mlo.log 2>&1
tail -40 gpurun_out/r3m/pytest_mlo.log
