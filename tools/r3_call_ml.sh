cd /root/repo
mkdir -p gpurun_out/r3f
timeout 1200 python -m pytest tests/test_multiline_gpu.py -m gpu -x -q > gpurun_out/r3f/pytest_ml.log 2>&1
tail -40 gpurun_out/r3f/pytest_ml.log
