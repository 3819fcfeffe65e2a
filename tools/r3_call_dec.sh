cd /root/repo
mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests/test_decoders_gpu.py tests/test_plugin_so.py tests/test_kv_gpu.py tests/test_json_gpu.py -m gpu -x -q > gpurun_out/r3e/pytest_dec.log 2>&1
tail -30 gpurun_out/r3e/pytest_dec.log
