cd /root/repo
mkdir -p gpurun_out/r3h
echo "--- tile"; python tools/perf_json.py 4000000 2>&1 | head -3
echo "--- direct"; FLBGPU_JSON_DIRECT=1 python tools/perf_json.py 4000000 2>&1 | head -3
