P='import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(round(d["value"]/1e6,1), {k:round(v["total_ms"]/v["launches"],2) for k,v in d["kernels"].items()})'
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 3 --no-cpu 2>&1 | python -c "$P"
