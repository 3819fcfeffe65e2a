B='python bench.py --records 1000000 --steps 3 --no-cpu'
P='import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(round(d["value"]/1e6,1), {k:round(v["total_ms"]/v["launches"],2) for k,v in d["kernels"].items()})'
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
for s in 0 1 2 4; do echo "skip=$s"; FLBGPU_DEBUG_SKIP=$s $B 2>&1 | python -c "$P"; done
python bench.py --steps 3 --no-cpu 2>&1 | python -c "$P"
