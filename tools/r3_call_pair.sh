cd /root/repo
mkdir -p gpurun_out/r3i
timeout 1500 python -m pytest tests/test_tile_gpu.py -m gpu -x -q > gpurun_out/r3i/pytest_pair.log 2>&1
tail -8 gpurun_out/r3i/pytest_pair.log
for v in 0 1; do
  if [ $v = 1 ]; then export FLBGPU_NO_PAIR2=1; fi; python bench.py --no-cpu --no-secondary --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('NO_PAIR2=$v', 'value', d['value'], 'ms', d['ms_per_step'], 'kernel', d['roofline']['kernel'], d['roofline']['avg_launch_ms'], 'verify', (d.get('verify') or {}).get('fused_equals_unfused'), (d.get('verify') or {}).get('oracle_sample_matches'))"
done
