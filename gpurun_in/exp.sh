for s in 0 1 2 3 7; do echo "skip=$s"; FLBGPU_DEBUG_SKIP=$s python bench.py --steps 3 --records 10000000 --no-cpu --no-secondary 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['ms_per_step'],2), {k: round(v['total_ms']/v['launches'],2) for k,v in d['kernels'].items()})"; done
