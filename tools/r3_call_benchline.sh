cd /root/repo
mkdir -p gpurun_out/r3l
timeout 1500 python bench.py > gpurun_out/r3l/bench_default.json 2> gpurun_out/r3l/bench_default.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r3l/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
"
