# tools/ab_fx.sh -- on the GPU box: k_parser_reg A/B runs of the round (same box, same data): skip masks of tools/perf_reg.py
# (1 walk, 2 value loads, 4 everything behind the walk, 16 rules, 32 time, 64 time text, 256 no rule prefetch), table forms
cd /root/repo
python -m pytest tests/test_tile_gpu.py -m gpu -x -q 2>&1 | tail -5
for rep in 1 2; do
echo "== fx5";        python tools/perf_reg.py 10000000 12 0,256,16,64,96,4,1
echo "== fx4";        FLBGPU_FX=4 python tools/perf_reg.py 10000000 12 0
done
