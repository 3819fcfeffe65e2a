# tools/ab_fx.sh -- on the GPU box: k_parser_reg with the four-port pair tables (FLBGPU_FX=4), the three-port ones (default, round 5)
# and the three-port build with hand-spelled high-half stores (FLBGPU_FX5_ASM=1), same box, same data; then the tile / parity tests
cd /root/repo
for rep in 1 2; do
echo "== fx4";        FLBGPU_FX=4 python tools/perf_reg.py 10000000 12 0
echo "== fx5";        python tools/perf_reg.py 10000000 12 0
echo "== fx5 asm hi"; FLBGPU_FX5_ASM=1 python tools/perf_reg.py 10000000 12 0
done
python -m pytest tests/test_tile_gpu.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
FLBGPU_FX5_ASM=1 python -m pytest tests/test_tile_gpu.py -m gpu -x -q 2>&1 | tail -3
