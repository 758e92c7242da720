# Round 6: 192 x 128 "tall" tiles for the N = 512 Linears whose 128 x 128 tiling is a few units past one round (PA_GEMM_TALL=1, default)
# against the two-blocks-per-CU kernel (PA_GEMM_TALL=0).  Same session, alternating.
OUT=gpurun_out/tall; mkdir -p $OUT
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | grep -E "passed|failed|rror" | tail -3
for v in 1 0 1 0; do echo "== PA_GEMM_TALL=$v"; PA_GEMM_TALL=$v timeout 300 python tools/gemm_tiles.py 2>&1 | grep -v amdgpu.ids | tail -6; done 2>&1 | tee $OUT/tiles.log
for v in 1 0 1 0; do
  PA_GEMM_TALL=$v timeout 600 python bench.py --steps 150 --warmup 20 --no-decode --no-cpu --no-kernels --no-f32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tall $v', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms')"
done 2>&1 | tee $OUT/step.log
