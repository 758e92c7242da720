for v in 128 192 128 192; do
  PA_GEMM_SMALL_MAX=$v timeout 600 python bench.py --steps 150 --warmup 20 --no-decode --no-cpu --no-kernels --no-f32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('small_max $v', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms')"
done
for v in 128 192; do echo "== PA_GEMM_SMALL_MAX=$v"; PA_GEMM_SMALL_MAX=$v timeout 300 python tools/gemm_small_k.py 2>&1 | grep -v amdgpu.ids | tail -2; done
