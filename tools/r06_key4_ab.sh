# A/B of the 16-row-wave kernels' LDS key in ONE session: tools/ubench/libplank_key4old.so (built with -DPA_KEY4_OLD) against the product
# library - stand-alone launches (tools/attn_balance.py) and the attention launches inside the train step (bench.py kernel_census).
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r06d
for rep in 1 2; do
for lib in tools/ubench/libplank_key4old.so plankassembly_amd/libplank_hip.so; do
  echo "== $lib (rep $rep)"
  PLANK_HIP_LIB=$PWD/$lib timeout 300 python tools/attn_balance.py 2>&1 | grep "seed 2022\|seed 7 \|16 x 1021"
  PLANK_HIP_LIB=$PWD/$lib timeout 400 python bench.py --steps 60 --warmup 15 --no-decode --no-cpu --no-f32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['kernel_census']
print('step', round(d['ms_per_step'],3), 'ms', round(d['value'],1), 'samples/s;', ' '.join(f\"{k} {c[k]['avg_launch_us']}us\" for k in c if k.startswith('attn')))"
done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06d/key4_ab.txt
