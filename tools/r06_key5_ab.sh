# A/B of the 32-row-wave kernels' LDS key (128-byte rows) in ONE session: tools/ubench/libplank_key5old.so (-DPA_KEY5_OLD) against the
# product library: packed / dense attention launches (bf16), the padded-shape kernel figures of bench.py, the bf16x3 train step.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r06d
for rep in 1 2; do
for lib in tools/ubench/libplank_key5old.so plankassembly_amd/libplank_hip.so; do
  echo "== $lib (rep $rep)"
  PLANK_HIP_LIB=$PWD/$lib timeout 300 python tools/attn_balance.py 2>&1 | grep "seed 2022\|16 x 1024\|64 x 299"
  PLANK_HIP_LIB=$PWD/$lib BWD=1 timeout 300 python tools/attn_sweep.py 2>&1 | grep "B  16 mask 1\|B  48 mask 0"
  PLANK_HIP_LIB=$PWD/$lib DTYPE=x3 STEPS=20 timeout 300 python tools/step_loop.py 2>&1 | grep "ms/step"
done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06d/key5_ab.txt
