# Round 6: register-staged attention kernels (bf16x3 and exact f32) - rows past the end zeroed in lstore instead of right behind the loads
# (the select made every "prefetch" a synchronous load).  Against tools/ubench/libplank_auxold.so (before it), same session, alternating.
mkdir -p gpurun_out/stage
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_headline_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -2
OLD=$PWD/tools/ubench/libplank_auxold.so
for v in new old new old; do
  if [ $v = old ]; then export PLANK_HIP_LIB=$OLD; else unset PLANK_HIP_LIB; fi
  timeout 900 python bench.py --steps 60 --warmup 10 --no-decode --no-cpu --no-kernels 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['train']; print('$v bf16', round(d['ms_per_step'],3), 'ms | x3', round(t['x3']['ms_per_step'],3), 'ms', round(t['x3']['value'],1), '| f32', round(t['f32']['ms_per_step'],3), 'ms', round(t['f32']['value'],1))"
done 2>&1 | tee gpurun_out/stage/step.log
