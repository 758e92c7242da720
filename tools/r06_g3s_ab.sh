# Round 6: gemm3s_kernel (64 x 64 tiles) - fragments a whole item ahead + no compiler vmcnt(0) in the K loop, against the old loop
# (-DPA_G3S_OLD build in tools/ubench/libplank_g3sold.so).  Same session, alternating.
R=$PWD; OUT=$R/gpurun_out/g3s; mkdir -p $OUT
OLD=$R/tools/ubench/libplank_g3sold.so
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or linear or ln" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for v in new old new old; do
  echo "== gemm_small_k $v"
  if [ $v = old ]; then PLANK_HIP_LIB=$OLD timeout 300 python tools/gemm_small_k.py; else timeout 300 python tools/gemm_small_k.py; fi
done 2>&1 | tee $OUT/small_k.log
for v in new old new old; do
  if [ $v = old ]; then export PLANK_HIP_LIB=$OLD; else unset PLANK_HIP_LIB; fi
  timeout 600 python bench.py --steps 150 --warmup 20 --no-decode --no-cpu --no-kernels --no-f32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms')"
done 2>&1 | tee $OUT/step.log
unset PLANK_HIP_LIB
