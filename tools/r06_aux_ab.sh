# Round 6: dK/dV kernels - per-row words (lse, delta) requested with the tile's DMA and written at the END of the step, against the
# library built before it (tools/ubench/libplank_auxold.so).  Same session, alternating.
R=$PWD; OUT=$R/gpurun_out/aux; mkdir -p $OUT
OLD=$R/tools/ubench/libplank_auxold.so
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attn or attention" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for v in new old new old; do
  if [ $v = old ]; then export PLANK_HIP_LIB=$OLD; else unset PLANK_HIP_LIB; fi
  timeout 600 python bench.py --steps 150 --warmup 20 --no-decode --no-cpu --no-kernels --no-f32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms')"
done 2>&1 | tee $OUT/step.log
export TMPDIR=/tmp; export STEPS=20
for v in new old; do
  if [ $v = old ]; then export PLANK_HIP_LIB=$OLD; else unset PLANK_HIP_LIB; fi
  cd /tmp; DTYPE=bf16 timeout 300 rocprofv3 --kernel-trace -d $OUT/kt_$v -o t -- python $R/tools/step_loop.py > $OUT/kt_$v.log 2>&1; cd $R
  DB=$(find $OUT/kt_$v -name "*.db" | head -1); echo "== $v"; python tools/rocpd_summary.py $DB 2>/dev/null | grep -E "attn|kernel time" | cut -c1-130
  rm -rf $OUT/kt_$v
done 2>&1 | tee $OUT/kernels.log
