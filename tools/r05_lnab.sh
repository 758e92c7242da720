# LayerNorm backward variants under rocprofv3: per-grid durations (tools/r05_lnab.sh)
R=$PWD; export TMPDIR=/tmp; export STEPS=20
for v in "0 libplank_hip.so" "1 libplank_hip.so" "1 libplank_hip_lnb16.so" "1 libplank_hip_lnb32.so"; do set -- $v
  for dt in bf16 x3; do
  OUT=$R/gpurun_out/lnab/$dt-$1-$2; mkdir -p $OUT; cd /tmp
  PA_LNB_512=$1 PLANK_HIP_LIB=$R/plankassembly_amd/$2 DTYPE=$dt timeout 300 rocprofv3 --kernel-trace -d $OUT/kt -o t -- python $R/tools/step_loop.py > $OUT/kt.log 2>&1
  cd $R; DB=$(find $OUT/kt -name "*.db" | head -1)
  echo "== $dt PA_LNB_512=$1 $2"; python tools/kernel_grid_breakdown.py $DB | grep "layernorm_bwd" | cut -c1-90
  rm -rf $OUT/kt
  done
done
