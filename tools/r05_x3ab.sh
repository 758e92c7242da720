# x3 step under rocprofv3 with / without the weight-image cache: per-kernel totals (tools/r05_x3ab.sh)
R=$PWD; export TMPDIR=/tmp; export STEPS=20 DTYPE=x3
for v in "0" "-1"; do
  OUT=$R/gpurun_out/x3ab/wc$v; mkdir -p $OUT; cd /tmp
  if [ $v = 0 ]; then export PLANK_X3_WCACHE_MB=0; else unset PLANK_X3_WCACHE_MB; fi
  timeout 300 rocprofv3 --kernel-trace -d $OUT/kt -o t -- python $R/tools/step_loop.py > $OUT/kt.log 2>&1
  cd $R; DB=$(find $OUT/kt -name "*.db" | head -1)
  echo "== WCACHE=$v"; python tools/rocpd_summary.py $DB | head -14 | cut -c1-150
  rm -rf $OUT/kt
done
