R=$PWD; export TMPDIR=/tmp; OUT=$R/gpurun_out/cold; mkdir -p $OUT; cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/kt -o t -- python $R/tools/gemm_cold.py > $OUT/run.log 2>&1
cd $R; tail -1 $OUT/run.log; python tools/gemm_cold.py --report $(find $OUT/kt -name "*.db" | head -1) | tee $OUT/gemm_cold.txt; rm -rf $OUT/kt
