OUT=$PWD/gpurun_out/${1:-r03o}; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD; cd /tmp
DTYPE=f32 GRAPH=0 STEPS=40 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/dec32 -o t -- python $R/tools/decode_prof.py > $OUT/dec32.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $OUT/dec32 -name "*.db" | head -1) > $OUT/decode_f32_kernel_trace_summary.txt 2>&1
find $OUT -name "*.db" -delete
cut -c1-160 $OUT/decode_f32_kernel_trace_summary.txt | head -30
