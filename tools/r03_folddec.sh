OUT=$PWD/gpurun_out/${1:-r03v}; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD; cd /tmp
PLANK_DECODE_FOLD_LN=1 GRAPH=0 STEPS=40 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/decfold -o t -- python $R/tools/decode_prof.py > $OUT/decfold.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $OUT/decfold -name "*.db" | head -1) > $OUT/decode_fold_kernel_trace_summary.txt 2>&1
find $OUT -name "*.db" -delete
cut -c1-160 $OUT/decode_fold_kernel_trace_summary.txt | head -14
