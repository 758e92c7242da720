# kernel trace of N steps of one dtype: tools/r05_trace.sh <tag> <dtype>
TAG=${1:-r05x3}; export DTYPE=${2:-x3}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o t -- python $R/tools/step_loop.py > $OUT/kt.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $OUT/kt -name "*.db" | head -1) > $OUT/${DTYPE}_train_kernel_trace_summary.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
tail -2 $OUT/kt.log; head -40 $OUT/${DTYPE}_train_kernel_trace_summary.txt | cut -c1-170
