# kernel trace of the decode step (B 256, S 1024; eager launches so that every kernel is named): tools/r05_dectrace.sh <tag> [env...]
TAG=${1:-r05dec}; shift; for kv in "$@"; do export "$kv"; done
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD; cd /tmp
GRAPH=0 STEPS=${STEPS:-60} timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o t -- python $R/tools/decode_prof.py > $OUT/kt.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $OUT/kt -name "*.db" | head -1) > $OUT/decode_kernel_trace_summary.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
tail -2 $OUT/kt.log; head -30 $OUT/decode_kernel_trace_summary.txt | cut -c1-170
