# kernel trace of the bf16 step with a per-grid-size breakdown of the row kernels: tools/r05_grid.sh <tag>
TAG=${1:-r05grid}; export DTYPE=${2:-bf16}; export STEPS=${STEPS:-30}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o t -- python $R/tools/step_loop.py > $OUT/kt.log 2>&1
cd $R
DB=$(find $OUT/kt -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $OUT/${DTYPE}_summary.txt 2>&1
python tools/kernel_grid_breakdown.py $DB > $OUT/${DTYPE}_grid.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
tail -1 $OUT/kt.log; grep -i "layernorm" $OUT/${DTYPE}_grid.txt
