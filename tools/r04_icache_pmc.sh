# Instruction-fetch counters of every kernel of the train step (are the per-launch fixed costs instruction-cache misses?)
TAG=${1:-r04ic}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; R=$PWD; cd /tmp
B="python $R/bench.py --steps 4 --warmup 2 --no-cpu --no-decode --no-kernels --no-f32 --long-steps 4"
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $OUT/ic_a -o t -- $B > $OUT/ica.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM --output-format csv -d $OUT/ic_b -o t -- $B > $OUT/icb.log 2>&1
cd $R
python tools/pmc_attn_summary.py $OUT/ic_a $OUT/ic_b --match _ > $OUT/icache_pmc.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
head -150 $OUT/icache_pmc.txt
