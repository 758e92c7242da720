# SQ counters of the encoder self-attention kernels at the padded benchmark shape (tools/attn_one.py), old (PA_ATTN_V5=0) and new
# forward kernel.  Two passes per variant (8 SQ counters each).  Usage: bash tools/r04_attn_pmc.sh <tag>
TAG=${1:-r04pmc}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
for v in 0 1; do
  PA_ATTN_V5=$v timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/a$v -o t -- python $R/tools/attn_one.py > $OUT/a$v.log 2>&1
  PA_ATTN_V5=$v timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/b$v -o t -- python $R/tools/attn_one.py > $OUT/b$v.log 2>&1
  echo "== PA_ATTN_V5=$v" >> $OUT/summary.txt
  python $R/tools/pmc_attn_summary.py $OUT/a$v $OUT/b$v --match fwd >> $OUT/summary.txt 2>&1
done
cat $OUT/summary.txt
