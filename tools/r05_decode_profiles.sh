# Round-5 decode evidence after the absorbed cross-attention (csrc/decode_mq.h): kernel traces of the bf16 and f32 decode steps under
# graph replay, HBM traffic of both (FETCH_SIZE / WRITE_SIZE in separate passes, MI355X_MICROARCH.md), SQ counters of the new kernels.
# Never combines --pmc with hip / hsa / memory-copy tracing.     bash tools/r05_decode_profiles.sh [tag]
TAG=${1:-r05dec}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD; cd /tmp
for DT in bf16 f32; do
  export DTYPE=$DT
  STEPS=48 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$DT -o t -- python $R/tools/decode_prof.py > $OUT/kt_$DT.log 2>&1
  STEPS=24 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$DT -o t -- python $R/tools/decode_prof.py > $OUT/f_$DT.log 2>&1
  STEPS=24 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write_$DT -o t -- python $R/tools/decode_prof.py > $OUT/w_$DT.log 2>&1
  STEPS=16 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/sqa_$DT -o t -- python $R/tools/decode_prof.py > $OUT/sa_$DT.log 2>&1
  STEPS=16 timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/sqb_$DT -o t -- python $R/tools/decode_prof.py > $OUT/sb_$DT.log 2>&1
done
cd $R
for DT in bf16 f32; do
  python tools/rocpd_summary.py $(find $OUT/kt_$DT -name "*.db" | head -1) > $OUT/decode_${DT}_kernel_trace_summary.txt 2>&1
  python tools/pmc_summary.py $OUT/fetch_$DT $OUT/write_$DT $OUT/decode_${DT}_pmc_traffic.json > $OUT/decode_${DT}_pmc_traffic.txt 2>&1
  python tools/pmc_attn_summary.py $OUT/sqa_$DT $OUT/sqb_$DT --match cross_mq > $OUT/decode_${DT}_cross_mq_pmc.txt 2>&1
done
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
du -sh $OUT | tail -1
for DT in bf16 f32; do head -12 $OUT/decode_${DT}_kernel_trace_summary.txt | cut -c1-150; tail -12 $OUT/decode_${DT}_pmc_traffic.txt; cat $OUT/decode_${DT}_cross_mq_pmc.txt | head -30; done
