# Round-6 evidence (run on the GPU box from the repo root; bash tools/r06_profiles.sh [tag]): kernel traces of the bf16 and x3 train
# steps and of the decode step (bf16 / f32 at batch 256 and at the reference's evaluation batch 16), HBM traffic of the train step
# (FETCH_SIZE / WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes), SQ counters of the GEMM families and of the
# attention kernels inside the step.  Never combines --pmc with hip / hsa / memory-copy tracing.
TAG=${1:-r06}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
BENCH="python $R/bench.py --steps 10 --warmup 3 --no-cpu --no-decode --no-kernels --no-f32"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o t -- $BENCH > $OUT/kt.log 2>&1
DTYPE=x3 STEPS=12 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/ktx3 -o t -- python $R/tools/step_loop.py > $OUT/ktx3.log 2>&1
for DT in bf16 f32; do
  DTYPE=$DT STEPS=48 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/dec_$DT -o t -- python $R/tools/decode_prof.py > $OUT/dec_$DT.log 2>&1
  DTYPE=$DT STEPS=100 BATCH=16 LANES=1 GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/dec16_$DT -o t -- python $R/tools/decode_prof.py > $OUT/dec16_$DT.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-decode --no-kernels --no-f32 > $OUT/pmcf.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-decode --no-kernels --no-f32 > $OUT/pmcw.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/sq_a -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu --no-decode --no-kernels --no-f32 > $OUT/sqa.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq_b -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu --no-decode --no-kernels --no-f32 > $OUT/sqb.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $OUT/kt -name "*.db" | head -1) > $OUT/train_kernel_trace_summary.txt 2>&1
python tools/rocpd_summary.py $(find $OUT/ktx3 -name "*.db" | head -1) > $OUT/x3_train_kernel_trace_summary.txt 2>&1
for DT in bf16 f32; do
  python tools/rocpd_summary.py $(find $OUT/dec_$DT -name "*.db" | head -1) > $OUT/decode_${DT}_kernel_trace_summary.txt 2>&1
  python tools/rocpd_summary.py $(find $OUT/dec16_$DT -name "*.db" | head -1) > $OUT/decode_${DT}_batch16_kernel_trace_summary.txt 2>&1
done
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
python tools/pmc_attn_summary.py $OUT/sq_a $OUT/sq_b --match gemm > $OUT/gemm_pmc.txt 2>&1
python tools/pmc_attn_summary.py $OUT/sq_a $OUT/sq_b --match attn > $OUT/attn_pmc_in_step.txt 2>&1
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +2M -delete
du -sh $OUT | tail -1
head -16 $OUT/train_kernel_trace_summary.txt | cut -c1-150
tail -3 $OUT/ktx3.log; head -12 $OUT/x3_train_kernel_trace_summary.txt | cut -c1-150
