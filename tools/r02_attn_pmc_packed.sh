# SQ counters of the packed (variable-length) encoder self-attention kernels: two PMC passes of tools/attn_packed.py
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/r02
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/r02/pmc_pk2a -o t -- python $R/tools/attn_packed.py > $R/gpurun_out/r02/pmc_pk2a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/r02/pmc_pk2b -o t -- python $R/tools/attn_packed.py > $R/gpurun_out/r02/pmc_pk2b.log 2>&1
cd $R
python tools/pmc_attn_summary.py gpurun_out/r02/pmc_pk2a gpurun_out/r02/pmc_pk2b > gpurun_out/r02/pmc_pk2.txt 2>&1
cat gpurun_out/r02/pmc_pk2.txt
