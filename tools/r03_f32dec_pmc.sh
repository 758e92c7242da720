OUT=$PWD/gpurun_out/${1:-r03ad}; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD; cd /tmp
DTYPE=f32 STEPS=24 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/dec32_fetch -o t -- python $R/tools/decode_prof.py > $OUT/dec32f.log 2>&1
DTYPE=f32 STEPS=24 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/dec32_write -o t -- python $R/tools/decode_prof.py > $OUT/dec32w.log 2>&1
cd $R
python tools/pmc_summary.py $OUT/dec32_fetch $OUT/dec32_write $OUT/decode_f32_pmc_traffic.json > $OUT/decode_f32_pmc_traffic.txt 2>&1
find $OUT -name "*.csv" -size +2M -delete
cat $OUT/decode_f32_pmc_traffic.txt | tail -12
