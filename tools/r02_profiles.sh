mkdir -p gpurun_out/r02
timeout 1100 python -m pytest tests -m gpu -q > gpurun_out/r02/t8_1.log 2>&1; grep -n "passed\|failed" gpurun_out/r02/t8_1.log | tail -2
timeout 1100 python -m pytest tests -m gpu -q > gpurun_out/r02/t8_2.log 2>&1; grep -n "passed\|failed" gpurun_out/r02/t8_2.log | tail -2
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02/kt8 -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu --no-decode --no-kernels > $R/gpurun_out/r02/kt8.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02/dec8 -o t -- python $R/tools/decode_prof.py > $R/gpurun_out/r02/dec8.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r02/pmc_fetch -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-decode --no-kernels > $R/gpurun_out/r02/pmcf.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r02/pmc_write -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-decode --no-kernels > $R/gpurun_out/r02/pmcw.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/r02/pmc_fetch gpurun_out/r02/pmc_write gpurun_out/r02/pmc_traffic.json > gpurun_out/r02/pmc_traffic.txt 2>&1
python tools/rocpd_summary.py $(find gpurun_out/r02/kt8 -name "*.db" | head -1) > gpurun_out/r02/kt8_summary.txt 2>&1
python tools/rocpd_summary.py $(find gpurun_out/r02/dec8 -name "*.db" | head -1) > gpurun_out/r02/dec8_summary.txt 2>&1
find gpurun_out/r02/pmc_fetch gpurun_out/r02/pmc_write -name "*.db" -delete
du -sh gpurun_out/r02 | tail -1
tail -3 gpurun_out/r02/pmc_traffic.txt
