# end-of-round evidence: full GPU suite, the default bench line, kernel traces of the train step and the decode step
mkdir -p gpurun_out/r02
timeout 1300 python -m pytest tests -m gpu -q > gpurun_out/r02/t_final.log 2>&1; grep -n "passed\|failed" gpurun_out/r02/t_final.log | tail -2
timeout 600 python bench.py > gpurun_out/r02/bench_final.json 2> gpurun_out/r02/bench_final.err; tail -c 400 gpurun_out/r02/bench_final.json
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02/kt_final -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu --no-decode --no-kernels > $R/gpurun_out/r02/kt_final.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02/dec_final -o t -- python $R/tools/decode_prof.py > $R/gpurun_out/r02/dec_final.log 2>&1
cd $R
python tools/rocpd_summary.py $(find gpurun_out/r02/kt_final -name "*.db" | head -1) > gpurun_out/r02/kt_final_summary.txt 2>&1
python tools/rocpd_summary.py $(find gpurun_out/r02/dec_final -name "*.db" | head -1) > gpurun_out/r02/dec_final_summary.txt 2>&1
find gpurun_out/r02/kt_final gpurun_out/r02/dec_final -name "*.db" -delete
head -12 gpurun_out/r02/kt_final_summary.txt
