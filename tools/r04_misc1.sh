TAG=${1:-r04m1}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 300 python tools/adam_overlap_probe.py > $OUT/adam_overlap.txt 2>&1; tail -4 $OUT/adam_overlap.txt
timeout 300 python tools/ln_time.py > $OUT/ln_time.txt 2>&1; cat $OUT/ln_time.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "group" 2>&1 | tail -2
