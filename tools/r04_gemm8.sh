# eight-wave big-tile GEMM against the existing kernels at the step's shapes (C micro-benchmarks, back-to-back launches)
TAG=${1:-r04g8}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
cd tools/ubench
{
for bk in 32 64; do
  for shp in "8704 1536 512" "8704 1024 512" "8704 512 512" "8704 512 1024" "8704 512 1536" "7940 1536 512" "10100 1024 512" "8704 1024 512 1 0 6"; do
    set -- $shp
    timeout 60 ./gemm8_lat $1 $2 $3 $bk ${4:-1} ${5:-0} ${6:-1} 2>&1 | grep -v amdgpu.ids
  done
done
echo "== existing kernels (gemm_lat)"
for shp in "8704 1536 512" "8704 1024 512" "8704 512 512" "8704 512 1024" "8704 512 1536" "7940 1536 512" "10100 1024 512"; do
  timeout 60 ./gemm_lat $shp 2>&1 | grep -v "amdgpu.ids\|check"
done
} | tee $OUT/gemm8.txt
