# four-wave whole-width gemm8 tiles (one round of <= 256 tiles at M ~ 8 700) against the default kernels; C ubench, back-to-back launches
TAG=${1:-r04g8b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
cd tools/ubench
{
for M in 8704 8300 7940; do
  for spec in "1536 512 6" "1536 512 3" "1536 512 0" "1024 512 7" "1024 512 3" "1024 512 0" "1024 1024 7"; do
    set -- $spec
    for res in 0 1; do timeout 60 ./gemm8_lat $M $1 $2 64 $res 0 1 $3 2>&1 | grep -v "amdgpu.ids\|^check rc 0 sync 0: 0 of"; done
  done
done
} | tee $OUT/gemm8.txt
