# gemm8.h with four waves per block and TWO blocks per CU (512 slots), taller tiles: does one round at 8 704 rows pay?  C ubench.
TAG=${1:-r04g8e}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
cd tools/ubench
{
for M in 8704 8300 7940; do
  for spec in "1024 512 9" "1024 512 10" "1024 512 11" "1024 512 12" "1536 512 9" "1536 512 10" "1536 512 11" "1536 512 12" "512 512 11" "512 512 12" "512 1024 12" "512 1536 12" "512 1536 9"; do
    set -- $spec
    for res in 0 1; do timeout 60 ./gemm8_lat $M $1 $2 64 $res 0 1 $3 2>&1 | grep -v "amdgpu.ids\|^check rc 0 sync 0: 0 of"; done
  done
done
} | tee $OUT/gemm8.txt
