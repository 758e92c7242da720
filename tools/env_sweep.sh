run() { echo "== $*"; env "$@" python bench.py --steps 40 --warmup 5 --no-cpu --no-decode --no-kernels 2>&1 | grep -E "train:" | cut -c1-90; }
run A=0
run PA_GEMM_SMALL_MAX=64
run PA_GEMM_SMALL_MAX=192
run PA_GEMM_SMALL_MAX=256
run PA_DW_BUDGET=512
run PA_DW_BUDGET=384
run A=0
run PA_ATTN_BWD_MERGE_MAX=256
run PA_ATTN_KSPLIT=0
run PA_GEMM_WIDE=1
run PA_GEMM_WIDE=0
run A=0
