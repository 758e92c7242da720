#!/bin/bash
# range blocks of the packed encoder self-attention: kmax / pmax sweep (one process per setting)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r06a
for cfg in ${CFGS:-"8 2" "6 2" "10 2" "8 4" "6 4" "4 4"}; do
  set -- $cfg
  PA_ATTN_SPLIT_KMAX=$1 PA_ATTN_SPLIT_PMAX=$2 timeout 300 python tools/attn_split_bench.py
done 2>&1 | tee gpurun_out/r06a/attn_split_sweep${TAG}.txt
