# bf16 greedy decode (B 256 x 1024 steps, graph replay, one lane): self-attention over K / V caches (PLANK_DECODE_MQ_SELF_BF16=0) against
# the absorbed form (layer-input rows cached, W_v behind the softmax), alternating in ONE session; then the small batches.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r06e
for rep in 1 2; do for v in 0 1; do echo "PLANK_DECODE_MQ_SELF_BF16=$v (rep $rep)"; PLANK_DECODE_MQ_SELF_BF16=$v timeout 400 python tools/decode_time.py 2>&1 | grep "lanes 1"; done; done 2>&1 | tee gpurun_out/r06e/decode_self_ab.txt
for v in 0 1; do echo "PLANK_DECODE_MQ_SELF_BF16=$v"; PLANK_DECODE_MQ_SELF_BF16=$v STEPS=1024 BATCHES=16,64 timeout 400 python tools/decode_small_batch.py 2>&1 | grep "bf16 B"; done 2>&1 | tee -a gpurun_out/r06e/decode_self_ab.txt
