# Round 6: W^T refresh on a side stream (PLANK_TRANSPOSE_STREAM=1, default) vs on the main stream (=0).  Same session, alternating.
mkdir -p gpurun_out/trs
timeout 1500 python -m pytest tests/test_headline_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -2
for v in 1 0 1 0; do    # (run when the side stream was the default; it is opt-in now)
  PLANK_TRANSPOSE_STREAM=$v timeout 600 python bench.py --steps 150 --warmup 20 --no-decode --no-cpu --no-kernels 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['train']; print('side stream $v', round(d['value'],1), 'samples/s', round(d['ms_per_step'],3), 'ms | x3', round(t['x3']['value'],1), ' f32', round(t['f32']['value'],1))"
done 2>&1 | tee gpurun_out/trs/step.log
