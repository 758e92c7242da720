# GradSync slice coalescing under a one-rank RCCL group (pure per-collective overhead): ms/step for PLANK_SYNC_COALESCE in elements.
TAG=${1:-r04co}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1
echo "# one-rank RCCL group, bench.py --steps 100 --warmup 10: bf16 ms/step by GradSync coalescing threshold (elements)" > $OUT/coalesce.txt
for co in 2097152 4194304 8388608 16777216 40000000; do
  MASTER_PORT=$((29600 + RANDOM % 200)) PLANK_SYNC_COALESCE=$co timeout 300 python bench.py --steps 100 --warmup 10 --no-decode --no-cpu --no-kernels --long-steps 100 > $OUT/b.json 2> $OUT/b.err
  python - $co $OUT/b.json >> $OUT/coalesce.txt <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][0])
print(f"PLANK_SYNC_COALESCE {int(sys.argv[1]):>9d}: bf16 {d['ms_per_step']:.3f} ms/step  (steady {d['steady_state']['ms_per_step']:.3f})   f32 {d['train']['f32']['ms_per_step']:.2f} ms/step")
PY
done
unset RANK LOCAL_RANK WORLD_SIZE
timeout 300 python bench.py --steps 100 --warmup 10 --no-decode --no-cpu --no-kernels --long-steps 100 > $OUT/b.json 2> $OUT/b.err
python - $OUT/b.json >> $OUT/coalesce.txt <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print(f"no process group             : bf16 {d['ms_per_step']:.3f} ms/step  (steady {d['steady_state']['ms_per_step']:.3f})   f32 {d['train']['f32']['ms_per_step']:.2f} ms/step")
PY
cat $OUT/coalesce.txt
