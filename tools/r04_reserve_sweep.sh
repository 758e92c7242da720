# One-GPU rehearsal of the 8-GPU step: bench.py under a one-rank process group with a stand-in collective per gradient slice
# (PLANK_FAKE_COLLECTIVE, distributed.py) and PA_RESERVE_CUS 0 / 16 / 32.  Usage: bash tools/r04_reserve_sweep.sh <tag>
TAG=${1:-r04sweep}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1
echo "# bench.py --steps 100 --warmup 10 --no-decode --no-cpu --no-kernels under a one-rank RCCL group; ms/step of the bf16 headline step (fresh batches)" > $OUT/sweep.txt
for fake in "" "32:200:8" "32:100:8" "64:200:8"; do
  for res in 0 16 32; do
    if [ -z "$fake" ] && [ "$res" != "0" ]; then continue; fi
    MASTER_PORT=$((29600 + RANDOM % 200)) PLANK_FAKE_COLLECTIVE=$fake PA_RESERVE_CUS=$res timeout 300 python bench.py --steps 100 --warmup 10 --no-decode --no-cpu --no-kernels --long-steps 100 > $OUT/b.json 2> $OUT/b.err
    python - "$fake" "$res" $OUT/b.json >> $OUT/sweep.txt <<'PY'
import json, sys
fake, res, path = sys.argv[1:4]
try:
    d = json.loads([l for l in open(path) if l.startswith("{")][0])
    print(f"fake collective {fake or 'off':>10s}  PA_RESERVE_CUS {res:>2s}:  bf16 {d['ms_per_step']:.3f} ms/step ({d['value']:.0f} samples/s)   f32 {d['train']['f32']['ms_per_step']:.2f} ms/step")
except Exception as e:
    print(f"fake collective {fake or 'off':>10s}  PA_RESERVE_CUS {res:>2s}:  FAILED {e}")
PY
  done
done
cat $OUT/sweep.txt
