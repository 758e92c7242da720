# Probe: does the chip have idle capacity that a SECOND independent kernel stream could use?  Two bench.py processes at
# batch 8 side by side on the one GPU against one process at batch 8 and one at batch 16 (the headline).  Aggregate samples/s
# of the pair above the batch-16 figure = launch gaps / tails / low-occupancy phases are fillable by a concurrent stream.
TAG=${1:-r04conc}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
F="--no-cpu --no-decode --no-kernels --no-f32 --dtype bf16 --steps 300 --warmup 20 --long-steps 300"
python bench.py $F --batch 8 > $OUT/solo8.json 2> $OUT/solo8.err
python bench.py $F --batch 8 > $OUT/pairA.json 2> $OUT/pairA.err &
PA=$!
python bench.py $F --batch 8 > $OUT/pairB.json 2> $OUT/pairB.err
wait $PA
python bench.py $F --batch 4 > $OUT/solo4.json 2> $OUT/solo4.err
for f in solo8 pairA pairB solo4; do python - $OUT/$f.json <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); print(sys.argv[1].split('/')[-1], 'value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'steady', round(d['steady_state']['ms_per_step'],3), round(d['steady_state']['value'],1))
PY
done
