mkdir -p gpurun_out/r04cfg
for c in complete visible sideface t1024 tiny; do
  timeout 500 python bench.py --config $c --no-decode --no-kernels --no-cpu > gpurun_out/r04cfg/bench_$c.json 2> gpurun_out/r04cfg/bench_$c.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04cfg/bench_$c.json').read().strip().splitlines()[-1])
print('$c', d['value'], d['unit'], d['ms_per_step'], d.get('cpu_baseline'), d['config']['workload'][:60])
PY
done
timeout 300 python bench.py --dtype f32 --no-decode --no-kernels --no-cpu > gpurun_out/r04cfg/bench_f32.json 2> gpurun_out/r04cfg/bench_f32.err; tail -c 600 gpurun_out/r04cfg/bench_f32.json

python - <<'PY'
import json
out = {}
for c in ("complete", "visible", "sideface", "t1024", "tiny", "f32"):
    try:
        out[c] = json.loads(open(f"gpurun_out/r04cfg/bench_{c}.json").read().strip().splitlines()[-1])
    except Exception as e:
        out[c] = {"error": str(e)}
json.dump(out, open("gpurun_out/r04cfg/bench_other_configs.json", "w"), indent=1)
PY
