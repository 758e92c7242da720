# Grouped weight-gradient launch: plain unit order (PA_DW_XCD=0) vs XCD-contiguous order (PA_DW_XCD=1): step time and the kernel's average duration.
TAG=${1:-r04dx}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "group" > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt
echo "# bench.py --steps 100 --warmup 10 --no-decode --no-cpu --no-kernels: bf16 ms/step by grouped-dW unit order" > $OUT/dw_xcd.txt
for rep in 1 2; do for x in 0 1; do
  PA_DW_XCD=$x timeout 300 python bench.py --steps 100 --warmup 10 --no-decode --no-cpu --no-kernels --long-steps 100 > $OUT/b.json 2> $OUT/b.err
  python - $x $OUT/b.json >> $OUT/dw_xcd.txt <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][0])
print(f"PA_DW_XCD={sys.argv[1]}: bf16 {d['ms_per_step']:.3f} ms/step  (steady {d['steady_state']['ms_per_step']:.3f})")
PY
done; done
cd /tmp; export TMPDIR=/tmp
for x in 0 1; do
  PA_DW_XCD=$x timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt$x -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu --no-decode --no-kernels > $OUT/kt$x.log 2>&1
  echo "PA_DW_XCD=$x kernel trace:" >> $OUT/dw_xcd.txt
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $OUT/kt$x -name "*.db" | head -1) | grep -i "GemmGroup\|^#\|total" | cut -c1-200 | head -6 >> $OUT/dw_xcd.txt
  for c in FETCH_SIZE WRITE_SIZE; do PA_DW_XCD=$x timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc${c}_$x -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-decode --no-kernels > $OUT/pmc$c$x.log 2>&1; done
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/pmcFETCH_SIZE_$x $OUT/pmcWRITE_SIZE_$x $OUT/pmc_traffic_$x.json 2>&1 | grep -i "GemmGroup\|kernel\|total" | cut -c1-200 | head -6 >> $OUT/dw_xcd.txt
  find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
done
cat $OUT/dw_xcd.txt
