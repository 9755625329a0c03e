# A/B of kernel variants built by tools/build_variant.sh: bench line of each (no CPU baseline).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
STEPS=${STEPS:-5}
for so in gpurun_variants/librtx_*.so; do
  n=$(basename $so .so)
  RTX_LIB=$GRAFT_REPO_ROOT/$so timeout -k 10 120 python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline $BENCH_ARGS > gpurun_out/var_$n.log 2>&1 || { echo "$n FAILED"; tail -5 gpurun_out/var_$n.log; exit 1; }
  python - "$n" gpurun_out/var_$n.log <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r=d["roofline"]
print(f"{sys.argv[1]:28s} ms/step {d['ms_per_step']:8.3f}  path {r['avg_launch_ms']:8.3f}  resolve {r['resolve_ms_per_step']:6.3f}  Mrays/s {d['value']:9.1f}")
PY
done
