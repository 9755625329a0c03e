cd $GRAFT_REPO_ROOT
for so in v2 v4 v5noxcd; do for sb in 1 4 16 64; do
  [ $so = v2 ] && [ $sb != 16 ] && continue
  RTX_STRIPE_BLOCKS=$sb RTX_LIB=$GRAFT_REPO_ROOT/gpurun_variants/librtx_$so.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-reference-visits --steps 8 --warmup 2 $BENCH_ARGS > gpurun_out/sb.log 2>&1 || { tail -3 gpurun_out/sb.log; exit 1; }
  python - "$so sb=$sb" gpurun_out/sb.log <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[2]).read().strip().splitlines() if l.startswith('{')][-1])
r=d["roofline"]
print(f"{sys.argv[1]:24s} ms/step {d['ms_per_step']:8.3f}  path {r['avg_launch_ms']:8.3f}  resolve {r['resolve_ms_per_step']:6.3f}")
PY
done; done
