cd $GRAFT_REPO_ROOT
for lb in 0 1; do for set in "" "--scene basic_suzanne --width 1024 --spp 4" "--scene scan_glass --width 2048 --spp 8 --steps 3"; do
RTX_LEAF_BANK=$lb timeout -k 10 200 python bench.py --no-cpu-baseline --no-reference-visits --steps 8 --warmup 2 $set > gpurun_out/bank.log 2>&1 || { tail -3 gpurun_out/bank.log; exit 1; }
python - "bank=$lb $set" gpurun_out/bank.log <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[2]).read().strip().splitlines() if l.startswith('{')][-1])
r=d["roofline"]
print(f"{sys.argv[1]:60s} ms/step {d['ms_per_step']:8.3f}  path {r['avg_launch_ms']:8.3f}  resolve {r['resolve_ms_per_step']:6.3f}")
PY
done; done
