# GPU box: bench line + rocprofv3 kernel trace / stats of the same command (copied to profiles/ afterwards).
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 python -u bench.py $BENCH_ARGS > gpurun_out/bench.log 2>&1 || (tail -40 gpurun_out/bench.log; exit 1)
tail -1 gpurun_out/bench.log > gpurun_out/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench.json"))
r = d["roofline"]
print("ms/step", d["ms_per_step"], "Mrays/s", d["value"], "| path launch", r["avg_launch_ms"], "resolve", r["resolve_ms_per_step"],
      "| frac", r["frac"], "achieved", r["achieved"], "delivery", r["record_delivery_GBps"], "| cpu", d["cpu_baseline"] and d["cpu_baseline"]["value"])
PY
rm -rf gpurun_out/prof
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-reference-visits $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1 || (tail -40 $GRAFT_REPO_ROOT/gpurun_out/prof.log; exit 1)
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof.log | cut -c1-200
head -12 $GRAFT_REPO_ROOT/gpurun_out/prof/bench_kernel_stats.csv
