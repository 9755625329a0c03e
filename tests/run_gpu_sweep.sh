# GPU box: A/B of kernel variants in gpurun_variants/ (built by tools/build_variant.sh):
# headline bench line per variant.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for f in gpurun_variants/*.so; do
  echo "== $f"
  RTX_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 120 python -u bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print(j['value'], 'Mrays/s', j['ms_per_step'], 'ms/step, trace span', r['avg_launch_ms'])
" || exit 1
done
