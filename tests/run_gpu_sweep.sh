cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for f in gpurun_variants/*.so; do
  echo "== $f"
  RTX_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 120 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print(j['value'], 'Mrays/s', j['ms_per_step'], 'ms/step', r['kernel'], r['avg_launch_ms']*r['launches_per_step'], 'other', r['other_kernel']['ms_per_step'])
" || exit 1
done
