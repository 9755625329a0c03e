# A/B on one box: the round-1 tree (old_r01/, a git worktree of a16c467) against the current one, same workloads.
cd $GRAFT_REPO_ROOT
show () { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"   {d['ms_per_step']:9.3f} ms/step {d['value']:9.1f} Mrays/s")
PY
}
for args in "--scene soup --tris 4000000 --width 1024 --spp 8 --steps 3" "--scene scan_glass --width 1024 --spp 8 --steps 3" "--scene basic_suzanne --width 1024 --spp 4 --steps 5" "--scene random_spheres --tris 1000 --width 1024 --spp 1 --steps 5" "--steps 5"; do
  echo "== $args"
  (cd old_r01 && timeout -k 10 200 python bench.py --no-cpu-baseline --warmup 2 $args > /tmp/old.log 2>&1) || { tail -5 /tmp/old.log; exit 1; }
  echo -n " old"; show /tmp/old.log
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-reference-visits --warmup 2 $args > /tmp/new.log 2>&1 || { tail -5 /tmp/new.log; exit 1; }
  echo -n " new"; show /tmp/new.log
done
