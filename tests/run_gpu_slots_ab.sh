cd $GRAFT_REPO_ROOT
show () { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"   {d['ms_per_step']:9.3f} ms/step {d['value']:9.1f} Mrays/s  launch {d['roofline']['avg_launch_ms']}")
PY
}
for args in "--scene soup --tris 4000000 --width 1024 --spp 8 --steps 4" "--scene scan_glass --width 1024 --spp 8 --steps 6" "--scene basic_suzanne --width 1024 --spp 4 --steps 10" "--scene random_spheres --tris 1000 --width 1024 --spp 1 --steps 10" "--steps 10"; do
  echo "== $args"
  for sl in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-reference-visits --warmup 2 --slots $sl $args > /tmp/new.log 2>&1 || { tail -5 /tmp/new.log; exit 1; }
  echo -n " slots $sl"; show /tmp/new.log
  done
done
