# GPU box: bench.py on the other BASELINE configs (not the headline) -> gpurun_out/configs.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/configs.log
run () { timeout -k 10 280 python -u bench.py --no-cpu-baseline --no-reference-visits --also-two-slots "$@" 2>/dev/null | grep '^{' >> gpurun_out/configs.log || exit 1; }
run --scene random_spheres --tris 1000 --width 1024 --spp 1 --steps 10 --warmup 2 && \
run --scene basic_suzanne --width 1024 --spp 4 --steps 10 --warmup 2 && \
run --scene scan_glass --width 4096 --spp 64 --steps 1 --warmup 0 && \
run --scene soup --width 2048 --spp 16 --steps 2 --warmup 1 && \
run --scene soup --tris 4000000 --width 1024 --spp 8 --steps 4 --warmup 1
python - <<'PY'
import json
for l in open('gpurun_out/configs.log'):
    j = json.loads(l); r = j['roofline']
    m = r.get('memory') or {}
    print(j['config']['workload'][:60], '|', j['value'], 'Mrays/s', j['ms_per_step'], 'ms | two slots', (j.get('two_slots') or {}).get('ms_per_step'), 'ms | bound', r['bound'],
          r['achieved'], r['unit'], 'frac', r['frac'], '| memory', m.get('achieved'), 'GB/s frac', m.get('frac'), 'B/ray', r['bytes_per_ray'])
PY
