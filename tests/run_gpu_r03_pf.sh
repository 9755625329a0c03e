#!/bin/bash
# Primaries-first launches (RTX_PRIMARIES_FIRST=1) against the single persistent launch (=0): parity tests with it
# forced on, bench lines of the headline and the two small configs, rank 0's 1/8 and 1/16 share
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "${RUN_TESTS:-1}" = 1 ]; then
  RTX_PRIMARIES_FIRST=1 timeout -k 10 600 python -u -m pytest ${TESTS:-tests/test_gpu_parity.py tests/test_gpu_fuzz.py} -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/pf_tests.log 2>&1 || { tail -40 gpurun_out/pf_tests.log; exit 1; }
  tail -2 gpurun_out/pf_tests.log
fi
for pf in 0 1; do
  for set in ${SETS:-headline suzanne spheres}; do
    case $set in
      headline) args="--steps 8 --warmup 2";;
      soup4m) args="--scene soup --tris 4000000 --width 1024 --spp 8 --steps 3 --warmup 1";;
      suzanne) args="--scene basic_suzanne --width 1024 --spp 4 --steps 10 --warmup 2";;
      spheres) args="--scene random_spheres --tris 1000 --width 1024 --spp 1 --steps 10 --warmup 2";;
      glass) args="--scene scan_glass --width 2048 --spp 8 --steps 2 --warmup 1";;
    esac
    RTX_PRIMARIES_FIRST=$pf timeout -k 10 200 python bench.py --no-cpu-baseline --no-reference-visits --also-two-slots $args > gpurun_out/pf_${pf}_$set.log 2>&1 || { echo "$pf $set FAILED"; tail -5 gpurun_out/pf_${pf}_$set.log; exit 1; }
    python - "pf=$pf $set" gpurun_out/pf_${pf}_$set.log <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[2]).read().strip().splitlines() if l.startswith('{')][-1])
r=d["roofline"]
print(f"{sys.argv[1]:24s} ms/step {d['ms_per_step']:8.3f}  path {r['avg_launch_ms']:8.3f}  resolve {r['resolve_ms_per_step']:6.3f}  two slots {(d.get('two_slots') or {}).get('ms_per_step')}  Mrays/s {d['value']:9.1f}")
PY
  done
  echo "== partition probe, RTX_PRIMARIES_FIRST=$pf"
  RTX_PRIMARIES_FIRST=$pf timeout -k 10 250 python tools/partition_probe.py ${PARTS:-8 16} 2>&1 | tail -3 | cut -c1-250 || exit 1
done
