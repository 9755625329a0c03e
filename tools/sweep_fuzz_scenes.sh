#!/bin/bash
# The exactness sweep (production vs verbatim traversal, hit ids / t / shaded colours) on the
# random scenes of tests/test_gpu_fuzz.py: usage tools/sweep_fuzz_scenes.sh FIRST LAST
set -e
mkdir -p gpurun_out/fuzz_scenes
python - "$1" "$2" <<'PY'
import sys
sys.path.insert(0, "tests")
import test_gpu_fuzz as fz
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    open(f"gpurun_out/fuzz_scenes/f{seed}.scene", "w").write(fz._scene_text(seed))
PY
: > gpurun_out/sweep_fuzz.log
for seed in $(seq $1 $(($2 - 1))); do
  timeout -k 10 120 python -u tools/exactness_sweep.py file:gpurun_out/fuzz_scenes/f$seed.scene 0 $((1000 + seed)) 1 262144 >> gpurun_out/sweep_fuzz.log 2>&1 || { tail -12 gpurun_out/sweep_fuzz.log; exit 1; }
done
grep -c "MISMATCHES 0" gpurun_out/sweep_fuzz.log
