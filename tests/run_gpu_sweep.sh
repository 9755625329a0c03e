#!/bin/bash
# Exactness sweep at scale (tools/exactness_sweep.py) over the scene kinds; one line per repetition.
set -e
mkdir -p gpurun_out
: > gpurun_out/sweep.log
for spec in "scan 0 1 6" "soup 2000000 2 4" "basic_suzanne 0 3 4" "random_spheres 1000 4 4" "scan_glass 300000 5 3"; do
  set -- $spec
  timeout -k 10 400 python -u tools/exactness_sweep.py $1 $2 $3 $4 >> gpurun_out/sweep.log 2>&1 || { tail -20 gpurun_out/sweep.log; exit 1; }
done
# objects that can be hit outside their boxes (non-planar quads, sheared spheres), through a scene file
python tools/make_warped_scene.py gpurun_out/warped.scene 40 9 >> gpurun_out/sweep.log
timeout -k 10 400 python -u tools/exactness_sweep.py file:gpurun_out/warped.scene 0 11 4 >> gpurun_out/sweep.log 2>&1 || { tail -20 gpurun_out/sweep.log; exit 1; }
# lights placed where shadow queries degenerate (axis-parallel directional lights, a point light on a vertex)
timeout -k 10 400 python -u tools/exactness_sweep.py file:assets/scenes/adversarial_lights.scene 0 51 4 >> gpurun_out/sweep.log 2>&1 || { tail -20 gpurun_out/sweep.log; exit 1; }
tail -56 gpurun_out/sweep.log
# the same families against the CPU oracle (moderate sizes: the oracle visits both children of every node)
: > gpurun_out/oracle_sweep.log
for spec in "basic_suzanne 0 61 4" "scan 200000 62 3" "random_spheres 1000 63 3" "file:gpurun_out/warped.scene 0 64 3" "file:assets/scenes/adversarial_lights.scene 0 65 3" "soup 200000 66 2"; do
  set -- $spec
  timeout -k 10 280 python -u tests/oracle_sweep.py $1 $2 $3 $4 >> gpurun_out/oracle_sweep.log 2>&1 || { tail -20 gpurun_out/oracle_sweep.log; exit 1; }
done
grep -E "TOTAL|param" gpurun_out/oracle_sweep.log
