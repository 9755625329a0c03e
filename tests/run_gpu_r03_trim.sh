#!/bin/bash
# Round 3, second half: A/B of the pass trims of kernels.hip (RTX_LIGHTS_K, RTX_FAST_THR, RTX_ONE_SQRT, RTX_ROOM_MUL,
# RTX_SHADE_IN_LDS, RTX_FAST_INV and the constants RTX_TOP_NODES, RTX_LDS_STACK, RTX_EVENT_STEPS, RTX_SHADE_MIN).
# On the CPU box, before the GPU call, build one library per variant into gpurun_variants/ (tools/build_variant.sh;
# HOSTDEFS as well for RTX_LDS_STACK, which sizes the spill buffer), e.g.
#   bash tools/build_variant.sh a_base -DRTX_LIGHTS_K=0 -DRTX_FAST_THR=0 -DRTX_ONE_SQRT=0 -DRTX_ROOM_MUL=0 -DRTX_SHADE_IN_LDS=0
#   bash tools/build_variant.sh b_lights -DRTX_FAST_THR=0 -DRTX_ONE_SQRT=0 -DRTX_ROOM_MUL=0 -DRTX_SHADE_IN_LDS=0
#   bash tools/build_variant.sh c_all
#   bash tools/build_variant.sh d_inv -DRTX_FAST_INV=1
#   bash tools/build_variant.sh e_top0 -DRTX_TOP_NODES=0
# then on the GPU box: every library on the five kinds of scene, one bench line each (tests/run_gpu_r03_ab.sh).
# tools/valu_ubench.hip is the issue-rate microbenchmark that goes with it (profiles/r03_valu_ubench.txt).
cd $GRAFT_REPO_ROOT
RUN_TESTS=${RUN_TESTS:-1} SETS="${SETS:-headline soup suzanne spheres glass}" bash tests/run_gpu_r03_ab.sh
