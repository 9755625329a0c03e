#!/bin/bash
# A/B of the work-stack order (RTX_DEFER): parity tests on the in-tree library, bench lines per variant, then rank 0's shares
cd $GRAFT_REPO_ROOT
RUN_TESTS=${RUN_TESTS:-1} SETS="${SETS:-headline suzanne spheres}" bash tests/run_gpu_r03_ab.sh || exit 1
PARTS="${PARTS:-8}" bash tests/run_gpu_r03_part.sh
