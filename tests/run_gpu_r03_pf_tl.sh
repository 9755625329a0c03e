#!/bin/bash
# kernel timelines of rank 0's 1/8 share and of config 1 with primaries-first launches forced on
cd $GRAFT_REPO_ROOT
export RTX_PRIMARIES_FIRST=1
bash tests/run_gpu_timeline.sh 8 || exit 1
