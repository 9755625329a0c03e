# GPU box: kernel timeline of one warm frame of rank 0's 1/N share (N = $1, default 8).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tl
rm -f gpurun_out/tl/*
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-8}
cd /tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o tl -- python3 $R/tools/partition_once.py $N > $R/gpurun_out/tl/run.log 2>&1 || { tail -20 $R/gpurun_out/tl/run.log; exit 1; }
python3 $R/tools/timeline.py $R/gpurun_out/tl/tl_kernel_trace.csv
