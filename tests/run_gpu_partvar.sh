# partition probe (frame split over N ranks, rank 0's share on one GPU) for every variant library
cd $GRAFT_REPO_ROOT
for so in gpurun_variants/librtx_*.so; do
  echo "== $(basename $so .so)"
  RTX_LIB=$GRAFT_REPO_ROOT/$so timeout -k 10 120 python tools/partition_probe.py 1 4 8 || exit 1
done
