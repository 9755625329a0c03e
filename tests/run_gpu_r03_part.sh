# rank 0's share of the headline frame for every library in gpurun_variants/ (tools/partition_probe.py)
cd $GRAFT_REPO_ROOT
for so in gpurun_variants/librtx_*.so; do
  echo "== $(basename $so .so)"
  RTX_LIB=$GRAFT_REPO_ROOT/$so timeout -k 10 250 python tools/partition_probe.py ${PARTS:-1 8 16} 2>&1 | tail -4 || exit 1
done
