# per-wave begin / dry / end times of the production kernel (diagnostic build -DRTX_WAVE_TIMES) on shares of the headline frame
cd $GRAFT_REPO_ROOT
for b in ${BOUNCES:-4}; do
echo "== bounces $b"
RTX_WAVE_TIMES=1 RTX_LIB=$GRAFT_REPO_ROOT/gpurun_variants_util/librtx_v5wt.so timeout -k 10 200 python tools/util_probe.py --bounces $b ${PARTS:-1 8} 2>&1 | grep "wave times\|path"
done
