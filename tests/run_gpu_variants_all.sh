# the variants of tests/run_gpu_variants.sh over the five kinds of scene
cd $GRAFT_REPO_ROOT
SL=${SLOTS:-1}
export STEPS=4 BENCH_ARGS="--scene soup --tris 4000000 --width 1024 --spp 8 --no-reference-visits --slots $SL"; echo "== soup"; bash tests/run_gpu_variants.sh || exit 1
export STEPS=10 BENCH_ARGS="--scene basic_suzanne --width 1024 --spp 4 --no-reference-visits --slots $SL"; echo "== suzanne"; bash tests/run_gpu_variants.sh || exit 1
export STEPS=8 BENCH_ARGS="--scene scan_glass --width 1024 --spp 8 --no-reference-visits --slots $SL"; echo "== glass"; bash tests/run_gpu_variants.sh || exit 1
export STEPS=10 BENCH_ARGS="--scene random_spheres --tris 1000 --width 1024 --spp 1 --no-reference-visits --slots $SL"; echo "== spheres"; bash tests/run_gpu_variants.sh || exit 1
export STEPS=10 BENCH_ARGS="--no-reference-visits --slots $SL"; echo "== headline"; bash tests/run_gpu_variants.sh || exit 1
