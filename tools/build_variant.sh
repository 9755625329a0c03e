#!/bin/bash
# Builds gpurun_variants/librtx_<name>.so: the backend with extra -D flags on kernels.hip
# (A/B runs of kernel variants on the GPU box: RTX_LIB=<path> python bench.py ...).
# Run `make -C rust-raytracer_amd/csrc` first: the host objects are shared.  Flags that the host
# side must see too (RTX_LDS_STACK sizes the spill buffer) go in HOSTDEFS as well: rtx_api.cpp is
# then recompiled for the variant.
set -e
name=$1; shift
cd "$(dirname "$0")/../rust-raytracer_amd/csrc"
mkdir -p ../../gpurun_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize "$@" -c kernels.hip -o /tmp/kernels_$name.o
api=rtx_api.o
if [ -n "$HOSTDEFS" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $HOSTDEFS -c rtx_api.cpp -o /tmp/rtx_api_$name.o
  api=/tmp/rtx_api_$name.o
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../gpurun_variants/librtx_$name.so /tmp/kernels_$name.o $api gpu_build.o bvh_build.o flatten.o host_scene.o host_io.o scene_file.o -lz -lpthread -lrccl
echo built $name
