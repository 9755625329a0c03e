#!/bin/bash
# Builds gpurun_variants/librtx_<name>.so: the backend with extra -D flags on kernels.hip
# (A/B runs of kernel variants on the GPU box: RTX_LIB=<path> python bench.py ...).
# Run `make -C rust-raytracer_amd/csrc` first: the host objects are shared.
set -e
name=$1; shift
cd "$(dirname "$0")/../rust-raytracer_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize "$@" -c kernels.hip -o /tmp/kernels_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../gpurun_variants/librtx_$name.so /tmp/kernels_$name.o rtx_api.o bvh_build.o flatten.o host_scene.o host_io.o scene_file.o -lz -lpthread -lrccl
echo built $name
