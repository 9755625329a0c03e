// GPU-assisted binning for the top of the BVH build: a TopSplitter (bvh_build.h) whose passes are HIP kernels.
#pragma once
#include "bvh_build.h"

namespace rtx {

// The splitter allocates on `device` at begin() and frees in its destructor; it uses a stream of its own.
TopSplitter *new_gpu_top_splitter(int device);

}  // namespace rtx
