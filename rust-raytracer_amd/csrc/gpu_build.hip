// GPU-assisted binning for the top of the BVH build (bvh_build.h, TopSplitter).
//
// build_bvh splits a node of millions of objects with three passes over them -- range of the centroids, the
// twelve buckets with their counts and boxes, a stable partition (bvh.rs:195-321) -- and near the root there is
// one node per level: one host thread streams 20 M objects while the others wait.  Here the three passes are
// kernels over the objects' boxes and centroids in HBM; the host keeps the decisions (widest axis, SAH cost of the
// eleven splits) in the reference's f32 arithmetic and gets back, per node, 6 + 12 x 7 numbers.  Every one of
// them is a min, a max or a count, and the partition is stable, so the result does not depend on the order the
// lanes work in: the tree is the host build's, bit for bit.  (A NaN coordinate is skipped as fmin / fmax skip it;
// +0 and -0 may come out either way, as from the host's own loop order, and nothing downstream looks at the sign
// of a zero: the range feeds differences and the bucket index, the bucket boxes feed surface areas.)
#include <hip/hip_runtime.h>

#include <cmath>
#include <limits>
#include <vector>

#include "gpu_build.h"

namespace rtx {

namespace {

constexpr int GB_BLOCK = 256;
constexpr int GB_SEG = 4096;       // objects per block of the partition passes
constexpr int GB_MAX_BLOCKS = 1024;  // blocks of the two reduction passes (their partial results are summed up on the host)

__device__ __forceinline__ uint32_t lane() { return __lane_id(); }

// floats as unsigned keys of the same order (-0 < +0; NaN never gets here)
__device__ __forceinline__ uint32_t fkey(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float funkey(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// range of the centroids of idx[b..e): per block (min xyz, max xyz)
__global__ void __launch_bounds__(GB_BLOCK) crange_kernel(const uint32_t *idx, const float *cents, uint32_t b, uint32_t e, float *out) {
    const float big = 3.402823466e+38f;
    float mn[3] = {big, big, big}, mx[3] = {-big, -big, -big};
    for (uint32_t i = b + blockIdx.x * GB_BLOCK + threadIdx.x; i < e; i += gridDim.x * GB_BLOCK) {
        const float *c = cents + (size_t)idx[i] * 3;
        for (int k = 0; k < 3; k++) {
            mn[k] = fminf(mn[k], c[k]);
            mx[k] = fmaxf(mx[k], c[k]);
        }
    }
    __shared__ float part[GB_BLOCK / 64][6];
    for (int k = 0; k < 3; k++)
        for (int d = 32; d > 0; d >>= 1) {
            mn[k] = fminf(mn[k], __shfl_down(mn[k], d, 64));
            mx[k] = fmaxf(mx[k], __shfl_down(mx[k], d, 64));
        }
    if (lane() == 0)
        for (int k = 0; k < 3; k++) {
            part[threadIdx.x / 64][k] = mn[k];
            part[threadIdx.x / 64][3 + k] = mx[k];
        }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = part[0][threadIdx.x];
        for (int w = 1; w < GB_BLOCK / 64; w++) v = threadIdx.x < 3 ? fminf(v, part[w][threadIdx.x]) : fmaxf(v, part[w][threadIdx.x]);
        out[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// Rust `as i32` of trunc(12 (c - min) / range), 12 -> 11 (bvh.rs:254-260; bvh_build.cpp, bucket_of): the same
// f32 operations -- a true division, -ffp-contract=off
__device__ __forceinline__ int bucket_of(float c, float cmin_dim, float dim_range) {
    const float f = 12.0f * (c - cmin_dim) / dim_range;
    const float ft = truncf(f);
    int bi;
    if (!(ft == ft)) bi = 0;
    else if (ft >= 2147483648.0f) bi = 2147483647;
    else if (ft <= -2147483648.0f) bi = -2147483647 - 1;
    else bi = (int)ft;
    if (bi == kSahBuckets) bi -= 1;
    return bi;
}

// buckets of idx[b..e): bucket[i], and per block the twelve counts and boxes (seeded with the origin, bvh.rs:251)
__global__ void __launch_bounds__(GB_BLOCK) bin_kernel(const uint32_t *idx, const float *cents, const float *boxes, uint32_t b, uint32_t e,
                                                      int dim, float cmin_dim, float dim_range, uint8_t *bucket, uint32_t *out_count,
                                                      float *out_box, uint32_t *bad) {
    __shared__ uint32_t cnt[kSahBuckets];
    __shared__ uint32_t key[kSahBuckets][6];
    if (threadIdx.x < kSahBuckets) {
        cnt[threadIdx.x] = 0;
        for (int k = 0; k < 6; k++) key[threadIdx.x][k] = fkey(0.0f);
    }
    __syncthreads();
    for (uint32_t i = b + blockIdx.x * GB_BLOCK + threadIdx.x; i < e; i += gridDim.x * GB_BLOCK) {
        const uint32_t o = idx[i];
        const int bi = bucket_of(cents[(size_t)o * 3 + dim], cmin_dim, dim_range);
        if (bi < 0 || bi >= kSahBuckets) {
            *bad = 1u;
            continue;
        }
        bucket[i] = (uint8_t)bi;
        atomicAdd(&cnt[bi], 1u);
        const float *bx = boxes + (size_t)o * 6;
        for (int k = 0; k < 3; k++) {
            const float lo = bx[k], hi = bx[3 + k];
            if (lo == lo) atomicMin(&key[bi][k], fkey(lo));        // fmin / fmax skip a NaN
            if (hi == hi) atomicMax(&key[bi][3 + k], fkey(hi));
        }
    }
    __syncthreads();
    if (threadIdx.x < kSahBuckets) {
        out_count[blockIdx.x * kSahBuckets + threadIdx.x] = cnt[threadIdx.x];
        for (int k = 0; k < 6; k++) out_box[(blockIdx.x * kSahBuckets + threadIdx.x) * 6 + k] = funkey(key[threadIdx.x][k]);
    }
}

// stable partition of idx[b..e) by bucket <= best: count per GB_SEG objects, scan, scatter into `scratch`
__global__ void __launch_bounds__(GB_BLOCK) pcount_kernel(const uint8_t *bucket, uint32_t b, uint32_t e, int best, uint32_t *blk) {
    __shared__ uint32_t cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const uint32_t seg = b + blockIdx.x * (uint32_t)GB_SEG;
    uint32_t c = 0;
    for (uint32_t k = 0; k < (uint32_t)GB_SEG / GB_BLOCK; k++) {
        const uint32_t i = seg + k * GB_BLOCK + threadIdx.x;
        if (i < e) c += (int)bucket[i] <= best;
    }
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if (lane() == 0) atomicAdd(&cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = cnt;
}
__global__ void __launch_bounds__(1024) pscan_kernel(uint32_t *blk, uint32_t nblocks) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x, per = (nblocks + 1023u) / 1024u;
    const uint32_t b0 = t * per < nblocks ? t * per : nblocks, b1 = b0 + per < nblocks ? b0 + per : nblocks;
    uint32_t s = 0;
    for (uint32_t k = b0; k < b1; k++) s += blk[k];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t a = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += a;
        __syncthreads();
    }
    uint32_t o = part[t] - s;
    for (uint32_t k = b0; k < b1; k++) {
        const uint32_t c = blk[k];
        blk[k] = o;
        o += c;
    }
    if (t == 1023u) blk[nblocks] = part[t];  // objects on the left side
}
__global__ void __launch_bounds__(GB_BLOCK) pscatter_kernel(const uint32_t *idx, const uint8_t *bucket, uint32_t b, uint32_t e, int best,
                                                           const uint32_t *blk, uint32_t nblocks, uint32_t *scratch) {
    __shared__ uint32_t wcnt[GB_BLOCK / 64];
    __shared__ uint32_t base;
    const uint32_t seg = b + blockIdx.x * (uint32_t)GB_SEG, w = threadIdx.x / 64u, nl = blk[nblocks];
    const unsigned long long lt = (1ull << lane()) - 1ull;
    if (threadIdx.x == 0) base = blk[blockIdx.x];
    for (uint32_t k = 0; k < (uint32_t)GB_SEG / GB_BLOCK; k++) {
        const uint32_t i0 = seg + k * GB_BLOCK, i = i0 + threadIdx.x;   // (block-uniform i0: the objects before this stretch)
        const bool in = i < e, left = in && (int)bucket[i] <= best;
        const unsigned long long m = __ballot(left);
        if (lane() == 0) wcnt[w] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = base, total = 0;
        for (uint32_t v = 0; v < GB_BLOCK / 64u; v++) {
            if (v < w) before += wcnt[v];
            total += wcnt[v];
        }
        const uint32_t lrank = before + (uint32_t)__popcll(m & lt);      // left-side objects before this one
        if (left) scratch[b + lrank] = idx[i];
        else if (in) scratch[b + nl + ((i - b) - lrank)] = idx[i];         // right side: position among the others
        const uint32_t nbase = base + total;
        __syncthreads();
        if (threadIdx.x == 0) base = nbase;
    }
}

struct GpuTopSplitter : TopSplitter {
    int device;
    hipStream_t stream = nullptr;
    uint32_t n = 0;
    float *d_boxes = nullptr, *d_cents = nullptr, *d_part = nullptr;
    uint32_t *d_idx = nullptr, *d_scratch = nullptr, *d_cnt = nullptr, *d_blk = nullptr, *d_bad = nullptr;
    uint8_t *d_bucket = nullptr;
    std::vector<float> h_part;
    std::vector<uint32_t> h_cnt;
    explicit GpuTopSplitter(int dev) : device(dev) {}
    ~GpuTopSplitter() override {
        (void)hipSetDevice(device);
        for (void *p : {(void *)d_boxes, (void *)d_cents, (void *)d_part, (void *)d_idx, (void *)d_scratch, (void *)d_cnt, (void *)d_blk,
                        (void *)d_bad, (void *)d_bucket})
            if (p) (void)hipFree(p);
        if (stream) (void)hipStreamDestroy(stream);
    }
    static int blocks_for(uint32_t cnt) { return (int)std::min<uint32_t>(GB_MAX_BLOCKS, (cnt + GB_BLOCK * 8 - 1) / (GB_BLOCK * 8)); }
    bool ok(hipError_t e) { return e == hipSuccess; }

    bool begin(const float *boxes, const float *cents, uint32_t count) override {
        n = count;
        if (!ok(hipSetDevice(device)) || !ok(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking))) return false;
        const uint32_t segs = (n + GB_SEG - 1) / GB_SEG;
        if (!ok(hipMalloc((void **)&d_boxes, (size_t)n * 24)) || !ok(hipMalloc((void **)&d_cents, (size_t)n * 12)) ||
            !ok(hipMalloc((void **)&d_idx, (size_t)n * 4)) || !ok(hipMalloc((void **)&d_scratch, (size_t)n * 4)) ||
            !ok(hipMalloc((void **)&d_bucket, (size_t)n)) || !ok(hipMalloc((void **)&d_part, (size_t)GB_MAX_BLOCKS * kSahBuckets * 6 * 4)) ||
            !ok(hipMalloc((void **)&d_cnt, (size_t)GB_MAX_BLOCKS * kSahBuckets * 4)) || !ok(hipMalloc((void **)&d_blk, ((size_t)segs + 1) * 4)) ||
            !ok(hipMalloc((void **)&d_bad, 4)))
            return false;
        std::vector<uint32_t> id(n);
        for (uint32_t i = 0; i < n; i++) id[i] = i;
        h_part.resize((size_t)GB_MAX_BLOCKS * kSahBuckets * 6);
        h_cnt.resize((size_t)GB_MAX_BLOCKS * kSahBuckets);
        return ok(hipMemcpyAsync(d_boxes, boxes, (size_t)n * 24, hipMemcpyHostToDevice, stream)) &&
               ok(hipMemcpyAsync(d_cents, cents, (size_t)n * 12, hipMemcpyHostToDevice, stream)) &&
               ok(hipMemcpyAsync(d_idx, id.data(), (size_t)n * 4, hipMemcpyHostToDevice, stream)) && ok(hipStreamSynchronize(stream));
    }
    bool centroid_range(uint32_t b, uint32_t e, float cmin[3], float cmax[3]) override {
        const int g = blocks_for(e - b);
        hipLaunchKernelGGL(crange_kernel, dim3(g), dim3(GB_BLOCK), 0, stream, d_idx, d_cents, b, e, d_part);
        if (!ok(hipMemcpyAsync(h_part.data(), d_part, (size_t)g * 6 * 4, hipMemcpyDeviceToHost, stream)) || !ok(hipStreamSynchronize(stream)))
            return false;
        const float big = std::numeric_limits<float>::max();
        for (int k = 0; k < 3; k++) {
            cmin[k] = big;
            cmax[k] = -big;
        }
        for (int j = 0; j < g; j++)
            for (int k = 0; k < 3; k++) {
                cmin[k] = std::fmin(cmin[k], h_part[(size_t)j * 6 + k]);
                cmax[k] = std::fmax(cmax[k], h_part[(size_t)j * 6 + 3 + k]);
            }
        return true;
    }
    bool bin(uint32_t b, uint32_t e, int dim, float cmin_dim, float dim_range, uint32_t count[kSahBuckets], float bb[kSahBuckets][6],
             bool *bad) override {
        const int g = blocks_for(e - b);
        uint32_t h_bad = 0;
        if (!ok(hipMemsetAsync(d_bad, 0, 4, stream))) return false;
        hipLaunchKernelGGL(bin_kernel, dim3(g), dim3(GB_BLOCK), 0, stream, d_idx, d_cents, d_boxes, b, e, dim, cmin_dim, dim_range, d_bucket,
                           d_cnt, d_part, d_bad);
        if (!ok(hipMemcpyAsync(h_cnt.data(), d_cnt, (size_t)g * kSahBuckets * 4, hipMemcpyDeviceToHost, stream)) ||
            !ok(hipMemcpyAsync(h_part.data(), d_part, (size_t)g * kSahBuckets * 6 * 4, hipMemcpyDeviceToHost, stream)) ||
            !ok(hipMemcpyAsync(&h_bad, d_bad, 4, hipMemcpyDeviceToHost, stream)) || !ok(hipStreamSynchronize(stream)))
            return false;
        *bad = h_bad != 0;
        for (int i = 0; i < kSahBuckets; i++) {
            count[i] = 0;
            for (int k = 0; k < 6; k++) bb[i][k] = 0.0f;
        }
        for (int j = 0; j < g; j++)
            for (int i = 0; i < kSahBuckets; i++) {
                count[i] += h_cnt[(size_t)j * kSahBuckets + i];
                const float *p = &h_part[((size_t)j * kSahBuckets + i) * 6];
                for (int k = 0; k < 3; k++) {
                    bb[i][k] = std::fmin(bb[i][k], p[k]);
                    bb[i][3 + k] = std::fmax(bb[i][3 + k], p[3 + k]);
                }
            }
        return true;
    }
    bool partition(uint32_t b, uint32_t e, int best_i) override {
        const uint32_t segs = (e - b + GB_SEG - 1) / GB_SEG;
        hipLaunchKernelGGL(pcount_kernel, dim3(segs), dim3(GB_BLOCK), 0, stream, d_bucket, b, e, best_i, d_blk);
        hipLaunchKernelGGL(pscan_kernel, dim3(1), dim3(1024), 0, stream, d_blk, segs);
        hipLaunchKernelGGL(pscatter_kernel, dim3(segs), dim3(GB_BLOCK), 0, stream, d_idx, d_bucket, b, e, best_i, d_blk, segs, d_scratch);
        return ok(hipMemcpyAsync(d_idx + b, d_scratch + b, (size_t)(e - b) * 4, hipMemcpyDeviceToDevice, stream)) && ok(hipGetLastError());
    }
    bool finish(uint32_t *idx_out) override {
        return ok(hipMemcpyAsync(idx_out, d_idx, (size_t)n * 4, hipMemcpyDeviceToHost, stream)) && ok(hipStreamSynchronize(stream));
    }
};

}  // namespace

TopSplitter *new_gpu_top_splitter(int device) { return new GpuTopSplitter(device); }

}  // namespace rtx
