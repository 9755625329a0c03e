// Microbenchmark: dependent per-lane fetches of 128-byte records (the wide BVH node),
//   mode A  every lane issues 7 dwordx4 loads of its own record (64 distinct lines per instruction)
//   mode B  8 lanes fetch one record together (8 distinct lines per instruction), the wave
//           transposes through an 8 KB LDS stage and every lane reads its 7 pieces back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(256) chase_own(const float4 *tab, uint32_t nrec, uint32_t steps, uint32_t *out) {
    uint32_t gtid = blockIdx.x * 256 + threadIdx.x;
    uint32_t idx = (gtid * 2654435761u) % nrec;
    float acc = 0.f;
    for (uint32_t s = 0; s < steps; s++) {
        const float4 *p = tab + (size_t)idx * 8;
        float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6];
        acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z;
        idx = (__float_as_uint(g.w) + gtid * 7u + s) % nrec;
    }
    out[gtid] = __float_as_uint(acc) + idx;
}

__global__ void __launch_bounds__(256) chase_coop(const float4 *tab, uint32_t nrec, uint32_t steps, uint32_t *out) {
    __shared__ float4 stage[4][64 * 8];
    uint32_t gtid = blockIdx.x * 256 + threadIdx.x;
    uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t g = lane >> 3, piece = lane & 7u;
    uint32_t idx = (gtid * 2654435761u) % nrec;
    float acc = 0.f;
    float4 *st = stage[wave];
    for (uint32_t s = 0; s < steps; s++) {
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            uint32_t o = 8u * j + g;
            uint32_t io = __shfl(idx, o, 64);
            float4 v = tab[(size_t)io * 8 + piece];
            st[o * 8 + (piece ^ g)] = v;
        }
        uint32_t sw = lane & 7u;
        float4 a = st[lane * 8 + (0 ^ sw)], b = st[lane * 8 + (1 ^ sw)], c = st[lane * 8 + (2 ^ sw)],
               d = st[lane * 8 + (3 ^ sw)], e = st[lane * 8 + (4 ^ sw)], f = st[lane * 8 + (5 ^ sw)],
               gg = st[lane * 8 + (6 ^ sw)];
        acc += a.x + b.y + c.z + d.w + e.x + f.y + gg.z;
        idx = (__float_as_uint(gg.w) + gtid * 7u + s) % nrec;
    }
    out[gtid] = __float_as_uint(acc) + idx;
}

int main(int argc, char **argv) {
    uint32_t mb = argc > 1 ? atoi(argv[1]) : 46;
    uint32_t nrec = (uint32_t)((size_t)mb * (1 << 20) / 128);
    std::vector<float4> h((size_t)nrec * 8);
    uint32_t x = 12345;
    for (size_t i = 0; i < h.size(); i++) {
        x = x * 1664525u + 1013904223u;
        h[i] = make_float4(1.f, 2.f, 3.f, __builtin_bit_cast(float, (x >> 4)));
    }
    float4 *d;
    uint32_t *out;
    hipMalloc(&d, h.size() * sizeof(float4));
    hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const uint32_t steps = 200;
    for (int mode = 0; mode < 2; mode++)
        for (int bpc = 2; bpc <= 4; bpc++) {
            int blocks = 256 * bpc;
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(chase_own, dim3(blocks), dim3(256), 0, 0, d, nrec, steps, out);
                else hipLaunchKernelGGL(chase_coop, dim3(blocks), dim3(256), 0, 0, d, nrec, steps, out);
            };
            launch();
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            double recs = (double)blocks * 256 * steps;
            printf("table %4u MB  %s  blocks/CU %d : %7.3f ms  %8.2f Grec/s\n", mb, mode ? "coop(8 lanes/rec + LDS)" : "own (7 loads/lane)     ", bpc, ms, recs / ms / 1e6);
        }
    return 0;
}
