// Microbenchmark: dependent per-lane gathers of 64-byte (or 32-byte) records from a table,
// the memory access pattern of BVH traversal.  Reports records/s for several occupancies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int NLOAD>
__global__ void __launch_bounds__(256) chase(const float4 *tab, uint32_t nrec, uint32_t steps, uint32_t *out, uint32_t coherent_shift) {
    uint32_t gtid = blockIdx.x * 256 + threadIdx.x;
    uint32_t idx = ((gtid >> coherent_shift) * 2654435761u) % nrec;
    float acc = 0.f;
    for (uint32_t s = 0; s < steps; s++) {
        const float4 *p = tab + (size_t)idx * 4;
        float4 a = p[0];
        float4 b = NLOAD > 1 ? p[1] : a;
        float4 c = NLOAD > 2 ? p[2] : a;
        float4 d = NLOAD > 3 ? p[3] : a;
        acc += a.x + b.y + c.z + d.w;
        idx = (__float_as_uint(a.w) + (gtid >> coherent_shift) * 7u + s) % nrec;
    }
    out[gtid] = __float_as_uint(acc) + idx;
}

int main(int argc, char **argv) {
    uint32_t mb = argc > 1 ? atoi(argv[1]) : 200;
    uint32_t nrec = (uint32_t)((size_t)mb * (1 << 20) / 64);
    std::vector<float4> h((size_t)nrec * 4);
    uint32_t x = 12345;
    for (size_t i = 0; i < h.size(); i++) {
        x = x * 1664525u + 1013904223u;
        h[i] = make_float4(1.f, 2.f, 3.f, __builtin_bit_cast(float, (x >> 4)));
    }
    float4 *d;
    uint32_t *out;
    hipMalloc(&d, h.size() * sizeof(float4));
    hipMemcpy(d, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipMalloc(&out, 4096 * 256 * 8 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const uint32_t steps = 400;
    for (int coh = 0; coh <= 6; coh += 3)
        for (int nload = 1; nload <= 4; nload++) {
            if (nload == 3) continue;
            for (int bpc = 2; bpc <= 8; bpc *= 2) {
                int blocks = 256 * bpc;
                auto launch = [&]() {
                    if (nload == 1) hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(256), 0, 0, d, nrec, steps, out, coh);
                    if (nload == 2) hipLaunchKernelGGL(chase<2>, dim3(blocks), dim3(256), 0, 0, d, nrec, steps, out, coh);
                    if (nload == 4) hipLaunchKernelGGL(chase<4>, dim3(blocks), dim3(256), 0, 0, d, nrec, steps, out, coh);
                };
                launch();
                hipEventRecord(e0);
                launch();
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                double recs = (double)blocks * 256 * steps;
                printf("table %u MB  lanes/addr %2d  loads/rec %d  blocks/CU %d : %7.3f ms  %8.2f Grec/s  %8.1f GB/s\n", mb, 1 << coh, nload, bpc, ms,
                       recs / ms / 1e6, recs * nload * 16 / ms / 1e6);
            }
        }
    return 0;
}
