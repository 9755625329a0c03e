// VALU issue-rate microbenchmark (gfx950): how many cycles does a SIMD spend per wave64 instruction of
// v_fma_f32 / v_mul_f32+v_sub_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_min_f32 / v_cndmask, with W waves per SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_ubench tools/valu_ubench.hip ; run: tools/valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096, UNROLL = 16;
template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, float seed) {
    float a[8];
    v2f p[4];
    for (int i = 0; i < 8; i++) a[i] = seed + (float)(threadIdx.x + i);
    for (int i = 0; i < 4; i++) p[i] = v2f{a[2 * i], a[2 * i + 1]};
    const float c = seed * 0.5f + 1.0f, ci = 1.0f / c;   // (c, 1/c alternate: the data stays finite and non-trivial)
    const v2f c2 = v2f{c, c + 1.0f}, c2i = v2f{ci, 1.0f / (c + 1.0f)};
    const unsigned long long mask = __ballot(a[0] > 100.0f);
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" ::"v"(a[0]), "v"(a[1]) : "vcc");
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNROLL / 8; u++) {
            if (KIND == 0) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            } else if (KIND == 1) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"((u & 1) ? ci : c));
            } else if (KIND == 2) {
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2i));
            } else if (KIND == 3) {
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(c2));
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "v"(c2));
            } else if (KIND == 4) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            } else if (KIND == 5) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c));
            } else if (KIND == 6) {
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c2));
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c2));
            } else if (KIND == 10) {  // the mask in an SGPR pair
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "s"(mask));
            } else if (KIND == 11) {  // compare + select, as the sort network does
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[i + 4]) : "vcc");
            } else if (KIND == 12) {  // two selects per compare
#pragma unroll
                for (int i = 0; i < 2; i++) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc\n\tv_cndmask_b32 %0, %2, %0, vcc" : "+v"(a[i]), "+v"(a[i + 2]) : "v"(a[i + 4]), "v"(a[i + 6]) : "vcc");
            } else if (KIND == 13) {  // compare-exchange of (key, ref) pairs as the compiler emits it: v_cmp -> vcc, four v_cndmask_b32_e32
                float k0, k1, r0, r1;
                asm volatile("v_cmp_lt_f32 vcc, %4, %5\n\tv_cndmask_b32 %0, %4, %5, vcc\n\tv_cndmask_b32 %1, %5, %4, vcc\n\t"
                             "v_cndmask_b32 %2, %6, %7, vcc\n\tv_cndmask_b32 %3, %7, %6, vcc"
                             : "=&v"(k0), "=&v"(k1), "=&v"(r0), "=&v"(r1) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "vcc");
                a[0] = k0; a[1] = k1; a[2] = r0; a[3] = r1;
                asm volatile("v_cmp_lt_f32 vcc, %4, %5\n\tv_cndmask_b32 %0, %4, %5, vcc\n\tv_cndmask_b32 %1, %5, %4, vcc\n\t"
                             "v_cndmask_b32 %2, %6, %7, vcc\n\tv_cndmask_b32 %3, %7, %6, vcc"
                             : "=&v"(k0), "=&v"(k1), "=&v"(r0), "=&v"(r1) : "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "vcc");
                a[4] = k0; a[5] = k1; a[6] = r0; a[7] = r1;
            } else if (KIND == 14) {  // the same with the mask in an SGPR pair
                float k0, k1, r0, r1;
                unsigned long long m;
                asm volatile("v_cmp_lt_f32_e64 %4, %5, %6\n\tv_cndmask_b32_e64 %0, %5, %6, %4\n\tv_cndmask_b32_e64 %1, %6, %5, %4\n\t"
                             "v_cndmask_b32_e64 %2, %7, %8, %4\n\tv_cndmask_b32_e64 %3, %8, %7, %4"
                             : "=&v"(k0), "=&v"(k1), "=&v"(r0), "=&v"(r1), "=&s"(m) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
                a[0] = k0; a[1] = k1; a[2] = r0; a[3] = r1;
                asm volatile("v_cmp_lt_f32_e64 %4, %5, %6\n\tv_cndmask_b32_e64 %0, %5, %6, %4\n\tv_cndmask_b32_e64 %1, %6, %5, %4\n\t"
                             "v_cndmask_b32_e64 %2, %7, %8, %4\n\tv_cndmask_b32_e64 %3, %8, %7, %4"
                             : "=&v"(k0), "=&v"(k1), "=&v"(r0), "=&v"(r1), "=&s"(m) : "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
                a[4] = k0; a[5] = k1; a[6] = r0; a[7] = r1;
            } else if (KIND == 15) {  // keys by min / max, references by two selects
                float k0, k1, r0, r1;
                asm volatile("v_cmp_lt_f32 vcc, %4, %5\n\tv_min_f32 %0, %4, %5\n\tv_max_f32 %1, %5, %4\n\t"
                             "v_cndmask_b32 %2, %6, %7, vcc\n\tv_cndmask_b32 %3, %7, %6, vcc"
                             : "=&v"(k0), "=&v"(k1), "=&v"(r0), "=&v"(r1) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "vcc");
                a[0] = k0; a[1] = k1; a[2] = r0; a[3] = r1;
                asm volatile("v_cmp_lt_f32 vcc, %4, %5\n\tv_min_f32 %0, %4, %5\n\tv_max_f32 %1, %5, %4\n\t"
                             "v_cndmask_b32 %2, %6, %7, vcc\n\tv_cndmask_b32 %3, %7, %6, vcc"
                             : "=&v"(k0), "=&v"(k1), "=&v"(r0), "=&v"(r1) : "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "vcc");
                a[4] = k0; a[5] = k1; a[6] = r0; a[7] = r1;
            } else if (KIND == 8) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"((u & 1) ? -c : c));
            } else if (KIND == 9) {   // the slab pattern: sub then mul, separate instructions
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_sub_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %2" : "+v"(a[i]) : "v"(c), "v"(ci));
            } else if (KIND == 7) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += a[i];
    for (int i = 0; i < 4; i++) s += p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}
template <int KIND>
void run(const char *name, int blocks_per_cu, float *d) {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount, grid = cus * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<KIND><<<grid, 256>>>(d, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<grid, 256>>>(d, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: blocks_per_cu waves (a block of 256 = one wave per SIMD), each ITER * UNROLL instructions
    const double instr_per_simd = (double)blocks_per_cu * ITER * UNROLL;
    const double ns_per = ms * 1e6 / instr_per_simd;
    printf("%-26s waves/SIMD %d: %.3f ms, %.3f ns per wave-instruction per SIMD = %.2f cycles at 2.4 GHz (%.2f at 2.1)\n", name,
           blocks_per_cu, ms, ns_per, ns_per * 2.4, ns_per * 2.1);
}
int main() {
    float *d;
    hipMalloc(&d, 4);
    for (int w : {1, 5, 8}) {
        run<0>("v_fma_f32", w, d);
        run<1>("v_mul_f32", w, d);
        run<4>("v_min_f32", w, d);
        run<5>("v_cndmask_b32", w, d);
        run<7>("v_max3_f32", w, d);
        run<10>("cndmask sgpr", w, d);
        run<11>("cmp+cndmask", w, d);
        run<12>("cmp+3cndmask", w, d);
        run<13>("cswap vcc (10 instr / 16)", w, d);
        run<14>("cswap sgpr (10 / 16)", w, d);
        run<15>("cswap minmax (10 / 16)", w, d);
        run<8>("v_sub_f32", w, d);
        run<9>("v_sub+v_mul", w, d);
        run<2>("v_pk_mul_f32", w, d);
        run<3>("v_pk_add_f32", w, d);
        run<6>("v_pk_fma_f32", w, d);
    }
    return 0;
}
