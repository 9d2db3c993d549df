// Does VALU / transcendental work of one wave overlap the fp32 MFMA stream of the other wave on the same SIMD?
// 512-thread workgroups (two waves per SIMD): waves 0-3 run NM v_mfma_f32_32x32x2_f32, waves 4-7 run NV rounds of a
// VALU mix (mode 1: v_exp_f32 + v_rcp_f32, mode 2: v_fma_f32, mode 3: v_pk_fma_f32).  Prints the time of
// MFMA only, VALU only and both: both ~ max => overlap, both ~ sum => the two share the SIMD's execution slots.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, bool BF16>
__global__ __launch_bounds__(512, 2) void k(float* out, int nm, int nv, int swap, int prio) {
    int w = threadIdx.x >> 6;
    if (swap) w = 7 - w;  // swap: the VALU waves are the OLDER ones
    if (prio == 1 && w < 4) __builtin_amdgcn_s_setprio(2);   // MFMA waves prioritised
    if (prio == 2 && w >= 4) __builtin_amdgcn_s_setprio(2);  // VALU waves prioritised
    float r = 0.f;
    if (w < 4) {
        f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
        float x = threadIdx.x * 1e-3f, y = 1.0f + x;
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        bf16x8 bx, by;
        for (int i = 0; i < 8; ++i) { bx[i] = (__bf16)x; by[i] = (__bf16)y; }
        for (int i = 0; i < nm; i += 4) {
            if (BF16) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a3, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (nv > 0) {
        float v[8];
        for (int j = 0; j < 8; ++j) v[j] = 0.5f + threadIdx.x * 1e-4f + j;
        for (int i = 0; i < nv; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE == 1) v[j] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[j]));
                else if (MODE == 2) { v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f); v[j] = __builtin_fmaf(v[j], 0.9999f, -0.5f); }
            }
            if (MODE == 3) {
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    f32x2 p = {v[j], v[j + 1]};
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"((f32x2){1.0001f, 1.0001f}), "v"((f32x2){0.5f, 0.5f}));
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"((f32x2){0.9999f, 0.9999f}), "v"((f32x2){-0.5f, -0.5f}));
                    v[j] = p[0]; v[j + 1] = p[1];
                }
            }
        }
        for (int j = 0; j < 8; ++j) r += v[j];
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int MODE, bool BF16>
float run(float* d, int nm, int nv, int swap = 0, int prio = 0) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE, BF16><<<256 * 4, 512>>>(d, nm, nv, swap, prio);
    hipEventRecord(a);
    k<MODE, BF16><<<256 * 4, 512>>>(d, nm, nv, swap, prio);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

template <int MODE, bool BF16>
void test(float* d, const char* name, int nm, int nv) {
    const float m = run<MODE, BF16>(d, nm, 0), v = run<MODE, BF16>(d, 0, nv), both = run<MODE, BF16>(d, nm, nv);
    printf("%-34s mfma %.3f ms  valu %.3f ms  both %.3f ms  (max %.3f, sum %.3f)", name, m, v, both, m > v ? m : v, m + v);
    printf("  | valu older: %.3f  | mfma prio: %.3f / valu older %.3f | valu prio: %.3f / valu older %.3f\n", run<MODE, BF16>(d, nm, nv, 1, 0),
           run<MODE, BF16>(d, nm, nv, 0, 1), run<MODE, BF16>(d, nm, nv, 1, 1), run<MODE, BF16>(d, nm, nv, 0, 2), run<MODE, BF16>(d, nm, nv, 1, 2));
}

int main() {
    float* d;
    hipMalloc(&d, 4096);
    const int nm = 16384;  // MFMAs per wave
    run<1, false>(d, nm, 2048); run<1, false>(d, nm, 2048); run<1, false>(d, nm, 2048);  // warm up the clocks
    test<1, false>(d, "f32 mfma || exp+rcp", nm, 2048);
    test<2, false>(d, "f32 mfma || v_fma_f32", nm, 4096);
    test<3, false>(d, "f32 mfma || v_pk_fma_f32", nm, 4096);
    test<1, true>(d, "bf16 mfma || exp+rcp", 2 * nm, 2048);
    test<2, true>(d, "bf16 mfma || v_fma_f32", 2 * nm, 4096);
    test<1, false>(d, "f32 mfma || exp+rcp (half the valu)", nm, 1024);
    return 0;
}
