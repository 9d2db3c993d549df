// Second cut of the split-bf16 feed test (see bf16_split_feed.hip): activations stay fp32 in memory (K4 layout
// unchanged for every other kernel) and are split into three bf16 pieces in registers after the load; weights come
// pre-split.  One unit tile x 2 site tiles per wave, 8 waves, late per-fragment refill of the weight pieces.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NQ = 48;  // K = 768 in stages of 16
constexpr int T = 13;

// x (8 fp32) -> hi, mid, lo (8 bf16 each, round to nearest even at every level; hi + mid + lo == x exactly)
__device__ __forceinline__ void split3(const f32x4 x0, const f32x4 x1, u32x4& hi, u32x4& mid, u32x4& lo) {
    float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const unsigned ha = (__builtin_bit_cast(unsigned, a) + 0x7fffu + ((__builtin_bit_cast(unsigned, a) >> 16) & 1u)) & 0xffff0000u;
        const unsigned hb = (__builtin_bit_cast(unsigned, b) + 0x7fffu + ((__builtin_bit_cast(unsigned, b) >> 16) & 1u)) & 0xffff0000u;
        const float ra = a - __builtin_bit_cast(float, ha), rb = b - __builtin_bit_cast(float, hb);
        const unsigned ma = (__builtin_bit_cast(unsigned, ra) + 0x7fffu + ((__builtin_bit_cast(unsigned, ra) >> 16) & 1u)) & 0xffff0000u;
        const unsigned mb = (__builtin_bit_cast(unsigned, rb) + 0x7fffu + ((__builtin_bit_cast(unsigned, rb) >> 16) & 1u)) & 0xffff0000u;
        const float sa = ra - __builtin_bit_cast(float, ma), sb = rb - __builtin_bit_cast(float, mb);
        const unsigned la = __builtin_bit_cast(unsigned, sa) & 0xffff0000u, lb = __builtin_bit_cast(unsigned, sb) & 0xffff0000u;
        hi[i] = (ha >> 16) | hb;
        mid[i] = (ma >> 16) | mb;
        lo[i] = (la >> 16) | lb;
    }
}

template <int NPROD>
__global__ __launch_bounds__(512, 2) void feed(const u32x4* __restrict__ w, const f32x4* __restrict__ x, float* out) {
    const int lane = threadIdx.x & 63;
    const int u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u32x4* wq = w + ((size_t)u * NQ * 12) * 64 + lane;                        // [ut][stage][gate][piece][lane]
    const f32x4* xq = x + ((size_t)(blockIdx.x & 255) * NQ * 4) * 64 + lane;        // [wg][stage][tile][2][lane] fp32
    f32x16 acc[4][2];
    for (int g = 0; g < 4; ++g)
        for (int m = 0; m < 2; ++m)
            for (int i = 0; i < 16; ++i) acc[g][m][i] = 0.f;
    u32x4 A[4][3];
    f32x4 X[2][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int p = 0; p < 3; ++p) A[g][p] = wq[(size_t)(g * 3 + p) * 64];
#pragma unroll
    for (int m = 0; m < 2; ++m) { X[m][0] = xq[(size_t)(m * 2) * 64]; X[m][1] = xq[(size_t)(m * 2 + 1) * 64]; }
    for (int step = 0; step < T; ++step) {
        for (int q = 0; q < NQ; ++q) {
            const int qn = q + 1 < NQ ? q + 1 : 0;
            u32x4 B[2][3];
#pragma unroll
            for (int m = 0; m < 2; ++m) split3(X[m][0], X[m][1], B[m][0], B[m][1], B[m][2]);
#pragma unroll
            for (int m = 0; m < 2; ++m) { X[m][0] = xq[(size_t)(qn * 4 + m * 2) * 64]; X[m][1] = xq[(size_t)(qn * 4 + m * 2 + 1) * 64]; }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // piece products, smallest first: (ll, lm, ml,) mm, lh, hl, mh, hm, hh
                const int pa[9] = {2, 2, 1, 1, 2, 0, 1, 0, 0}, pb[9] = {2, 1, 2, 1, 0, 2, 0, 1, 0};
#pragma unroll
                for (int k = 9 - NPROD; k < 9; ++k) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, A[g][pa[k]]);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        acc[g][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, B[m][pb[k]]), acc[g][m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 3; ++p) A[g][p] = wq[(size_t)(qn * 12 + g * 3 + p) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int g = 0; g < 4; ++g)
        for (int m = 0; m < 2; ++m)
            for (int i = 0; i < 16; ++i) s += acc[g][m][i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const size_t wbytes = (size_t)8 * NQ * 12 * 64 * 16, xbytes = (size_t)256 * NQ * 4 * 64 * 16;
    void *w, *x; float* o;
    hipMalloc(&w, wbytes); hipMalloc(&x, xbytes); hipMalloc(&o, (size_t)4096 * 512 * 4);
    hipMemset(w, 0, wbytes); hipMemset(x, 0, xbytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 2048;
    for (int nprod : {6, 9}) {
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            if (nprod == 6) hipLaunchKernelGGL(feed<6>, dim3(grid), dim3(512), 0, 0, (const u32x4*)w, (const f32x4*)x, o);
            else hipLaunchKernelGGL(feed<9>, dim3(grid), dim3(512), 0, 0, (const u32x4*)w, (const f32x4*)x, o);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double passes = grid / 256.0;
        const double ideal_ms = passes * (double)T * NQ * 8 * nprod * 32 * 2 / 2.4e9 * 1e3;
        const double fp32_ms = passes * (double)T * 96 * 32 * 64 * 2 / 2.4e9 * 1e3;
        printf("bf16x%d (in-register activation split): %.3f ms  (MFMA ideal %.3f ms -> %.1f %%; fp32 K-loop ideal %.3f ms -> x%.2f)\n",
               nprod, ms, ideal_ms, 100.0 * ideal_ms / ms, fp32_ms, fp32_ms / ms);
    }
    return 0;
}
