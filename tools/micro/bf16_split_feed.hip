// micro-benchmark for a LATER round (not part of the product): can the LSTM K-loop be fed fast enough if the fp32
// products are emulated with split-bf16 MFMAs (v_mfma_f32_32x32x16_bf16; 3 pieces per operand, NPROD = 6 or 9
// piece products per fp32 product)?  Same operand scheme as dsp_lstm4_kernel: a wave owns one unit tile (4 gate
// fragments) x 2 site tiles, weights stream from L2 (one 4.5 MB image shared by every workgroup, re-read every
// "step"), activations come from a small per-workgroup buffer; no cell phase, no barrier -- this isolates the feed.
// Prints achieved MFMA utilisation and the implied speed-up over the fp32 MFMA K-loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NQ = 48;   // K = 768 in stages of 16
constexpr int T = 13;

template <int NPROD>
__global__ __launch_bounds__(512, 2) void feed(const f32x4* __restrict__ w, const f32x4* __restrict__ x, float* out,
                                               long long* cyc) {
    const int lane = threadIdx.x & 63;
    const int u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // unit tile of this wave
    // weights image: [unit tile][stage][gate][piece][lane] 16 B ; activations: [wg][stage][tile][piece][lane] 16 B
    const f32x4* wq = w + ((size_t)u * NQ * 12) * 64 + lane;
    const f32x4* xq = x + ((size_t)(blockIdx.x & 255) * NQ * 6) * 64 + lane;
    f32x16 acc[4][2];
    for (int g = 0; g < 4; ++g)
        for (int m = 0; m < 2; ++m)
            for (int i = 0; i < 16; ++i) acc[g][m][i] = 0.f;
    f32x4 A[4][3], B[2][3];
    auto loadA = [&](int q) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int p = 0; p < 3; ++p) A[g][p] = wq[(size_t)(q * 12 + g * 3 + p) * 64];
    };
    auto loadB = [&](int q) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int p = 0; p < 3; ++p) B[m][p] = xq[(size_t)(q * 6 + m * 3 + p) * 64];
    };
    loadA(0); loadB(0);
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int step = 0; step < T; ++step) {
        for (int q = 0; q < NQ; ++q) {
            const int qn = q + 1 < NQ ? q + 1 : 0;
            f32x4 Bc[2][3];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) Bc[m][p] = B[m][p];
            loadB(qn);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // piece products in decreasing weight: hh, hm, mh, hl, lh, mm, (ml, lm, ll)
                const int pa[9] = {0, 0, 1, 0, 2, 1, 1, 2, 2}, pb[9] = {0, 1, 0, 2, 0, 1, 2, 1, 2};
#pragma unroll
                for (int k = 0; k < NPROD; ++k) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, A[g][pa[k]]);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        acc[g][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, Bc[m][pb[k]]), acc[g][m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < 3; ++p) A[g][p] = wq[(size_t)(qn * 12 + g * 3 + p) * 64];  // late refill of this fragment
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int g = 0; g < 4; ++g)
        for (int m = 0; m < 2; ++m)
            for (int i = 0; i < 16; ++i) s += acc[g][m][i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    const size_t wbytes = (size_t)8 * NQ * 12 * 64 * 16, xbytes = (size_t)256 * NQ * 6 * 64 * 16;
    void *w, *x; float* o; long long* c;
    hipMalloc(&w, wbytes); hipMalloc(&x, xbytes); hipMalloc(&o, (size_t)4096 * 512 * 4); hipMalloc(&c, 8);
    hipMemset(w, 0, wbytes); hipMemset(x, 0, xbytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 2048;  // 8 rounds of 256 workgroups, like one direction pair of a 65,536-site launch
    for (int nprod : {6, 9}) {
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            if (nprod == 6) hipLaunchKernelGGL(feed<6>, dim3(grid), dim3(512), 0, 0, (const f32x4*)w, (const f32x4*)x, o, c);
            else hipLaunchKernelGGL(feed<9>, dim3(grid), dim3(512), 0, 0, (const f32x4*)w, (const f32x4*)x, o, c);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        // per workgroup-pass: T*NQ stages x (4 gates x 2 tiles x NPROD) MFMAs x 32 cycles per wave, 2 waves per SIMD
        const double mfma_cycles_per_simd = (double)T * NQ * 8 * nprod * 32 * 2;
        const double passes = grid / 256.0;
        const double ideal_ms = passes * mfma_cycles_per_simd / 2.4e9 * 1e3;
        // fp32 K-loop of the same shape: T * 96 stages x 32 MFMAs x 64 cycles x 2 waves
        const double fp32_ms = passes * (double)T * 96 * 32 * 64 * 2 / 2.4e9 * 1e3;
        printf("bf16x%d: %.3f ms  (MFMA-bound ideal %.3f ms -> %.1f %% utilisation; fp32 MFMA K-loop ideal %.3f ms -> x%.2f)\n", nprod, ms,
               ideal_ms, 100.0 * ideal_ms / ms, fp32_ms, fp32_ms / ms);
        const double abytes = passes * 256 * 8 * (double)T * NQ * 12 * 1024;
        printf("        weight stream %.1f TB/s out of L2 (%.1f B/clk/CU)\n", abytes / (ms * 1e-3) / 1e12, abytes / (ms * 1e-3) / 256 / 2.4e9);
    }
    return 0;
}
