// Does the LSTM cell phase (tools/micro/cell_phase.hip, the shipped formulation) of one wave overlap the fp32 MFMA stream
// of the other wave on the same SIMD?  512-thread workgroups: waves 0-3 issue NM v_mfma_f32_32x32x2_f32 on 8 accumulator
// tiles (the k-loop's shape), waves 4-7 run ITERS cell phases.  MODE: 0 full cell, 1 without LDS, 2 without the
// transcendentals (FMAs instead), 3 only transcendentals + arithmetic on registers (no LDS, no stores).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_cell_overlap mfma_cell_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoid_pre(float x, float bp) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(x, -1.4426950408889634f, bp)));
}
__device__ __forceinline__ float tanh_pre(float x, float bp) {
    return __builtin_fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(x, -2.8853900817779268f, bp))), -1.0f);
}
__device__ __forceinline__ float fast_tanh(float x) {
    return __builtin_fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)), -1.0f);
}
__device__ __forceinline__ float fsig(float x, float bp) { float t = __builtin_fmaf(x, -1.44f, bp); t = __builtin_fmaf(t, t, 1.0f); return __builtin_fmaf(t, 0.25f, 0.1f); }

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const float* in, float* out, int nm, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, nthr = blockDim.x, w = tid >> 6;
    f32x4* c_lds = (f32x4*)smem;
    f32x4* b_lds = c_lds + 8 * nthr;
    for (int i = tid; i < 8 * nthr + 256; i += nthr) c_lds[i] = (f32x4){0.1f, 0.2f, 0.3f, 0.4f};
    f32x16 acc[4][2];
    for (int g = 0; g < 4; ++g)
        for (int m = 0; m < 2; ++m)
            for (int r = 0; r < 16; ++r) acc[g][m][r] = in[((g * 2 + m) * 16 + r) * 64 + (tid & 63)];
    __syncthreads();
    if (w < 4) {
        float x = tid * 1e-3f, y = 1.0f + x;
        for (int i = 0; i < nm; i += 8) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[g][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[g][m], 0, 0, 0);
        }
    } else {
        const f32x4* b_my = b_lds + (w & 3) * 32 + ((tid & 63) >> 5);
        f32x4* o4 = (f32x4*)out + (size_t)blockIdx.x * 8 * nthr;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                f32x4 bi, bf, bg, bo;
                if (MODE == 1 || MODE == 3) { bi = bf = bg = bo = (f32x4){0.1f, 0.2f, 0.3f, 0.4f}; }
                else { bi = b_my[aa * 8 + 0]; bf = b_my[aa * 8 + 2]; bg = b_my[aa * 8 + 4]; bo = b_my[aa * 8 + 6]; }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f32x4 cv;
                    if (MODE == 1 || MODE == 3) cv = (f32x4){acc[0][m][aa], acc[1][m][aa], acc[2][m][aa], acc[3][m][aa]};
                    else cv = c_lds[(m * 4 + aa) * nthr + tid];
                    f32x4 hv;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * aa + i;
                        float ig, fg, gg, og, th;
                        if (MODE == 2) {
                            ig = fsig(acc[0][m][r], bi[i]); fg = fsig(acc[1][m][r], bf[i]); gg = fsig(acc[2][m][r], bg[i]); og = fsig(acc[3][m][r], bo[i]);
                        } else {
                            ig = sigmoid_pre(acc[0][m][r], bi[i]); fg = sigmoid_pre(acc[1][m][r], bf[i]);
                            gg = tanh_pre(acc[2][m][r], bg[i]); og = sigmoid_pre(acc[3][m][r], bo[i]);
                        }
                        const float cn = __builtin_fmaf(fg, cv[i], ig * gg);
                        cv[i] = cn;
                        th = MODE == 2 ? fsig(cn, 0.3f) : fast_tanh(cn);
                        hv[i] = og * th;
                    }
                    if (MODE == 1 || MODE == 3) { acc[0][m][aa] = cv[0]; acc[1][m][4 + aa] = hv[1] + hv[0]; acc[2][m][8 + aa] = hv[2] + hv[3]; }
                    else { c_lds[(m * 4 + aa) * nthr + tid] = cv; }
                    if (MODE != 3) o4[(m * 4 + aa) * nthr + tid] = hv;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0.f;
    for (int g = 0; g < 4; ++g) for (int m = 0; m < 2; ++m) r += acc[g][m][3] + acc[g][m][9];
    if (r == 12345.678f) out[tid] = r;
}

static int g_grid = 512;
template <int MODE>
float run(const float* in, float* out, int nm, int iters) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const size_t lds = (8 * 512 + 256) * 16;
    k<MODE><<<g_grid, 512, lds>>>(in, out, nm, iters);
    (void)hipEventRecord(a);
    k<MODE><<<g_grid, 512, lds>>>(in, out, nm, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}
template <int MODE>
void test(const float* in, float* out, const char* name, int nm, int iters) {
    const float m = run<MODE>(in, out, nm, 0), v = run<MODE>(in, out, 0, iters), both = run<MODE>(in, out, nm, iters);
    printf("%-44s mfma %.3f ms  cell %.3f ms  both %.3f ms  (max %.3f, sum %.3f): %.0f %% of the cell work hidden\n", name, m, v, both,
           m > v ? m : v, m + v, 100.0 * (m + v - both) / (m < v ? m : v));
}
int main() {
    float *in, *out;
    (void)hipMalloc(&in, 8 * 16 * 64 * 4); (void)hipMalloc(&out, (size_t)512 * 8 * 512 * 16);
    (void)hipMemset(in, 0, 8 * 16 * 64 * 4);
    (void)hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int nm = 65536;  // MFMAs per wave: 4.2 M cycles
    run<0>(in, out, nm, 200); run<0>(in, out, nm, 200);
    for (int grid : {256, 512})
    for (int iters : {400}) {
        g_grid = grid;
        printf("-- grid %d: %d cell phases per wave against %d MFMAs\n", grid, iters, nm);
        test<0>(in, out, "full cell (LDS c + bias, stores)", nm, iters);
        test<1>(in, out, "without LDS", nm, iters);
        test<2>(in, out, "FMAs in place of the transcendentals", nm, iters);
        test<3>(in, out, "registers only", nm, iters);
    }
    return 0;
}
