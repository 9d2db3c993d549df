// Do independent VALU / transcendental instructions of the SAME wave issue in the shadow of its fp32 MFMAs?
// One wave per SIMD; per v_mfma_f32_32x32x2_f32 (64 cycles in the pipe) the wave also issues NV independent
// v_fma_f32 (MODE 0) or NV v_exp_f32 (MODE 1).  Time vs NV: flat => they overlap.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_same_wave mfma_same_wave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float* out, int nm) {
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = 0.5f + threadIdx.x * 1e-4f + j;
    const float x = threadIdx.x * 1e-3f, y = 1.0f + x;
    for (int i = 0; i < nm; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < NV; ++e) {
                const int s = (j * NV + e) & 15;
                if (MODE == 0) v[s] = __builtin_fmaf(v[s], 1.0001f, 0.5f);
                else v[s] = __builtin_amdgcn_exp2f(v[s]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0.f;
    for (int j = 0; j < 8; ++j) r += acc[j][3];
    for (int j = 0; j < 16; ++j) r += v[j];
    if (r == 12345.678f) out[threadIdx.x] = r;
}
template <int MODE, int NV>
float run(float* d, int nm) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<MODE, NV><<<256, 256>>>(d, nm);
    (void)hipEventRecord(a);
    k<MODE, NV><<<256, 256>>>(d, nm);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}
int main() {
    float* d;
    (void)hipMalloc(&d, 4096);
    const int nm = 65536;
    run<0, 0>(d, nm); run<0, 0>(d, nm);
    printf("v_fma_f32 per MFMA:  0: %.3f  2: %.3f  4: %.3f  8: %.3f  12: %.3f  16: %.3f ms\n", run<0, 0>(d, nm), run<0, 2>(d, nm), run<0, 4>(d, nm),
           run<0, 8>(d, nm), run<0, 12>(d, nm), run<0, 16>(d, nm));
    printf("v_exp_f32 per MFMA:  0: %.3f  1: %.3f  2: %.3f  3: %.3f  4: %.3f  6: %.3f ms\n", run<1, 0>(d, nm), run<1, 1>(d, nm), run<1, 2>(d, nm),
           run<1, 3>(d, nm), run<1, 4>(d, nm), run<1, 6>(d, nm));
    return 0;
}
