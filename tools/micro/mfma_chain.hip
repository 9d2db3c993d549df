// Does a lone wave per SIMD lose MFMA-pipe cycles when consecutive v_mfma_f32_32x32x2_f32 accumulate into the SAME tile?
// The LSTM kernels issue a fragment's four MFMAs back to back on one accumulator (a dependent chain of 4), then the next
// fragment's.  CHAIN = length of the dependent run (1 = round robin over the NACC accumulators, 4 = the kernels' order,
// 0 = all on one accumulator).  WAVES per SIMD 1 or 2.  Prints cycles per MFMA (64 = the pipe limit).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAIN, int NACC>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float x = threadIdx.x * 1e-3f, y = 1.0f + x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (CHAIN == 0) {
#pragma unroll
            for (int e = 0; e < 4 * NACC; ++e) { acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[0], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
        } else if (CHAIN == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < NACC; ++j) { acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[j], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
        } else {
#pragma unroll
            for (int j = 0; j < NACC; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[j], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int j = 0; j < NACC; ++j) r += acc[j][3];
    if (r == 12345.678f) out[threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CHAIN, int NACC>
void run(float* d, unsigned long long* c, int threads, const char* what) {
    const int iters = 4096;
    k<CHAIN, NACC><<<256, threads>>>(d, c, iters);
    k<CHAIN, NACC><<<256, threads>>>(d, c, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h = 0;
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / ((double)iters * 4 * NACC) / (threads / 256);   // s_memtime ticks at 100 MHz: converted below
    printf("%-34s waves/SIMD %d  accumulators %d: %.2f memtime ticks per MFMA and wave slot\n", what, threads / 256, NACC, per);
}
int main() {
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, 4096); (void)hipMalloc(&c, 8);
    for (int threads : {256, 512}) {
        run<0, 4>(d, c, threads, "all on one accumulator");
        run<4, 4>(d, c, threads, "runs of 4 per accumulator (kernels)");
        run<1, 4>(d, c, threads, "round robin");
        run<4, 8>(d, c, threads, "runs of 4 per accumulator");
        run<1, 8>(d, c, threads, "round robin");
    }
    // reference: the wall time of the same loops gives cycles = ms * clock; print the event time of one config too
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int mode = 0; mode < 2; ++mode) {
        (void)hipEventRecord(a);
        if (mode == 0) k<4, 4><<<256, 256>>>(d, c, 4096); else k<1, 4><<<256, 256>>>(d, c, 4096);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        printf("%s, one wave per SIMD: %.3f ms for %d MFMAs per wave = %.1f ns per MFMA (64 cycles at 2.4 GHz = 26.7 ns)\n",
               mode == 0 ? "runs of 4" : "round robin", ms, 4096 * 16, ms * 1e6 / (4096 * 16));
    }
    return 0;
}
