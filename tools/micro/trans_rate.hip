// micro-benchmark: issue cost of v_exp_f32 / v_rcp_f32 / v_fma_f32 for ONE wave per SIMD (the regime of the LSTM
// cell phase).  Prints cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) x[i] = __builtin_amdgcn_exp2f(x[i]);
            if (MODE == 1) x[i] = __builtin_amdgcn_rcpf(x[i]);
            if (MODE == 2) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
            if (MODE == 3) { x[i] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x[i] * -1.44f)); }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* o; long long* c; hipMalloc(&o, 1 << 20); hipMalloc(&c, 8);
    const int iters = 2000;
    const char* names[] = {"v_exp_f32", "v_rcp_f32", "v_fma_f32", "sigmoid(4 instr)"};
    for (int waves = 1; waves <= 2; ++waves) {
        for (int m = 0; m < 4; ++m) {
            long long h = 0;
            for (int rep = 0; rep < 2; ++rep) {
                if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * waves), 0, 0, o, c, iters);
                if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * waves), 0, 0, o, c, iters);
                if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256 * waves), 0, 0, o, c, iters);
                if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256 * waves), 0, 0, o, c, iters);
                hipDeviceSynchronize();
                hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
            }
            printf("%d wave(s)/SIMD  %-18s %6.2f ticks per wave-instruction-group (16 independent chains)\n", waves, names[m],
                   (double)h / (iters * 16.0));
        }
    }
    return 0;
}
