// which element of A (32 x 16), B (16 x 32) and C (32 x 32) does a lane hold for v_mfma_f32_32x32x16_bf16 ?
// Assumed: A row = lane % 32, k = 8 * (lane / 32) + j;  B col = lane % 32, k = 8 * (lane / 32) + j;
//          C[r]: row = 8 * (r / 4) + 4 * (lane / 32) + r % 4, col = lane % 32.   Prints the max error.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* C) {
    const int lane = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)A[(lane % 32) * 16 + 8 * (lane / 32) + j];
        b[j] = (__bf16)B[(8 * (lane / 32) + j) * 32 + lane % 32];
    }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[(8 * (r / 4) + 4 * (lane / 32) + r % 4) * 32 + lane % 32] = acc[r];
}
int main() {
    float hA[32 * 16], hB[16 * 32], hC[32 * 32];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
    float *A, *B, *C; hipMalloc(&A, sizeof hA); hipMalloc(&B, sizeof hB); hipMalloc(&C, sizeof hC);
    hipMemcpy(A, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(B, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, C);
    hipMemcpy(hC, C, sizeof hC, hipMemcpyDeviceToHost);
    double err = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { double s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[m * 16 + kk] * hB[kk * 32 + n]; err = fmax(err, fabs(s - hC[m * 32 + n])); }
    printf("max |C - A.B| with the assumed layout: %g\n", err);
    return 0;
}
