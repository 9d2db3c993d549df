// The LSTM cell phase of dsp_lstm_kernel in isolation (one wave per SIMD): 8 accumulator tiles (4 gates x 2 site tiles)
// -> c', h for 32 (unit, site) elements per lane; c in LDS, pre-scaled biases in LDS, h stored with buffer-like 16-byte
// stores.  Variant 0: the shipped formulation (3 sigmoid + 2 tanh = 5 v_exp + 5 v_rcp per element).  Variant 1: one
// common denominator per output (5 v_exp + 2 v_rcp), packed fp32 arithmetic.  Prints shader cycles per cell phase and
// the largest difference between the two on random and on saturating inputs.
// build: hipcc --offload-arch=gfx950 -O3 -o cell_phase cell_phase.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float sigmoid_pre(float x, float bp) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(x, -1.4426950408889634f, bp)));
}
__device__ __forceinline__ float tanh_pre(float x, float bp) {
    return __builtin_fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(x, -2.8853900817779268f, bp))), -1.0f);
}
__device__ __forceinline__ float fast_tanh(float x) {
    return __builtin_fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)), -1.0f);
}

template <int V>
__device__ __forceinline__ void cell(const f32x16 (&acc)[4][2], f32x4* c_lds, const f32x4* b_my, int tid, int nthr, f32x4* out) {
#pragma unroll
    for (int aa = 0; aa < 4; ++aa) {
        const f32x4 bi = b_my[aa * 8 + 0], bf = b_my[aa * 8 + 2], bg = b_my[aa * 8 + 4], bo = b_my[aa * 8 + 6];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            f32x4 cv = c_lds[(m * 4 + aa) * nthr + tid];
            f32x4 hv;
            if (V == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * aa + i;
                    const float ig = sigmoid_pre(acc[0][m][r], bi[i]);
                    const float fg = sigmoid_pre(acc[1][m][r], bf[i]);
                    const float gg = tanh_pre(acc[2][m][r], bg[i]);
                    const float og = sigmoid_pre(acc[3][m][r], bo[i]);
                    const float cn = __builtin_fmaf(fg, cv[i], ig * gg);
                    cv[i] = cn;
                    hv[i] = og * fast_tanh(cn);
                }
            } else if (V == 1) {
                // sigmoid(a) = 1/(1+ea), tanh(b) = (1-eb)/(1+eb) with ea = exp2(-log2e*a), eb = exp2(-2*log2e*b):
                //   c' = f*c + i*g = [c*(1+ea)(1+eb) + (1-eb)(1+ef)] / [(1+ef)(1+ea)(1+eb)]
                //   h  = o*tanh(c') = (1-ec) / [(1+eo)(1+ec)]
                // the exponents are capped at 2^30 (a sigmoid below 1e-9 becomes 1e-9), so no product overflows
                const f32x2 k1 = {-1.4426950408889634f, -1.4426950408889634f}, k2 = {-2.8853900817779268f, -2.8853900817779268f};
                const f32x2 one = {1.f, 1.f};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int r = 4 * aa + 2 * j;
                    f32x2 xa = {acc[0][m][r], acc[0][m][r + 1]}, xf = {acc[1][m][r], acc[1][m][r + 1]};
                    f32x2 xg = {acc[2][m][r], acc[2][m][r + 1]}, xo = {acc[3][m][r], acc[3][m][r + 1]};
                    f32x2 ba = {bi[2 * j], bi[2 * j + 1]}, bff = {bf[2 * j], bf[2 * j + 1]}, bgg = {bg[2 * j], bg[2 * j + 1]}, boo = {bo[2 * j], bo[2 * j + 1]};
                    f32x2 c = {cv[2 * j], cv[2 * j + 1]};
                    f32x2 ta = __builtin_elementwise_fma(xa, k1, ba), tf = __builtin_elementwise_fma(xf, k1, bff);
                    f32x2 tg = __builtin_elementwise_fma(xg, k2, bgg), to = __builtin_elementwise_fma(xo, k1, boo);
                    f32x2 ea, ef, eb, eo;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        ea[e] = __builtin_amdgcn_exp2f(__builtin_fminf(ta[e], 30.f));
                        ef[e] = __builtin_amdgcn_exp2f(__builtin_fminf(tf[e], 30.f));
                        eb[e] = __builtin_amdgcn_exp2f(__builtin_fminf(tg[e], 30.f));
                        eo[e] = __builtin_amdgcn_exp2f(__builtin_fminf(to[e], 30.f));
                    }
                    const f32x2 A = one + ea, F = one + ef, B = one + eb, nb = one - eb;
                    const f32x2 P = A * B, den = F * P, num = __builtin_elementwise_fma(c, P, nb * F);
                    f32x2 rd;
                    rd[0] = __builtin_amdgcn_rcpf(den[0]); rd[1] = __builtin_amdgcn_rcpf(den[1]);
                    const f32x2 cn = num * rd;
                    const f32x2 tc = cn * k2;
                    f32x2 ec;
                    ec[0] = __builtin_amdgcn_exp2f(__builtin_fminf(tc[0], 30.f)); ec[1] = __builtin_amdgcn_exp2f(__builtin_fminf(tc[1], 30.f));
                    const f32x2 O = one + eo, C = one + ec, nc = one - ec;
                    const f32x2 d2 = O * C;
                    f32x2 r2;
                    r2[0] = __builtin_amdgcn_rcpf(d2[0]); r2[1] = __builtin_amdgcn_rcpf(d2[1]);
                    const f32x2 h = nc * r2;
                    cv[2 * j] = cn[0]; cv[2 * j + 1] = cn[1];
                    hv[2 * j] = h[0]; hv[2 * j + 1] = h[1];
                }
            }
            if (V == 2) {
                // c' = c / (1+ef) + (1-eb) / ((1+ea)(1+eb)),  h = (1-ec) / ((1+eo)(1+ec)): 5 v_exp + 3 v_rcp; only the two
                // tanh exponentials need a cap (an infinite sigmoid exponential gives rcp = 0, times a finite factor)
                const f32x2 k1 = {-1.4426950408889634f, -1.4426950408889634f}, k2 = {-2.8853900817779268f, -2.8853900817779268f};
                const f32x2 one = {1.f, 1.f};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int r = 4 * aa + 2 * j;
                    f32x2 xa = {acc[0][m][r], acc[0][m][r + 1]}, xf = {acc[1][m][r], acc[1][m][r + 1]};
                    f32x2 xg = {acc[2][m][r], acc[2][m][r + 1]}, xo = {acc[3][m][r], acc[3][m][r + 1]};
                    f32x2 ba = {bi[2 * j], bi[2 * j + 1]}, bff = {bf[2 * j], bf[2 * j + 1]}, bgg = {bg[2 * j], bg[2 * j + 1]}, boo = {bo[2 * j], bo[2 * j + 1]};
                    f32x2 c = {cv[2 * j], cv[2 * j + 1]};
                    f32x2 ta = __builtin_elementwise_fma(xa, k1, ba), tf = __builtin_elementwise_fma(xf, k1, bff);
                    f32x2 tg = __builtin_elementwise_fma(xg, k2, bgg), to = __builtin_elementwise_fma(xo, k1, boo);
                    f32x2 ea, ef, eb, eo;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        ea[e] = __builtin_amdgcn_exp2f(ta[e]);
                        ef[e] = __builtin_amdgcn_exp2f(tf[e]);
                        eb[e] = __builtin_amdgcn_exp2f(__builtin_fminf(tg[e], 30.f));
                        eo[e] = __builtin_amdgcn_exp2f(to[e]);
                    }
                    const f32x2 A = one + ea, F = one + ef, B = one + eb, nb = one - eb;
                    const f32x2 P = A * B;
                    f32x2 r1, r2;
                    r1[0] = __builtin_amdgcn_rcpf(F[0]); r1[1] = __builtin_amdgcn_rcpf(F[1]);
                    r2[0] = __builtin_amdgcn_rcpf(P[0]); r2[1] = __builtin_amdgcn_rcpf(P[1]);
                    const f32x2 cn = __builtin_elementwise_fma(c, r1, nb * r2);
                    const f32x2 tc = cn * k2;
                    f32x2 ec;
                    ec[0] = __builtin_amdgcn_exp2f(__builtin_fminf(tc[0], 30.f)); ec[1] = __builtin_amdgcn_exp2f(__builtin_fminf(tc[1], 30.f));
                    const f32x2 O = one + eo, C = one + ec, nc = one - ec;
                    const f32x2 d2 = O * C;
                    f32x2 r3;
                    r3[0] = __builtin_amdgcn_rcpf(d2[0]); r3[1] = __builtin_amdgcn_rcpf(d2[1]);
                    const f32x2 h = nc * r3;
                    cv[2 * j] = cn[0]; cv[2 * j + 1] = cn[1];
                    hv[2 * j] = h[0]; hv[2 * j + 1] = h[1];
                }
            }
            c_lds[(m * 4 + aa) * nthr + tid] = cv;
            out[(m * 4 + aa) * nthr + tid] = hv;
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void k(const float* in, const float* bias, const float* c0, float* out, float* cout, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    f32x4* c_lds = (f32x4*)smem;
    f32x4* b_lds = c_lds + 8 * nthr;
    for (int i = tid; i < 128; i += nthr) b_lds[i] = ((const f32x4*)bias)[i];
    for (int j = 0; j < 8; ++j) c_lds[j * nthr + tid] = ((const f32x4*)c0)[(size_t)blockIdx.x * 8 * nthr + j * nthr + tid];
    f32x16 acc[4][2];
    for (int g = 0; g < 4; ++g)
        for (int m = 0; m < 2; ++m)
            for (int r = 0; r < 16; ++r) acc[g][m][r] = in[(((size_t)blockIdx.x * 8 + g * 2 + m) * 16 + r) * nthr + tid];
    __syncthreads();
    const f32x4* b_my = b_lds + (tid >> 6) * 32 + ((tid & 63) >> 5);
    f32x4* o4 = (f32x4*)out + (size_t)blockIdx.x * 8 * nthr;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        cell<V>(acc, c_lds, b_my, tid, nthr, o4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g)  // keep the inputs loop-variant at no real cost
            acc[g][0][it & 15] += 0.0f * acc[g][1][0];
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < 8; ++j) ((f32x4*)cout)[(size_t)blockIdx.x * 8 * nthr + j * nthr + tid] = c_lds[j * nthr + tid];
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    const int nthr = 256, blocks = 256;
    const size_t nin = (size_t)blocks * 8 * 16 * nthr, nc = (size_t)blocks * 8 * nthr * 4;
    std::vector<float> in(nin), bias(512), c0(nc);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (int pass = 0; pass < 2; ++pass) {
        const float scale = pass == 0 ? 3.f : 80.f;  // pass 1: saturating pre-activations
        for (auto& v : in) v = rnd() * scale;
        for (auto& v : bias) v = rnd() * -1.44f;
        for (auto& v : c0) v = rnd() * (pass == 0 ? 1.f : 12.f);
        float *din, *db, *dc, *dout[3], *dcout[3]; long long* dcy;
        hipMalloc(&din, nin * 4); hipMalloc(&db, 512 * 4); hipMalloc(&dc, nc * 4); hipMalloc(&dcy, 8);
        hipMemcpy(din, in.data(), nin * 4, hipMemcpyHostToDevice); hipMemcpy(db, bias.data(), 512 * 4, hipMemcpyHostToDevice);
        hipMemcpy(dc, c0.data(), nc * 4, hipMemcpyHostToDevice);
        std::vector<float> h[3], c[3];
        for (int v = 0; v < 3; ++v) {
            hipMalloc(&dout[v], nc * 4); hipMalloc(&dcout[v], nc * 4);
            const size_t lds = 8 * nthr * 16 + 128 * 16;
            long long cy = 0;
            for (int iters : {1, 201}) {
                if (v == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(nthr), lds, 0, din, db, dc, dout[v], dcout[v], dcy, iters);
                else if (v == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(nthr), lds, 0, din, db, dc, dout[v], dcout[v], dcy, iters);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(nthr), lds, 0, din, db, dc, dout[v], dcout[v], dcy, iters);
                hipDeviceSynchronize();
                long long t; hipMemcpy(&t, dcy, 8, hipMemcpyDeviceToHost);
                if (iters == 1) { h[v].resize(nc); c[v].resize(nc); hipMemcpy(h[v].data(), dout[v], nc * 4, hipMemcpyDeviceToHost); hipMemcpy(c[v].data(), dcout[v], nc * 4, hipMemcpyDeviceToHost); cy = t; }
                else cy = (t - cy) / 200;
            }
            printf("pass %d variant %d: %lld cycles per cell phase (32 elements per lane, one wave per SIMD)\n", pass, v, cy);
        }
        for (int v = 1; v < 3; ++v) {
            double dh = 0, dcc = 0; int bad = 0;
            for (size_t i = 0; i < nc; ++i) {
                if (!std::isfinite(h[v][i]) || !std::isfinite(c[v][i])) ++bad;
                dh = std::fmax(dh, std::fabs((double)h[0][i] - h[v][i]));
                dcc = std::fmax(dcc, std::fabs((double)c[0][i] - c[v][i]) / (1.0 + std::fabs((double)c[0][i])));
            }
            printf("pass %d variant %d vs 0: max |dh| %.3e, max rel |dc| %.3e, non-finite %d\n", pass, v, dh, dcc, bad);
        }
    }
    return 0;
}
