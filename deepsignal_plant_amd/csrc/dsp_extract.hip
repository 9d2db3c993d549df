// Per-read feature extraction on gfx950: raw DAQ samples + resquiggle events -> the tensors dsp_forward consumes
// (SURVEY.md 8(f) next-3).  Replaces the arithmetic of deepsignal_plant/extract_features.py:
//   _rescale_signals :273-274, _normalize_signals :179-190, the event slicing and per-base np.mean / np.std of
//   _extract_features :331-335, :363-368, and _get_signals_rect :232-251.
// Everything is float64 and reproduces numpy's evaluation order bit for bit (this file is compiled with
// -ffp-contract=off):
//   * np.add.reduce over a contiguous float64 array = 0.0 + pairwise sums of 8192-element buffer chunks, each chunk
//     summed by numpy's pairwise scheme (blocks of <= 128 with 8 accumulators, halves rounded down to a multiple
//     of 8 above that);
//   * np.median = k-th order statistic (mean of the two middle ones for even n), found here with an 8-pass
//     radix select over the order-preserving 64-bit image of the doubles -- no sort, no scratch arrays;
//   * np.around(x, 6) = rint(x * 1e6) / 1e6.
// HBM-bound integer/byte work: one workgroup per read for the read statistics, one thread per base for the base
// statistics, one thread per (site, base) for the window gather; no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dsp_amd.h"
#include "dsp_kernels.h"

namespace {

constexpr double kMadC = 0.6744897501960817;  // norm.ppf(3/4): statsmodels robust.mad's c
constexpr int kChunk = 8192;                  // numpy's reduction buffer (np.getbufsize())

struct ReadView {
    const int16_t* raw;
    int64_t n;
    double scaling, offset;
    __device__ __forceinline__ double pa(int64_t i) const { return scaling * ((double)raw[i] + offset); }
};

// normalised, 6-decimal-rounded sample (extract_features.py:186-190)
struct NormView {
    ReadView rd;
    double shift, scale;
    __device__ __forceinline__ double operator()(int64_t i) const {
        double v = rd.pa(i);
        if (scale != 0.0) v = (v - shift) / scale;
        return rint(v * 1e6) / 1e6;
    }
};

// ---- numpy pairwise summation -------------------------------------------------------------------------------
template <class F>
__device__ double pw_block(const F& f, int64_t lo, int n) {  // n <= 128
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += f(lo + i);
        return r;
    }
    double r0 = f(lo), r1 = f(lo + 1), r2 = f(lo + 2), r3 = f(lo + 3);
    double r4 = f(lo + 4), r5 = f(lo + 5), r6 = f(lo + 6), r7 = f(lo + 7);
    int i = 8;
    const int lim = n - (n % 8);
    for (; i < lim; i += 8) {
        r0 += f(lo + i);     r1 += f(lo + i + 1); r2 += f(lo + i + 2); r3 += f(lo + i + 3);
        r4 += f(lo + i + 4); r5 += f(lo + i + 5); r6 += f(lo + i + 6); r7 += f(lo + i + 7);
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += f(lo + i);
    return res;
}

template <class F>
__device__ double pw_chunk(const F& f, int64_t lo, int n) {  // n <= 8192; explicit-stack post-order walk
    if (n <= 128) return pw_block(f, lo, n);
    int s_lo[16], s_n[16], s_phase[16];
    double s_left[16];
    int sp = 0;
    s_lo[0] = 0; s_n[0] = n; s_phase[0] = 0;
    double ret = 0.0;
    bool have = false;
    while (true) {
        if (!have) {
            if (s_n[sp] <= 128) {
                ret = pw_block(f, lo + s_lo[sp], s_n[sp]);
                have = true;
                sp--;
            } else {
                int n2 = s_n[sp] / 2;
                n2 -= n2 % 8;
                s_phase[sp] = 1;
                s_lo[sp + 1] = s_lo[sp]; s_n[sp + 1] = n2; s_phase[sp + 1] = 0;
                sp++;
            }
        } else {
            if (sp < 0) return ret;
            if (s_phase[sp] == 1) {
                int n2 = s_n[sp] / 2;
                n2 -= n2 % 8;
                s_left[sp] = ret;
                s_phase[sp] = 2;
                s_lo[sp + 1] = s_lo[sp] + n2; s_n[sp + 1] = s_n[sp] - n2; s_phase[sp + 1] = 0;
                sp++;
                have = false;
            } else {
                ret = s_left[sp] + ret;
                sp--;
            }
        }
    }
}

// np.add.reduce of f(lo .. lo+n) by ONE thread
template <class F>
__device__ double np_sum_thread(const F& f, int64_t lo, int64_t n) {
    double total = 0.0;
    for (int64_t c = 0; c < n; c += kChunk) {
        const int m = (int)((n - c) < kChunk ? (n - c) : kChunk);
        total = total + pw_chunk(f, lo + c, m);
    }
    return total;
}

// The same sums by a group of 8 adjacent lanes (lane l8 owns accumulator r[l8] of numpy's 8-way unrolled loop);
// all 8 lanes follow the same control flow and end up with the same value.  IEEE addition is commutative, so
// the xor-butterfly reproduces ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) on every lane.
template <class F>
__device__ double pw_block8(const F& f, int64_t lo, int n, int l8) {  // n <= 128
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += f(lo + i);
        return r;
    }
    double r = f(lo + l8);
    const int lim = n - (n % 8);
    for (int i = 8; i < lim; i += 8) r += f(lo + i + l8);
    r = r + __shfl_xor(r, 1);
    r = r + __shfl_xor(r, 2);
    r = r + __shfl_xor(r, 4);
    for (int i = lim; i < n; i++) r += f(lo + i);
    return r;
}

template <class F>
__device__ double pw_chunk8(const F& f, int64_t lo, int n, int l8) {  // n <= 8192
    if (n <= 128) return pw_block8(f, lo, n, l8);
    int s_lo[16], s_n[16], s_phase[16];
    double s_left[16];
    int sp = 0;
    s_lo[0] = 0; s_n[0] = n; s_phase[0] = 0;
    double ret = 0.0;
    bool have = false;
    while (true) {
        if (!have) {
            if (s_n[sp] <= 128) {
                ret = pw_block8(f, lo + s_lo[sp], s_n[sp], l8);
                have = true;
                sp--;
            } else {
                int n2 = s_n[sp] / 2;
                n2 -= n2 % 8;
                s_phase[sp] = 1;
                s_lo[sp + 1] = s_lo[sp]; s_n[sp + 1] = n2; s_phase[sp + 1] = 0;
                sp++;
            }
        } else {
            if (sp < 0) return ret;
            if (s_phase[sp] == 1) {
                int n2 = s_n[sp] / 2;
                n2 -= n2 % 8;
                s_left[sp] = ret;
                s_phase[sp] = 2;
                s_lo[sp + 1] = s_lo[sp] + n2; s_n[sp + 1] = s_n[sp] - n2; s_phase[sp + 1] = 0;
                sp++;
                have = false;
            } else {
                ret = s_left[sp] + ret;
                sp--;
            }
        }
    }
}

template <class F>
__device__ double np_sum8(const F& f, int64_t lo, int64_t n, int l8) {
    double total = 0.0;
    for (int64_t c = 0; c < n; c += kChunk) {
        const int m = (int)((n - c) < kChunk ? (n - c) : kChunk);
        total = total + pw_chunk8(f, lo + c, m, l8);
    }
    return total;
}

// np.add.reduce of f(0 .. n) by a whole workgroup: one chunk per thread, chunk sums accumulated in order
template <class F>
__device__ double np_sum_block(const F& f, int64_t n, double* lds /*[blockDim.x]*/) {
    double total = 0.0;
    const int64_t nchunks = (n + kChunk - 1) / kChunk;
    for (int64_t c0 = 0; c0 < nchunks; c0 += blockDim.x) {
        const int64_t c = c0 + threadIdx.x;
        __syncthreads();
        if (c < nchunks) {
            const int64_t lo = c * kChunk;
            lds[threadIdx.x] = pw_chunk(f, lo, (int)((n - lo) < kChunk ? (n - lo) : kChunk));
        }
        __syncthreads();
        const int m = (int)((nchunks - c0) < (int64_t)blockDim.x ? (nchunks - c0) : (int64_t)blockDim.x);
        for (int i = 0; i < m; i++) total = total + lds[i];  // every thread: identical order, identical result
    }
    __syncthreads();
    return total;
}

// ---- radix select ----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t ord_bits(double v) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double ord_value(uint64_t k) {
    const uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

struct SelectLds {
    uint32_t hist[256];
    uint32_t sel[2];
    int64_t k[2];
};

// One radix pass over key(0..n) restricted to the candidates whose bits above `shift + 8` equal `prefix`:
// fills s->hist with the histogram of the next byte.  All threads of the workgroup take part.
template <class K>
__device__ void select_pass(const K& key, int64_t n, uint64_t prefix, int shift, int top_shift, SelectLds* s) {
    const int lane = threadIdx.x & 63;
    const uint64_t mask = shift == top_shift ? 0ull : (~0ull << (shift + 8));
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s->hist[i] = 0;
    __syncthreads();
    const int64_t nround = (n + blockDim.x - 1) / blockDim.x * blockDim.x;  // whole waves stay converged
    for (int64_t i = threadIdx.x; i < nround; i += blockDim.x) {
        bool match = false;
        uint32_t d = 0;
        if (i < n) {
            const uint64_t u = key(i);
            match = (u & mask) == prefix;
            d = (uint32_t)(u >> shift) & 255u;
        }
        // samples of one read share their high bytes: count a unanimous wave with one atomic
        const uint64_t m = __ballot(match);
        if (m == 0) continue;
        const int first = __ffsll((long long)m) - 1;
        const uint32_t d0 = (uint32_t)__shfl((int)d, first);
        const uint64_t same = __ballot(match && d == d0);
        if (same == m) {
            if (lane == first) atomicAdd(&s->hist[d0], (uint32_t)__popcll(m));
        } else if (match) {
            atomicAdd(&s->hist[d], 1u);
        }
    }
    __syncthreads();
}

// k-th smallest key (0-based) among the candidates of `prefix`, continuing at byte `shift`
template <class K>
__device__ uint64_t select_from(const K& key, int64_t n, uint64_t prefix, int64_t k, int shift, int top_shift, SelectLds* s) {
    for (; shift >= 0; shift -= 8) {
        select_pass(key, n, prefix, shift, top_shift, s);
        if (threadIdx.x == 0) {
            int64_t cum = 0;
            uint32_t sel = 255;
            for (uint32_t b = 0; b < 256; b++) {
                const int64_t h = s->hist[b];
                if (k < cum + h) { sel = b; break; }
                cum += h;
            }
            s->sel[0] = sel;
            s->k[0] = k - cum;
        }
        __syncthreads();
        prefix |= (uint64_t)s->sel[0] << shift;
        k = s->k[0];
    }
    return prefix;
}

// the k-th and (k+1)-th smallest keys: one walk while both fall into the same bucket, two after they part
template <class K>
__device__ void select_pair(const K& key, int64_t n, int64_t k, int top_shift, SelectLds* s, uint64_t* lo, uint64_t* hi) {
    uint64_t prefix = 0;
    for (int shift = top_shift; shift >= 0; shift -= 8) {
        select_pass(key, n, prefix, shift, top_shift, s);
        if (threadIdx.x == 0) {
            int64_t cum = 0;
            int found = 0;
            for (uint32_t b = 0; b < 256 && found < 2; b++) {
                const int64_t h = s->hist[b];
                while (found < 2 && k + found < cum + h) {
                    s->sel[found] = b;
                    s->k[found] = k + found - cum;
                    found++;
                }
                cum += h;
            }
        }
        __syncthreads();
        const uint32_t b0 = s->sel[0], b1 = s->sel[1];
        const int64_t k0 = s->k[0], k1 = s->k[1];
        if (b0 != b1) {
            const uint64_t p0 = prefix | ((uint64_t)b0 << shift), p1 = prefix | ((uint64_t)b1 << shift);
            *lo = select_from(key, n, p0, k0, shift - 8, top_shift, s);
            *hi = select_from(key, n, p1, k1, shift - 8, top_shift, s);
            return;
        }
        prefix |= (uint64_t)b0 << shift;
        k = k0;
    }
    *lo = *hi = prefix;  // equal keys
}

// np.median of value(key) over n >= 1 elements; `bits` = 64 (doubles through ord_bits) or 16 (raw DAQ codes)
template <class K, class V>
__device__ double np_median(const K& key, const V& value, int64_t n, int top_shift, SelectLds* s) {
    if (n & 1) return value(select_from(key, n, 0, n / 2, top_shift, top_shift, s));
    if (n < 2) return value(select_from(key, n, 0, 0, top_shift, top_shift, s));
    uint64_t lo, hi;
    select_pair(key, n, n / 2 - 1, top_shift, s, &lo, &hi);
    return ((0.0 + value(lo)) + value(hi)) / 2.0;  // np.mean of the two middle values
}

// ---- kernel 1: per-read shift / scale (extract_features.py:179-185) --------------------------------------------
__global__ __launch_bounds__(256) void dsp_ext_normalize_kernel(dsp_read_batch b, int method, double* shift_out,
                                                                double* scale_out) {
    __shared__ SelectLds sel;
    __shared__ double sums[256];
    const int64_t r = blockIdx.x;
    ReadView rd = {b.raw + b.raw_off[r], b.raw_off[r + 1] - b.raw_off[r], b.scaling[r], b.offset[r]};
    const int64_t n = rd.n;
    double shift, scale;
    if (n <= 0) {
        shift = 0.0; scale = 0.0;
    } else if (method == 0) {  // mad
        if (rd.scaling > 0.0) {
            // pA is a non-decreasing function of the DAQ code: take the order statistics on the 16-bit codes
            auto kraw = [&](int64_t i) { return (uint64_t)(uint16_t)(rd.raw[i] ^ (int16_t)0x8000); };
            auto vraw = [&](uint64_t u) { return rd.scaling * ((double)(int16_t)((uint16_t)u ^ 0x8000u) + rd.offset); };
            shift = np_median(kraw, vraw, n, 8, &sel);
        } else {
            auto kx = [&](int64_t i) { return ord_bits(rd.pa(i)); };
            auto vx = [&](uint64_t u) { return ord_value(u); };
            shift = np_median(kx, vx, n, 56, &sel);
        }
        // median(|x - med| / c): x / c is monotone in x, so select on |x - med| and divide the selected values
        auto ke = [&](int64_t i) { return ord_bits(fabs(rd.pa(i) - shift)); };
        auto ve = [&](uint64_t u) { return ord_value(u) / kMadC; };
        scale = np_median(ke, ve, n, 56, &sel);
    } else {  // zscore: np.mean, np.std
        auto fx = [&](int64_t i) { return rd.pa(i); };
        shift = np_sum_block(fx, n, sums) / (double)n;
        auto fd = [&](int64_t i) { const double d = rd.pa(i) - shift; return d * d; };
        scale = sqrt(np_sum_block(fd, n, sums) / (double)n);
    }
    if (threadIdx.x == 0) {
        shift_out[r] = shift;
        scale_out[r] = scale;
    }
}

// ---- kernel 2: per-base statistics (extract_features.py:331-335, :363-365) -------------------------------------
__device__ __forceinline__ int64_t read_of_event(const int64_t* ev_off, int64_t n_reads, int64_t e) {
    int64_t lo = 0, hi = n_reads;  // last r with ev_off[r] <= e
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (ev_off[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void dsp_ext_base_stats_kernel(dsp_read_batch b, const double* shift, const double* scale,
                                                                 double* base_mean, double* base_std, int32_t* base_len,
                                                                 int64_t* base_lo) {
    // 8 adjacent lanes per base (32 bases per 256-thread workgroup)
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t e = t >> 3;
    const int l8 = (int)(t & 7);
    if (e >= b.n_events) return;
    const int64_t r = read_of_event(b.ev_off, b.n_reads, e);
    const int64_t n_raw = b.raw_off[r + 1] - b.raw_off[r];
    // norm_signals[start:start+length] with Python's clamping of the slice ends
    int64_t lo = b.ev_start[e], hi = lo + b.ev_len[e];
    lo = lo < 0 ? 0 : (lo > n_raw ? n_raw : lo);
    hi = hi < lo ? lo : (hi > n_raw ? n_raw : hi);
    const int64_t n = hi - lo;
    NormView nv = {{b.raw + b.raw_off[r], n_raw, b.scaling[r], b.offset[r]}, shift[r], scale[r]};
    const double mean = np_sum8(nv, lo, n, l8) / (double)n;  // n == 0 -> nan, like np.mean([])
    auto dev2 = [&](int64_t i) { const double d = nv(i) - mean; return d * d; };
    const double var = np_sum8(dev2, lo, n, l8) / (double)n;
    if (l8 == 0) {
        base_mean[e] = mean;
        base_std[e] = sqrt(var);
        base_len[e] = (int32_t)n;
        base_lo[e] = lo;
    }
}

// ---- kernel 3: k-mer window gather (extract_features.py:360-368, :232-251) -------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

// base2code_dna (utils/process_utils.py:25-29); 255 = not in the alphabet (rejected on the host, dsp_extract_sites)
__device__ __forceinline__ uint8_t base_code(uint8_t c) {
    switch (c) {
        case 'A': return 0;  case 'C': return 1;  case 'G': return 2;  case 'T': return 3;
        case 'N': return 4;  case 'W': return 5;  case 'S': return 6;  case 'M': return 7;
        case 'K': return 8;  case 'R': return 9;  case 'Y': return 10; case 'B': return 11;
        case 'V': return 12; case 'D': return 13; case 'H': return 14; case 'Z': return 15;
        default: return 255;
    }
}

struct GatherArgs {
    dsp_read_batch b;
    const double *shift, *scale, *base_mean, *base_std;
    const int32_t* base_len;
    const int64_t* base_lo;
    int64_t n_sites;
    const int32_t *site_read, *site_loc;
    int L, S, round_stats;
    uint64_t seed;
    const uint64_t* read_uid;
    uint8_t* kmer;
    float *means, *stds;
    int32_t* lens;
    float* signals;
};

__global__ __launch_bounds__(256) void dsp_ext_gather_kernel(GatherArgs a) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n_sites * a.L) return;
    const int64_t site = t / a.L;
    const int j = (int)(t - site * a.L);
    const int64_t r = a.site_read[site];
    const int64_t bi = (int64_t)a.site_loc[site] - (a.L - 1) / 2 + j;  // base index within the read
    const int64_t e = a.b.ev_off[r] + bi;
    a.kmer[t] = base_code(a.b.ev_base[e]);
    double m = a.base_mean[e], sd = a.base_std[e];
    if (a.round_stats) {  // the TSV route rounds them when it prints the row (:389-390)
        m = rint(m * 1e6) / 1e6;
        sd = rint(sd * 1e6) / 1e6;
    }
    a.means[t] = (float)m;
    a.stds[t] = (float)sd;
    const int n = a.base_len[e];
    a.lens[t] = n;
    NormView nv = {{a.b.raw + a.b.raw_off[r], 0, a.b.scaling[r], a.b.offset[r]}, a.shift[r], a.scale[r]};
    const int64_t lo = a.base_lo[e];
    float* out = a.signals + t * a.S;
    const int S = a.S;
    if (n <= S) {  // centred zero padding, left = pad // 2
        const int left = (S - n) / 2;
        for (int i = 0; i < left; i++) out[i] = 0.f;
        for (int i = 0; i < n; i++) out[left + i] = (float)nv(lo + i);
        for (int i = left + n; i < S; i++) out[i] = 0.f;
    } else {  // S of n samples in time order: Floyd's subset sampling over a counter-based stream, O(S^2)
        const uint64_t h = mix64((a.seed ^ (a.read_uid[r] * 0x9E3779B97F4A7C15ull)) + (uint64_t)bi * 0xD1B54A32D192ED03ull);
        int* sel = reinterpret_cast<int*>(out);  // the sorted sample indices live in the output slots first
        for (int q = 0; q < S; q++) {
            const int jj = n - S + q;
            const uint64_t rnd = mix64(h + (uint64_t)q) >> 32;
            int v = (int)((rnd * (uint64_t)(jj + 1)) >> 32);  // uniform on [0, jj]
            for (int k = 0; k < q; k++)
                if (sel[k] == v) { v = jj; break; }
            int k = q;
            for (; k > 0 && sel[k - 1] > v; k--) sel[k] = sel[k - 1];
            sel[k] = v;
        }
        for (int k = 0; k < S; k++) out[k] = (float)nv(lo + sel[k]);
    }
}

}  // namespace

extern "C" void dsp_set_error_(const char* msg);

static int32_t ext_fail(int32_t code, const char* msg) {
    dsp_set_error_(msg);
    return code;
}

static int32_t ext_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[256];
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        return ext_fail(DSP_EHIP, buf);
    }
    return DSP_OK;
}

extern "C" {

int32_t dsp_extract_normalize(void* stream, const dsp_read_batch* b, int32_t method, double* shift, double* scale) {
    if (!b || !shift || !scale || (method != DSP_NORM_MAD && method != DSP_NORM_ZSCORE))
        return ext_fail(DSP_EINVAL, "dsp_extract_normalize: bad arguments");
    if (b->n_reads == 0) return DSP_OK;
    hipLaunchKernelGGL(dsp_ext_normalize_kernel, dim3((unsigned)b->n_reads), dim3(256), 0, (hipStream_t)stream, *b,
                       (int)method, shift, scale);
    return ext_check_launch("dsp_extract_normalize");
}

int32_t dsp_extract_base_stats(void* stream, const dsp_read_batch* b, const double* shift, const double* scale,
                               double* base_mean, double* base_std, int32_t* base_len, int64_t* base_lo) {
    if (!b || !shift || !scale || !base_mean || !base_std || !base_len || !base_lo)
        return ext_fail(DSP_EINVAL, "dsp_extract_base_stats: bad arguments");
    if (b->n_events == 0) return DSP_OK;
    const unsigned grid = (unsigned)((b->n_events * 8 + 255) / 256);
    hipLaunchKernelGGL(dsp_ext_base_stats_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *b, shift, scale,
                       base_mean, base_std, base_len, base_lo);
    return ext_check_launch("dsp_extract_base_stats");
}

int32_t dsp_extract_gather(void* stream, const dsp_read_batch* b, const double* shift, const double* scale,
                           const double* base_mean, const double* base_std, const int32_t* base_len,
                           const int64_t* base_lo, int64_t n_sites, const int32_t* site_read, const int32_t* site_loc,
                           int32_t seq_len, int32_t signal_len, int32_t round_stats, uint64_t seed,
                           const uint64_t* read_uid, uint8_t* kmer, float* means, float* stds, int32_t* lens,
                           float* signals) {
    if (!b || n_sites < 0 || seq_len <= 0 || !(seq_len & 1) || signal_len <= 0)
        return ext_fail(DSP_EINVAL, "dsp_extract_gather: bad arguments (seq_len must be odd)");
    if (n_sites == 0) return DSP_OK;
    if (!shift || !scale || !base_mean || !base_std || !base_len || !base_lo || !site_read || !site_loc || !read_uid ||
        !kmer || !means || !stds || !lens || !signals)
        return ext_fail(DSP_EINVAL, "dsp_extract_gather: NULL array");
    GatherArgs a = {*b, shift, scale, base_mean, base_std, base_len, base_lo, n_sites, site_read, site_loc,
                    (int)seq_len, (int)signal_len, (int)round_stats, seed, read_uid, kmer, means, stds, lens, signals};
    const int64_t threads = n_sites * seq_len;
    hipLaunchKernelGGL(dsp_ext_gather_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a);
    return ext_check_launch("dsp_extract_gather");
}

}  // extern "C"
