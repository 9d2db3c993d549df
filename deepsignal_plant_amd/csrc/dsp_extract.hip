// Per-read feature extraction on gfx950: raw DAQ samples + resquiggle events -> the tensors dsp_forward consumes
// (SURVEY.md 8(f) next-3).  Replaces the arithmetic of deepsignal_plant/extract_features.py:
//   _rescale_signals :273-274, _normalize_signals :179-190, the event slicing and per-base np.mean / np.std of
//   _extract_features :331-335, :363-368, and _get_signals_rect :232-251.
// Everything is float64 and reproduces numpy's evaluation order bit for bit (this file is compiled with
// -ffp-contract=off):
//   * np.add.reduce over a contiguous float64 array = 0.0 + pairwise sums of 8192-element buffer chunks, each chunk
//     summed by numpy's pairwise scheme (blocks of <= 128 with 8 accumulators, halves rounded down to a multiple
//     of 8 above that);
//   * np.median = k-th order statistic (mean of the two middle ones for even n): read off code histograms in two
//     streamed passes when pA is monotone in the DAQ code (dsp_ext_mad_kernel), else found with an 8-pass radix
//     select over the order-preserving 64-bit image of the doubles -- no sort, no scratch arrays;
//   * np.around(x, 6) = rint(x * 1e6) / 1e6.
// HBM-bound integer/byte work, no MFMA: one 1024-thread workgroup per read for the read statistics; 8 adjacent
// lanes per base for the base statistics (lane j = accumulator r[j] of numpy's unrolled loop; stalls of more than
// 128 samples by the whole workgroup); 4 output samples per lane (one 16-byte store) for the window gather, the bases
// longer than signal_len compacted into full waves for the subset draw.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dsp_amd.h"
#include "dsp_kernels.h"

// (HIP's dynamic-LDS declaration as a macro: the test-suite's SIMT interpreter, tests/native/emu, gives it its own meaning when it
// runs these kernels on the host under AddressSanitizer; the product build sees exactly the declaration it names)
#ifndef DSP_EMU
#define DSP_DYN_LDS_T(type, name) extern __shared__ type name[]
// (a wave runs in lockstep: what one lane stored to LDS an instruction ago is there for the next lane's load.  The interpreter runs
// a wave's lanes one after another between cross-lane instructions and needs to be TOLD where the kernel relies on that.)
#define DSP_WAVE_LOCKSTEP() ((void)0)
// (... and a piece of LDS that the lanes of a group all write with the SAME values in lockstep -- read-then-write in one
// instruction each -- is given to every lane as its own copy there)
#define DSP_LOCKSTEP_SHARED(type, ptr) ((void)0)
#endif

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
    __device__ __forceinline__ double of_code(int16_t code) const {  // the same for a DAQ code already in a register
        double v = rd.scaling * ((double)code + rd.offset);
        if (scale != 0.0) v = (v - shift) / scale;
        return rint(v * 1e6) / 1e6;
    }
};

// ---- numpy pairwise summation -------------------------------------------------------------------------------
// The same sums by a group of 8 adjacent lanes (lane l8 owns accumulator r[l8] of numpy's 8-way unrolled loop);
// all 8 lanes follow the same control flow and end up with the same value.  IEEE addition is commutative, so
// the xor-butterfly reproduces ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) on every lane.
template <class F>
__device__ double pw_block8(const F& f, int64_t lo, int n, int l8) {  // n <= 128
    if (n < 8) {  // r = 0; r += a[0]; r += a[1]; ...  -- one evaluation per lane, the adds replayed on every lane
        const double v = l8 < n ? f(lo + l8) : 0.0;
        double r = 0.0;
        for (int i = 0; i < n; i++) r += __shfl(v, i, 8);
        return r;
    }
    double r = f(lo + l8);
    const int lim = n - (n % 8);
    for (int i = 8; i < lim; i += 8) r += f(lo + i + l8);
    r = r + __shfl_xor(r, 1);
    r = r + __shfl_xor(r, 2);
    r = r + __shfl_xor(r, 4);
    const int tail = n - lim;
    const double v = l8 < tail ? f(lo + lim + l8) : 0.0;
    for (int i = 0; i < tail; i++) r += __shfl(v, i, 8);
    return r;
}

// walk stack of one 8-lane group (n <= 8192 -> at most 7 levels); lives in LDS to keep the kernels' VGPR count low
struct PwStack {
    int lo[8], n[8], phase[8];
    double left[8];
};

template <class F>
__device__ double pw_chunk8(const F& f, int64_t lo, int n, int l8, PwStack* st) {  // n <= 8192
    if (n <= 128) return pw_block8(f, lo, n, l8);
    // post-order walk of numpy's recursion; all 8 lanes execute it identically (same-value LDS writes are benign)
    DSP_LOCKSTEP_SHARED(PwStack, st);
    int sp = 0;
    st->lo[0] = 0; st->n[0] = n; st->phase[0] = 0;
    double ret = 0.0;
    bool have = false;
    while (true) {
        if (!have) {
            const int cn = st->n[sp], clo = st->lo[sp];
            if (cn <= 128) {
                ret = pw_block8(f, lo + clo, cn, l8);
                have = true;
                sp--;
            } else {
                int n2 = cn / 2;
                n2 -= n2 % 8;
                st->phase[sp] = 1;
                st->lo[sp + 1] = clo; st->n[sp + 1] = n2; st->phase[sp + 1] = 0;
                sp++;
            }
        } else {
            if (sp < 0) return ret;
            const int cn = st->n[sp], clo = st->lo[sp];
            if (st->phase[sp] == 1) {
                int n2 = cn / 2;
                n2 -= n2 % 8;
                st->left[sp] = ret;
                st->phase[sp] = 2;
                st->lo[sp + 1] = clo + n2; st->n[sp + 1] = cn - n2; st->phase[sp + 1] = 0;
                sp++;
                have = false;
            } else {
                ret = st->left[sp] + ret;
                sp--;
            }
        }
    }
}

template <class F>
__device__ double np_sum8(const F& f, int64_t lo, int64_t n, int l8, PwStack* st) {
    double total = 0.0;
    for (int64_t c = 0; c < n; c += kChunk) {
        const int m = (int)((n - c) < kChunk ? (n - c) : kChunk);
        total = total + pw_chunk8(f, lo + c, m, l8, st);
    }
    return total;
}

// np.add.reduce of f(0 .. n) by a whole workgroup: one 8192-element chunk per group of 8 lanes, chunk sums
// accumulated in order
template <class F>
__device__ double np_sum_block(const F& f, int64_t n, double* lds /*[blockDim.x / 8]*/, PwStack* stacks) {
    double total = 0.0;
    const int groups = blockDim.x >> 3, grp = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    const int64_t nchunks = (n + kChunk - 1) / kChunk;
    for (int64_t c0 = 0; c0 < nchunks; c0 += groups) {
        const int64_t c = c0 + grp;
        __syncthreads();
        if (c < nchunks) {
            const int64_t lo = c * kChunk;
            const double v = pw_chunk8(f, lo, (int)((n - lo) < kChunk ? (n - lo) : kChunk), l8, &stacks[grp]);
            if (l8 == 0) lds[grp] = v;
        }
        __syncthreads();
        const int m = (int)((nchunks - c0) < (int64_t)groups ? (nchunks - c0) : (int64_t)groups);
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
    uint32_t wsum[4];
    uint32_t sel[2];
    int64_t k[2];
};

// After a pass: the bucket holding the k-th candidate (and the (k+1)-th when nk == 2) and its rank inside the
// bucket, by a 256-wide prefix scan (waves 0..3).  Leaves the answer in s->sel / s->k for every thread.
__device__ void select_pick(SelectLds* s, int64_t k, int nk) {
    const int lane = threadIdx.x & 63;
    uint32_t h = 0, x = 0;
    if (threadIdx.x < 256) {
        h = s->hist[threadIdx.x];
        x = h;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d);
            if (lane >= d) x += y;
        }
        if (lane == 63) s->wsum[threadIdx.x >> 6] = x;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        uint32_t base = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); w++) base += s->wsum[w];
        const int64_t incl = (int64_t)base + x, excl = incl - h;
        for (int q = 0; q < nk; q++)
            if (excl <= k + q && k + q < incl) {
                s->sel[q] = threadIdx.x;
                s->k[q] = k + q - excl;
            }
    }
    __syncthreads();
}

// histogram one candidate: samples of one read share their high bytes, so a unanimous wave costs one atomic
__device__ __forceinline__ void select_count(bool match, uint32_t d, int lane, uint32_t* hist) {
    const uint64_t m = __ballot(match);
    if (m == 0) return;
    const int first = __ffsll((long long)m) - 1;
    const uint32_t d0 = (uint32_t)__shfl((int)d, first);
    const uint64_t same = __ballot(match && d == d0);
    if (same == m) {
        if (lane == first) atomicAdd(&hist[d0], (uint32_t)__popcll(m));
    } else if (match) {
        atomicAdd(&hist[d], 1u);
    }
}

// One radix pass over key(raw[0..n)) restricted to the candidates whose bits above `shift + 8` equal `prefix`:
// fills s->hist with the histogram of the next byte.  All threads of the workgroup take part; the body of the
// read is streamed as 16-byte vectors (8 samples per lane and load).
template <class K>
__device__ void select_pass(const int16_t* raw, int64_t n, const K& key, uint64_t prefix, int shift, int top_shift,
                            SelectLds* s) {
    const int lane = threadIdx.x & 63;
    const uint64_t mask = shift == top_shift ? 0ull : (~0ull << (shift + 8));
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s->hist[i] = 0;
    __syncthreads();
    int64_t head = (int64_t)((16 - ((uintptr_t)raw & 15)) & 15) >> 1;  // samples before the first 16 B boundary
    if (head > n) head = n;
    const int64_t nvec = (n - head) >> 3;
    const int64_t tail0 = head + (nvec << 3);
    {   // head + tail: fewer than 16 samples, one per thread
        const int64_t t = threadIdx.x;
        const int64_t i = t < head ? t : tail0 + (t - head);
        if (threadIdx.x < 64) {  // wave 0, converged
            bool match = false;
            uint32_t d = 0;
            if (t < head + (n - tail0)) {
                const uint64_t u = key(raw[i]);
                match = (u & mask) == prefix;
                d = (uint32_t)(u >> shift) & 255u;
            }
            select_count(match, d, lane, s->hist);
        }
    }
    const int4* vec = reinterpret_cast<const int4*>(raw + head);
    const int64_t nround = (nvec + blockDim.x - 1) / blockDim.x * blockDim.x;  // whole waves stay converged
    for (int64_t g = threadIdx.x; g < nround; g += blockDim.x) {
        const bool live = g < nvec;
        int4 v = make_int4(0, 0, 0, 0);
        if (live) v = vec[g];
        const int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int16_t code = (int16_t)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffff));
            const uint64_t u = key(code);
            select_count(live && (u & mask) == prefix, (uint32_t)(u >> shift) & 255u, lane, s->hist);
        }
    }
    __syncthreads();
}

// k-th smallest key (0-based) among the candidates of `prefix`, continuing at byte `shift`
template <class K>
__device__ uint64_t select_from(const int16_t* raw, int64_t n, const K& key, uint64_t prefix, int64_t k, int shift, int top_shift,
                               SelectLds* s) {
    for (; shift >= 0; shift -= 8) {
        select_pass(raw, n, key, prefix, shift, top_shift, s);
        select_pick(s, k, 1);
        prefix |= (uint64_t)s->sel[0] << shift;
        k = s->k[0];
    }
    return prefix;
}

// the k-th and (k+1)-th smallest keys: one walk while both fall into the same bucket, two after they part
template <class K>
__device__ void select_pair(const int16_t* raw, int64_t n, const K& key, int64_t k, int top_shift, SelectLds* s, uint64_t* lo,
                            uint64_t* hi) {
    uint64_t prefix = 0;
    for (int shift = top_shift; shift >= 0; shift -= 8) {
        select_pass(raw, n, key, prefix, shift, top_shift, s);
        select_pick(s, k, 2);
        const uint32_t b0 = s->sel[0], b1 = s->sel[1];
        const int64_t k0 = s->k[0], k1 = s->k[1];
        if (b0 != b1) {
            const uint64_t p0 = prefix | ((uint64_t)b0 << shift), p1 = prefix | ((uint64_t)b1 << shift);
            *lo = select_from(raw, n, key, p0, k0, shift - 8, top_shift, s);
            *hi = select_from(raw, n, key, p1, k1, shift - 8, top_shift, s);
            return;
        }
        prefix |= (uint64_t)b0 << shift;
        k = k0;
    }
    *lo = *hi = prefix;  // equal keys
}

// np.median of value(key(raw[i])) over n >= 1 samples; top_shift = 56 for doubles through ord_bits, 8 for the
// 16-bit DAQ codes themselves
template <class K, class V>
__device__ double np_median(const int16_t* raw, int64_t n, const K& key, const V& value, int top_shift, SelectLds* s) {
    if ((n & 1) || n < 2) return value(select_from(raw, n, key, 0, n / 2, top_shift, top_shift, s));
    uint64_t lo, hi;
    select_pair(raw, n, key, n / 2 - 1, top_shift, s, &lo, &hi);
    return ((0.0 + value(lo)) + value(hi)) / 2.0;  // np.mean of the two middle values
}

// ---- kernel 1: per-read shift / scale (extract_features.py:179-185); one workgroup per read ----------------------
//
// Fast MAD path (scaling > 0, i.e. pA non-decreasing in the DAQ code): two streamed passes instead of ~12.
//   pass 1  histogram of the codes' high bytes -> where the median lies;
//   pass 2  fine histogram of the 32,768 codes around it (LDS), codes outside only counted;
//   median  = order statistics read off the prefix sums of the fine histogram;
//   MAD     = order statistics of dev(v) = |x(v) - med| read off the same histogram by LEVELS: level a holds the
//             codes m1 - a and m2 + a (m1 <= m2 the median codes).  dev is non-decreasing in a on either side, and
//             consecutive levels are one code step (= scaling, guarded to be >> rounding) apart, so the k-th
//             smallest deviation sits in the level where the cumulative count passes k; the at most two distinct
//             codes of that level are ordered by their float deviations.
// Anything the window cannot decide (codes outside it that the ranks reach, a degenerate scaling) falls back to
// the generic radix select below.
constexpr int kWin = 32768;

struct MadLds {
    uint32_t wave_sum[16];
    uint32_t below, above;
    int found_bin[2];
    int ok;
};

// exclusive prefix of `val` over the 1024 threads of the workgroup (thread order)
__device__ uint32_t block_excl_scan(uint32_t val, MadLds* m, uint32_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t x = val;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    __syncthreads();
    if (lane == 63) m->wave_sum[w] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int i = 0; i < 16; i++) {
        const uint32_t v = m->wave_sum[i];
        if (i < w) base += v;
        tot += v;
    }
    *total = tot;
    return base + x - val;
}

__device__ bool mad_by_histogram(const int16_t* raw, int64_t n, double scaling, double offset, uint32_t* fine, SelectLds* sel,
                                 MadLds* m, double* shift_out, double* scale_out) {
    const int64_t k1 = (n - 1) / 2, k2 = n / 2;
    auto kraw = [](int16_t c) { return (uint64_t)(uint16_t)(c ^ (int16_t)0x8000); };
    // pass 1: high byte of the lower median
    select_pass(raw, n, kraw, 0, 8, 8, sel);
    select_pick(sel, k1, 1);
    const int hb = (int)sel->sel[0];
    int ub0 = hb * 256 + 128 - kWin / 2;  // window [ub0, ub0 + kWin) in biased-code space 0..65535
    ub0 = ub0 < 0 ? 0 : (ub0 > 65536 - kWin ? 65536 - kWin : ub0);
    // pass 2: fine histogram
    for (int i = threadIdx.x; i < kWin; i += blockDim.x) fine[i] = 0;
    if (threadIdx.x == 0) { m->below = 0; m->above = 0; m->ok = 1; }
    __syncthreads();
    {
        const int lane = threadIdx.x & 63;
        uint32_t lo_cnt = 0, hi_cnt = 0;
        auto take = [&](int16_t c) {
            const int d = (int)(uint16_t)(c ^ (int16_t)0x8000) - ub0;
            if (d < 0) lo_cnt++;
            else if (d >= kWin) hi_cnt++;
            else atomicAdd(&fine[d], 1u);
        };
        int64_t head = (int64_t)((16 - ((uintptr_t)raw & 15)) & 15) >> 1;
        if (head > n) head = n;
        const int64_t nvec = (n - head) >> 3;
        const int64_t tail0 = head + (nvec << 3);
        if ((int64_t)threadIdx.x < head) take(raw[threadIdx.x]);
        else if ((int64_t)threadIdx.x - head < n - tail0) take(raw[tail0 + threadIdx.x - head]);
        const int4* vec = reinterpret_cast<const int4*>(raw + head);
        for (int64_t g = threadIdx.x; g < nvec; g += blockDim.x) {
            const int4 v = vec[g];
            const int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 8; e++) take((int16_t)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffff)));
        }
        for (int d = 32; d >= 1; d >>= 1) { lo_cnt += __shfl_xor(lo_cnt, d); hi_cnt += __shfl_xor(hi_cnt, d); }
        if (lane == 0) { if (lo_cnt) atomicAdd(&m->below, lo_cnt); if (hi_cnt) atomicAdd(&m->above, hi_cnt); }
    }
    __syncthreads();
    const uint32_t below = m->below, above = m->above;
    // median codes: thread t owns bins [32 t, 32 t + 32)
    const int t32 = threadIdx.x * 32;
    uint32_t mine = 0;
    for (int i = 0; i < 32; i++) mine += fine[t32 + i];
    uint32_t in_window;
    const uint32_t off = block_excl_scan(mine, m, &in_window) + below;
    for (int q = 0; q < 2; q++) {
        const int64_t k = q ? k2 : k1;
        if (k < (int64_t)below || k >= (int64_t)below + in_window) { if (threadIdx.x == 0) m->ok = 0; }
        else if ((int64_t)off <= k && k < (int64_t)off + mine) {
            int64_t c = off;
            for (int i = 0; i < 32; i++) {
                c += fine[t32 + i];
                if (k < c) { m->found_bin[q] = t32 + i; break; }
            }
        }
    }
    __syncthreads();
    if (!m->ok) return false;
    const int m1 = m->found_bin[0], m2 = m->found_bin[1];
    auto x_of = [&](int bin) { return scaling * ((double)(int16_t)((uint16_t)(ub0 + bin) ^ 0x8000u) + offset); };
    const double shift = (n & 1) ? x_of(m1) : ((0.0 + x_of(m1)) + x_of(m2)) / 2.0;
    // level spacing (one code step) must dwarf the rounding of the deviations
    if (!(scaling > 1e-9 * (fabs(x_of(m1)) + fabs(x_of(m2)) + fabs(shift) + 1.0))) return false;
    // levels: thread t owns levels [32 t, 32 t + 32)
    auto cnt_l = [&](int a) { const int v = m1 - a; return v >= 0 ? fine[v] : 0u; };
    auto cnt_r = [&](int a) { const int v = m2 + a; return v < kWin ? fine[v] : 0u; };
    auto level = [&](int a) { return (a == 0 && m1 == m2) ? fine[m1] : cnt_l(a) + cnt_r(a); };
    uint32_t lmine = 0;
    for (int i = 0; i < 32; i++) lmine += level(t32 + i);
    uint32_t covered;
    const uint32_t loff = block_excl_scan(lmine, m, &covered);
    __syncthreads();
    if (threadIdx.x == 0) { m->found_bin[0] = -1; m->found_bin[1] = -1; }
    __syncthreads();
    for (int q = 0; q < 2; q++) {
        const int64_t k = q ? k2 : k1;
        if ((int64_t)loff <= k && k < (int64_t)loff + lmine) {
            int64_t c = loff;
            for (int i = 0; i < 32; i++) {
                const int64_t c0 = c;
                c += level(t32 + i);
                if (k < c) { m->found_bin[q] = t32 + i; sel->k[q] = k - c0; break; }
            }
        }
    }
    __syncthreads();
    double e[2];
    for (int q = 0; q < 2; q++) {
        const int a = m->found_bin[q];
        if (a < 0) return false;  // the rank reaches codes outside the window
        // a level is decided only if neither side of it (nor of the levels before) hides samples outside the window
        if ((below && m1 - a < 0) || (above && m2 + a >= kWin)) return false;
        const int64_t r = sel->k[q];
        double dev;
        if (a == 0 && m1 == m2) {
            dev = fabs(x_of(m1) - shift);
        } else {
            const uint32_t cl = cnt_l(a), cr = cnt_r(a);
            const double dl = cl ? fabs(x_of(m1 - a) - shift) : 0.0, dr = cr ? fabs(x_of(m2 + a) - shift) : 0.0;
            if (!cl) dev = dr;
            else if (!cr) dev = dl;
            else if (dl <= dr) dev = r < (int64_t)cl ? dl : dr;
            else dev = r < (int64_t)cr ? dr : dl;
        }
        e[q] = dev / kMadC;
    }
    *shift_out = shift;
    *scale_out = (n & 1) ? e[0] : ((0.0 + e[0]) + e[1]) / 2.0;
    return true;
}

__global__ __launch_bounds__(1024) void dsp_ext_mad_kernel(dsp_read_batch b, double* shift_out, double* scale_out) {
    DSP_DYN_LDS_T(uint32_t, fine_lds);  // kWin bins
    __shared__ SelectLds sel;
    __shared__ MadLds mad;
    const int64_t r = blockIdx.x;
    const int16_t* raw = b.raw + b.raw_off[r];
    const int64_t n = b.raw_off[r + 1] - b.raw_off[r];
    const double scaling = b.scaling[r], offset = b.offset[r];
    double shift = 0.0, scale = 0.0;
    if (n > 0) {
        const bool fast = scaling > 0.0 && n < (1ll << 31) &&
                          mad_by_histogram(raw, n, scaling, offset, fine_lds, &sel, &mad, &shift, &scale);
        if (!fast) {
            if (scaling > 0.0) {
                // pA is a non-decreasing function of the DAQ code: take the order statistics on the 16-bit codes
                auto kraw = [](int16_t c) { return (uint64_t)(uint16_t)(c ^ (int16_t)0x8000); };
                auto vraw = [&](uint64_t u) { return scaling * ((double)(int16_t)((uint16_t)u ^ 0x8000u) + offset); };
                shift = np_median(raw, n, kraw, vraw, 8, &sel);
            } else {
                auto kx = [&](int16_t c) { return ord_bits(scaling * ((double)c + offset)); };
                auto vx = [](uint64_t u) { return ord_value(u); };
                shift = np_median(raw, n, kx, vx, 56, &sel);
            }
            // median(|x - med| / c): x / c is monotone in x, so select on |x - med| and divide the selected values
            auto ke = [&](int16_t c) { return ord_bits(fabs(scaling * ((double)c + offset) - shift)); };
            auto ve = [](uint64_t u) { return ord_value(u) / kMadC; };
            scale = np_median(raw, n, ke, ve, 56, &sel);
        }
    }
    if (threadIdx.x == 0) {
        shift_out[r] = shift;
        scale_out[r] = scale;
    }
}

__global__ __launch_bounds__(256) void dsp_ext_zscore_kernel(dsp_read_batch b, double* shift_out, double* scale_out) {
    __shared__ double sums[32];
    __shared__ PwStack stacks[32];
    const int64_t r = blockIdx.x;
    ReadView rd = {b.raw + b.raw_off[r], b.raw_off[r + 1] - b.raw_off[r], b.scaling[r], b.offset[r]};
    const int64_t n = rd.n;
    double shift = 0.0, scale = 0.0;
    if (n > 0) {  // np.mean, np.std
        auto fx = [&](int64_t i) { return rd.pa(i); };
        shift = np_sum_block(fx, n, sums, stacks) / (double)n;
        auto fd = [&](int64_t i) { const double d = rd.pa(i) - shift; return d * d; };
        scale = sqrt(np_sum_block(fd, n, sums, stacks) / (double)n);
    }
    if (threadIdx.x == 0) {
        shift_out[r] = shift;
        scale_out[r] = scale;
    }
}

// ---- kernel 2: per-base statistics (extract_features.py:331-335, :363-365) -------------------------------------
// Workgroups own 256 consecutive bases of ONE read (8 rounds of 32 bases x 8 lanes), so everything per read is
// wave-uniform (scalar loads) and the read lookup is paid once per 256 bases; blk_off[r] = first workgroup of
// read r, built by the scan kernel below.
constexpr int kBasesPerBlock = 256;

__global__ __launch_bounds__(1024) void dsp_ext_blkoff_kernel(const int64_t* ev_off, int64_t n_reads, int64_t* blk_off) {
    __shared__ int64_t part[1024];
    const int64_t per = (n_reads + blockDim.x - 1) / blockDim.x;
    const int64_t r0 = (int64_t)threadIdx.x * per, r1 = (r0 + per < n_reads) ? r0 + per : n_reads;
    int64_t sum = 0;
    for (int64_t r = r0; r < r1; r++) sum += (ev_off[r + 1] - ev_off[r] + kBasesPerBlock - 1) / kBasesPerBlock;
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t run = 0;
        for (int i = 0; i < (int)blockDim.x; i++) { const int64_t v = part[i]; part[i] = run; run += v; }
        blk_off[n_reads] = run;
    }
    __syncthreads();
    int64_t run = part[threadIdx.x];
    for (int64_t r = r0; r < r1; r++) {
        blk_off[r] = run;
        run += (ev_off[r + 1] - ev_off[r] + kBasesPerBlock - 1) / kBasesPerBlock;
    }
}

// numpy's recursion over one chunk (n <= 8192), walked by ONE thread: first to list the leaf blocks (<= 128 samples
// each, in order), later to add up their sums in numpy's association
struct LeafList {
    int lo[128], n[128];
    double sum[128];
    int count;
    double result;
};

struct WalkScratch {  // stacks of the single-thread walks; in LDS so that they cost the kernel no registers
    int lo[16], n[16], phase[16];
    double left[16];
};

__device__ void chunk_leaves(int n, LeafList* L, WalkScratch* w) {
    int sp = 0, cnt = 0;
    w->lo[0] = 0; w->n[0] = n;
    while (sp >= 0) {  // pre-order, left child first = leaves in sample order
        const int clo = w->lo[sp], cn = w->n[sp];
        sp--;
        if (cn <= 128) {
            L->lo[cnt] = clo; L->n[cnt] = cn; cnt++;
        } else {
            int n2 = cn / 2;
            n2 -= n2 % 8;
            w->lo[sp + 1] = clo + n2; w->n[sp + 1] = cn - n2;  // right below left on the stack
            w->lo[sp + 2] = clo;      w->n[sp + 2] = n2;
            sp += 2;
        }
    }
    L->count = cnt;
}

__device__ double chunk_combine(int n, const LeafList* L, WalkScratch* w) {  // sum(n) = n <= 128 ? leaf : sum(left) + sum(right)
    int next = 0;
    if (n <= 128) return L->sum[next];
    int sp = 0;
    w->n[0] = n; w->phase[0] = 0;
    double ret = 0.0;
    bool have = false;
    while (true) {
        if (!have) {
            const int cn = w->n[sp];
            if (cn <= 128) { ret = L->sum[next++]; have = true; sp--; }
            else { int n2 = cn / 2; n2 -= n2 % 8; w->phase[sp] = 1; w->n[sp + 1] = n2; w->phase[sp + 1] = 0; sp++; }
        } else {
            if (sp < 0) return ret;
            const int cn = w->n[sp];
            if (w->phase[sp] == 1) {
                int n2 = cn / 2; n2 -= n2 % 8;
                w->left[sp] = ret; w->phase[sp] = 2;
                w->n[sp + 1] = cn - n2; w->phase[sp + 1] = 0; sp++;
                have = false;
            } else { ret = w->left[sp] + ret; sp--; }
        }
    }
}

// np.add.reduce of f(lo .. lo+n) by a whole 256-thread workgroup: the leaf blocks of every chunk are summed by the
// 32 eight-lane groups in parallel and combined by thread 0; every thread returns the same value.
template <class F>
__device__ double np_sum_wg(const F& f, int64_t lo, int64_t n, LeafList* L, WalkScratch* w) {
    const int grp = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    double total = 0.0;
    for (int64_t c = 0; c < n; c += kChunk) {
        const int m = (int)((n - c) < kChunk ? (n - c) : kChunk);
        __syncthreads();
        if (threadIdx.x == 0) chunk_leaves(m, L, w);
        __syncthreads();
        for (int j = grp; j < L->count; j += 32) {
            const double v = pw_block8(f, lo + c + L->lo[j], L->n[j], l8);
            if (l8 == 0) L->sum[j] = v;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            total = total + chunk_combine(m, L, w);
            L->result = total;
        }
    }
    __syncthreads();
    total = L->result;
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(256) void dsp_ext_base_stats_kernel(dsp_read_batch b, const double* shift, const double* scale,
                                                                 const int64_t* blk_off, double* base_mean,
                                                                 double* base_std, int32_t* base_len, int64_t* base_lo) {
    __shared__ PwStack stacks[32];
    __shared__ double cache[32][32];
    __shared__ LeafList leaves;
    __shared__ WalkScratch walk;
    __shared__ int n_long;
    __shared__ int long_e[kBasesPerBlock];  // bases of more than one numpy block, left to the whole workgroup
    const int64_t blk = blockIdx.x;
    if (blk >= blk_off[b.n_reads]) return;
    int64_t rlo = 0, rhi = b.n_reads;  // last r with blk_off[r] <= blk (uniform: scalar loads)
    while (rhi - rlo > 1) {
        const int64_t mid = (rlo + rhi) >> 1;
        if (blk_off[mid] <= blk) rlo = mid; else rhi = mid;
    }
    const int64_t r = rlo;
    const int grp = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    const int64_t e0 = b.ev_off[r], n_ev = b.ev_off[r + 1] - e0;
    const int64_t n_raw = b.raw_off[r + 1] - b.raw_off[r];
    const NormView nv = {{b.raw + b.raw_off[r], n_raw, b.scaling[r], b.offset[r]}, shift[r], scale[r]};
    if (threadIdx.x == 0) n_long = 0;
    __syncthreads();
    for (int round = 0; round < kBasesPerBlock / 32; round++) {
        const int64_t e_local = (blk - blk_off[r]) * kBasesPerBlock + round * 32 + grp;
        if (e_local >= n_ev) break;  // whole 8-lane groups leave together
        const int64_t e = e0 + e_local;
        // norm_signals[start:start+length] with Python's clamping of the slice ends
        int64_t lo = b.ev_start[e], hi = lo + b.ev_len[e];
        lo = lo < 0 ? 0 : (lo > n_raw ? n_raw : lo);
        hi = hi < lo ? lo : (hi > n_raw ? n_raw : hi);
        const int64_t n = hi - lo;
        if (l8 == 0) {
            base_len[e] = (int32_t)n;
            base_lo[e] = lo;
        }
        if (n > 128) {  // a stall: 1 % of the bases, a third of the samples -- it would hold this wave's other groups up
            if (l8 == 0) long_e[atomicAdd(&n_long, 1)] = (int)e_local;
            continue;
        }
        double mean, var;
        if (n <= 32) {
            // a short base (one numpy block): every sample is normalised once (two f64 divisions) by the lane that owns it in
            // both passes, and parked in LDS between the mean and the variance pass
            double* c = cache[grp];
            auto first = [&](int64_t i) { const double v = nv(i); c[i - lo] = v; return v; };
            mean = np_sum8(first, lo, n, l8, &stacks[grp]) / (double)n;  // n == 0 -> nan, like np.mean([])
            auto dev2 = [&](int64_t i) { const double d = c[i - lo] - mean; return d * d; };
            var = np_sum8(dev2, lo, n, l8, &stacks[grp]) / (double)n;
        } else {
            mean = np_sum8(nv, lo, n, l8, &stacks[grp]) / (double)n;
            auto dev2 = [&](int64_t i) { const double d = nv(i) - mean; return d * d; };
            var = np_sum8(dev2, lo, n, l8, &stacks[grp]) / (double)n;
        }
        if (l8 == 0) {
            base_mean[e] = mean;
            base_std[e] = sqrt(var);
        }
    }
    __syncthreads();
    const int nl = n_long;
    for (int k = 0; k < nl; k++) {  // uniform: the whole workgroup takes the long bases one by one
        const int64_t e = e0 + long_e[k];
        int64_t lo = b.ev_start[e], hi = lo + b.ev_len[e];
        lo = lo < 0 ? 0 : (lo > n_raw ? n_raw : lo);
        hi = hi < lo ? lo : (hi > n_raw ? n_raw : hi);
        const int64_t n = hi - lo;
        const double mean = np_sum_wg(nv, lo, n, &leaves, &walk) / (double)n;
        auto dev2 = [&](int64_t i) { const double d = nv(i) - mean; return d * d; };
        const double var = np_sum_wg(dev2, lo, n, &leaves, &walk) / (double)n;
        if (threadIdx.x == 0) {
            base_mean[e] = mean;
            base_std[e] = sqrt(var);
        }
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

template <typename T>
struct GatherArgsT {
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
    T *means, *stds;
    int32_t* lens;
    T* signals;
};
typedef GatherArgsT<float> GatherArgs;

// One thread per OUTPUT SAMPLE (element of signals[n][L][S]): stores are contiguous across the wave.  The thread
// of sample 0 of a (site, base) pair is its leader: it writes the pair's k-mer code / mean / std / length and, for
// a base longer than S, draws the sorted subset into LDS for the pair's other threads.
template <typename T>
__global__ __launch_bounds__(256) void dsp_ext_gather_kernel(GatherArgsT<T> a) {
    DSP_DYN_LDS_T(int, sel_lds);  // [pairs touched by this workgroup][S]
    const int S = a.S;
    const int64_t total = a.n_sites * a.L * (int64_t)S;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x;
    const int64_t t = t0 + threadIdx.x;
    const int64_t pair0 = t0 / S;  // first (site, base) pair this workgroup touches
    const bool live = t < total;
    const int64_t pair = live ? t / S : 0;
    const int sidx = (int)(t - pair * S);
    const int64_t site = pair / a.L;
    const int j = (int)(pair - site * a.L);
    int64_t r = 0, bi = 0, e = 0, lo = 0;
    int n = 0;
    if (live) {
        r = a.site_read[site];
        bi = (int64_t)a.site_loc[site] - (a.L - 1) / 2 + j;  // base index within the read
        e = a.b.ev_off[r] + bi;
        n = a.base_len[e];
        lo = a.base_lo[e];
    }
    // leaders: sample 0 of a pair, or the first thread of the workgroup when its pair started in the previous one
    const bool leader = live && (sidx == 0 || threadIdx.x == 0);
    int* sel = sel_lds + (pair - pair0) * S;
    if (leader) {
        if (sidx == 0) {
            a.kmer[pair] = base_code(a.b.ev_base[e]);
            double m = a.base_mean[e], sd = a.base_std[e];
            if (a.round_stats) {  // the TSV route rounds them when it prints the row (:389-390)
                m = rint(m * 1e6) / 1e6;
                sd = rint(sd * 1e6) / 1e6;
            }
            a.means[pair] = (T)m;
            a.stds[pair] = (T)sd;
            a.lens[pair] = n;
        }
        if (n > S) {  // S of n samples in time order: Floyd's subset sampling over a counter-based stream, O(S^2)
            const uint64_t h = mix64((a.seed ^ (a.read_uid[r] * 0x9E3779B97F4A7C15ull)) + (uint64_t)bi * 0xD1B54A32D192ED03ull);
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
        }
    }
    __syncthreads();
    if (!live) return;
    NormView nv = {{a.b.raw + a.b.raw_off[r], 0, a.b.scaling[r], a.b.offset[r]}, a.shift[r], a.scale[r]};
    T out = (T)0;
    if (n <= S) {  // centred zero padding, left = pad // 2
        const int left = (S - n) / 2;
        if (sidx >= left && sidx < left + n) out = (T)nv(lo + (sidx - left));
    } else {
        out = (T)nv(lo + sel[sidx]);
    }
    a.signals[t] = out;
}

constexpr int kSelWords = 16;  // membership bitset of the subset sampler: bases of up to 512 samples
constexpr int kLongMax = kSelWords * 32;
constexpr int kSampleMaxS = 64;  // widest signal_len the sampling pass serves (LDS index columns)

// Fast path for signal_len = 4, 8, ..., 256 (a power of two): S/4 adjacent lanes per (site, base) pair, each lane
// producing four consecutive samples and storing them as one 16-byte vector.
__global__ __launch_bounds__(256) void dsp_ext_gather4_kernel(GatherArgs a, int log2_lpp) {
    DSP_DYN_LDS_T(int, sel_lds);  // [pairs per workgroup][S]
    const int S = a.S;
    const int lpp = 1 << log2_lpp;                 // lanes per pair = S / 4
    const int ppb = 256 >> log2_lpp;               // pairs per workgroup
    const int pl = threadIdx.x >> log2_lpp;        // pair within the workgroup
    const int sub = threadIdx.x & (lpp - 1);       // which four samples of the pair
    const int64_t pair = (int64_t)blockIdx.x * ppb + pl;
    const bool live = pair < a.n_sites * a.L;
    const uint32_t site = live ? (uint32_t)(pair / a.L) : 0;
    const int j = (int)(pair - (int64_t)site * a.L);
    int64_t r = 0, bi = 0, e = 0, lo = 0;
    int n = 0;
    if (live) {
        r = a.site_read[site];
        bi = (int64_t)a.site_loc[site] - (a.L - 1) / 2 + j;
        e = a.b.ev_off[r] + bi;
        n = a.base_len[e];
        lo = a.base_lo[e];
    }
    const int long_max = S <= kSampleMaxS ? kLongMax : 0;  // 0: no sampling pass, every long base is drawn here
    int* sel = sel_lds + pl * S;  // sorted sample indices of this pair (only for bases beyond the sampling pass)
    if (live && sub == 0) {
        a.kmer[pair] = base_code(a.b.ev_base[e]);
        double m = a.base_mean[e], sd = a.base_std[e];
        if (a.round_stats) {
            m = rint(m * 1e6) / 1e6;
            sd = rint(sd * 1e6) / 1e6;
        }
        a.means[pair] = (float)m;
        a.stds[pair] = (float)sd;
        a.lens[pair] = n;
        if (n > long_max) {  // a stall of more than 512 samples (or signal_len > 64): Floyd's subset sampling with a sorted list, O(S^2)
            const uint64_t h = mix64((a.seed ^ (a.read_uid[r] * 0x9E3779B97F4A7C15ull)) + (uint64_t)bi * 0xD1B54A32D192ED03ull);
            for (int q = 0; q < S; q++) {
                const int jj = n - S + q;
                const uint64_t rnd = mix64(h + (uint64_t)q) >> 32;
                int v = (int)((rnd * (uint64_t)(jj + 1)) >> 32);
                for (int k = 0; k < q; k++)
                    if (sel[k] == v) { v = jj; break; }
                int k = q;
                for (; k > 0 && sel[k - 1] > v; k--) sel[k] = sel[k - 1];
                sel[k] = v;
            }
        }
    }
    __syncthreads();
    if (!live) return;
    NormView nv = {{a.b.raw + a.b.raw_off[r], 0, a.b.scaling[r], a.b.offset[r]}, a.shift[r], a.scale[r]};
    if (n > S && n <= long_max) return;  // S < n <= 512: written by dsp_ext_sample_kernel, all lanes busy
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    const int s0 = sub * 4;
    if (n <= S) {
        const int left = (S - n) / 2;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int k = s0 + i - left;
            if (k >= 0 && k < n) o[i] = (float)nv(lo + k);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = (float)nv(lo + sel[s0 + i]);
    }
    reinterpret_cast<float4*>(a.signals)[pair * lpp + sub] = make_float4(o[0], o[1], o[2], o[3]);
}

// Subsampling pass of the fast path: 15 % of the (site, base) pairs are longer than S but nearly every wave of the
// gather holds one, so drawing them there leaves 3 of 4 (or 63 of 64) lanes idle for the whole serial draw.  Here a
// wave scans 64 pairs at a time (their lengths were just written to `lens`), queues the long ones in LDS and, as
// soon as 64 are waiting, draws them with every lane busy: Floyd's subset sampling with a per-lane membership
// bitset in LDS, whose set bits in ascending order are the sorted sample.
constexpr int kPairsPerWave = 512;

__device__ void sample_one(const GatherArgs& a, int64_t pair, uint32_t* bits /* [kSelWords][64], this lane's column */,
                           int* idx /* [kSampleMaxS][64], this lane's column */) {
    const int S = a.S;
    const int64_t site = pair / a.L;
    const int j = (int)(pair - site * a.L);
    const int64_t r = a.site_read[site];
    const int64_t bi = (int64_t)a.site_loc[site] - (a.L - 1) / 2 + j;
    const int64_t e = a.b.ev_off[r] + bi;
    const int n = a.base_len[e];
    const int64_t lo = a.base_lo[e];
    const int nw = (n + 31) >> 5;
    for (int w = 0; w < nw; w++) bits[w * 64] = 0;
    const uint64_t h = mix64((a.seed ^ (a.read_uid[r] * 0x9E3779B97F4A7C15ull)) + (uint64_t)bi * 0xD1B54A32D192ED03ull);
    for (int q = 0; q < S; q++) {
        const int jj = n - S + q;
        const uint64_t rnd = mix64(h + (uint64_t)q) >> 32;
        int v = (int)((rnd * (uint64_t)(jj + 1)) >> 32);  // uniform on [0, jj]
        if (bits[(v >> 5) * 64] & (1u << (v & 31))) v = jj;
        bits[(v >> 5) * 64] |= 1u << (v & 31);
    }
    // set bits in ascending order = the sorted sample; the indices go through LDS so that the 4 sample loads of
    // an output vector are independent (a pop-load-store loop would serialise S scattered loads)
    int c = 0;
    for (int w = 0; w < nw; w++) {
        uint32_t x = bits[w * 64];
        while (x) {
            idx[(c++) * 64] = (w << 5) + __ffs((int)x) - 1;
            x &= x - 1;
        }
    }
    const NormView nv = {{a.b.raw + a.b.raw_off[r], 0, a.b.scaling[r], a.b.offset[r]}, a.shift[r], a.scale[r]};
    float4* out = reinterpret_cast<float4*>(a.signals + pair * S);
    for (int k = 0; k < S; k += 4) {
        const int i0 = idx[k * 64], i1 = idx[(k + 1) * 64], i2 = idx[(k + 2) * 64], i3 = idx[(k + 3) * 64];
        const int16_t r0 = nv.rd.raw[lo + i0], r1 = nv.rd.raw[lo + i1], r2 = nv.rd.raw[lo + i2], r3 = nv.rd.raw[lo + i3];
        out[k >> 2] = make_float4((float)nv.of_code(r0), (float)nv.of_code(r1), (float)nv.of_code(r2), (float)nv.of_code(r3));
    }
}

__global__ __launch_bounds__(256) void dsp_ext_sample_kernel(GatherArgs a) {
    __shared__ uint32_t bits_lds[4][kSelWords * 64];
    __shared__ int64_t queue_lds[4][128];
    DSP_DYN_LDS_T(int, idx_lds);  // [4 waves][S][64]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t* bits = bits_lds[w] + lane;
    int* idx = idx_lds + (size_t)w * a.S * 64 + lane;
    int64_t* queue = queue_lds[w];
    const int64_t total = a.n_sites * a.L;
    const int64_t p0 = ((int64_t)blockIdx.x * 4 + w) * kPairsPerWave;
    int qn = 0;  // wave-uniform
    int len[kPairsPerWave / 64];  // all length loads of this wave's pairs in flight at once
#pragma unroll
    for (int c = 0; c < kPairsPerWave / 64; c++) {
        const int64_t pair = p0 + c * 64 + lane;
        len[c] = pair < total ? a.lens[pair] : 0;
    }
#pragma unroll
    for (int c = 0; c < kPairsPerWave / 64; c++) {
        const int64_t pair = p0 + c * 64 + lane;
        const bool is_long = len[c] > a.S && len[c] <= kLongMax;
        const uint64_t m = __ballot(is_long);
        if (is_long) queue[qn + __popcll(m & ((1ull << lane) - 1))] = pair;
        qn += __popcll(m);
        DSP_WAVE_LOCKSTEP();   // every lane's queue entry is in LDS before any lane takes one
        if (qn >= 64) {
            sample_one(a, queue[lane], bits, idx);
            qn -= 64;
            DSP_WAVE_LOCKSTEP();
            if (lane < qn) {
                const int64_t keep = queue[64 + lane];
                queue[lane] = keep;
            }
            DSP_WAVE_LOCKSTEP();
        }
    }
    if (lane < qn) sample_one(a, queue[lane], bits, idx);
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
    if (method == DSP_NORM_MAD)
    {
        static bool attr_set = false;  // 128 KB of dynamic LDS needs the opt-in (set once; idempotent if raced)
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)dsp_ext_mad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kWin * 4) != hipSuccess)
                return ext_fail(DSP_EHIP, "dsp_extract_normalize: cannot reserve LDS for the MAD kernel");
            attr_set = true;
        }
        hipLaunchKernelGGL(dsp_ext_mad_kernel, dim3((unsigned)b->n_reads), dim3(1024), (size_t)kWin * 4, (hipStream_t)stream, *b,
                           shift, scale);
    }
    else
        hipLaunchKernelGGL(dsp_ext_zscore_kernel, dim3((unsigned)b->n_reads), dim3(256), 0, (hipStream_t)stream, *b, shift, scale);
    return ext_check_launch("dsp_extract_normalize");
}

int32_t dsp_extract_base_stats(void* stream, const dsp_read_batch* b, const double* shift, const double* scale,
                               int64_t* blk_off, double* base_mean, double* base_std, int32_t* base_len,
                               int64_t* base_lo) {
    if (!b || !shift || !scale || !blk_off || !base_mean || !base_std || !base_len || !base_lo)
        return ext_fail(DSP_EINVAL, "dsp_extract_base_stats: bad arguments");
    if (b->n_events == 0 || b->n_reads == 0) return DSP_OK;
    hipLaunchKernelGGL(dsp_ext_blkoff_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, b->ev_off, b->n_reads, blk_off);
    // sum over reads of ceil(bases / 256) <= n_events / 256 + n_reads; surplus workgroups exit at once
    const unsigned grid = (unsigned)(b->n_events / kBasesPerBlock + b->n_reads);
    hipLaunchKernelGGL(dsp_ext_base_stats_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *b, shift, scale,
                       (const int64_t*)blk_off, base_mean, base_std, base_len, base_lo);
    return ext_check_launch("dsp_extract_base_stats");
}

int32_t dsp_extract_gather(void* stream, const dsp_read_batch* b, const double* shift, const double* scale,
                           const double* base_mean, const double* base_std, const int32_t* base_len,
                           const int64_t* base_lo, int64_t n_sites, const int32_t* site_read, const int32_t* site_loc,
                           int32_t seq_len, int32_t signal_len, int32_t round_stats, uint64_t seed,
                           const uint64_t* read_uid, uint8_t* kmer, float* means, float* stds, int32_t* lens,
                           float* signals) {
    if (!b || n_sites < 0 || seq_len <= 0 || !(seq_len & 1) || signal_len <= 0 || signal_len > 4096)
        return ext_fail(DSP_EINVAL, "dsp_extract_gather: bad arguments (seq_len must be odd)");
    if (n_sites == 0) return DSP_OK;
    if (!shift || !scale || !base_mean || !base_std || !base_len || !base_lo || !site_read || !site_loc || !read_uid ||
        !kmer || !means || !stds || !lens || !signals)
        return ext_fail(DSP_EINVAL, "dsp_extract_gather: NULL array");
    GatherArgs a = {*b, shift, scale, base_mean, base_std, base_len, base_lo, n_sites, site_read, site_loc,
                    (int)seq_len, (int)signal_len, (int)round_stats, seed, read_uid, kmer, means, stds, lens, signals};
    if (signal_len >= 4 && signal_len <= 256 && (signal_len & (signal_len - 1)) == 0 && ((uintptr_t)signals & 15) == 0) {
        int log2_lpp = 0;
        while ((4 << log2_lpp) < signal_len) log2_lpp++;
        const int ppb = 256 >> log2_lpp;
        const int64_t pairs = n_sites * seq_len;
        hipLaunchKernelGGL(dsp_ext_gather4_kernel, dim3((unsigned)((pairs + ppb - 1) / ppb)), dim3(256),
                           (size_t)ppb * signal_len * sizeof(int), (hipStream_t)stream, a, log2_lpp);
        const int64_t per_block = 4 * (int64_t)kPairsPerWave;
        if (signal_len <= kSampleMaxS)
            hipLaunchKernelGGL(dsp_ext_sample_kernel, dim3((unsigned)((pairs + per_block - 1) / per_block)), dim3(256),
                               (size_t)4 * signal_len * 64 * sizeof(int), (hipStream_t)stream, a);
        return ext_check_launch("dsp_extract_gather");
    }
    const int64_t threads = n_sites * seq_len * (int64_t)signal_len;
    const size_t lds = (size_t)(256 / signal_len + 2) * signal_len * sizeof(int);
    hipLaunchKernelGGL(dsp_ext_gather_kernel<float>, dim3((unsigned)((threads + 255) / 256)), dim3(256), lds,
                       (hipStream_t)stream, a);
    return ext_check_launch("dsp_extract_gather");
}

int32_t dsp_extract_gather_f64(void* stream, const dsp_read_batch* b, const double* shift, const double* scale,
                               const double* base_mean, const double* base_std, const int32_t* base_len,
                               const int64_t* base_lo, int64_t n_sites, const int32_t* site_read,
                               const int32_t* site_loc, int32_t seq_len, int32_t signal_len, int32_t round_stats,
                               uint64_t seed, const uint64_t* read_uid, uint8_t* kmer, double* means, double* stds,
                               int32_t* lens, double* signals) {
    if (!b || n_sites < 0 || seq_len <= 0 || !(seq_len & 1) || signal_len <= 0 || signal_len > 4096)
        return ext_fail(DSP_EINVAL, "dsp_extract_gather_f64: bad arguments (seq_len must be odd)");
    if (n_sites == 0) return DSP_OK;
    if (!shift || !scale || !base_mean || !base_std || !base_len || !base_lo || !site_read || !site_loc || !read_uid ||
        !kmer || !means || !stds || !lens || !signals)
        return ext_fail(DSP_EINVAL, "dsp_extract_gather_f64: NULL array");
    GatherArgsT<double> a = {*b, shift, scale, base_mean, base_std, base_len, base_lo, n_sites, site_read, site_loc,
                             (int)seq_len, (int)signal_len, (int)round_stats, seed, read_uid, kmer, means, stds, lens,
                             signals};
    const int64_t threads = n_sites * seq_len * (int64_t)signal_len;
    const size_t lds = (size_t)(256 / signal_len + 2) * signal_len * sizeof(int);
    hipLaunchKernelGGL(dsp_ext_gather_kernel<double>, dim3((unsigned)((threads + 255) / 256)), dim3(256), lds,
                       (hipStream_t)stream, a);
    return ext_check_launch("dsp_extract_gather_f64");
}

}  // extern "C"
