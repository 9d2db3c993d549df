// dsp_kernels.hip -- hand-written gfx950 (CDNA4) kernels for the deepsignal-plant call_mods forward.
//
// Reference arithmetic being implemented: ModelBiLSTM.forward, deepsignal_plant/models.py:178-240
// (torch.nn.LSTM cell semantics: gates i,f,g,o; both biases; c' = s(f)c + s(i)tanh(g); h' = s(o)tanh(c')).
//
// Formulation (see DESIGN.md "Data layout" / "Kernels"):
//   * Everything is computed TRANSPOSED:  gates^T[4H, sites] = W[4H, K] * act^T[K, sites], so the weight
//     matrix is the MFMA A operand (pre-packed on the host in fragment order, streamed from L2 with
//     coalesced 16-B loads) and a 32-site tile is the MFMA N dimension.
//   * Activations live in the "K4" layout  act[tile][t][F/4][32 sites][4 feats]  (fp32): one float4 per
//     lane is exactly 4 consecutive k-steps of the B operand of v_mfma_f32_32x32x2_f32, AND exactly what
//     one lane holds in 4 consecutive accumulator registers of the 32x32 C/D layout
//     (row = 8*(r/4) + 4*(lane/32) + r%4, col = lane%32).  So a layer's output registers are stored with
//     plain float4 stores and re-loaded by the next layer / next time step as B fragments with no
//     transposition, shuffle or bank conflict.
//   * fp32 MFMA (v_mfma_f32_32x32x2_f32): bf16/fp16 operands break the 1e-4 parity contract on sharp
//     models, so the binding roofline is the fp32 matrix peak (157.3 TFLOP/s).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <initializer_list>
#include <string>
#include <type_traits>
#include <utility>

#include "dsp_cluster_protocol.h"
#include "dsp_kernels.h"

// The five places where this file speaks gfx950 assembly or HIP's LDS declaration syntax, as macros -- so that the test-suite's
// SIMT interpreter (tests/native/emu: the kernels of this file compiled for the HOST and run lane by lane, workgroups
// concurrently, buffer descriptors with the hardware's range-check semantics; tests/test_kernel_emu.py; never the product) can
// give them its own meaning.  The product build sees exactly the statements these macros name.
#ifndef DSP_EMU
#define DSP_STORE_GUARD(v) asm volatile("s_nop 1" : : "v"(v))
#define DSP_KEEP_SGPR2(a, b) asm("" : "+s"(a), "+s"(b))
#define DSP_DRAIN_STORES() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define DSP_READ_XCC_ID(x) asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x))
#define DSP_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) float name[]
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// explicit address spaces keep hipcc from merging a global and an LDS load of the same variable into one
// flat_load (which counts on both vmcnt and lgkmcnt and forces s_waitcnt 0 on everything in flight)
typedef __attribute__((address_space(1))) const f32x4 gf32x4;  // global
typedef __attribute__((address_space(3))) const f32x4 lf32x4;  // LDS

// gfx950 store-data hazard (measured: tools/micro/h_exchange.hip, profiles/r3/micro_h_exchange.txt): a VMEM store of
// more than 64 bits followed IMMEDIATELY by a VALU write of its data registers stores the new values for the last quad
// of every 16-lane row (lanes 12-15, 28-31, 44-47, 60-63), depending on what the SIMD's other wave is doing.  With an
// immediate soffset (the case the ISA manual documents) hipcc pads two wait states; with an SGPR soffset -- every store
// of these kernels -- ONE wait state is needed and hipcc (ROCm 7.2) pads none: it assumes the SGPR operand hides the
// hazard.  So every wide store is followed by `s_nop 1` (two wait states) in an asm statement that takes the data
// registers as an input, i.e. keeps them live until after the nop: nothing that writes them can be scheduled in between.  tools/check_store_hazard.py
// scans the assembly of every build for the pattern (the csrc Makefile fails without it).
__device__ __forceinline__ void store_data_guard(const f32x4& v) {
    DSP_STORE_GUARD(v);
}
__device__ __forceinline__ void gst16(f32x4* p, f32x4 v) {  // a plain 16-byte global store, same guard
    *p = v;
    store_data_guard(v);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// Buffer descriptors with REAL extents (round 6).  Until round 5 every descriptor was a 2 GiB window from
// wherever the workgroup's base landed (num_records 0x7ffffff0): the hardware range check was switched off, and an offset
// gone wrong would have faulted the process (HSA's memory-access-fault abort: no test name, no message from this library).
// Now num_records = the bytes left in the allocation behind the base (`end` travels in the argument blocks: the region of
// the handle's workspace the pointer lies in, or the weight upload): a load past it returns 0 and a store past it is
// dropped.  Costs one s_sub / s_min per descriptor in the prologue, nothing in the MFMA stream (the descriptor words are
// SGPRs either way).  A base at or past its end gives num_records 0: every access through it is out of range.
//   DSP_BOUNDS builds (make bounds -> libdsp_amd_bounds.so, never the product): every access through a descriptor is also
// compared with the extent in software, exactly as the address unit forms it -- voffset + soffset without wrapping at 32 bits
// -- and the first offender is recorded (source line, operand, workgroup, thread, offset, extent) for dsp_forward to return as
// DSP_EBOUNDS; the host passes the TIGHT logical extents of the call there (this call's tiles, this layer's weights).
constexpr long long kRsrcMax = 0x7ffffff0ll;   // (a workgroup never reaches 2 GiB past its base)
__device__ __forceinline__ uint32_t rsrc_records(const void* base, const void* end) {
    // (32-bit scalar compares on the two halves of end - base: a 64-bit signed compare is a VALU instruction on this target,
    // its result a lane mask in an SGPR pair -- enough extra scalar pressure to spill SGPRs in the widest kernels)
    const unsigned long long left = (unsigned long long)end - (unsigned long long)base;   // wraps above 2^63 when base > end
    uint32_t lo = (uint32_t)left, hi = (uint32_t)(left >> 32);
    DSP_KEEP_SGPR2(lo, hi);   // (the halves stay apart, in SGPRs: hipcc otherwise folds the tests back into 64-bit VALU compares)
    const uint32_t in32 = lo < (uint32_t)kRsrcMax ? lo : (uint32_t)kRsrcMax;
    return hi == 0u ? in32 : ((int32_t)hi < 0 ? 0u : (uint32_t)kRsrcMax);
}
#ifdef DSP_BOUNDS
struct rsrc_t { __amdgpu_buffer_rsrc_t r; uint32_t nrec; int kind; };
__device__ unsigned int g_bounds[8];
__device__ __forceinline__ void bounds_note(int line, int kind, unsigned long long off, uint32_t nrec) {
    if (atomicAdd(&g_bounds[0], 1u) == 0u) {
        g_bounds[1] = (unsigned)line; g_bounds[2] = (unsigned)kind; g_bounds[3] = blockIdx.x; g_bounds[4] = threadIdx.x;
        g_bounds[5] = (unsigned)off; g_bounds[6] = (unsigned)(off >> 32); g_bounds[7] = nrec;
    }
}
__device__ __forceinline__ void bounds_check(const rsrc_t& r, uint32_t voff, uint32_t soff, int line) {
    const unsigned long long off = (unsigned long long)voff + (unsigned long long)soff;   // as the address unit adds them
    if (off + 16ull > (unsigned long long)r.nrec) bounds_note(line, r.kind, off, r.nrec);
}
// a flat (pointer) access of `bytes` at p against the end of its allocation
__device__ __forceinline__ void bounds_flat(const void* p, size_t bytes, const void* base, const void* end, int kind, int line) {
    if ((const char*)p < (const char*)base || (const char*)p + bytes > (const char*)end)
        bounds_note(line, kind, (unsigned long long)((const char*)p - (const char*)base), rsrc_records(base, end));
}
#define BOUNDS_FLAT(p, bytes, base, end, kind) bounds_flat((p), (bytes), (base), (end), (kind), __LINE__)
#define RS(x) ((x).r)
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, const void* end, int kind) {
    const uint32_t n = rsrc_records(base, end);
    return rsrc_t{__builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)n, 0x00020000), n, kind};
}
#else
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define BOUNDS_FLAT(p, bytes, base, end, kind) do { } while (0)
#define RS(x) (x)
__device__ __forceinline__ void bounds_check(const rsrc_t&, uint32_t, uint32_t, int) {}
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, const void* end, int /*kind*/) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)rsrc_records(base, end), 0x00020000);
}
#endif
__device__ __forceinline__ f32x4 bld16_(const rsrc_t& r, uint32_t voff, uint32_t soff, int line) {
    bounds_check(r, voff, soff, line);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(RS(r), (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void bst16_(const rsrc_t& r, uint32_t voff, uint32_t soff, f32x4 v, int line) {
    bounds_check(r, voff, soff, line);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), RS(r), (int)voff, (int)soff, 0);
    store_data_guard(v);
}
#define bld16(r, voff, soff) bld16_((r), (voff), (soff), __LINE__)
#define bst16(r, voff, soff, v) bst16_((r), (voff), (soff), (v), __LINE__)

// ------------------------------------------------------------------------------------------------
// math helpers: v_exp_f32 / v_rcp_f32 based (about 1 ulp each); abs error of sigmoid/tanh ~1e-7
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = 2*sigmoid(2x) - 1 ; exp overflow -> rcp(inf)=0 -> -1 ; underflow -> 2*1-1 = 1
    return __builtin_fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)), -1.0f);
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11) + Box-Muller: the in-kernel stand-in for torch.randn in
// init_hidden (models.py:169-176).  Key = seed; counter = (site_lo, site_hi, stream, unit/4).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void sincos2pi(float u, float& sn, float& cs) {  // u in [0, 1)
    const float k = __builtin_rintf(u * 4.0f);              // quadrant 0..4
    const float t = 6.283185307179586f * (u - 0.25f * k);   // the subtraction is exact; |t| <= pi/4
    const float t2 = t * t;
    const float ps = t * __builtin_fmaf(t2, __builtin_fmaf(t2, __builtin_fmaf(t2, __builtin_fmaf(t2, 2.7557319e-6f, -1.9841270e-4f),
                                                                            8.3333333e-3f), -1.6666667e-1f), 1.0f);
    const float pc = __builtin_fmaf(t2, __builtin_fmaf(t2, __builtin_fmaf(t2, __builtin_fmaf(t2, __builtin_fmaf(t2, -2.7557319e-7f, 2.4801587e-5f),
                                                                                          -1.3888889e-3f), 4.1666667e-2f), -0.5f), 1.0f);
    const int q = (int)k & 3;   // angle = t + q * pi/2
    sn = q == 0 ? ps : (q == 1 ? pc : (q == 2 ? -ps : -pc));
    cs = q == 0 ? pc : (q == 1 ? -ps : (q == 2 ? -pc : ps));
}
__device__ __forceinline__ void bm_pair(uint32_t a, uint32_t b, float& z0, float& z1) {
    const float u1 = ((float)(a >> 9) + 0.5f) * (1.0f / 8388608.0f);
    const float u2 = ((float)(b >> 9) + 0.5f) * (1.0f / 8388608.0f);
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // sqrt(-2 ln u1), v_log_f32 = log2
    float sn, cs;
    sincos2pi(u2, sn, cs);
    z0 = r * cs; z1 = r * sn;
}

__device__ __forceinline__ f32x4 philox_normal4(uint64_t seed, uint64_t site, uint32_t stream, uint32_t group) {
    uint32_t c0 = (uint32_t)site, c1 = (uint32_t)(site >> 32), c2 = stream, c3 = group;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    // Box-Muller on (c0, c1) and (c2, c3).  The transcendental work runs once per workgroup and time step 0, but for
    // 16 such calls per lane it was 0.9 % of the forward with the OCML logf / sincosf (generic range reduction): here
    // ln(u) = ln2 * v_log_f32(u), v_sqrt_f32, and sin / cos of 2*pi*u by quadrant reduction + two short polynomials on
    // [-pi/4, pi/4] (errors < 1 ulp of the result: within 3e-7 of a double-precision libm evaluation of the same formula).
    float z0, z1, z2, z3;
    bm_pair(c0, c1, z0, z1);
    bm_pair(c2, c3, z2, z3);
    return f32x4{z0, z1, z2, z3};
}

// activations with the bias folded into the exp2 argument: bp = -log2e*b (sigmoid) / -2*log2e*b (tanh)
__device__ __forceinline__ float sigmoid_pre(float x, float bp) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(x, -1.4426950408889634f, bp)));
}
__device__ __forceinline__ float tanh_pre(float x, float bp) {
    return __builtin_fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(x, -2.8853900817779268f, bp))), -1.0f);
}

__device__ __forceinline__ float load_code(const void* p, int dt, size_t i) {
    switch (dt) {
        case 1: return (float)((const uint8_t*)p)[i];
        case 2: return (float)((const uint16_t*)p)[i];
        case 3: return (float)((const int32_t*)p)[i];
        default: return ((const float*)p)[i];
    }
}

// ------------------------------------------------------------------------------------------------
// pack_kernel: reference-layout feature rows -> K4 activations for the two front-end LSTMs.
//   seq features (models.py:182-195): [embed(kmer) (E) | mean | std | (len)] zero-padded to Fseq
//   signal features (models.py:206): signals[.., S] zero-padded to Fsig
// One thread per (tile, t, site).  HBM-bound, ~1 KB in / ~1.2 KB out per site: negligible.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dsp_pack_kernel(PackArgs a) {
    // the arrival counters of this forward's clustered LSTM launches start from zero (first launch of the forward: the
    // kernel boundary orders these stores before every later launch)
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < a.n_zero_words; i += 256) {
            BOUNDS_FLAT(&a.zero_words[i], 4, a.zero_words, a.zero_words_end, DSP_BND_FLAGS);
            a.zero_words[i] = 0u;
        }
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int sl = (int)(idx & 31);
    const long long rem = idx >> 5;
    const int t = (int)(rem % a.T);
    const long long tile = rem / a.T;
    if (tile >= a.NTp) return;
    const long long site = tile * 32 + sl;
    const bool live = site < a.n;
    const size_t r = (size_t)site * a.T + t;
    if (a.xseq) {
        float mean = 0.f, sd = 0.f, len = 0.f;
        long code = 0;
        if (live) {
            mean = a.means[r]; sd = a.stds[r];
            if (a.is_siglen) len = load_code(a.lens, a.ldt, r);
            if (a.is_base) code = (long)load_code(a.kmer, a.kdt, r);  // kmer.long(): truncation
            // a code the table does not hold is the caller's error (nn.Embedding raises IndexError, models.py:186; the CLI
            // checks its rows on the host): here the index is only kept inside the table
            code = code < 0 ? 0 : (code >= a.V ? a.V - 1 : code);
        }
        f32x4* dst = (f32x4*)a.xseq + ((size_t)(tile * a.T + t) * (a.Fseq >> 2)) * 32 + sl;
        const int E = a.is_base ? a.E : 0;
        for (int g = 0; g < (a.Fseq >> 2); ++g) {
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = 4 * g + i - a.xoff_seq;  // the features sit at the END of the padded block
                float x = 0.f;
                if (live && f >= 0) {
                    if (f < E) x = a.embed[(size_t)code * a.E + f];
                    else if (f == E) x = mean;
                    else if (f == E + 1) x = sd;
                    else if (f == E + 2 && a.is_siglen) x = len;
                }
                v[i] = x;
            }
            BOUNDS_FLAT(&dst[(size_t)g * 32], 16, a.xseq, a.xseq_end, DSP_BND_FLAT_OUT);
            gst16(&dst[(size_t)g * 32], v);
        }
    }
    if (a.xsig) {
        f32x4* dst = (f32x4*)a.xsig + ((size_t)(tile * a.T + t) * (a.Fsig >> 2)) * 32 + sl;
        const float* src = a.signals + r * a.S;
        for (int g = 0; g < (a.Fsig >> 2); ++g) {
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = 4 * g + i - a.xoff_sig;
                v[i] = (live && f >= 0 && f < a.S) ? src[f] : 0.f;
            }
            BOUNDS_FLAT(&dst[(size_t)g * 32], 16, a.xsig, a.xsig_end, DSP_BND_FLAT_OUT);
            gst16(&dst[(size_t)g * 32], v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// initial LSTM state of 4 consecutive hidden units of one site (init_hidden, models.py:169-176):
// zeros, explicit buffers in the reference layout (2*layers, n, H), or in-kernel Philox N(0,1)
// ------------------------------------------------------------------------------------------------
// (skey = the Philox counter key of the site: site_keys[site] when the caller names its sites, else site_offset + site)
__device__ __forceinline__ uint64_t philox_site_key(const unsigned long long* keys, uint64_t site_offset, long long n,
                                                    long long site) {
    return (keys && site < n) ? (uint64_t)keys[site] : site_offset + (uint64_t)site;
}
__device__ __forceinline__ f32x4 init_state4(const float* src, int mode, long long n, long long site, int dir, int H,
                                             int k4, uint64_t seed, uint64_t skey, uint32_t stream) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (site >= n || 4 * k4 >= H) return v;
    if (mode == 1) {
        const float* p = src + ((size_t)dir * (size_t)n + (size_t)site) * H + 4 * k4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * k4 + i < H) v[i] = p[i];
    } else if (mode == 2) {
        v = philox_normal4(seed, skey, stream, (uint32_t)k4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * k4 + i >= H) v[i] = 0.f;
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// dsp_lstm_kernel<SPARSE, NP, XL>: one direction of one LSTM layer, all T steps, for 64*SG sites per workgroup.
// blockIdx.x & 1 = direction: with the observed block -> XCD (b % 8) placement even XCDs run the forward and odd XCDs
// the backward direction, so each XCD's 4 MiB L2 holds one direction's weights.
// (History, all measured on MI355X: first kernel -- 8 waves, h in LDS, one-deep prefetch -- 76 % of the fp32 MFMA
// peak; ablation showed operand loads, not MFMA issue, cost ~25 %.  One wave per SIMD with register rings 90.4 %;
// two waves per SIMD 93.6 %; round 2 94.9 %; profiles/README.md.)
//   * TWO waves per SIMD, 256 registers each.  A wave owns one unit tile (32 hidden units x 4 gates) x two site tiles
//     (32 sites each): 8 accumulator tiles (128 registers), 32 MFMAs per k-group (8 k); a weight fragment feeds 8
//     MFMAs; a workgroup has UT waves per 64 sites (UT = Hp/32 unit tiles) and SG such site groups.
//     A <2 unit tiles, 1 site tile> tiling (a fragment feeds 4 MFMAs, twice the weight stream out of L2, but only UT/2
//     waves per 32 sites: two INDEPENDENT 4-wave workgroups per CU at hidden 256) was measured in round 2: +0.6 % on the
//     combined-stack launches, but with non-zero initial states the h0 read-back of the second wave of a SIMD came out
//     wrong for sites 12-15 / 28-31 of a tile, non-deterministically, in some builds and not in others.  ROOT CAUSE
//     (round 3, tools/micro/h_exchange.hip, profiles/r3/micro_h_exchange.txt): not the exchange -- a store-data hazard of
//     gfx950 that hipcc does not pad.  A buffer_store_dwordx4 with an SGPR soffset followed AT ONCE by a VALU write of
//     its data registers stores the new values for the last quad of every 16-lane row (exactly those lanes), depending on
//     the SIMD's other wave; one wait state cures it; the compiler pads only the immediate-soffset form.  In that build
//     the instruction after an h0 store happened to reuse its registers.  Every wide store of this file now goes through
//     bst16 / gst16 (store, s_nop 1, data registers held live across the nop) and the build fails if the pattern appears
//     in the assembly (tools/check_store_hazard.py, tests/test_store_hazard.py).  The shipped round-2 binaries held the
//     pattern twice (the NP = 2 kernels, zeros overwriting stored zeros: harmless) and 19 times at one wait state (safe).
//     Rebuilt in round 3 with guarded stores (dsp_lstm21_kernel below): bit-identical results, 0.25 % SLOWER than this
//     tiling on full batches in an alternating same-box A/B -- the +0.6 % of round 2 did not survive -- but HALF the
//     latency on batches that leave CUs idle, which is where it runs.
//     The exchange itself -- stores, ONE s_barrier without s_waitcnt vmcnt(0), loads by the other waves -- showed 0
//     mismatches in 3e11 checked values with guarded stores, for 4- and 8-wave workgroups, with and without vmcnt(0), and
//     also when the destination rows had been pulled into the CU's L1 before the stores (no stale lines).
//   * EVERY operand is a coalesced BUFFER load: a wave-uniform 128-bit descriptor (SGPRs) + a wave-uniform byte
//     offset (SGPR soffset) + lane*16 (the only address VGPR of the kernel): weights (A), x_t (B) and also h_{t-1}
//     (B), which is read back from the K4 output the workgroup itself stored one step earlier (same CU, visible
//     after the per-step workgroup barrier; L2-resident).  Compared with flat/global addressing this removes all
//     64-bit VALU address arithmetic from the MFMA stream (measured: ~5 % of the MFMA issue rate).
//   * Register rings: an A fragment (weights of one unit tile x gate, 4 VGPRs) is re-requested for k-group q+4
//     after the MFMAs of the NEXT fragment of group q (one fragment late, so the load never writes registers
//     that the MFMA issued just before it is still reading: a WAR interlock that otherwise stalls the in-order
//     issue); B fragments four k-groups deep.  The k-group count is padded to a multiple of 4 with zero weights
//     (host side), so the loop body is branch-free with exact vmcnt counts; the last groups of step t request the
//     first groups of step t+1 (weights and x_{t+1} do not depend on h_t) before the cell phase.
//   * The cell state c and the (pre-scaled) biases live in LDS (own-lane float4 slots, conflict-free), the
//     accumulators start from literal zero (first MFMA of a step takes C = 0) and the biases are folded into
//     the exp2 arguments of the activations.  The cell phase of a wave is 5.1 k cycles of VALU work (10 v_exp/v_rcp +
//     15 VALU per element, tools/micro/cell_phase.hip) and fp32 MFMAs do NOT overlap VALU work of either wave of a SIMD
//     on gfx950 (tools/micro/mfma_cell_overlap.hip: both = sum), so a step costs MFMA cycles + cell cycles whatever the
//     schedule: 393.2 k + 2 x 5.1 k for the combined stack, measured 403.9 k.
// SPARSE != 0 (front-end layers): the x part is padded from 8/16 to 32 features so that the first four k-groups never
// depend on h_t; k-groups that are pure zero padding keep their operand requests but issue no MFMAs.
//   SPARSE = 2, XL = 1..3 (the shipped front ends: hidden <= 256, features at the END of the 32-wide block): which of the
//     first four k-groups are padding is a template parameter.  A step opens with 4 - XL refill-only stages (the requests
//     for the first h-part k-groups go out right behind the barrier), runs the XL live x-part k-groups while those are
//     in flight and continues with the same branch-free stages as the dense kernel.  No spills, exact s_waitcnt counts.
//   SPARSE = 1 (any other padded shape, e.g. hidden > 256 or a signal window wider than 32): a wave-uniform test
//     around the MFMAs of every k-group.  Measured on the front ends before SPARSE = 2 existed: 2,490 cycles per
//     k-group instead of 2,110 (a branch around every fragment's MFMAs, s_waitcnt counts merged conservatively at every
//     join, two ring fragments spilled to scratch at the end of every step).
// The padding to four x-part k-groups keeps every cross-step request off rows that do not exist yet: the B ring
// requests four k-groups across a step boundary, i.e. before the cell phase of step t has stored h_t -- such a request
// would put h_{t-1}'s row (or garbage) into the ring's REGISTERS.  (Round 2 also worried about stale lines in the CU's
// vector L1; measured in round 3: rows read before they are stored come back updated after store + barrier, the L1 is
// not the issue.  A 16-feature padding with a re-request of the h-part slots after the barrier was measured in round 2:
// +0.6 % on the front-end launches, noise level; not kept.)
// ------------------------------------------------------------------------------------------------

template <int N> using ic = std::integral_constant<int, N>;

// Workgroup barrier of the h exchange through global memory.  hipcc's __syncthreads() is a workgroup-scope release /
// acquire; not being in threadgroup-split mode it emits NO s_waitcnt vmcnt(0) before s_barrier (checked in the ISA):
// the waves of a workgroup share one CU's vector memory pipeline and L1, which keeps a wave's buffer_store ahead of
// another wave's buffer_load issued after the barrier.  What the exchange relies on, and what guards it:
//   (1) that in-order property of the CU's vector memory path -- measured in isolation by tools/micro/h_exchange.hip
//       (0 mismatches in 3e11 values, with and without vmcnt(0), stale-L1 variant included); an explicit vmcnt(0) costs
//       2 % on the combined-stack launches and changes no result, so it is not inserted;
//   (2) intact store DATA -- the gfx950 store-data hazard described at bst16 above, guarded in the source and gated in
//       the build;
//   (3) tests that would see either break: contiguous 2,304-site windows at five depths of 300k-site batches with
//       N(0,1) states against the CPU restatement of the reference, bit-exact permutation equivariance at full batch under non-zero states
//       (tests/test_gpu_windows.py), five-fold determinism at 300 k sites (tests/test_gpu_parity.py).
__device__ __forceinline__ void barrier_after_global_stores() { __syncthreads(); }

// DSP_TRACE builds only (make trace -> libdsp_amd_trace.so, tools/trace_lstm.py): shader-clock stamps of wave 0 of
// every workgroup of ONE chosen LSTM launch, at the start / after the k-loop / after the cell phase of every step
#ifdef DSP_TRACE
#define DSP_TRACE_WGS 8192
__device__ unsigned long long g_trace[DSP_TRACE_WGS][16][8];
__device__ unsigned int g_trace_hw[DSP_TRACE_WGS][4];
#define TSTAMP(k) do { if ((a.flags & 256) && tid == ((a.flags >> 9) & 7) * 64 && blockIdx.x < DSP_TRACE_WGS && step < 16) g_trace[blockIdx.x][step][k] = __builtin_amdgcn_s_memtime(); } while (0)
#define TSTAMP_AT(s, k) do { const int step = (s); TSTAMP(k); } while (0)
extern "C" int dsp_k_trace_read(unsigned long long* t, unsigned int* hw) {
    hipError_t e = hipMemcpyFromSymbol(t, HIP_SYMBOL(g_trace), sizeof(g_trace));
    if (e == hipSuccess) e = hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_trace_hw), sizeof(g_trace_hw));
    return (int)e;
}
#else
#define TSTAMP(k) do { } while (0)
#define TSTAMP_AT(s, k) do { } while (0)
#endif

template <int SPARSE, int NP, int XL = 0>
__global__ __launch_bounds__(512, 2) void dsp_lstm_kernel(LstmArgs a) {
    // NP = passes over the unit tiles per time step: 1 for hidden sizes up to 256 (8 unit tiles, one per wave);
    // 0 = a.NP passes (hidden sizes above 256, 8 unit tiles per pass: a wave computes unit tile w in pass 0, w + 8 in pass 1,
    // ..., one barrier per step): the pass loop is not unrolled and the cell state -- 64 KiB per pass, beyond the LDS from
    // three passes on -- lives in a global scratch with own-lane slots (no exchange between waves, so no barrier for it).
    // (NP = 2, unrolled with the cell state in LDS, compiles too but is no longer instantiated: same speed, 21 spills.)
    constexpr bool CG = NP == 0;
    const int np = CG ? a.NP : NP;
    constexpr int NF = 4;                  // A fragments (gates) per k-group
    constexpr int DA = 4, DB = 4;          // ring depths in k-groups
    DSP_DYN_LDS(smem);
#ifdef DSP_TRACE
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
    const int nthr = blockDim.x;
    f32x4* c_lds = (f32x4*)smem;           // [NP][2 site tiles][4 groups][nthr] float4 (not with CG)
    f32x4* b_lds = c_lds + (CG ? 0 : NP * 8 * nthr);  // [unit tile][aa][gate][half] float4
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t voff = (uint32_t)lane * 16u;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int UTW = a.UT / np;             // waves per site group = unit tiles per pass
    const int ug = w % UTW, sg = w / UTW;
    const int dir = blockIdx.x & 1;
    const int grp = blockIdx.x >> 1;
    const int half = lane >> 5, ls = lane & 31;
    const int HQ = a.Hp >> 2;
    const int nqx = a.Ipad >> 3, nq = nqx + (a.Hp >> 3), NQ = a.NQ;
    const int T = a.T;
    const int F4 = a.Fout >> 2;
    const uint32_t xrow = (uint32_t)(a.Ipad >> 2) * 512u;  // bytes of one (tile, t) block of the input
    const uint32_t orow = (uint32_t)F4 * 512u;             // bytes of one (tile, t) block of the output
    const uint32_t pstride = (uint32_t)UTW * (uint32_t)NQ * 4096u;  // bytes between a wave's unit tiles of two passes
    const bool prio = (a.flags & 1) != 0;

    // first site tile of this wave; its two tiles are adjacent, so one descriptor per buffer serves both
    const long long gt0 = ((long long)grp * a.SG + sg) * 2;
    const rsrc_t rw = make_rsrc((const char*)(dir ? a.wpk1 : a.wpk0) + (size_t)ug * NQ * 4096, dir ? a.wpk1_end : a.wpk0_end, DSP_BND_W);
    const rsrc_t rx = make_rsrc((const char*)a.x + (size_t)gt0 * T * xrow, a.x_end, DSP_BND_X);
    const rsrc_t ro = make_rsrc((const char*)a.out + (size_t)gt0 * T * orow + (size_t)dir * HQ * 512, a.out_end, DSP_BND_OUT);
    const rsrc_t rh0 = make_rsrc((const char*)a.h0buf + (size_t)gt0 * orow + (size_t)dir * HQ * 512, a.h0buf_end, DSP_BND_H0);
    const f32x4* bias4 = (const f32x4*)(dir ? a.sbias1 : a.sbias0);
    // CG: this workgroup's slice of the cell-state scratch, [pass][2 site tiles][4 groups][nthr] float4
    const rsrc_t rc = make_rsrc((const char*)a.cbuf + (CG ? (size_t)blockIdx.x * (size_t)np * 8 * nthr * 16 : 0), a.cbuf_end, DSP_BND_C);
    auto c_off = [&](int p, int m, int aa) __attribute__((always_inline)) {
        return (uint32_t)((((p * 2 + m) * 4 + aa) * nthr + w * 64) * 16);
    };

    // ---- initial state: c0 -> LDS, h0 -> the K4 scratch that step 0 reads as "h_{-1}"; biases -> LDS
    for (int i = tid; i < a.Hp; i += nthr) {
        const int h = i & 1, g = (i >> 1) & 3, aa = (i >> 3) & 3, ut = i >> 5;
        b_lds[i] = bias4[g * HQ + ut * 8 + 2 * aa + h];
    }
    for (int p = 0; p < np; ++p) {   // (one pass, or a loop that is not unrolled)
        const int u = ug + p * UTW;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const long long site = (gt0 + m) * 32 + ls;
            const uint64_t skey = a.init_mode == 2 ? philox_site_key(a.site_keys, a.site_offset, a.n, site) : 0;
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                const int k4 = u * 8 + 2 * aa + half;
                f32x4 hv = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f};
                if (a.init_mode != 0) {
                    hv = init_state4(a.h0, a.init_mode, a.n, site, dir, a.H, k4, a.seed, skey,
                                     (uint32_t)(a.stream_base + dir * 2 + 0));
                    cv = init_state4(a.c0, a.init_mode, a.n, site, dir, a.H, k4, a.seed, skey,
                                     (uint32_t)(a.stream_base + dir * 2 + 1));
                }
                bst16(rh0, voff + aa * 1024u, (uint32_t)m * orow + (uint32_t)u * 4096u, hv);
                if constexpr (CG) bst16(rc, voff, c_off(p, m, aa), cv);
                else c_lds[((p * 2 + m) * 4 + aa) * nthr + tid] = cv;
            }
        }
    }
    barrier_after_global_stores();  // h0 stored before any wave reads it back

    // B-operand source of a step (all uniform): x_t from rx, h_{t-1} from the K4 output of the previous step
    // (ro) or, at step 0, from the h0 scratch (rh0).  Offsets are biased so that both parts are "base + q*1024".
    rsrc_t rhp = rh0;
    uint32_t xo[2], ho[2];
    auto set_bases = [&](int step) __attribute__((always_inline)) {
        const int t = dir ? (T - 1 - step) : step;
        const int tp = dir ? (t + 1) : (t - 1);
        rhp = step == 0 ? rh0 : ro;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            xo[m] = (uint32_t)(m * T + t) * xrow;
            ho[m] = (step == 0 ? (uint32_t)m * orow : (uint32_t)(m * T + tp) * orow) - (uint32_t)nqx * 1024u;
        }
    };

    f32x4 A[DA][NF], B[DB][2];
    f32x16 acc[4][2];
    auto loadB = [&](f32x4 (&Bs)[2], int q) __attribute__((always_inline)) {
        const int qc = q < nq ? q : nq - 1;  // padded k-groups have zero weights: any finite B will do
        const bool isx = qc < nqx;
        const rsrc_t r = isx ? rx : rhp;
#pragma unroll
        for (int m = 0; m < 2; ++m) Bs[m] = bld16(r, voff, (isx ? xo[m] : ho[m]) + (uint32_t)qc * 1024u);
    };
    // weights of (this pass's unit tile, gate g, k-group q); k-groups >= NQ belong to the NEXT pass (or the next step's
    // first pass): po_cur / po_next are the byte offsets of the two unit tiles
    uint32_t po_cur = 0, po_next = NP != 1 ? pstride : 0;
    auto ldA = [&](int g, int q) __attribute__((always_inline)) {
        const uint32_t so = q < NQ ? po_cur + (uint32_t)q * 4096u : po_next + (uint32_t)(q - NQ) * 4096u;
        return bld16(rw, voff + (uint32_t)g * 1024u, so);
    };
#define QW(x) ((x) < NQ ? (x) : (x) - NQ)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nqx_used = a.nqx_used, nqx_lo = a.nqx_lo;
    // one k-group (ring slots are compile-time: QS = q mod 4): the 8 MFMAs of gate fragment g, then the refill of
    // fragment g-1 for k-group q+4; fragment 3 of the PREVIOUS stage's slot is refilled after fragment 0.
    // first = true: the very first k-step of a time step accumulates onto literal zero.
    // mode (compile time) 0: a live k-group; 1: may be pure padding (wave-uniform test); 2: is padding (refills only)
    auto stage = [&](auto qs, int q, auto first, auto mode) __attribute__((always_inline)) {
        constexpr int QS = decltype(qs)::value;
        constexpr int sa = QS % DA, sp = (QS + DA - 1) % DA, sb = QS % DB;
        constexpr int MODE = decltype(mode)::value;
        const bool live = MODE != 1 || (q >= nqx_lo && q < nqx_used) || (q >= nqx && q < nq);  // wave-uniform
#pragma unroll
        for (int g = 0; g < NF; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 2 || (MODE == 1 && !live)) break;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    if (decltype(first)::value && i == 0)
                        acc[g][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[sa][g][i], B[sb][m][i], zero16, 0, 0, 0);
                    else
                        acc[g][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[sa][g][i], B[sb][m][i], acc[g][m], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep "MFMAs of a fragment, then one refill" in program order
            if (g == 0) A[sp][NF - 1] = ldA(NF - 1, q + DA - 1);
            else A[sa][g - 1] = ldA(g - 1, q + DA);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto stage4 = [&](int q, auto first, auto mode) __attribute__((always_inline)) {
        stage(ic<0>{}, q + 0, first, mode); loadB(B[0], QW(q + 0 + DB)); __builtin_amdgcn_sched_barrier(0);
        stage(ic<1>{}, q + 1, std::false_type{}, mode); loadB(B[1], QW(q + 1 + DB)); __builtin_amdgcn_sched_barrier(0);
        stage(ic<2>{}, q + 2, std::false_type{}, mode); loadB(B[2], QW(q + 2 + DB)); __builtin_amdgcn_sched_barrier(0);
        stage(ic<3>{}, q + 3, std::false_type{}, mode); loadB(B[3], QW(q + 3 + DB)); __builtin_amdgcn_sched_barrier(0);
    };
    // SPARSE = 2 (the front ends: 7 or 16 features in a 32-wide block, no tail padding): the XL live x-part k-groups are
    // the LAST of the first four, everything known at compile time.  A step then opens with 4 - XL refill-only stages
    // -- the requests for the first h-part k-groups go out right behind the barrier -- runs the live x-part MFMAs while
    // those are in flight, and continues with the branch-free dense stages.  (With a wave-uniform test in every stage,
    // SPARSE = 1, the k-loop ran at 2,490 cycles per k-group instead of 2,110: a branch around every fragment's MFMAs,
    // s_waitcnt counts merged conservatively at every join, two ring fragments spilled.)
    auto stage4_first = [&]() __attribute__((always_inline)) {
        if constexpr (SPARSE == 2) {
            constexpr int D = 4 - XL;  // dead stages
            stage(ic<0>{}, 0, std::integral_constant<bool, D == 0>{}, ic<(0 < D ? 2 : 0)>{}); loadB(B[0], QW(0 + DB)); __builtin_amdgcn_sched_barrier(0);
            stage(ic<1>{}, 1, std::integral_constant<bool, D == 1>{}, ic<(1 < D ? 2 : 0)>{}); loadB(B[1], QW(1 + DB)); __builtin_amdgcn_sched_barrier(0);
            stage(ic<2>{}, 2, std::integral_constant<bool, D == 2>{}, ic<(2 < D ? 2 : 0)>{}); loadB(B[2], QW(2 + DB)); __builtin_amdgcn_sched_barrier(0);
            stage(ic<3>{}, 3, std::integral_constant<bool, D == 3>{}, ic<0>{}); loadB(B[3], QW(3 + DB)); __builtin_amdgcn_sched_barrier(0);
        } else {
            stage4(0, std::true_type{}, ic<SPARSE>{});
        }
    };
    using rest_mode = ic<(SPARSE == 1 ? 1 : 0)>;

    set_bases(0);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
        for (int g = 0; g < NF; ++g) A[d][g] = ldA(g, d);
        loadB(B[d], d);
    }

#ifdef DSP_TRACE
    if ((a.flags & 256) && tid == ((a.flags >> 9) & 7) * 64 && blockIdx.x < DSP_TRACE_WGS) {
        g_trace_hw[blockIdx.x][0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        g_trace_hw[blockIdx.x][1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        g_trace_hw[blockIdx.x][2] = (unsigned int)t_entry;                         // kernel entry of this workgroup
        g_trace_hw[blockIdx.x][3] = (unsigned int)(t_entry >> 32);
    }
#endif
    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        TSTAMP(0);
        if (step > 0) barrier_after_global_stores();  // h_{t-1} of every wave stored before anyone reads it back
        TSTAMP(1);
        // issue priority by phase: a wave in its k-loop outranks its SIMD partner's cell phase, so that the partner's
        // VALU / transcendental stream takes the issue slots the MFMA stream leaves and not the other way round
        // (measured on the combined stack: +0.45 %; no setting of the two priorities changes the front ends)
        if (prio) __builtin_amdgcn_s_setprio(2);
        for (int p = 0; p < np; ++p) {
            const int u = ug + p * UTW;
            if (NP != 1) { po_cur = (uint32_t)p * pstride; po_next = p + 1 < np ? (uint32_t)(p + 1) * pstride : 0u; }
            stage4_first();
            TSTAMP(4);
            for (int q = 4; q < NQ - 4; q += 4) stage4(q, std::false_type{}, rest_mode{});
            TSTAMP(5);
            // B requests from here on belong to the next pass of this step (same rows) or to the next step
            if (p == np - 1) set_bases(step + 1 < T ? step + 1 : step);
            stage4(NQ - 4, std::false_type{}, rest_mode{});
            // (stage NQ-1 leaves the last fragment of its ring slot, for the next k-loop's k-group 3, to "the next
            // stage": request it here, ahead of the cell phase)
            A[DA - 1][NF - 1] = ldA(NF - 1, NQ + DA - 1);
            if (p == np - 1) TSTAMP(2);
            if (prio) __builtin_amdgcn_s_setprio(0);

            // LSTM cell.  b_lds holds the PRE-SCALED biases (-log2e*b for i,f,o; -2*log2e*b for g), so
            // sigmoid(x+b) = rcp(1 + exp2(fma(x, -log2e, b'))) costs no extra instruction for the bias.
            const f32x4* b_my = b_lds + (size_t)u * 32 + half;  // + aa*8 + gate*2
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                const f32x4 bi = b_my[aa * 8 + 0], bf = b_my[aa * 8 + 2], bg = b_my[aa * 8 + 4], bo = b_my[aa * 8 + 6];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f32x4 cv;
                    if constexpr (CG) cv = bld16(rc, voff, c_off(p, m, aa));
                    else cv = c_lds[((p * 2 + m) * 4 + aa) * nthr + tid];
                    f32x4 hv;
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
                    if constexpr (CG) bst16(rc, voff, c_off(p, m, aa), cv);
                    else c_lds[((p * 2 + m) * 4 + aa) * nthr + tid] = cv;
                    bst16(ro, voff + aa * 1024u, (uint32_t)(m * T + t) * orow + (uint32_t)u * 4096u, hv);
                }
#ifdef DSP_TRACE
                __builtin_amdgcn_sched_barrier(0);
                if (aa == 0) TSTAMP(6);
                if (aa == 2) TSTAMP(7);
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
        TSTAMP(3);
    }
#undef QW
}

// ------------------------------------------------------------------------------------------------
// dsp_lstm21_kernel (round 3; dense one-pass layers with an even unit-tile count): the other way to cut a step's
// 8 unit tiles x 2 site tiles among waves -- a wave owns TWO unit tiles x ONE site tile (still 8 accumulator tiles), a
// workgroup is UT/2 waves for 32 sites.  Same data layout, weights, cell arithmetic and summation order per accumulator
// as dsp_lstm_kernel<0, 1>: bit-identical results (test_small_batch_tiling_does_not_change_a_bit).
// WHERE IT RUNS: batches whose 32-site tiles x 2 directions fit the CUs at once (<= 4,096 sites on 256 CUs; the host
// decides per call, dsp_capi.cpp).  A forward of up to 8,192 sites is ONE round of 64-site workgroups and takes 6.6 ms
// whatever its size; with 32-site workgroups of one wave per SIMD a round takes half as long: 3.7 ms per forward, 512
// sites 77 k -> 137 k sites/s, 2,048 sites 0.31 -> 0.54 M (profiles/r3/batch_sweep.jsonl).  In that mode the launch asks
// for more than half a CU's LDS (flags bit 2), so that one workgroup sits on a CU: forwards issued concurrently on other
// streams then spread over the idle CUs instead of doubling up on busy ones (4 x 512 sites: 0.30 -> 0.49 M sites/s,
// 16 x 512: 0.88 -> 0.99 M; profiles/r3/small_batches.jsonl).
// ON FULL BATCHES (DSP_LSTM_TILING=21 forces it, =0 forbids it: A/B switch) two independent 4-wave workgroups share a CU
// and one's barrier / cell phase runs under the other's k-loop, but an A fragment feeds 4 MFMAs instead of 8 (twice the
// weight stream out of L2 per site): 0.25 % slower than <1, 2>, same-box A/B.  Round 2 measured +0.6 % with a build whose
// h0 read-back was corrupted by the store-data hazard (see bst16); with guarded stores it is sound.
// 8 A fragments per k-group, ring two k-groups deep (64 VGPRs), B ring four deep (16 VGPRs); 246 VGPRs, no spills.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void dsp_lstm21_kernel(LstmArgs a) {
    constexpr int NF = 8;                  // A fragments per k-group: 2 unit tiles x 4 gates
    constexpr int DA = 2, DB = 4;          // ring depths in k-groups
    DSP_DYN_LDS(smem);
    const int nthr = blockDim.x;
    f32x4* c_lds = (f32x4*)smem;           // [2 unit tiles][4 groups][nthr] float4
    f32x4* b_lds = c_lds + 8 * nthr;       // [unit tile][aa][gate][half] float4
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t voff = (uint32_t)lane * 16u;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int UTW = a.UT / 2;              // waves per site tile
    const int ug = w % UTW, sg = w / UTW;
    const int dir = blockIdx.x & 1;
    const int grp = blockIdx.x >> 1;
    const int half = lane >> 5, ls = lane & 31;
    const int HQ = a.Hp >> 2;
    const int nqx = a.Ipad >> 3, nq = nqx + (a.Hp >> 3), NQ = a.NQ;
    const int T = a.T;
    const int F4 = a.Fout >> 2;
    const uint32_t xrow = (uint32_t)(a.Ipad >> 2) * 512u;
    const uint32_t orow = (uint32_t)F4 * 512u;
    const uint32_t ustride = (uint32_t)NQ * 4096u;   // bytes between the wave's two (adjacent) unit tiles in wpk
    const bool prio = (a.flags & 1) != 0;
    const int u0 = 2 * ug;

    const long long gt0 = (long long)grp * a.SG + sg;   // this wave's site tile
    const rsrc_t rw = make_rsrc((const char*)(dir ? a.wpk1 : a.wpk0) + (size_t)u0 * NQ * 4096, dir ? a.wpk1_end : a.wpk0_end, DSP_BND_W);
    const rsrc_t rx = make_rsrc((const char*)a.x + (size_t)gt0 * T * xrow, a.x_end, DSP_BND_X);
    const rsrc_t ro = make_rsrc((const char*)a.out + (size_t)gt0 * T * orow + (size_t)dir * HQ * 512, a.out_end, DSP_BND_OUT);
    const rsrc_t rh0 = make_rsrc((const char*)a.h0buf + (size_t)gt0 * orow + (size_t)dir * HQ * 512, a.h0buf_end, DSP_BND_H0);
    const f32x4* bias4 = (const f32x4*)(dir ? a.sbias1 : a.sbias0);

    for (int i = tid; i < a.Hp; i += nthr) {
        const int h = i & 1, g = (i >> 1) & 3, aa = (i >> 3) & 3, ut = i >> 5;
        b_lds[i] = bias4[g * HQ + ut * 8 + 2 * aa + h];
    }
    {
        const long long site = gt0 * 32 + ls;
        const uint64_t skey = a.init_mode == 2 ? philox_site_key(a.site_keys, a.site_offset, a.n, site) : 0;
#pragma unroll
        for (int ut = 0; ut < 2; ++ut) {
            const int u = u0 + ut;
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                const int k4 = u * 8 + 2 * aa + half;
                f32x4 hv = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f};
                if (a.init_mode != 0) {
                    hv = init_state4(a.h0, a.init_mode, a.n, site, dir, a.H, k4, a.seed, skey, (uint32_t)(a.stream_base + dir * 2 + 0));
                    cv = init_state4(a.c0, a.init_mode, a.n, site, dir, a.H, k4, a.seed, skey, (uint32_t)(a.stream_base + dir * 2 + 1));
                }
                bst16(rh0, voff + aa * 1024u, (uint32_t)u * 4096u, hv);
                c_lds[(ut * 4 + aa) * nthr + tid] = cv;
            }
        }
    }
    barrier_after_global_stores();

    rsrc_t rhp = rh0;
    uint32_t xo = 0, ho = 0;
    auto set_bases = [&](int step) __attribute__((always_inline)) {
        const int t = dir ? (T - 1 - step) : step;
        const int tp = dir ? (t + 1) : (t - 1);
        rhp = step == 0 ? rh0 : ro;
        xo = (uint32_t)t * xrow;
        ho = (step == 0 ? 0u : (uint32_t)tp * orow) - (uint32_t)nqx * 1024u;
    };
    f32x4 A[DA][NF], B[DB];
    f32x16 acc[NF];
    auto loadB = [&](f32x4& Bs, int q) __attribute__((always_inline)) {
        const int qc = q < nq ? q : nq - 1;
        const bool isx = qc < nqx;
        Bs = bld16(isx ? rx : rhp, voff, (isx ? xo : ho) + (uint32_t)qc * 1024u);
    };
    auto ldA = [&](int f, int q) __attribute__((always_inline)) {   // fragment f = unit tile f / 4, gate f % 4
        const uint32_t so = (uint32_t)(f >> 2) * ustride + (uint32_t)(q < NQ ? q : q - NQ) * 4096u;
        return bld16(rw, voff + (uint32_t)(f & 3) * 1024u, so);
    };
#define QW(x) ((x) < NQ ? (x) : (x) - NQ)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto stage = [&](auto qs, int q, auto first) __attribute__((always_inline)) {
        constexpr int QS = decltype(qs)::value;
        constexpr int sa = QS % DA, sp = (QS + DA - 1) % DA, sb = QS % DB;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (decltype(first)::value && i == 0)
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[sa][f][i], B[sb][i], zero16, 0, 0, 0);
                else
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[sa][f][i], B[sb][i], acc[f], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (f == 0) A[sp][NF - 1] = ldA(NF - 1, q + DA - 1);
            else A[sa][f - 1] = ldA(f - 1, q + DA);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto stage4 = [&](int q, auto first) __attribute__((always_inline)) {
        stage(ic<0>{}, q + 0, first); loadB(B[0], QW(q + 0 + DB)); __builtin_amdgcn_sched_barrier(0);
        stage(ic<1>{}, q + 1, std::false_type{}); loadB(B[1], QW(q + 1 + DB)); __builtin_amdgcn_sched_barrier(0);
        stage(ic<2>{}, q + 2, std::false_type{}); loadB(B[2], QW(q + 2 + DB)); __builtin_amdgcn_sched_barrier(0);
        stage(ic<3>{}, q + 3, std::false_type{}); loadB(B[3], QW(q + 3 + DB)); __builtin_amdgcn_sched_barrier(0);
    };

    set_bases(0);
#pragma unroll
    for (int d = 0; d < DB; ++d) {
        if (d < DA) {
#pragma unroll
            for (int f = 0; f < NF; ++f) A[d][f] = ldA(f, d);
        }
        loadB(B[d], d);
    }
    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        if (step > 0) barrier_after_global_stores();
        if (prio) __builtin_amdgcn_s_setprio(2);
        stage4(0, std::true_type{});
        for (int q = 4; q < NQ - 4; q += 4) stage4(q, std::false_type{});
        set_bases(step + 1 < T ? step + 1 : step);
        stage4(NQ - 4, std::false_type{});
        A[DA - 1][NF - 1] = ldA(NF - 1, NQ + DA - 1);
        if (prio) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int ut = 0; ut < 2; ++ut) {
            const int u = u0 + ut;
            const f32x4* b_my = b_lds + (size_t)u * 32 + half;
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                const f32x4 bi = b_my[aa * 8 + 0], bf = b_my[aa * 8 + 2], bg = b_my[aa * 8 + 4], bo = b_my[aa * 8 + 6];
                f32x4 cv = c_lds[(ut * 4 + aa) * nthr + tid];
                f32x4 hv;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * aa + i;
                    const float ig = sigmoid_pre(acc[ut * 4 + 0][r], bi[i]);
                    const float fg = sigmoid_pre(acc[ut * 4 + 1][r], bf[i]);
                    const float gg = tanh_pre(acc[ut * 4 + 2][r], bg[i]);
                    const float og = sigmoid_pre(acc[ut * 4 + 3][r], bo[i]);
                    const float cn = __builtin_fmaf(fg, cv[i], ig * gg);
                    cv[i] = cn;
                    hv[i] = og * fast_tanh(cn);
                }
                c_lds[(ut * 4 + aa) * nthr + tid] = cv;
                bst16(ro, voff + aa * 1024u, (uint32_t)t * orow + (uint32_t)u * 4096u, hv);
            }
        }
    }
#undef QW
}

// ------------------------------------------------------------------------------------------------
// dsp_lstmc_kernel<G, D> (round 4): the dense one-pass layers for batches that leave most of the chip idle.
// A batch of 512 sites is 16 site tiles x 2 directions = 32 independent recurrences; dsp_lstm21_kernel gives each of them ONE
// compute unit (4 waves, 2 unit tiles each) and a time step then costs 96 k-groups x 2,048 cycles = 82 us on that CU while
// 224 CUs idle: 3.7 ms per forward, whatever the batch below 4,096 sites.  Here the 8 unit tiles x 4 gates of a (site tile,
// direction) are spread over a CLUSTER of P = 8 / G workgroups on P compute units: a wave owns G gates of one unit tile
// (G accumulator tiles), a workgroup of 4 waves owns G unit tiles:
//      G = 4: P = 2 (2,048 sites fill 256 CUs)    G = 2: P = 4 (1,024 sites)    G = 1: P = 8 (512 sites)
//   * SAME arithmetic: an accumulator tile still sums its k-groups in order with the same MFMAs, the cell phase is the same
//     code on the same values -> results are bit-identical to dsp_lstm_kernel<0, 1> / dsp_lstm21_kernel.
//   * Gates of one unit tile that sit in different waves (G < 4) meet in LDS after the k-loop: every wave writes its
//     accumulator tiles as [unit tile][gate][row group aa][lane] float4, one workgroup barrier, every wave reads the four gates
//     of the row groups it owns (wave (unit tile, slice s) owns row groups [s G, s G + G): cell state in registers).
//   * h_t crosses compute units through memory.  Producer: the h stores are WRITE-THROUGH (sc1), every wave drains them
//     (s_waitcnt vmcnt(0)), one workgroup barrier, one lane adds 1 to the cluster's arrival counter (relaxed, agent scope).
//     Consumer: every wave polls that ONE word (relaxed agent-scope load) until P x (step + 1) arrivals are in, then reads
//     h_{t-1} with sc1 loads (served by L2 / memory, never by a CU's L1) -- the placement-independent R1 hand-off of
//     /opt/skills/guides/cdna_hip_programming.md, Guideline 16.  No h row is requested before the counter says it exists:
//     the k-loop runs the x part first (k-groups [0, nqx), 2/3 of a step in layers 1+) and polls where the B ring would
//     first reach into the h part, so the hop is hidden behind the x part of the NEXT step as long as the cluster's
//     workgroups run in step.  The counters are zeroed by the forward's first launch (dsp_pack_kernel).
//   * Residency: the host only picks a cluster size whose whole grid fits the compute units at once (one workgroup per CU:
//     the launch asks for more than half a CU's LDS); members of a cluster are P consecutive workgroups of ONE XCD's
//     dispatch list (block b runs on XCD b % 8, observed), so that with in-order dispatch the lowest unfinished cluster
//     always has its missing members at the head of the list.  A poll that sees nothing for seconds traps (loud failure,
//     never a hang, never silent garbage).
//   * Rings D k-groups deep for both operands (D x 256 G cycles of MFMA work in flight: 4 / 8 / 16 for G = 4 / 2 / 1).
// Requirements (host-checked): 8 unit tiles (hidden 193..256), no padded k-groups, nqx a multiple of D and >= 2 D.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) unsigned int gu32;
__device__ __forceinline__ f32x4 bld16_sc1_(const rsrc_t& r, uint32_t voff, uint32_t soff, int line) {
    bounds_check(r, voff, soff, line);
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(RS(r), (int)voff, (int)soff, 16));   // aux 16 = sc1
}
__device__ __forceinline__ void bst16_sc1_(const rsrc_t& r, uint32_t voff, uint32_t soff, f32x4 v, int line) {
    bounds_check(r, voff, soff, line);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), RS(r), (int)voff, (int)soff, 16);
    store_data_guard(v);
}
#define bld16_sc1(r, voff, soff) bld16_sc1_((r), (voff), (soff), __LINE__)
#define bst16_sc1(r, voff, soff, v) bst16_sc1_((r), (voff), (soff), (v), __LINE__)
// The two lock-free functions of the protocol -- admission of the members, the wait for the arrivals of a step -- live in
// dsp_cluster_protocol.h, written against an Ops policy: the same source runs here (relaxed agent-scope atomics, s_sleep,
// s_memtime) and, with std::atomic under ThreadSanitizer, in tests/native/cluster_model.cpp.
struct ClusterOpsDev {
    gu32* flag;   // word 0: arrivals of the steps; word 1: admission (count | abandoned)
    static constexpr unsigned kCheckMask = 0xfffu;       // the abandoned bit is looked at every 4,096 polls
    static constexpr unsigned kSpinLimit = 1u << 24;     // polls without progress (seconds) before a member gives the cluster up
    __device__ __forceinline__ unsigned load_arrivals() const {
        return __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    __device__ __forceinline__ unsigned load_state() const {
        return __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    __device__ __forceinline__ void or_state(unsigned bits) const {
        __hip_atomic_fetch_or(flag + 1, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ bool cas_state(unsigned& expected, unsigned desired) const {
        return __hip_atomic_compare_exchange_strong(flag + 1, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void pause() const { __builtin_amdgcn_s_sleep(1); }
    __device__ __forceinline__ unsigned long long now() const { return __builtin_amdgcn_s_memtime(); }
};
__device__ __forceinline__ bool wait_arrivals(gu32* flag, unsigned target) { return dsp_wait_arrivals(ClusterOpsDev{flag}, target); }
__device__ __forceinline__ bool cluster_admit(gu32* flag, unsigned P, unsigned long long limit) {
    return dsp_cluster_admit(ClusterOpsDev{flag}, P, limit);
}

template <int LO, int HI, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (LO < HI) { f(ic<LO>{}); static_for<LO + 1, HI>(f); }
}

// One layer of one (site tile, direction) on this workgroup: prologue (ring fill, bias table, initial states, h0 hand-off)
// and the T steps.  `pi` = this workgroup's index in the cluster (0 when LOCAL), `flag` = the cluster's counters of THIS layer.
// Returns false when the cluster was given up on the way (wait_arrivals).
template <int G, int D, bool LOCAL, int DEAD, int NW, bool XSHORT, bool XA>
__device__ __forceinline__ bool lstmc_layer(const LstmArgs& a, const int dir, const long long gt0, const int pi, gu32* flag) {
    constexpr int WPU = 4 / G;             // waves per unit tile = gate slices; unit tiles per workgroup = G
    DSP_DYN_LDS(smem);
    f32x4* b_lds = (f32x4*)smem;           // [unit tile][aa][gate][half] float4 (Hp float4)
    f32x4* xch = b_lds + a.Hp;             // G < 4: [local unit tile][gate][aa][64 lanes] float4
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t voff = (uint32_t)lane * 16u;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = LOCAL ? 1 : a.UT / G;
    const int ul = w / WPU, gs = w % WPU;
    const int u = pi * G + ul;             // this wave's unit tile
    const int half = lane >> 5, ls = lane & 31;
    const int HQ = a.Hp >> 2;
    const int nqx = a.Ipad >> 3, NQ = a.NQ;
    const int T = a.T;
    const int F4 = a.Fout >> 2;
    const uint32_t xrow = (uint32_t)(a.Ipad >> 2) * 512u;
    const uint32_t orow = (uint32_t)F4 * 512u;
    const bool prio = (a.flags & 1) != 0;
    // XA ("x ahead", LstmArgs::xs): the k-groups [0, xs) of every step were summed by dsp_xahead_kernel; this launch's own x part
    // is the ring [xs, xs + D) = [nqx - D, nqx), its k-loop wraps from NQ back to xs
    int xs = 0;
    if constexpr (XA) xs = a.xs;

    const rsrc_t rw = make_rsrc((const char*)(dir ? a.wpk1 : a.wpk0) + (size_t)u * NQ * 4096, dir ? a.wpk1_end : a.wpk0_end, DSP_BND_W);
    const rsrc_t rx = make_rsrc((const char*)a.x + (size_t)gt0 * T * xrow, a.x_end, DSP_BND_X);
    const rsrc_t ro = make_rsrc((const char*)a.out + (size_t)gt0 * T * orow + (size_t)dir * HQ * 512, a.out_end, DSP_BND_OUT);
    const rsrc_t rh0 = make_rsrc((const char*)a.h0buf + (size_t)gt0 * orow + (size_t)dir * HQ * 512, a.h0buf_end, DSP_BND_H0);
    // (XA) this unit tile's x-ahead sums of the cluster: [t][unit tile][gate][row group][lane] float4
    const rsrc_t rxa = XA ? make_rsrc((const char*)a.xacc + ((size_t)(gt0 * 2 + dir) * T * a.UT + u) * 16384, a.xacc_end, DSP_BND_XACC) : rw;
    const f32x4* bias4 = (const f32x4*)(dir ? a.sbias1 : a.sbias0);
    const uint32_t voffA = voff + (uint32_t)(gs * G) * 1024u;   // this wave's gates within a k-group's 4 KiB of weights
    const uint32_t voffO = voff + (uint32_t)(gs * G) * 1024u;   // this wave's row groups within a unit tile's 4 KiB of h

    // (the rings are filled FIRST: weights and x rows depend on nothing computed here, and their round trip runs under the
    // Philox / Box-Muller work of the initial states below instead of behind it)
    rsrc_t rhp = rh0;
    uint32_t xo = 0, ho = 0;
    auto set_bases = [&](int step) __attribute__((always_inline)) {
        const int t = dir ? (T - 1 - step) : step;
        const int tp = dir ? (t + 1) : (t - 1);
        rhp = step == 0 ? rh0 : ro;
        xo = (uint32_t)t * xrow;
        ho = (step == 0 ? 0u : (uint32_t)tp * orow) - (uint32_t)nqx * 1024u;
    };
    f32x4 A[D][G], B[D];
    f32x16 acc[G];
    f32x16 accn[XA ? G : 1];   // XA: what the step's accumulators start from (requested a step early)
    auto ldA = [&](int f, int q) __attribute__((always_inline)) {
        return bld16(rw, voffA + (uint32_t)f * 1024u, (uint32_t)(q < NQ ? q : q - NQ + xs) * 4096u);
    };
    auto load_accn = [&](int step) __attribute__((always_inline)) {
        if constexpr (XA) {
            const int t = dir ? (T - 1 - step) : step;
            const uint32_t so = (uint32_t)t * (uint32_t)a.UT * 16384u;
#pragma unroll
            for (int f = 0; f < G; ++f)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = bld16(rxa, voff + (uint32_t)((gs * G + f) * 4 + i) * 1024u, so);
                    accn[f][4 * i] = v[0]; accn[f][4 * i + 1] = v[1]; accn[f][4 * i + 2] = v[2]; accn[f][4 * i + 3] = v[3];
                }
        }
    };
    // B fragments: x part (plain loads: written by an earlier launch), h part (sc1: written by the cluster during this one)
    auto ldBx = [&](int q) __attribute__((always_inline)) { return bld16(rx, voff, xo + (uint32_t)q * 1024u); };
    auto ldBh = [&](int q) __attribute__((always_inline)) {
        if constexpr (LOCAL) return bld16(rhp, voff, ho + (uint32_t)q * 1024u);
        else return bld16_sc1(rhp, voff, ho + (uint32_t)q * 1024u);
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // one k-group: per fragment its 4 MFMAs, then the (late) refill of the fragment before it for k-group q + D; the B
    // fragment of the slot is refilled behind the stage.  kind (compile time): 0 = the refill is an x row of THIS step's
    // bases, 1 = an h row
    // the refills of stage qs at k-group q, in the stage's own order (f == 0: the last fragment of the slot before it)
    auto refill_A = [&](auto qs, auto fi, int q) __attribute__((always_inline)) {
        constexpr int S = decltype(qs)::value % D, SP = (decltype(qs)::value + D - 1) % D, f = decltype(fi)::value;
        if constexpr (f == 0) A[SP][G - 1] = ldA(G - 1, q + D - 1);
        else A[S][f - 1] = ldA(f - 1, q + D);
    };
    auto refill_B = [&](auto qs, int q, auto kind) __attribute__((always_inline)) {
        constexpr int S = decltype(qs)::value % D;
        const int qn = q + D < NQ ? q + D : q + D - NQ + xs;
        if constexpr (decltype(kind)::value == 1) B[S] = ldBh(qn); else B[S] = ldBx(qn);
    };
    // norefill (compile time): the stage issues NO memory request -- its refills are made up for later, in the same order,
    // by refill_stage (the deferred arrival of the per-wave hand-off: nothing may be requested between the h stores and
    // their drain, or the drain would wait for it too)
    auto stage = [&](auto qs, int q, auto first, auto kind, auto dead, auto norefill) __attribute__((always_inline)) {
        constexpr int S = decltype(qs)::value % D;
#pragma unroll
        for (int f = 0; f < G; ++f) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (decltype(dead)::value) break;
                if (decltype(first)::value && i == 0)
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[S][f][i], B[S][i], XA ? accn[XA ? f : 0] : zero16, 0, 0, 0);
                else
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[S][f][i], B[S][i], acc[f], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!decltype(norefill)::value) {
                if (f == 0) refill_A(qs, ic<0>{}, q);
                else if (f == 1) refill_A(qs, ic<1 % G>{}, q);
                else if (f == 2) refill_A(qs, ic<2 % G>{}, q);
                else refill_A(qs, ic<3 % G>{}, q);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!decltype(norefill)::value) refill_B(qs, q, kind);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto refill_stage = [&](auto qs, int q, auto kind) __attribute__((always_inline)) {
        refill_A(qs, ic<0>{}, q);
        if constexpr (G > 1) refill_A(qs, ic<1 % G>{}, q);
        if constexpr (G > 2) { refill_A(qs, ic<2 % G>{}, q); refill_A(qs, ic<3 % G>{}, q); }
        refill_B(qs, q, kind);
        __builtin_amdgcn_sched_barrier(0);
    };
    using live = std::false_type;
    using refills = std::false_type;   // (the usual stage: refills right behind the MFMAs)
    auto stages = [&](int q, auto first, auto kind) __attribute__((always_inline)) {
        stage(ic<0>{}, q, first, kind, live{}, refills{});
        if constexpr (D > 1) stage(ic<1>{}, q + 1, std::false_type{}, kind, live{}, refills{});
        if constexpr (D > 2) { stage(ic<2>{}, q + 2, std::false_type{}, kind, live{}, refills{}); stage(ic<3>{}, q + 3, std::false_type{}, kind, live{}, refills{}); }
        if constexpr (D > 4) { stage(ic<4>{}, q + 4, std::false_type{}, kind, live{}, refills{}); stage(ic<5>{}, q + 5, std::false_type{}, kind, live{}, refills{});
                               stage(ic<6>{}, q + 6, std::false_type{}, kind, live{}, refills{}); stage(ic<7>{}, q + 7, std::false_type{}, kind, live{}, refills{}); }
        if constexpr (D > 8) { stage(ic<8>{}, q + 8, std::false_type{}, kind, live{}, refills{}); stage(ic<9>{}, q + 9, std::false_type{}, kind, live{}, refills{});
                               stage(ic<10>{}, q + 10, std::false_type{}, kind, live{}, refills{}); stage(ic<11>{}, q + 11, std::false_type{}, kind, live{}, refills{});
                               stage(ic<12>{}, q + 12, std::false_type{}, kind, live{}, refills{}); stage(ic<13>{}, q + 13, std::false_type{}, kind, live{}, refills{});
                               stage(ic<14>{}, q + 14, std::false_type{}, kind, live{}, refills{}); stage(ic<15>{}, q + 15, std::false_type{}, kind, live{}, refills{}); }
    };
    // the first block of a front-end step (D == 4 == nqx): DEAD refill-only stages, then the live x-part k-groups, the first
    // of which starts the accumulators from zero
    auto stages_first_sparse = [&](auto kind) __attribute__((always_inline)) {
        static_assert(DEAD == 0 || D == 4, "front-end shape");
        stage(ic<0>{}, 0, std::integral_constant<bool, DEAD == 0>{}, kind, std::integral_constant<bool, (0 < DEAD)>{}, refills{});
        stage(ic<1>{}, 1, std::integral_constant<bool, DEAD == 1>{}, kind, std::integral_constant<bool, (1 < DEAD)>{}, refills{});
        stage(ic<2>{}, 2, std::integral_constant<bool, DEAD == 2>{}, kind, std::integral_constant<bool, (2 < DEAD)>{}, refills{});
        stage(ic<3>{}, 3, std::integral_constant<bool, DEAD == 3>{}, kind, live{}, refills{});
    };

    TSTAMP_AT(13, 0);   // (DSP_TRACE builds: prologue start / ring fill issued / states stored / h0 published)
    set_bases(0);
#pragma unroll
    for (int d = 0; d < D; ++d) {
#pragma unroll
        for (int f = 0; f < G; ++f) A[d][f] = ldA(f, xs + d);
        B[d] = ldBx(xs + d);
    }
    load_accn(0);
    TSTAMP_AT(13, 1);
    for (int i = tid; i < a.Hp; i += NW * 64) {
        const int h = i & 1, g = (i >> 1) & 3, aa = (i >> 3) & 3, ut = i >> 5;
        b_lds[i] = bias4[g * HQ + ut * 8 + 2 * aa + h];
    }
    // initial state of the row groups this wave owns: c0 -> registers, h0 -> the K4 scratch step 0 reads as "h_{-1}"
    f32x4 creg[G];
    {
        const long long site = gt0 * 32 + ls;
        const uint64_t skey = a.init_mode == 2 ? philox_site_key(a.site_keys, a.site_offset, a.n, site) : 0;
#pragma unroll
        for (int al = 0; al < G; ++al) {
            const int k4 = u * 8 + 2 * (gs * G + al) + half;
            f32x4 hv = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f};
            if (a.init_mode != 0) {
                hv = init_state4(a.h0, a.init_mode, a.n, site, dir, a.H, k4, a.seed, skey, (uint32_t)(a.stream_base + dir * 2 + 0));
                cv = init_state4(a.c0, a.init_mode, a.n, site, dir, a.H, k4, a.seed, skey, (uint32_t)(a.stream_base + dir * 2 + 1));
            }
            if constexpr (LOCAL) bst16(rh0, voffO + al * 1024u, (uint32_t)u * 4096u, hv);
            else bst16_sc1(rh0, voffO + al * 1024u, (uint32_t)u * 4096u, hv);
            creg[al] = cv;
        }
    }
    // Hand-off, round 5 (flags bit 6; DSP_LSTM_HANDOFF=0 keeps round 4's): every WAVE counts its own arrival -- it drains its
    // own write-through stores and adds 1; no workgroup barrier (the poll target is P x NW x (step + 1); the barrier that kept
    // the next step's LDS writes behind this step's LDS reads is implied: nobody passes the next poll before every wave has
    // arrived, i.e. finished its cell phase).  In the layers with a real x part the arrival is DEFERRED into the next step:
    // the first E stages of the x part run on what the ring already holds and request nothing, so that by the time the wave
    // drains, its h stores have long been acknowledged (round 4: 750 cycles per step in drain + barrier + atomic, every one of
    // them with the matrix pipe idle); the E stages' refills follow in one burst -- the ring is D stages deep, E of them
    // cover the store's round trip and D - E the refills'.  And the poll's own round trip (460 cycles with the wave stalled in
    // front of D stages of x rows it already holds) is taken out of the way by requesting the counter one block early.
    // (not for the front ends' XSHORT form: with no x part to defer into, 16 waves' atomics on one word lengthen the hop the
    // step is waiting for -- 51.9 -> 53.1 us per launch at 512 sites -- so they keep one arrival per workgroup)
    const bool wavepub = !LOCAL && !XSHORT && (a.flags & 64) != 0;
    const unsigned per_step = wavepub ? (unsigned)(P * NW) : (unsigned)P;
    // stages without requests in front of a deferred arrival (>= 1: stage 0 starts the accumulators).  Same-box A/B of the
    // depth (profiles/r5/handoff_ab.txt): 3 / 2 / 1 as here 0.5697 / 0.9767 / 1.7729 ms per forward of 512 / 1,024 / 2,048 sites;
    // 1 / 1 / 1: 0.5726 / 0.9792 / 1.7716; 6 / 3 / 2: +8 / +6 / +5 us; 10 / 5 / 2: +17 / +10 / +3 us (the burst of refills
    // behind the drain is issued with no MFMA in flight); round 4's hand-off 0.5716 / 0.9943 / 1.7962
    constexpr int E = D == 4 ? 1 : (G == 1 ? 3 : 2);   // (rings four deep -- 4 unit tiles, or G = 4 -- spare one stage)
    auto arrive = [&]() __attribute__((always_inline)) {
        DSP_DRAIN_STORES();
        if (lane == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // publish: every wave drains its write-through stores, the workgroup meets, one lane counts the arrival
    auto publish = [&]() __attribute__((always_inline)) {
        if constexpr (LOCAL) {
            barrier_after_global_stores();   // same CU: the stores of this workgroup's waves are ahead of the loads issued after it
        } else {
            DSP_DRAIN_STORES();
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    TSTAMP_AT(13, 2);
    if (!wavepub) publish();
    else {
        __syncthreads();              // (the bias table is complete before any wave's cell phase reads it)
    }
    TSTAMP_AT(13, 3);

    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        TSTAMP(0);
        if (prio) __builtin_amdgcn_s_setprio(2);
        if constexpr (LOCAL) {
            // (the barrier behind the previous step's -- or the prologue's -- h stores has been passed: every request may go out)
            if (nqx > D) {
                stages(0, std::true_type{}, ic<0>{});
                for (int q = D; q < nqx - D; q += D) stages(q, std::false_type{}, ic<0>{});
                for (int q = nqx - D; q < NQ - D; q += D) stages(q, std::false_type{}, ic<1>{});
            } else {
                if constexpr (DEAD > 0) stages_first_sparse(ic<1>{});
                else stages(0, std::true_type{}, ic<1>{});
                for (int q = D; q < NQ - D; q += D) stages(q, std::false_type{}, ic<1>{});
            }
        } else if constexpr (XSHORT) {
            // every refill of this step is an h row: h_{t-1} (h0 at step 0) of every member of the cluster must be in memory
            asm volatile("" ::: "memory");
            TSTAMP(1);
            if (!wait_arrivals(flag, per_step * (unsigned)(step + 1))) return false;   // (given up: the clean-up launch computes this cluster)
            TSTAMP(2);
            asm volatile("" ::: "memory");
            if constexpr (DEAD > 0) stages_first_sparse(ic<1>{});
            else stages(0, std::true_type{}, ic<1>{});
            for (int q = D; q < NQ - D; q += D) stages(q, std::false_type{}, ic<1>{});
        } else if constexpr (XA) {
            // the x part is ONE ring, [xs, xs + D): its stages run on what the ring already holds (x rows requested by the
            // previous step's last block), but every refill they make is an h row of THIS step -- the poll stands in front of
            // the first request.  Per-wave hand-off: E stages that request nothing, this wave's deferred arrival, the poll, the
            // E stages' refills in one burst, the rest of the ring; else (one arrival per workgroup, counted behind the cell
            // phase) the poll first.
            if (wavepub) {
                static_for<0, E>([&](auto i) __attribute__((always_inline)) {
                    stage(i, xs + decltype(i)::value, std::integral_constant<bool, decltype(i)::value == 0>{}, ic<1>{}, live{}, std::true_type{});
                });
                arrive();
            }
            asm volatile("" ::: "memory");
            TSTAMP(1);
            if (!wait_arrivals(flag, per_step * (unsigned)(step + 1))) return false;   // (given up: the clean-up launch computes this cluster)
            TSTAMP(2);
            asm volatile("" ::: "memory");
            if (wavepub) {
                static_for<0, E>([&](auto i) __attribute__((always_inline)) { refill_stage(i, xs + decltype(i)::value, ic<1>{}); });
                static_for<E, D>([&](auto i) __attribute__((always_inline)) {
                    stage(i, xs + decltype(i)::value, std::false_type{}, ic<1>{}, live{}, refills{});
                });
            } else {
                stages(xs, std::true_type{}, ic<1>{});
            }
            for (int q = xs + D; q < NQ - D; q += D) stages(q, std::false_type{}, ic<1>{});
        } else {
            // x part: nothing here depends on h_{t-1}
            unsigned seen = 0;
            if (!wavepub) {
                stages(0, std::true_type{}, ic<0>{});
                for (int q = D; q < nqx - D; q += D) stages(q, std::false_type{}, ic<0>{});
            } else {
                // the first block: E stages that request nothing, the deferred arrival of the previous step's (the prologue's)
                // h stores, the E stages' refills in one burst, the rest of the block
                static_for<0, E>([&](auto i) __attribute__((always_inline)) {
                    stage(i, decltype(i)::value, std::integral_constant<bool, decltype(i)::value == 0>{}, ic<0>{}, live{}, std::true_type{});
                });
                arrive();
                static_for<0, E>([&](auto i) __attribute__((always_inline)) { refill_stage(i, decltype(i)::value, ic<0>{}); });
                static_for<E, D>([&](auto i) __attribute__((always_inline)) {
                    stage(i, decltype(i)::value, std::false_type{}, ic<0>{}, live{}, refills{});
                });
                for (int q = D; q < nqx - 2 * D; q += D) stages(q, std::false_type{}, ic<0>{});
                // the counter, one block early (nqx == 2 D: this was the only block -- the value may be too early to be final)
                if (nqx > 2 * D) {
                    seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    stages(nqx - 2 * D, std::false_type{}, ic<0>{});
                }
            }
            // the B ring is about to reach into the h part: h_{t-1} (h0 at step 0) of every member of the cluster must be in memory
            asm volatile("" ::: "memory");   // (no h load may be moved above the poll by the compiler either)
            TSTAMP(1);
            if ((unsigned)__builtin_amdgcn_readfirstlane(seen) < per_step * (unsigned)(step + 1))
                if (!wait_arrivals(flag, per_step * (unsigned)(step + 1))) return false;   // (given up: the clean-up launch computes this cluster)
            TSTAMP(2);
            asm volatile("" ::: "memory");
            for (int q = nqx - D; q < NQ - D; q += D) stages(q, std::false_type{}, ic<1>{});
        }
        set_bases(step + 1 < T ? step + 1 : step);   // the last D refills are the next step's first x rows
        stages(NQ - D, std::false_type{}, ic<0>{});
        if constexpr (XA) { if (step + 1 < T) load_accn(step + 1); }
        if (prio) __builtin_amdgcn_s_setprio(0);
        TSTAMP(3);

        // the four gates of a row group meet: in this wave's registers (G == 4) or through LDS
        if constexpr (G < 4) {
#pragma unroll
            for (int gl = 0; gl < G; ++gl)
#pragma unroll
                for (int aa = 0; aa < 4; ++aa)
                    xch[((ul * 4 + gs * G + gl) * 4 + aa) * 64 + lane] =
                        f32x4{acc[gl][4 * aa], acc[gl][4 * aa + 1], acc[gl][4 * aa + 2], acc[gl][4 * aa + 3]};
            __syncthreads();
        }
        TSTAMP(4);
        const f32x4* b_my = b_lds + (size_t)u * 32 + half;
#pragma unroll
        for (int al = 0; al < G; ++al) {
            const int aa = gs * G + al;
            const f32x4 bi = b_my[aa * 8 + 0], bf = b_my[aa * 8 + 2], bg = b_my[aa * 8 + 4], bo = b_my[aa * 8 + 6];
            f32x4 gi, gf, gg4, go;
            if constexpr (G < 4) {
                gi = xch[((ul * 4 + 0) * 4 + aa) * 64 + lane]; gf = xch[((ul * 4 + 1) * 4 + aa) * 64 + lane];
                gg4 = xch[((ul * 4 + 2) * 4 + aa) * 64 + lane]; go = xch[((ul * 4 + 3) * 4 + aa) * 64 + lane];
            } else {
                gi = f32x4{acc[0][4 * al], acc[0][4 * al + 1], acc[0][4 * al + 2], acc[0][4 * al + 3]};
                gf = f32x4{acc[1 % G][4 * al], acc[1 % G][4 * al + 1], acc[1 % G][4 * al + 2], acc[1 % G][4 * al + 3]};
                gg4 = f32x4{acc[2 % G][4 * al], acc[2 % G][4 * al + 1], acc[2 % G][4 * al + 2], acc[2 % G][4 * al + 3]};
                go = f32x4{acc[3 % G][4 * al], acc[3 % G][4 * al + 1], acc[3 % G][4 * al + 2], acc[3 % G][4 * al + 3]};
            }
            f32x4 cv = creg[al], hv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float ig = sigmoid_pre(gi[i], bi[i]);
                const float fg = sigmoid_pre(gf[i], bf[i]);
                const float gt = tanh_pre(gg4[i], bg[i]);
                const float og = sigmoid_pre(go[i], bo[i]);
                const float cn = __builtin_fmaf(fg, cv[i], ig * gt);
                cv[i] = cn;
                hv[i] = og * fast_tanh(cn);
            }
            creg[al] = cv;
            if constexpr (LOCAL) bst16(ro, voffO + al * 1024u, (uint32_t)t * orow + (uint32_t)u * 4096u, hv);
            else bst16_sc1(ro, voffO + al * 1024u, (uint32_t)t * orow + (uint32_t)u * 4096u, hv);
        }
        TSTAMP(5);
        if (!wavepub) publish();   // (also the barrier between this step's LDS reads and the next step's LDS writes)
        // (wavepub: the arrival is counted E stages into the next step; the last step's is nobody's to wait for)
        TSTAMP(6);
    }
    TSTAMP_AT(14, 0);
    return true;
}

template <int G, int D, bool LOCAL = false, int DEAD = 0, int NW = 4, bool XSHORT = false, bool XA = false>
__global__ __launch_bounds__(NW * 64, 1) void dsp_lstmc_kernel(LstmArgs a) {
    static_assert(!XA || (!LOCAL && !XSHORT && DEAD == 0), "x ahead: the clustered dense forms");
    // XSHORT (round 5; clustered, layers of 4 unit tiles whose x part is exactly one ring: nqx == D == 4 -- the front ends at
    // hidden 128): a (site tile, direction) of a front end spread over P = 4 / G workgroups.  Until round 5 the front ends of a
    // small batch ran workgroup-local: 32 recurrences on 32 CUs at 512 sites, 13 steps x 17 k-groups x 4 gates x 256 cycles =
    // 0.12-0.13 ms per launch whatever the batch, a fifth of a 512-site forward, with 192 CUs idle.  There is no x part to hide
    // the hop behind (its k-groups are the ring's first fill, requested a step early): the poll comes first, then the
    // (partly dead) x-part stages whose refills are the first h rows.
    // NW = 8 (LOCAL, layers of 8 unit tiles): eight waves, two per SIMD, a wave 1 unit tile x 1 site tile -- the alternative
    // to dsp_lstm21_kernel's four waves of 2 unit tiles for batches of 2,049..4,096 sites (A/B: DSP_LSTM_LOCAL8)
    // DEAD (LOCAL, nqx == D: the front ends, 7 or 16 features at the end of a 32-wide block): the first DEAD k-groups of a
    // step are pure zero padding -- their stages keep the ring turning (refills only) and issue no MFMAs
    // LOCAL (G = 4, layers of 4 unit tiles: the front ends at hidden 128): the workgroup holds the whole hidden state, P = 1 --
    // the h exchange is the step barrier of dsp_lstm21_kernel (plain stores, one s_barrier, plain loads), no counter; the x
    // part may be as short as the ring (nqx == D: the h part's first requests then leave right behind the barrier)
    DSP_DYN_LDS(smem);
    const int tid = threadIdx.x;
    const int P = LOCAL ? 1 : a.UT / G;
    // cluster c = (site tile, direction); its P members are consecutive entries of one XCD's dispatch list
    const int xs = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int pi = j % P;
    const long long c = (long long)(j / P) * 8 + xs;
    if (c >= a.NTp * 2) return;
    const int dir = (int)(c & 1);
    const long long gt0 = c >> 1;
    // (XA: tiles without a live site are not computed -- by dsp_xahead_kernel, by this launch, by its clean-up launch; nothing
    // reads their rows but the same tiles of later launches, and no output row comes from them.  All members of such a cluster
    // leave here, before any of them touches a counter.)
    if constexpr (XA) { if (gt0 * 32 >= a.n) return; }
    gu32* flag = (gu32*)a.cflags + c * 32;   // word 0: arrivals of the steps; word 1: admission (count | abandoned)
    if (!LOCAL || (a.flags & 16)) BOUNDS_FLAT((const unsigned*)a.cflags + c * 32, 8, a.cflags, a.cflags_end, DSP_BND_FLAGS);
    if constexpr (LOCAL) {
        // the clean-up launch behind a clustered one (flags bit 4): only the clusters that were abandoned are computed here
        if ((a.flags & 16) && !(__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kClusterAbandon)) return;
    } else {
        // (the verdict travels through the first word of the dynamic LDS -- no static LDS next to a 160 KiB dynamic limit --,
        // which the bias table overwrites only after everybody has read it)
        int* verdict = (int*)smem;
        if (tid == 0) *verdict = cluster_admit(flag, (unsigned)P, a.cluster_timeout) ? 1 : 0;
        __syncthreads();
        const int admitted = *(volatile int*)verdict;
        __syncthreads();
        if (!admitted) return;
    }
    lstmc_layer<G, D, LOCAL, DEAD, NW, XSHORT, XA>(a, dir, gt0, pi, flag);
}

// ------------------------------------------------------------------------------------------------
// dsp_xahead_kernel (round 6, opt-in: DSP_LSTM_XAHEAD=1): the part of an LSTM layer that is NOT a recurrence, taken out of it.
// A step of a dense layer sums k-groups [0, nqx) of x_t W_ih^T and [nqx, NQ) of h_{t-1} W_hh^T into one accumulator tile; only
// the second half waits for the step before.  On a batch of <= 256 sites the clustered launch keeps 8 CUs per (site tile,
// direction) busy with a chain of 13 steps x 96 k-groups while most of the chip has nothing to do, and two thirds of that chain
// (layers 1+: nqx = 64 of NQ = 96) is x part.  Here every (cluster, step t, unit tile, gate) gets its own wave -- T x 32 of them
// per cluster instead of 32 -- which sums the k-groups [0, xs) with the SAME MFMAs in the SAME order, from zero, and leaves the
// accumulator tile in a.xacc.  The recurrent launch (dsp_lstmc_kernel<.., XA>) starts the step's accumulators from it (the C
// operand of its first MFMA) and carries on with k-group xs: every sum is made of the same terms in the same order, so the
// results are bit-identical to the undivided launch -- unlike a split of K across workgroups, which would reorder them.
// Grid: live clusters x T x UT workgroups of four waves (one gate each); unit tile = blockIdx % UT, so that with 8 unit tiles
// the weights of unit tile u are read through XCD u's L2 only.  Tiles without a live site are skipped (see dsp_lstmc_kernel).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void dsp_xahead_kernel(LstmArgs a) {
    constexpr int R = 16;                  // ring depth in k-groups (one k-group = 4 MFMAs = 256 cycles of this wave)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t voff = (uint32_t)lane * 16u;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int UT = a.UT, T = a.T, NQ = a.NQ, xs = a.xs;
    const int u = (int)(blockIdx.x % (unsigned)UT);
    const unsigned r = blockIdx.x / (unsigned)UT;
    const int t = (int)(r % (unsigned)T);
    const long long c = (long long)(r / (unsigned)T);
    const int dir = (int)(c & 1);
    const long long gt0 = c >> 1;
    if (gt0 * 32 >= a.n) return;
    const uint32_t xrow = (uint32_t)(a.Ipad >> 2) * 512u;
    const rsrc_t rw = make_rsrc((const char*)(dir ? a.wpk1 : a.wpk0) + (size_t)u * NQ * 4096, dir ? a.wpk1_end : a.wpk0_end, DSP_BND_W);
    const rsrc_t rx = make_rsrc((const char*)a.x + (size_t)gt0 * T * xrow, a.x_end, DSP_BND_X);
    const rsrc_t ro = make_rsrc((const char*)a.xacc + (size_t)c * T * UT * 16384, a.xacc_end, DSP_BND_XACC);
    const uint32_t xo = (uint32_t)t * xrow;
    const uint32_t voffA = voff + (uint32_t)g * 1024u;
    f32x4 A[R], B[R];
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < R; ++d) {
        const int q = d < xs ? d : 0;      // (a ring deeper than the x part: the spare slots hold k-group 0, unused)
        A[d] = bld16(rw, voffA, (uint32_t)q * 4096u);
        B[d] = bld16(rx, voff, xo + (uint32_t)q * 1024u);
    }
    for (int q0 = 0; q0 < xs; q0 += R) {
#pragma unroll
        for (int sl = 0; sl < R; ++sl) {
            const int q = q0 + sl;
            if (q < xs) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[sl][i], B[sl][i], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int qn = q + R < xs ? q + R : q;   // (past the end: the slot's own k-group again, unused)
                A[sl] = bld16(rw, voffA, (uint32_t)qn * 4096u);
                B[sl] = bld16(rx, voff, xo + (uint32_t)qn * 1024u);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const uint32_t so = ((uint32_t)t * (uint32_t)UT + (uint32_t)u) * 16384u;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        bst16(ro, voff + (uint32_t)(g * 4 + i) * 1024u, so, (f32x4{acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]}));
}

// ------------------------------------------------------------------------------------------------
// dsp_lstm6_kernel<NPROD> (opt-in, DSP_PRECISION=bf16x6 | bf16x9 | fp16x3): dsp_lstm_kernel<0, 1> with every fp32 product
// emulated on the bf16 matrix cores.  Both operands are split into three bf16 pieces (hi + mid + lo == x exactly
// for an fp32 x); a product keeps the NPROD largest piece products (9 = all of them, exact; 6 = without ml, lm,
// ll, i.e. about 2^-24 relative -- below the rounding of an fp32 accumulation), smallest first, accumulated in
// fp32 by v_mfma_f32_32x32x16_bf16 (8x the fp32 MFMA rate per piece product).  Weights come pre-split from the
// host ([unit tile][k-stage of 16][gate][piece][lane] 16 B); activations stay fp32 in the K4 layout -- nothing
// changes for the other kernels -- and are split in registers after the load.  Same wave/tile mapping, cell phase,
// h exchange and initial-state handling as dsp_lstm_kernel; one k-stage = 16 k = 4 K4 groups.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// 8 fp32 (two float4 = the lane's 8 consecutive k of a k-stage) -> hi / mid / lo as packed bf16x8
__device__ __forceinline__ void split_bf16x3(const f32x4 x0, const f32x4 x1, u32x4& hi, u32x4& mid, u32x4& lo) {
    const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));  // RNE
        const float ra = a - __builtin_bit_cast(float, hp << 16), rb = b - __builtin_bit_cast(float, hp & 0xffff0000u);
        const unsigned mp = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){ra, rb}, bf16x2_t));
        const float sa = ra - __builtin_bit_cast(float, mp << 16), sb = rb - __builtin_bit_cast(float, mp & 0xffff0000u);
        const unsigned lp = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){sa, sb}, bf16x2_t));
        hi[i] = hp; mid[i] = mp; lo[i] = lp;
    }
}

// 8 fp32 -> hi / lo as packed fp16x8 (NPROD == 3: two fp16 pieces, 11 + 11 mantissa bits, products lh + hl + hh:
// about 2^-22 relative; activations and weights must stay inside the fp16 range, which saturates visibly to inf)
__device__ __forceinline__ void split_f16x2(const f32x4 x0, const f32x4 x1, u32x4& hi, u32x4& lo) {
    const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const f16x2_t hp = __builtin_convertvector((f32x2_t){a, b}, f16x2_t);  // RNE
        const f32x2_t hf = __builtin_convertvector(hp, f32x2_t);
        const f16x2_t lp = __builtin_convertvector((f32x2_t){a - hf[0], b - hf[1]}, f16x2_t);
        hi[i] = __builtin_bit_cast(unsigned, hp);
        lo[i] = __builtin_bit_cast(unsigned, lp);
    }
}

template <int NPROD>
__global__ __launch_bounds__(512, 2) void dsp_lstm6_kernel(LstmArgs a) {
    constexpr bool F16 = NPROD == 3;     // two fp16 pieces instead of three bf16 pieces
    constexpr int NP = F16 ? 2 : 3;
    DSP_DYN_LDS(smem);
    f32x4* c_lds = (f32x4*)smem;           // [2 site tiles][4 groups][512 threads] float4
    f32x4* b_lds = c_lds + 8 * 512;        // [unit tile][aa][gate][half] float4
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t voff = (uint32_t)lane * 16u;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u = w % a.UT, sg = w / a.UT;
    const int dir = blockIdx.x & 1;
    const int grp = blockIdx.x >> 1;
    const int half = lane >> 5, ls = lane & 31;
    const int HQ = a.Hp >> 2;
    const int nqx = a.Ipad >> 4, NQ = a.NQ;  // k-stages of 16: x part, total
    const int T = a.T;
    const int F4 = a.Fout >> 2;
    const uint32_t xrow = (uint32_t)(a.Ipad >> 2) * 512u;
    const uint32_t orow = (uint32_t)F4 * 512u;
    // the lane's two K4 groups of a stage: 2*half and 2*half+1 (k = 8*half + j inside the 16-wide stage)
    const uint32_t xvoff = (uint32_t)half * 1024u + (uint32_t)ls * 16u;

    const long long gt0 = (long long)grp * (a.SG * 2) + sg * 2;
    const rsrc_t rw = make_rsrc((const char*)(dir ? a.wpk1 : a.wpk0) + (size_t)u * NQ * (4096 * NP), dir ? a.wpk1_end : a.wpk0_end, DSP_BND_W);
    const rsrc_t rx = make_rsrc((const char*)a.x + (size_t)gt0 * T * xrow, a.x_end, DSP_BND_X);
    const rsrc_t ro = make_rsrc((const char*)a.out + (size_t)gt0 * T * orow + (size_t)dir * HQ * 512, a.out_end, DSP_BND_OUT);
    const rsrc_t rh0 = make_rsrc((const char*)a.h0buf + (size_t)gt0 * orow + (size_t)dir * HQ * 512, a.h0buf_end, DSP_BND_H0);
    const f32x4* bias4 = (const f32x4*)(dir ? a.sbias1 : a.sbias0);

    for (int i = tid; i < a.Hp; i += blockDim.x) {
        const int h = i & 1, g = (i >> 1) & 3, aa = (i >> 3) & 3, ut = i >> 5;
        b_lds[i] = bias4[g * HQ + ut * 8 + 2 * aa + h];
    }
    const f32x4* b_my = b_lds + (size_t)u * 32 + half;  // + aa*8 + gate*2
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const long long site = (gt0 + m) * 32 + ls;
        const uint64_t skey = a.init_mode == 2 ? philox_site_key(a.site_keys, a.site_offset, a.n, site) : 0;
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) {
            const int k4 = u * 8 + 2 * aa + half;
            f32x4 hv = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f};
            if (a.init_mode != 0) {
                hv = init_state4(a.h0, a.init_mode, a.n, site, dir, a.H, k4, a.seed, skey,
                                 (uint32_t)(a.stream_base + dir * 2 + 0));
                cv = init_state4(a.c0, a.init_mode, a.n, site, dir, a.H, k4, a.seed, skey,
                                 (uint32_t)(a.stream_base + dir * 2 + 1));
            }
            bst16(rh0, voff + aa * 1024u, (uint32_t)m * orow + (uint32_t)u * 4096u, hv);
            c_lds[(m * 4 + aa) * 512 + tid] = cv;
        }
    }
    barrier_after_global_stores();  // h0 stored before any wave reads it back

    rsrc_t rhp = rh0;
    uint32_t xo[2], ho[2];
    auto set_bases = [&](int step) __attribute__((always_inline)) {
        const int t = dir ? (T - 1 - step) : step;
        const int tp = dir ? (t + 1) : (t - 1);
        rhp = step == 0 ? rh0 : ro;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            xo[m] = (uint32_t)(m * T + t) * xrow;
            ho[m] = step == 0 ? (uint32_t)m * orow : (uint32_t)(m * T + tp) * orow;
        }
    };

    u32x4 A[4][NP];
    f32x4 X[2][2];
    f32x16 acc[4][2];
    auto loadX = [&](int q) __attribute__((always_inline)) {
        const bool isx = q < nqx;
        const rsrc_t r = isx ? rx : rhp;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            // never form a negative intermediate: the buffer unit adds voffset + soffset + imm without wrapping at 32 bits
            const uint32_t so = isx ? xo[m] + (uint32_t)q * 2048u : ho[m] + (uint32_t)(q - nqx) * 2048u;
            X[m][0] = bld16(r, xvoff, so);
            X[m][1] = bld16(r, xvoff + 512u, so);
        }
    };
    auto loadA1 = [&](int g, int q) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
            A[g][p] = __builtin_bit_cast(u32x4, bld16(rw, voff, (uint32_t)((q * 4 + g) * NP + p) * 1024u));
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // piece products, smallest first.  bf16 (pieces h, m, l = 0, 1, 2): (ll, lm, ml,) mm, lh, hl, mh, hm, hh;
    // fp16 (pieces h, l = 0, 1): lh, hl, hh
    constexpr int pa9[9] = {2, 2, 1, 1, 2, 0, 1, 0, 0}, pb9[9] = {2, 1, 2, 1, 0, 2, 0, 1, 0};
    constexpr int pa3[3] = {1, 0, 0}, pb3[3] = {0, 1, 0};
    // one k-stage: split the activations loaded a stage ago, request the next ones, then per gate fragment the
    // NPROD x 2 MFMAs followed by the (late) refill of that fragment for the next stage
    auto stage = [&](int q, int qn, auto first) __attribute__((always_inline)) {
        u32x4 B[2][NP];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            if constexpr (F16) split_f16x2(X[m][0], X[m][1], B[m][0], B[m][1]);
            else split_bf16x3(X[m][0], X[m][1], B[m][0], B[m][1], B[m][NP - 1]);
        }
        loadX(qn);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int k = 0; k < NPROD; ++k) {
                const int ia = F16 ? pa3[k] : pa9[9 - NPROD + k], ib = F16 ? pb3[k] : pb9[9 - NPROD + k];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const bool z = decltype(first)::value && k == 0;
                    if constexpr (F16)
                        acc[g][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[g][ia]),
                                                                             __builtin_bit_cast(f16x8, B[m][ib]),
                                                                             z ? zero16 : acc[g][m], 0, 0, 0);
                    else
                        acc[g][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[g][ia]),
                                                                              __builtin_bit_cast(bf16x8, B[m][ib]),
                                                                              z ? zero16 : acc[g][m], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            loadA1(g, qn);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    set_bases(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) loadA1(g, 0);
    loadX(0);

    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        if (step > 0) barrier_after_global_stores();  // h_{t-1} of every wave stored before anyone reads it back
        stage(0, 1, std::true_type{});
        for (int q = 1; q < NQ - 1; ++q) stage(q, q + 1, std::false_type{});
        set_bases(step + 1 < T ? step + 1 : step);  // the activation request of the last stage belongs to the next step
        stage(NQ - 1, 0, std::false_type{});

        // LSTM cell (see dsp_lstm_kernel): pre-scaled biases folded into the exp2 arguments
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) {
            const f32x4 bi = b_my[aa * 8 + 0], bf = b_my[aa * 8 + 2], bg = b_my[aa * 8 + 4], bo = b_my[aa * 8 + 6];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                f32x4 cv = c_lds[(m * 4 + aa) * 512 + tid];
                f32x4 hv;
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
                c_lds[(m * 4 + aa) * 512 + tid] = cv;
                bst16(ro, voff + aa * 1024u, (uint32_t)(m * T + t) * orow + (uint32_t)u * 4096u, hv);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// linear_kernel: out[., out_off + o] = act( W[o,:] . x[., :] + b[o] ) on K4 activations, per (tile, t)
// column block of 32 sites.  Used for fc_seq / fc_signal (+ReLU; models.py:199-201, :215-217).
// ------------------------------------------------------------------------------------------------
// A wave computes 4 row tiles (128 output rows) x 2 column blocks (2 x 32 sites): 8 accumulator tiles, 32 MFMAs
// per k-group against 6 fragment loads (the first version -- one tile per wave, 4 MFMAs per 2 loads -- was bound
// by the L1 fill rate at 32 B/clk/CU and reached 88 TFLOP/s).  Operands come through the same buffer-load register
// rings as in dsp_lstm_kernel: the weights (L2-resident) two k-groups deep, the activations four -- they stream
// from HBM (the LSTM layer's [B,13,2H] output is 0.87 GB, beyond the 256 MiB Infinity Cache), and with the first
// version's one-deep prefetch every k-group waited out part of an HBM round trip (66 % of the MFMA peak).
__global__ __launch_bounds__(256, 2) void dsp_linear_kernel(LinArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t voff = (uint32_t)lane * 16u;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rt0 = blockIdx.y * 4;
    const int half = lane >> 5, ls = lane & 31;
    const bool second = blockIdx.x >= a.nbx;   // (workgroup-uniform: the second problem of a fused launch)
    const long long col0 = ((long long)(blockIdx.x - (second ? a.nbx : 0u)) * 4 + w) * 2;
    if (col0 >= a.ncols) return;
    const bool two = col0 + 1 < a.ncols;
    const int nq = a.Fin >> 3;   // a multiple of 8 (Fin = 2*Hp, Hp a multiple of 32)
    const int nrt = a.ORT - rt0 < 4 ? a.ORT - rt0 : 4;
    const uint32_t xrow = (uint32_t)(a.Fin >> 2) * 512u;  // bytes of one column block of the input
    const rsrc_t rw = make_rsrc((const char*)(second ? a.wpk2 : a.wpk) + (size_t)rt0 * nq * 1024, second ? a.wpk2_end : a.wpk_end, DSP_BND_W);
    const rsrc_t rx = make_rsrc((const char*)(second ? a.x2 : a.x) + (size_t)col0 * xrow, second ? a.x2_end : a.x_end, DSP_BND_X);
    const uint32_t wrow = (uint32_t)nq * 1024u;           // bytes between row tiles
    const uint32_t x1 = two ? xrow : 0u;                  // a lone last column block is computed twice, stored once
    const f32x4* bias4 = (const f32x4*)(second ? a.bias2 : a.bias);
    const int out_off = second ? a.out_off2 : a.out_off;
    f32x16 acc[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) {
            const f32x4 b = r < nrt ? bias4[(rt0 + r) * 8 + 2 * aa + half] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[r][0][4 * aa + i] = b[i]; acc[r][1][4 * aa + i] = b[i]; }
        }
    uint32_t ro[4];  // row tiles beyond ORT re-read tile 0 (results dropped)
#pragma unroll
    for (int r = 0; r < 4; ++r) ro[r] = (uint32_t)(r < nrt ? r : 0) * wrow;
    f32x4 A[2][4], B[4][2];
    auto ldA = [&](int r, int q) __attribute__((always_inline)) { return bld16(rw, voff, ro[r] + (uint32_t)q * 1024u); };
    auto ldB = [&](f32x4 (&Bs)[2], int q) __attribute__((always_inline)) {
        Bs[0] = bld16(rx, voff, (uint32_t)q * 1024u);
        Bs[1] = bld16(rx, voff, x1 + (uint32_t)q * 1024u);
    };
#define QC(x) ((x) < nq ? (x) : nq - 1)
// (a compiler-level memory barrier keeps the IR passes from moving the loads, sched_barrier the machine scheduler)
#define PIN_ORDER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    auto stage = [&](auto qs, int q) __attribute__((always_inline)) {
        constexpr int QS = decltype(qs)::value;
        constexpr int sa = QS % 2, sp = (QS + 1) % 2, sb = QS % 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[r][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[sa][r][i], B[sb][0][i], acc[r][0], 0, 0, 0);
                acc[r][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[sa][r][i], B[sb][1][i], acc[r][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (r == 0) A[sp][3] = ldA(3, QC(q + 1));
            else A[sa][r - 1] = ldA(r - 1, QC(q + 2));
            __builtin_amdgcn_sched_barrier(0);
        }
        ldB(B[sb], QC(q + 4));
        __builtin_amdgcn_sched_barrier(0);
    };
    // The ring fill in the order the stages use it, PINNED (round 4): hipcc otherwise moves fragments of the first k-group to
    // the end of the fill, and the wait it then needs at the head of the k-loop -- taken over all ways into the loop: vmcnt(2)
    // -- drained both rings once every four k-groups for the rest of the kernel (80.5 -> 86 % of the MFMA peak)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (d < 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { A[d][r] = ldA(r, d); PIN_ORDER(); }
        }
        ldB(B[d], d);
        PIN_ORDER();
    }
    for (int q = 0; q < nq; q += 4) {
        stage(ic<0>{}, q); stage(ic<1>{}, q + 1); stage(ic<2>{}, q + 2); stage(ic<3>{}, q + 3);
    }
#undef QC
#undef PIN_ORDER
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (c == 1 && !two) break;
        f32x4* out4 = (f32x4*)a.out + ((size_t)(col0 + c) * (a.Fout >> 2) + (out_off >> 2) + half) * 32 + ls;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r >= nrt) break;
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float y = acc[r][c][4 * aa + i];
                    v[i] = a.relu ? fmaxf(y, 0.f) : y;
                }
                BOUNDS_FLAT(&out4[(size_t)((rt0 + r) * 8 + aa * 2) * 32], 16, a.out, a.out_end, DSP_BND_FLAT_OUT);
                gst16(&out4[(size_t)((rt0 + r) * 8 + aa * 2) * 32], v);
            }
        }
    }
}

// Round 5, small batches: the same projection with ONE accumulator tile per wave (1 row tile x 1 column block).  The wave of
// dsp_linear_kernel above runs 32 k-groups x 32 MFMAs = 65 k cycles = 28 us whatever the batch; at 512 sites that is 26
// workgroups on 26 CUs and a fifth of the front-end phase.  Here: 4 MFMAs per k-group, 8 k cycles per wave, rings four
// k-groups deep for both operands (pure L2 latency otherwise), 4 x as many workgroups.  An accumulator tile still starts
// from its bias and sums its k-groups in order with the same MFMAs: bit-identical to dsp_linear_kernel.
__global__ __launch_bounds__(256, 2) void dsp_linear1_kernel(LinArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t voff = (uint32_t)lane * 16u;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, ls = lane & 31;
    const bool second = blockIdx.x >= a.nbx;   // (workgroup-uniform: the second problem of a fused launch)
    const long long col = (long long)(blockIdx.x - (second ? a.nbx : 0u));
    const int rt = blockIdx.y * 4 + w;          // a workgroup = 4 row tiles of one column block (they share the B rows in L1)
    if (col >= a.ncols || rt >= a.ORT) return;
    const int nq = a.Fin >> 3;   // a multiple of 8
    const uint32_t xrow = (uint32_t)(a.Fin >> 2) * 512u;
    const rsrc_t rw = make_rsrc((const char*)(second ? a.wpk2 : a.wpk) + (size_t)rt * nq * 1024, second ? a.wpk2_end : a.wpk_end, DSP_BND_W);
    const rsrc_t rx = make_rsrc((const char*)(second ? a.x2 : a.x) + (size_t)col * xrow, second ? a.x2_end : a.x_end, DSP_BND_X);
    const f32x4* bias4 = (const f32x4*)(second ? a.bias2 : a.bias);
    const int out_off = second ? a.out_off2 : a.out_off;
    f32x16 acc;
#pragma unroll
    for (int aa = 0; aa < 4; ++aa) {
        const f32x4 b = bias4[rt * 8 + 2 * aa + half];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * aa + i] = b[i];
    }
    f32x4 A[4], B[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        A[d] = bld16(rw, voff, (uint32_t)d * 1024u);
        B[d] = bld16(rx, voff, (uint32_t)d * 1024u);
    }
    auto stage = [&](auto qs, int q) __attribute__((always_inline)) {
        constexpr int S = decltype(qs)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[S][i], B[S][i], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int qn = q + 4 < nq ? q + 4 : nq - 1;
        A[S] = bld16(rw, voff, (uint32_t)qn * 1024u);
        B[S] = bld16(rx, voff, (uint32_t)qn * 1024u);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int q = 0; q < nq; q += 4) {
        stage(ic<0>{}, q); stage(ic<1>{}, q + 1); stage(ic<2>{}, q + 2); stage(ic<3>{}, q + 3);
    }
    f32x4* out4 = (f32x4*)a.out + ((size_t)col * (a.Fout >> 2) + (out_off >> 2) + half) * 32 + ls;
#pragma unroll
    for (int aa = 0; aa < 4; ++aa) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float y = acc[4 * aa + i];
            v[i] = a.relu ? fmaxf(y, 0.f) : y;
        }
        BOUNDS_FLAT(&out4[(size_t)(rt * 8 + aa * 2) * 32], 16, a.out, a.out_end, DSP_BND_FLAT_OUT);
        gst16(&out4[(size_t)(rt * 8 + aa * 2) * 32], v);
    }
}

// ------------------------------------------------------------------------------------------------
// head_kernel: [h_fwd(t=T-1) | h_bwd(t=0)] -> fc1 + ReLU -> fc2 -> softmax (+argmax)
// (models.py:229-240; dropouts are identity in eval).  One 4-wave workgroup per FOUR 32-site tiles.
// Round 3: a wave computes 2 row tiles (64 hidden rows of fc1) x 4 site tiles = 8 accumulator tiles, 32 MFMAs per
// k-group against 6 fragment loads (round 2: one site tile per workgroup, 8 MFMAs per 3 loads: every workgroup streamed
// the 512 KB of fc1 weights out of L2 for 32 sites -- 5.5 TB/s of L2 traffic, 57 % of the MFMA peak, and fc2 was a
// 256-iteration scalar loop on a quarter of the threads).  fc2 never leaves the registers: a lane holds 16 rows of each
// of its row tiles for one site, multiplies them by the matching fc2 weights, the two half-waves are added with one
// cross-lane move, the row-tile groups with a 4 KB LDS exchange; softmax on the first 128 threads.
// Hidden sizes above 256 (more than 8 row tiles) loop over pairs of row tiles per wave.
// ------------------------------------------------------------------------------------------------
// Round 4: kHeadST = site tiles per workgroup is a template parameter.  4 = full batches (above).  1 = batches of a few
// thousand sites: four times the workgroups (512 sites: 16 instead of 4), and the k-loop -- two MFMAs per fragment there,
// i.e. pure load latency -- runs on rings four k-groups deep instead of the one-deep prefetch (0.081 -> 0.03 ms per
// forward of 512 sites; same sums in the same order: bit-identical).
template <int kHeadST>
__global__ __launch_bounds__(256, 2) void dsp_head_kernel(HeadArgs a) {
    DSP_DYN_LDS(smem);
    float* part = smem;                         // [4 waves][C][kHeadST * 32]
    float* lg = smem + 4 * a.C * kHeadST * 32;  // [C][kHeadST * 32]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, ls = lane & 31;
    const size_t tile0 = (size_t)blockIdx.x * kHeadST;
    const int F4 = (2 * a.Hp) >> 2;
    const int nq = (2 * a.Hp) >> 3;
    const int nqf = a.Hp >> 3;  // k-groups that come from the forward half (t = T-1)
    const f32x4* bias4 = (const f32x4*)a.b1;
    const int NRT = a.Hp >> 5;
    // every operand through a buffer descriptor + SGPR offset + lane*16, like the LSTM kernel: no 64-bit pointer VGPRs
    const uint32_t voff = (uint32_t)lane * 16u;
    const uint32_t xrow = (uint32_t)F4 * 512u;                           // bytes of one (tile, t) block
    const rsrc_t rx = make_rsrc((const char*)a.x + tile0 * a.T * (size_t)xrow, a.x_end, DSP_BND_X);
    const rsrc_t rw = make_rsrc(a.w1pk, a.w1pk_end, DSP_BND_W);
    uint32_t xfo[kHeadST], xro[kHeadST];                                 // h_fwd at t = T-1, h_bwd at t = 0
#pragma unroll
    for (int st = 0; st < kHeadST; ++st) {
        xfo[st] = (uint32_t)(st * a.T + (a.T - 1)) * xrow;
        xro[st] = (uint32_t)(st * a.T) * xrow;
    }
    // per class partial sums need C x kHeadST registers: classes are processed in chunks of 2 (the reference has 2)
    for (int c0 = 0; c0 < a.C; c0 += 2) {
        float s0[kHeadST], s1[kHeadST];
#pragma unroll
        for (int st = 0; st < kHeadST; ++st) { s0[st] = 0.f; s1[st] = 0.f; }
        const bool has1 = c0 + 1 < a.C;
        for (int rt = w; rt < NRT; rt += 8) {  // row tiles rt and rt + 4 (when present) share the B fragments
            const bool two = rt + 4 < NRT;
            const uint32_t wo0 = (uint32_t)rt * (uint32_t)nq * 1024u;
            const uint32_t wo1 = two ? wo0 + 4u * (uint32_t)nq * 1024u : wo0;
            f32x16 acc0[kHeadST], acc1[kHeadST];
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                const f32x4 b0 = bias4[rt * 8 + 2 * aa + half];
                const f32x4 b1 = two ? bias4[(rt + 4) * 8 + 2 * aa + half] : b0;
#pragma unroll
                for (int st = 0; st < kHeadST; ++st)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { acc0[st][4 * aa + i] = b0[i]; acc1[st][4 * aa + i] = b1[i]; }
            }
            if constexpr (kHeadST == 1) {
                constexpr int DP = 4;   // ring depth; nq = Hp / 4 is a multiple of 8
                f32x4 A0r[DP], A1r[DP], Br[DP];
#pragma unroll
                for (int dq = 0; dq < DP; ++dq) {
                    A0r[dq] = bld16(rw, voff, wo0 + (uint32_t)dq * 1024u); A1r[dq] = bld16(rw, voff, wo1 + (uint32_t)dq * 1024u);
                    Br[dq] = bld16(rx, voff, ((dq < nqf) ? xfo[0] : xro[0]) + (uint32_t)dq * 1024u);
                }
                for (int q = 0; q < nq; q += DP) {
#pragma unroll
                    for (int dq = 0; dq < DP; ++dq) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            acc0[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0r[dq][i], Br[dq][i], acc0[0], 0, 0, 0);
                            acc1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1r[dq][i], Br[dq][i], acc1[0], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const int qn = q + dq + DP < nq ? q + dq + DP : nq - 1;
                        A0r[dq] = bld16(rw, voff, wo0 + (uint32_t)qn * 1024u); A1r[dq] = bld16(rw, voff, wo1 + (uint32_t)qn * 1024u);
                        Br[dq] = bld16(rx, voff, ((qn < nqf) ? xfo[0] : xro[0]) + (uint32_t)qn * 1024u);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
            f32x4 A0n = bld16(rw, voff, wo0), A1n = bld16(rw, voff, wo1), Bn[kHeadST];
#pragma unroll
            for (int st = 0; st < kHeadST; ++st) Bn[st] = bld16(rx, voff, xfo[st]);
            for (int q = 0; q < nq; ++q) {
                const f32x4 A0 = A0n, A1 = A1n;
                f32x4 B[kHeadST];
#pragma unroll
                for (int st = 0; st < kHeadST; ++st) B[st] = Bn[st];
                const int qn = q + 1 < nq ? q + 1 : q;
                A0n = bld16(rw, voff, wo0 + (uint32_t)qn * 1024u); A1n = bld16(rw, voff, wo1 + (uint32_t)qn * 1024u);
#pragma unroll
                for (int st = 0; st < kHeadST; ++st) Bn[st] = bld16(rx, voff, ((qn < nqf) ? xfo[st] : xro[st]) + (uint32_t)qn * 1024u);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int st = 0; st < kHeadST; ++st) {
                        acc0[st] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[i], B[st][i], acc0[st], 0, 0, 0);
                        acc1[st] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[i], B[st][i], acc1[st], 0, 0, 0);
                    }
            }
            }
            // fc2 on the registers: rows 8*aa + 4*half + i of row tile rt (and rt + 4) of this lane's site
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                const int r0 = rt * 32 + 8 * aa + 4 * half;
                const f32x4 u0 = *(const f32x4*)(a.w2 + (size_t)c0 * a.Hp + r0);
                const f32x4 u1 = has1 ? *(const f32x4*)(a.w2 + (size_t)(c0 + 1) * a.Hp + r0) : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
                if (two) {
                    v0 = *(const f32x4*)(a.w2 + (size_t)c0 * a.Hp + r0 + 128);
                    if (has1) v1 = *(const f32x4*)(a.w2 + (size_t)(c0 + 1) * a.Hp + r0 + 128);
                }
#pragma unroll
                for (int st = 0; st < kHeadST; ++st)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float h0 = fmaxf(acc0[st][4 * aa + i], 0.f);
                        const float h1 = two ? fmaxf(acc1[st][4 * aa + i], 0.f) : 0.f;
                        s0[st] = __builtin_fmaf(h0, u0[i], s0[st]);
                        s0[st] = __builtin_fmaf(h1, v0[i], s0[st]);
                        s1[st] = __builtin_fmaf(h0, u1[i], s1[st]);
                        s1[st] = __builtin_fmaf(h1, v1[i], s1[st]);
                    }
            }
        }
        // lanes l and l + 32 hold the two halves of a site's rows: add them, then the four waves through LDS
#pragma unroll
        for (int st = 0; st < kHeadST; ++st) {
            s0[st] += __shfl_xor(s0[st], 32);
            s1[st] += __shfl_xor(s1[st], 32);
            if (half == 0) {
                part[(w * a.C + c0) * (kHeadST * 32) + st * 32 + ls] = s0[st];
                if (has1) part[(w * a.C + c0 + 1) * (kHeadST * 32) + st * 32 + ls] = s1[st];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < a.C * kHeadST * 32; i += 256) {
        const int cc = i / (kHeadST * 32), col = i % (kHeadST * 32);
        float s = a.b2[cc];
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) s += part[(ww * a.C + cc) * (kHeadST * 32) + col];  // fixed order: deterministic
        lg[cc * (kHeadST * 32) + col] = s;
    }
    __syncthreads();
    if (tid < kHeadST * 32) {
        const long long site = (long long)tile0 * 32 + tid;
        if (site < a.n) {
            const int stride = kHeadST * 32;
            float mx = -INFINITY;
            for (int cc = 0; cc < a.C; ++cc) mx = fmaxf(mx, lg[cc * stride + tid]);
            float sum = 0.f;
            for (int cc = 0; cc < a.C; ++cc) sum += expf(lg[cc * stride + tid] - mx);
            const float inv = 1.0f / sum;
            int best = 0;
            float bp = -1.f;
            for (int cc = 0; cc < a.C; ++cc) {
                const float l = lg[cc * stride + tid];
                const float p = expf(l - mx) * inv;
                if (a.logits) a.logits[(size_t)site * a.C + cc] = l;
                if (a.probs) a.probs[(size_t)site * a.C + cc] = p;
                if (p > bp) { bp = p; best = cc; }  // first maximum, as torch.max(dim) (call_modifications.py:163)
            }
            if (a.labels) a.labels[site] = (uint8_t)best;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (called from dsp_capi.cpp; keep all <<<>>> syntax in this translation unit)
// ------------------------------------------------------------------------------------------------
// Which XCD does block b of a launch run on?  The clustered launches place the members of a cluster on consecutive entries of
// ONE XCD's dispatch list (block b on XCD b % 8: observed on MI355X in SPX mode, promised by nobody).  dsp_model_create asks once.
__global__ void dsp_xcc_probe_kernel(unsigned* out) {
    unsigned x;
    DSP_READ_XCC_ID(x);
    if (threadIdx.x == 0) out[blockIdx.x] = x & 0xf;
}
extern "C" int dsp_k_probe_xcc(unsigned* dev_out, int blocks, hipStream_t s) {
    hipLaunchKernelGGL(dsp_xcc_probe_kernel, dim3((unsigned)blocks), dim3(64), 0, s, dev_out);
    return (int)hipGetLastError();
}

// What does the hardware do with an access past a descriptor's num_records?  (dsp_debug_range_probe: the test that the range
// check the extents rely on is really there, and that the SGPR offset -- where these kernels carry almost all of an address --
// takes part in it.)  buf: 4 KiB of 1.0f, all mapped; the descriptor covers its first 256 bytes.
__global__ void dsp_range_probe_kernel(float* buf, unsigned* out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, (int)rsrc_records(buf, (const char*)buf + 256), 0x00020000);
    const uint32_t lane16 = threadIdx.x * 16u;   // one wave: lanes 0..63 -> voffset 0..1008
    const uint32_t s256 = __builtin_amdgcn_readfirstlane(256u), s512 = __builtin_amdgcn_readfirstlane(512u);
    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane16, 0, 0));        // lanes 0..15 in range
    const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane16, (int)s256, 0)); // all past the end by soffset
    // stores: every lane writes 2.0f at voffset + 512 (soffset): all out of range -> dropped
    const f32x4 two = {2.f, 2.f, 2.f, 2.f};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, two), r, (int)lane16, (int)s512, 0);
    store_data_guard(two);
    unsigned in_ok = 0, v_oob_zero = 0, s_oob_zero = 0;
    if (threadIdx.x < 16) in_ok = (a[0] == 1.f && a[1] == 1.f && a[2] == 1.f && a[3] == 1.f) ? 1u : 0u;
    else v_oob_zero = (a[0] == 0.f && a[1] == 0.f && a[2] == 0.f && a[3] == 0.f) ? 1u : 0u;
    s_oob_zero = (b[0] == 0.f && b[1] == 0.f && b[2] == 0.f && b[3] == 0.f) ? 1u : 0u;
    atomicAdd(&out[0], in_ok);        // 16: every in-range lane read its data
    atomicAdd(&out[1], v_oob_zero);   // 48: every lane past the end by VOFFSET read zeros
    atomicAdd(&out[2], s_oob_zero);   // 64: every lane past the end by SOFFSET read zeros
}
extern "C" int dsp_k_range_probe(float* dev_buf4k, unsigned* dev_out, hipStream_t s) {
    hipLaunchKernelGGL(dsp_range_probe_kernel, dim3(1), dim3(64), 0, s, dev_buf4k, dev_out);
    return (int)hipGetLastError();
}

extern "C" int dsp_k_init(void) {
    const void* fns[] = {(const void*)dsp_lstm_kernel<0, 1>, (const void*)dsp_lstm_kernel<1, 1>,
                         (const void*)dsp_lstm_kernel<2, 1, 1>, (const void*)dsp_lstm_kernel<2, 1, 2>, (const void*)dsp_lstm_kernel<2, 1, 3>,
                         (const void*)dsp_lstm_kernel<0, 0>, (const void*)dsp_lstm_kernel<1, 0>, (const void*)dsp_lstm21_kernel,
                         (const void*)dsp_lstm6_kernel<6>, (const void*)dsp_lstm6_kernel<9>, (const void*)dsp_lstm6_kernel<3>,
                         (const void*)dsp_lstmc_kernel<4, 4>, (const void*)dsp_lstmc_kernel<2, 8>, (const void*)dsp_lstmc_kernel<1, 16>,
                         (const void*)dsp_lstmc_kernel<4, 4, true>, (const void*)dsp_lstmc_kernel<4, 4, true, 2>,
                         (const void*)dsp_lstmc_kernel<4, 4, true, 3>, (const void*)dsp_lstmc_kernel<4, 4, true, 0, 8>,
                         (const void*)dsp_lstmc_kernel<1, 4, false, 0, 4, true>, (const void*)dsp_lstmc_kernel<1, 4, false, 2, 4, true>,
                         (const void*)dsp_lstmc_kernel<1, 4, false, 3, 4, true>, (const void*)dsp_lstmc_kernel<2, 4, false, 0, 4, true>,
                         (const void*)dsp_lstmc_kernel<2, 4, false, 2, 4, true>, (const void*)dsp_lstmc_kernel<2, 4, false, 3, 4, true>,
                         (const void*)dsp_lstmc_kernel<1, 4>, (const void*)dsp_lstmc_kernel<2, 4>};
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
    }
    const hipError_t e1 = hipFuncSetAttribute((const void*)dsp_head_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e1 != hipSuccess) return (int)e1;
    return (int)hipFuncSetAttribute((const void*)dsp_head_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
}

// Dry runs (round 6; dsp_debug_dry_run, no device): the wrappers below run every check they make and work out every launch's
// geometry, then NOTE the launch -- kernel, grid, block, dynamic LDS -- instead of making it.  The host half of a forward (which
// kernel form a batch size takes, the extents behind every descriptor, the cut into pieces) is thereby testable on a machine
// without a GPU: tests/test_dry_run.py.  Off unless dsp_k_set_dry(1) was called on this thread.
static thread_local int g_dry = 0;
static thread_local std::string* g_dry_log = nullptr;
extern "C" void dsp_k_set_dry(int on, void* log_string) { g_dry = on; g_dry_log = (std::string*)log_string; }
static bool dry_note(const char* kern, dim3 grid, dim3 block, size_t lds) {
    if (!g_dry) return false;
    if (g_dry_log) {
        char buf[256];
        snprintf(buf, sizeof buf, "%s grid %u,%u block %u lds %zu\n", kern, grid.x, grid.y, block.x, lds);
        *g_dry_log += buf;
    }
    return true;
}
#define DSP_LAUNCH(kern, grid, block, lds, stream, ...)                               \
    do {                                                                              \
        if (dry_note(#kern, grid, block, lds)) break;                                 \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);              \
    } while (0)
#define DSP_LAUNCH_RESULT() (g_dry ? 0 : (int)hipGetLastError())

// every pointer that is addressed through a buffer descriptor comes with the end of its allocation: a NULL or inverted end
// would make a descriptor of zero records (loads of zeros, dropped stores -- silently wrong results), so it is refused here
static bool ends_ok(std::initializer_list<std::pair<const void*, const void*>> v) {
    for (const auto& pe : v)
        if (pe.first && (!pe.second || (const char*)pe.second <= (const char*)pe.first)) return false;
    return true;
}
static bool lstm_ends_ok(const LstmArgs* a) {
    return a->x && a->out && a->wpk0 && a->wpk1 && a->h0buf &&
           ends_ok({{a->x, a->x_end}, {a->out, a->out_end}, {a->wpk0, a->wpk0_end}, {a->wpk1, a->wpk1_end}, {a->h0buf, a->h0buf_end},
                    {a->cflags, a->cflags_end}}) &&
           (!a->cbuf || (a->cbuf_end && (const char*)a->cbuf_end >= (const char*)a->cbuf));   // (an empty cell-state scratch is legal)
}

extern "C" int dsp_k_bounds_build(void) {
#ifdef DSP_BOUNDS
    return 1;
#else
    return 0;
#endif
}
extern "C" int dsp_k_bounds_read(unsigned rec[8]) {
#ifdef DSP_BOUNDS
    hipError_t e = hipMemcpyFromSymbol(rec, HIP_SYMBOL(g_bounds), sizeof(unsigned) * 8);
    if (e != hipSuccess) return (int)e;
    if (rec[0]) {
        const unsigned zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_bounds), zero, sizeof zero);
    }
    return (int)e;
#else
    for (int i = 0; i < 8; ++i) rec[i] = 0;
    return 0;
#endif
}

extern "C" int dsp_k_pack(const PackArgs* a, hipStream_t s) {
    const long long threads = (long long)a->NTp * a->T * 32;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    DSP_LAUNCH(dsp_pack_kernel, dim3(blocks), dim3(256), 0, s, *a);
    return DSP_LAUNCH_RESULT();
}

// a wave owns one unit tile (per pass) x two site tiles; a->SG site groups (of two tiles) per workgroup; a->NP passes
extern "C" int dsp_k_lstm(const LstmArgs* a, hipStream_t s) {
    if (!lstm_ends_ok(a)) return (int)hipErrorInvalidValue;
    if (a->CG > 0) {
        // a (site tile, direction) spread over a cluster of UT / CG workgroups (dsp_lstmc_kernel); the caller has checked the
        // residency.  CG = 4 with 4 unit tiles: the workgroup holds the whole layer (the front ends at hidden 128), no counters
        const int G = a->CG;
        const int nqx = a->Ipad >> 3;
        const bool local8 = G == 4 && a->UT == 8 && (a->flags & 8);   // eight waves hold the whole layer of 8 unit tiles
        const bool local = (G == 4 && a->UT == 4) || local8;
        // a front end (4 unit tiles, an x part of exactly four k-groups) spread over 4 / G workgroups: rings four deep
        const bool xshort = !local && (a->UT == 4 || a->UT == 8) && nqx == 4 && (G == 1 || G == 2);   // (8 unit tiles: the seq front end of
        // a seq-only model at hidden 193..256 -- BASELINE configs[2]'s shape)
        // dense layers of 4 unit tiles (hidden 97..128: the combined stack of a hid_rnn-128 model) clustered like those of 8:
        // P = 4 / G workgroups, rings four deep
        const bool dense4 = !local && !xshort && a->UT == 4 && (G == 1 || G == 2);
        // (flags bit 7, x ahead at one gate per wave only: rings 8 deep instead of 16 -- the launch keeps 8 k-groups of x part)
        const bool ring8 = a->xs > 0 && G == 1 && (a->flags & 128);
        const int D = (xshort || dense4) ? 4 : (G == 4 ? 4 : (G == 2 || ring8 ? 8 : 16));
        if ((G != 1 && G != 2 && G != 4) || a->UT % G || a->NP > 1 || (!local && !a->cflags) || a->NQ != ((a->Ipad + a->Hp) >> 3) ||
            nqx % D || nqx < ((local || xshort) ? D : 2 * D) || a->NQ % D || a->NQ < 2 * D || (!local && !xshort && !dense4 && a->UT != 8))
            return (int)hipErrorInvalidValue;
        // x ahead (a->xs > 0): the clustered dense forms (8 unit tiles, or 4: hidden 97..128); this launch keeps exactly one ring of x part
        const bool xa = a->xs > 0;
        if (xa && (local || xshort || a->xs != nqx - D || !a->xacc || !ends_ok({{a->xacc, a->xacc_end}}) || a->nqx_lo != 0 ||
                   a->nqx_used != nqx))
            return (int)hipErrorInvalidValue;
        // (zero-padded x-part k-groups -- nqx_lo, nqx_used -- are computed like live ones here: their weights are zero)
        const int P = local ? 1 : a->UT / G;
        const unsigned clusters = (unsigned)(a->NTp * 2);
        const unsigned grid = (clusters + 7) / 8 * 8 * (unsigned)P;
        size_t lds = (size_t)a->Hp * 16 + (G < 4 ? (size_t)G * 16384 : 0);
        if (a->flags & 4) lds = lds < 84 * 1024 ? 84 * 1024 : lds;   // more than half a CU's LDS: one workgroup per CU
        // the front ends' zero-padded leading k-groups (features at the end of the 32-wide block) issue no MFMAs
        const int dead = ((local || xshort) && nqx == 4 && a->nqx_used == 4 && (a->nqx_lo == 2 || a->nqx_lo == 3)) ? a->nqx_lo : 0;
        if (local8) DSP_LAUNCH((dsp_lstmc_kernel<4, 4, true, 0, 8>), dim3(grid), dim3(512), lds, s, *a);
        else if (local && dead == 3) DSP_LAUNCH((dsp_lstmc_kernel<4, 4, true, 3>), dim3(grid), dim3(256), lds, s, *a);
        else if (local && dead == 2) DSP_LAUNCH((dsp_lstmc_kernel<4, 4, true, 2>), dim3(grid), dim3(256), lds, s, *a);
        else if (local) DSP_LAUNCH((dsp_lstmc_kernel<4, 4, true>), dim3(grid), dim3(256), lds, s, *a);
        else if (xshort && G == 1 && dead == 3) DSP_LAUNCH((dsp_lstmc_kernel<1, 4, false, 3, 4, true>), dim3(grid), dim3(256), lds, s, *a);
        else if (xshort && G == 1 && dead == 2) DSP_LAUNCH((dsp_lstmc_kernel<1, 4, false, 2, 4, true>), dim3(grid), dim3(256), lds, s, *a);
        else if (xshort && G == 1) DSP_LAUNCH((dsp_lstmc_kernel<1, 4, false, 0, 4, true>), dim3(grid), dim3(256), lds, s, *a);
        else if (xshort && dead == 3) DSP_LAUNCH((dsp_lstmc_kernel<2, 4, false, 3, 4, true>), dim3(grid), dim3(256), lds, s, *a);
        else if (xshort && dead == 2) DSP_LAUNCH((dsp_lstmc_kernel<2, 4, false, 2, 4, true>), dim3(grid), dim3(256), lds, s, *a);
        else if (xshort) DSP_LAUNCH((dsp_lstmc_kernel<2, 4, false, 0, 4, true>), dim3(grid), dim3(256), lds, s, *a);
        else if (xa) {
            // the x part ahead of the recurrence: live clusters x T x UT workgroups (clusters are numbered tile-major: the live
            // ones come first)
            const unsigned live = (unsigned)((a->n + 31) / 32) * 2;
            DSP_LAUNCH(dsp_xahead_kernel, dim3(live * (unsigned)a->T * (unsigned)a->UT), dim3(256), 0, s, *a);
            if (dense4 && G == 1) DSP_LAUNCH((dsp_lstmc_kernel<1, 4, false, 0, 4, false, true>), dim3(grid), dim3(256), lds, s, *a);
            else if (dense4) DSP_LAUNCH((dsp_lstmc_kernel<2, 4, false, 0, 4, false, true>), dim3(grid), dim3(256), lds, s, *a);
            else if (G == 4) DSP_LAUNCH((dsp_lstmc_kernel<4, 4, false, 0, 4, false, true>), dim3(grid), dim3(256), lds, s, *a);
            else if (G == 2) DSP_LAUNCH((dsp_lstmc_kernel<2, 8, false, 0, 4, false, true>), dim3(grid), dim3(256), lds, s, *a);
            else if (ring8) DSP_LAUNCH((dsp_lstmc_kernel<1, 8, false, 0, 4, false, true>), dim3(grid), dim3(256), lds, s, *a);
            else DSP_LAUNCH((dsp_lstmc_kernel<1, 16, false, 0, 4, false, true>), dim3(grid), dim3(256), lds, s, *a);
        }
        else if (dense4 && G == 1) DSP_LAUNCH((dsp_lstmc_kernel<1, 4>), dim3(grid), dim3(256), lds, s, *a);
        else if (dense4) DSP_LAUNCH((dsp_lstmc_kernel<2, 4>), dim3(grid), dim3(256), lds, s, *a);
        else if (G == 4) DSP_LAUNCH((dsp_lstmc_kernel<4, 4>), dim3(grid), dim3(256), lds, s, *a);
        else if (G == 2) DSP_LAUNCH((dsp_lstmc_kernel<2, 8>), dim3(grid), dim3(256), lds, s, *a);
        else DSP_LAUNCH((dsp_lstmc_kernel<1, 16>), dim3(grid), dim3(256), lds, s, *a);
        if (!local) {
            // the clean-up launch: clusters whose members did not all become resident in time were abandoned by them and are
            // computed here, one workgroup each (eight waves for 8 unit tiles, four for 4), nothing waiting on another
            // workgroup (a few microseconds when none was)
            LstmArgs b = *a;
            b.CG = 4;
            b.xs = 0;   // (an abandoned cluster is computed whole, x part included)
            const unsigned g1 = (clusters + 7) / 8 * 8;
            if (a->UT == 8) {
                b.flags = (a->flags | 8 | 16) & ~4;
                DSP_LAUNCH((dsp_lstmc_kernel<4, 4, true, 0, 8>), dim3(g1), dim3(512), (size_t)a->Hp * 16, s, b);
            } else {
                b.flags = (a->flags | 16) & ~(4 | 8);
                if (dead == 3) DSP_LAUNCH((dsp_lstmc_kernel<4, 4, true, 3>), dim3(g1), dim3(256), (size_t)a->Hp * 16, s, b);
                else if (dead == 2) DSP_LAUNCH((dsp_lstmc_kernel<4, 4, true, 2>), dim3(g1), dim3(256), (size_t)a->Hp * 16, s, b);
                else DSP_LAUNCH((dsp_lstmc_kernel<4, 4, true>), dim3(g1), dim3(256), (size_t)a->Hp * 16, s, b);
            }
        }
        return DSP_LAUNCH_RESULT();
    }
    if ((a->flags & 2) && a->NP <= 1 && a->UT >= 2 && a->UT % 2 == 0 && a->nqx_lo == 0 && a->nqx_used == (a->Ipad >> 3) &&
        a->NQ == ((a->Ipad + a->Hp) >> 3) && (a->Ipad >> 3) >= 4) {
        // <2 unit tiles, 1 site tile> per wave: UT/2 waves x SG single-tile site groups per workgroup
        const int sg = a->UT >= 8 ? 1 : 8 / a->UT;
        LstmArgs b = *a;
        b.SG = sg;
        const int threads = (a->UT / 2) * sg * 64;
        size_t lds = (size_t)8 * threads * 16 + (size_t)a->Hp * 16;
        if (a->flags & 4) lds = lds < 84 * 1024 ? 84 * 1024 : lds;   // more than half a CU's LDS: one workgroup per CU
        DSP_LAUNCH(dsp_lstm21_kernel, dim3((unsigned)(a->NTp / sg) * 2), dim3(threads), lds, s, b);
        return DSP_LAUNCH_RESULT();
    }
    const int np = a->NP < 1 ? 1 : a->NP;   // 1, 2: the cell state in LDS; more: in a->cbuf (dsp_lstm_kernel<., 0>)
    const int threads = (a->UT / np) * a->SG * 64;
    const unsigned groups = (unsigned)(a->NTp / (a->SG * 2));
    // Several passes (hidden > 256): dsp_lstm_kernel<., 0>, the pass loop not unrolled and the cell state in a->cbuf.  (Until
    // round 3 two passes had their own unrolled instantiation with the cell state in 128 KiB of LDS: 21 spilled registers,
    // the same speed -- 35.62 vs 35.63 ms per launch at hidden 512 -- and one more kernel to keep correct: removed.)
    const bool many = np >= 2;
    const size_t lds = (!many ? (size_t)np * 8 * threads * 16 : 0) + (size_t)a->Hp * 16;
    if (many && !a->cbuf) return (int)hipErrorInvalidValue;
    const int nqx = a->Ipad >> 3, nq = (a->Ipad + a->Hp) >> 3;
    if (nqx < 4 || threads > 512 || a->UT % np) return (int)hipErrorInvalidValue;  // see the SPARSE note: four x-part k-groups are required
    // 0: no padded k-groups; 2: the front-end shape (dead k-groups first, inside the first four); 1: padding anywhere else
    const bool padded = a->nqx_lo > 0 || a->nqx_used < nqx || a->NQ > nq;
    const int sparse = !padded ? 0 : (np == 1 && nqx == 4 && a->NQ == nq && a->nqx_used == 4 && a->nqx_lo > 0) ? 2 : 1;
    if (sparse == 1 && a->nqx_lo != 0) return (int)hipErrorInvalidValue;  // the tested k-loop starts on a live k-group
    const dim3 g(groups * 2), b(threads);
    if (sparse == 2) {
        if (a->nqx_lo == 3) DSP_LAUNCH((dsp_lstm_kernel<2, 1, 1>), g, b, lds, s, *a);
        else if (a->nqx_lo == 2) DSP_LAUNCH((dsp_lstm_kernel<2, 1, 2>), g, b, lds, s, *a);
        else DSP_LAUNCH((dsp_lstm_kernel<2, 1, 3>), g, b, lds, s, *a);
    } else if (many) {
        if (sparse) DSP_LAUNCH((dsp_lstm_kernel<1, 0>), g, b, lds, s, *a);
        else DSP_LAUNCH((dsp_lstm_kernel<0, 0>), g, b, lds, s, *a);
    } else {
        if (sparse) DSP_LAUNCH((dsp_lstm_kernel<1, 1>), g, b, lds, s, *a);
        else DSP_LAUNCH((dsp_lstm_kernel<0, 1>), g, b, lds, s, *a);
    }
    return DSP_LAUNCH_RESULT();
}

// split variants: a->wpk0/1 = split weights, a->NQ = k-stages of 16, nprod = 6 / 9 (bf16 pieces) or 3 (fp16 pieces)
extern "C" int dsp_k_lstm6(const LstmArgs* a, int nprod, hipStream_t s) {
    if (!lstm_ends_ok(a)) return (int)hipErrorInvalidValue;
    const int waves = a->UT * a->SG;
    const unsigned groups = (unsigned)(a->NTp / (a->SG * 2));
    const size_t lds = (size_t)8 * 512 * 16 + (size_t)a->Hp * 16;
    if (nprod == 9) DSP_LAUNCH(dsp_lstm6_kernel<9>, dim3(groups * 2), dim3(waves * 64), lds, s, *a);
    else if (nprod == 3) DSP_LAUNCH(dsp_lstm6_kernel<3>, dim3(groups * 2), dim3(waves * 64), lds, s, *a);
    else DSP_LAUNCH(dsp_lstm6_kernel<6>, dim3(groups * 2), dim3(waves * 64), lds, s, *a);
    return DSP_LAUNCH_RESULT();
}

extern "C" int dsp_k_linear(const LinArgs* a, hipStream_t s) {
    if (!a->x || !a->wpk || !a->out || (a->x2 && !a->wpk2) ||
        !ends_ok({{a->x, a->x_end}, {a->wpk, a->wpk_end}, {a->x2, a->x2_end}, {a->wpk2, a->wpk2_end}, {a->out, a->out_end}}))
        return (int)hipErrorInvalidValue;
    if (a->small) {   // batches that leave CUs idle: one accumulator tile per wave (dsp_linear1_kernel)
        LinArgs b = *a;
        b.nbx = (unsigned)a->ncols;
        DSP_LAUNCH(dsp_linear1_kernel, dim3(a->x2 ? 2 * b.nbx : b.nbx, (unsigned)((a->ORT + 3) / 4)), dim3(256), 0, s, b);
        return DSP_LAUNCH_RESULT();
    }
    const unsigned bx = (unsigned)((a->ncols + 7) / 8);  // 4 waves x 2 column blocks per workgroup
    LinArgs b = *a;
    b.nbx = bx;
    DSP_LAUNCH(dsp_linear_kernel, dim3(a->x2 ? 2 * bx : bx, (unsigned)((a->ORT + 3) / 4)), dim3(256), 0, s, b);
    return DSP_LAUNCH_RESULT();
}

extern "C" int dsp_k_head(const HeadArgs* a, hipStream_t s) {
    if (!a->x || !a->w1pk || !ends_ok({{a->x, a->x_end}, {a->w1pk, a->w1pk_end}})) return (int)hipErrorInvalidValue;
    // batches of a few thousand sites: one site tile per workgroup (four times the workgroups, deeper operand rings)
    const bool small = a->n <= 4096 && !(a->flags & 1);
    const int st = small ? 1 : 4;
    const size_t lds = (size_t)(5 * a->C * st * 32) * sizeof(float);
    const unsigned groups = (unsigned)((a->n + 32 * st - 1) / (32 * st));  // the K4 buffers are padded to 16 tiles
    if (small) DSP_LAUNCH(dsp_head_kernel<1>, dim3(groups), dim3(256), lds, s, *a);
    else DSP_LAUNCH(dsp_head_kernel<4>, dim3(groups), dim3(256), lds, s, *a);
    return DSP_LAUNCH_RESULT();
}
