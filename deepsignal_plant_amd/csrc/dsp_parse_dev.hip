// dsp_parse_dev.hip -- the feature-row parser on the GPU (SURVEY.md 8(f)-2 "SIMD float parsing", MI355X-first).
//
// Replaces, for plain rows, the per-row work of _read_features_file (deepsignal_plant/call_modifications.py:76-86: split
// on tabs, base2code_dna per k-mer letter, float() per number) -- and this build's own host parser threads
// (csrc/dsp_text.cpp: 1.1 GB/s of text per thread, i.e. 2.3 host threads per GPU at the forward's rate; eight GPUs wanted
// 32 cores).  The host now only copies the block into a page-locked buffer and notes the row starts in the same pass
// (dsp_copy_rows_index); the text crosses PCIe as it is (2.08 kB per row: 2.6 GB/s per GPU) and ONE thread per row walks it
// here.
//
// Same values as the host parser, by construction: this is the plain-row path of csrc/dsp_text.cpp (parse_row_fast) --
// digits accumulated into an integer mantissa, value = mantissa * or / an exact power of ten in float64 (one correctly
// rounded IEEE operation: v_mul_f64 / the compiler's IEEE division), then the float64 -> float32 conversion the
// reference's FloatTensor does.  Whatever that path does not accept (blanks, '+', inf / nan, mantissas beyond 18 digits or
// 2^53, exponents beyond +-22, another field count, a letter outside base2code_dna, ...) is not guessed at: the row is
// flagged and the caller gives the whole block to the host parser, which also owns the error messages.
//
// Three versions (the third runs; the second is kept for rows too long for its LDS).  A row is about 2.1 kB in 13 + 13 + 13
// + 208 numbers.  The first version -- one thread per row walking all of it -- took
// 0.67 ms per block of 32,768 rows (102 GB/s): 512 waves for 1,024 SIMDs, and a cursor whose 16-byte refills happen at a
// different step in every lane, so that the wave waited out a memory round trip at nearly every byte (vmcnt counts
// instructions, not lanes: reading the prefetch register waits for the refill another lane issued a step ago).  Now two
// kernels: (1) one thread per row walks it UNIFORMLY (every lane loads its next word in the same iteration, four ahead)
// and only looks for the delimiters -- tabs and the ';' of the signals field --, parses the short pieces (k-mer, lengths,
// label) and writes the row's segment table; (2) one thread per (row, float list) -- 15 lists per row, 7.7 waves per SIMD:
// the occupancy hides part of the divergent refills: 0.30 ms per block (0.13 + 0.17), 1.2 % of the block's forward.  (Measured
// and not kept: requesting a list's lines up front so that the refills hit the cache -- 0.185 instead of 0.169 ms, the
// kernels are bound by their ~90 instructions per byte, not by the round trips; an eight-deep instead of a four-deep
// prefetch in the scan: no change.)  Third version (dsp_parse_tokens_kernel below): a workgroup stages 4 rows in LDS with
// coalesced loads, numbers their delimiters with one scan and parses ONE TOKEN PER THREAD, so that neighbouring lanes read
// neighbouring bytes and store neighbouring words: 0.17 ms per block = 400 GB/s of text, 0.65 % of the block's forward
// (4 / 8 / 16 / 2 / 1 rows per workgroup: 0.171 / 0.174-0.199 / 0.243 / 0.203 / 0.242 ms; bound by the ~250 instructions a
// number costs -- every one with a decimal point is an IEEE float64 division).  Integer / byte work: no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "dsp_amd.h"
#include "dsp_parse_arith.h"   // fast_float / fast_int / base_code / is_space / eq_mask4 / delim_mask4: shared with the host sanitizer test

// (the one spot of HIP's LDS declaration syntax, as a macro: the test-suite's SIMT interpreter -- tests/native/emu -- gives it its own
// meaning when it runs these kernels on the host under AddressSanitizer; the product build sees exactly the declaration it names)
#ifndef DSP_EMU
#define DSP_DYN_LDS_T(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#endif

namespace {

using namespace dsp_parse_arith;

// a byte cursor over global memory: 16-byte words, the next one requested while the current one is consumed.  The pointers
// carry the GLOBAL address space: through a generic `const char*` hipcc emits flat_load, which counts on vmcnt AND lgkmcnt
// and made every refill wait out its own request at once (checked in the ISA)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const u32x4 guint4;
struct Reader {
    guint4* next;          // address of the word after `ahead`
    u32x4 ahead;           // the word behind the current one (already loaded)
    uint64_t lo, hi;       // the unconsumed bytes of the current word, lowest byte first in `lo`
    int left;              // bytes left in lo | hi
    uint32_t pos;          // offset of the cursor from the row start
    __device__ __forceinline__ void init(const char* p) {
        const uintptr_t a = (uintptr_t)p;
        guint4* w = (guint4*)(a & ~(uintptr_t)15);
        const int skip = (int)(a & 15);
        const u32x4 cur = w[0];
        ahead = w[1];
        next = w + 2;
        lo = (uint64_t)cur.x | ((uint64_t)cur.y << 32);
        hi = (uint64_t)cur.z | ((uint64_t)cur.w << 32);
        left = 16;
        pos = 0;
        for (int i = 0; i < skip; ++i) shift();   // (at most 15 steps, once per row)
        pos = 0;
    }
    __device__ __forceinline__ void shift() {
        lo = (lo >> 8) | (hi << 56);
        hi >>= 8;
        if (--left == 0) {
            lo = (uint64_t)ahead.x | ((uint64_t)ahead.y << 32);
            hi = (uint64_t)ahead.z | ((uint64_t)ahead.w << 32);
            ahead = *next++;
            left = 16;
        }
    }
    __device__ __forceinline__ unsigned cur() const { return (unsigned)(lo & 0xffu); }
    __device__ __forceinline__ void adv() { shift(); ++pos; }
};

struct ParseArgs {
    const char* text; const uint64_t* row_off; long long n; int L, S;
    uint8_t* kmer; float* means; float* stds; int* lens; float* signals; int* labels;
    uint32_t* info_len; uint32_t* read_off; uint32_t* read_len; uint8_t* status; uint32_t* n_flagged;
    unsigned long long text_bytes;   // bytes staged at `text` (offsets beyond them are never followed)
    uint32_t* seg;        // [n][L + 4]: row-relative starts of means, stds, signal groups 0 .. L-1; one past the '\t' behind the
                          // signals field; one past the '\t' behind stds
};

__device__ __forceinline__ void flag_row(const ParseArgs& a, long long r) {
    a.status[r] = 1;                 // (several threads of a row may say so: same value)
    atomicAdd(a.n_flagged, 1u);      // only zero / non-zero is used
}

// ---- kernel 1: one thread per row.  A UNIFORM walk over the row in 16-byte words (every lane loads its next word in the
// same iteration, four iterations ahead: no lane-divergent refills) that only looks for delimiters: the first 11 tabs and
// the ';' of the signals field.  Then the short pieces -- k-mer, the L lengths, the label, the sampleinfo addressing -- and
// the segment table of the float lists for kernel 2.
__global__ __launch_bounds__(64) void dsp_parse_scan_kernel(ParseArgs a) {
    const long long r = (long long)blockIdx.x * 64 + threadIdx.x;
    if (r >= a.n) return;
    a.status[r] = 0;
    const uint64_t o0 = a.row_off[r], o1 = a.row_off[r + 1];
    if (o1 <= o0 || o1 > a.text_bytes || o1 - o0 > (1u << 28)) { flag_row(a, r); return; }   // not a row of this text: nothing is read
    const uint32_t len = (uint32_t)(o1 - o0 - 1);              // without the '\n'
    const int L = a.L;
    const int NSEG = 2 + L;
    uint32_t* seg = a.seg + (size_t)r * (NSEG + 2);
    bool ok = len >= (uint32_t)(12 + L);
    const char* row = a.text + o0;
    // delimiters: tabs 0..10, and the ';' between tab 9 and tab 10 (or the row end)
    uint32_t tab[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) tab[k] = 0xffffffffu;
    int ntab = 0, nsemi = 0;
    const uintptr_t base = (uintptr_t)row & ~(uintptr_t)15;
    const int skip = (int)((uintptr_t)row & 15);
    guint4* wp = (guint4*)base;
    const uint32_t nwords = ok ? (len + (uint32_t)skip + 15u) / 16u : 0u;
    // eight words = one 128-byte line ahead (the staged text has 64 readable bytes behind its end: the first loads are
    // clamped to the row's words as well)
    auto wclamp = [&](uint32_t i) __attribute__((always_inline)) { return wp[i < nwords ? i : (nwords ? nwords - 1 : 0)]; };
    u32x4 q0 = wclamp(0), q1 = wclamp(1), q2 = wclamp(2), q3 = wclamp(3), q4 = wclamp(4), q5 = wclamp(5), q6 = wclamp(6), q7 = wclamp(7);
    for (uint32_t wi = 0; wi < nwords; ++wi) {
        const u32x4 w = q0;
        q0 = q1; q1 = q2; q2 = q3; q3 = q4; q4 = q5; q5 = q6; q6 = q7;
        q7 = wclamp(wi + 8);                                    // uniform prefetch, clamped inside the row's words
        const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t mt = eq_mask4(ws[j], 0x09090909u);
            uint32_t ms = eq_mask4(ws[j], 0x3b3b3b3bu);
            uint32_t m = mt | ms;
            while (m) {                                         // rare: ~23 delimiters in 2.1 kB
                const int bit = __builtin_ctz(m);
                const int byte = bit >> 3;
                const int32_t pos = (int32_t)(wi * 16u + (uint32_t)j * 4u + (uint32_t)byte) - skip;
                m &= m - 1;
                if (pos < 0 || (uint32_t)pos >= len) continue;
                if ((mt >> bit) & 1u) {
                    if (ntab < 11) {
                        // (static indexing keeps tab[] in registers)
#pragma unroll
                        for (int k = 0; k < 11; ++k)
                            if (k == ntab) tab[k] = (uint32_t)pos;
                    }
                    ++ntab;
                } else if (ntab == 10) {                        // inside the signals field
                    if (nsemi < L - 1) seg[3 + nsemi] = (uint32_t)pos + 1u;
                    ++nsemi;
                }
            }
        }
    }
    ok = ok && ntab >= 10 && nsemi == L - 1;
    // field 10 (signals) ends at tab 10 when there is a label column behind it -- there must be
    ok = ok && ntab >= 11;
    if (!ok) { flag_row(a, r); return; }
    const unsigned c0 = (unsigned)(unsigned char)row[0];
    if (is_space(c0)) { flag_row(a, r); return; }
    a.info_len[r] = tab[5];
    a.read_off[r] = tab[3] + 1;
    a.read_len[r] = tab[4] - tab[3] - 1;
    seg[0] = tab[6] + 1;            // means
    seg[1] = tab[7] + 1;            // stds
    seg[2] = tab[9] + 1;            // signal group 0 (groups 1 .. L-1 were noted at their ';')
    seg[2 + L] = tab[10] + 1;       // one past the '\t' that ends the signals field
    seg[3 + L] = tab[8] + 1;        // one past the '\t' that ends stds
    // the k-mer: exactly L letters of base2code_dna between tab 5 and tab 6
    if (tab[6] - tab[5] - 1 != (uint32_t)L) { flag_row(a, r); return; }
    Reader rd;
    rd.init(row + tab[5] + 1);
    uint8_t* km = a.kmer + r * L;
    for (int i = 0; i < L; ++i) {
        const int c = base_code(rd.cur());
        if (c < 0) { flag_row(a, r); return; }
        km[i] = (uint8_t)c;
        rd.adv();
    }
    // the L lengths between tab 8 and tab 9
    rd.init(row + tab[8] + 1);
    int* ln = a.lens + r * L;
    for (int i = 0; i < L; ++i) {
        int v;
        if (!fast_int(rd, &v) || rd.cur() != (i == L - 1 ? '\t' : ',')) { flag_row(a, r); return; }
        ln[i] = v;
        rd.adv();
    }
    if (rd.pos != tab[9] - tab[8]) { flag_row(a, r); return; }  // the list ended exactly at tab 9
    // the label behind tab 10: ends at the line end (LF or CRLF) or at a tab (extra columns are ignored)
    rd.init(row + tab[10] + 1);
    int lab;
    if (!fast_int(rd, &lab)) { flag_row(a, r); return; }
    const uint32_t p = tab[10] + 1 + rd.pos;
    if (!(p == len || rd.cur() == '\t' || (rd.cur() == '\r' && p + 1 == len)) || p > len) { flag_row(a, r); return; }
    a.labels[r] = lab;
}

// ---- kernel 2: one thread per (row, float list): means, stds and the L signal groups, L or S numbers each, ',' between,
// the list ending exactly one byte before the next segment's start.  15 lists per row: 7.7 waves per SIMD for a block
// of 32,768 rows, which is what hides the lane-divergent 16-byte refills of the cursor.
__global__ __launch_bounds__(256) void dsp_parse_lists_kernel(ParseArgs a) {
    const int NSEG = 2 + a.L;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long r = t / NSEG;
    const int sgi = (int)(t - r * NSEG);
    if (r >= a.n || a.status[r]) return;     // (rows kernel 1 flagged have no segment table)
    const uint32_t* seg = a.seg + (size_t)r * (NSEG + 2);
    const uint32_t s0 = seg[sgi];
    // one past the list's terminator: means -> the start of stds; stds -> seg[3 + L] (the lengths lie behind it); group i ->
    // the start of group i + 1; the last group -> seg[2 + L]
    const uint32_t s1 = sgi == 1 ? seg[3 + a.L] : seg[sgi + 1];
    const uint64_t o0 = a.row_off[r], o1 = a.row_off[r + 1];
    if (o1 <= o0 || o1 > a.text_bytes || s0 >= o1 - o0 || s1 > o1 - o0) { flag_row(a, r); return; }   // (a table kernel 1 did not write)
    const int cnt = sgi < 2 ? a.L : a.S;
    float* dst = sgi == 0 ? a.means + r * a.L : (sgi == 1 ? a.stds + r * a.L : a.signals + ((size_t)r * a.L + (sgi - 2)) * a.S);
    const unsigned term = sgi < 2 ? '\t' : (sgi == NSEG - 1 ? '\t' : ';');
    Reader rd;
    rd.init(a.text + o0 + s0);
    for (int i = 0; i < cnt; ++i) {
        float v;
        if (!fast_float(rd, &v) || rd.cur() != (i == cnt - 1 ? term : ',')) { flag_row(a, r); return; }
        dst[i] = v;
        rd.adv();
    }
    if (s0 + rd.pos != s1) flag_row(a, r);   // the list ended exactly where the table says
}

// ---- the token-parallel parser (round 4, third version): one workgroup per RB rows, the rows' text staged in LDS.
//   (1) the RB rows are one contiguous piece of the block: copied into LDS with coalesced 16-byte loads;
//   (2) every thread counts the delimiters (',' ';' '\t' '\n') of its slice of that piece (SWAR on LDS words), one
//       workgroup scan numbers them, a second pass writes their positions into a table: token j of the piece ends at
//       delimiter j.  A plain row has EXACTLY NTOK = 7 + 3 L + L S + 1 tokens whose terminators' kinds are fixed by the
//       grammar (a row whose sampleinfo holds a ',' or ';', or extra columns, has another count: flagged -> host parser);
//   (3) one thread per token: lane k of a row parses token k from LDS -- 6 sampleinfo fields (addressing only), the k-mer,
//       L + L floats, L integers, L S floats, the label -- and stores its value: neighbouring lanes write neighbouring
//       words (the thread-per-row kernels stored with a 1 kB stride between lanes).
// Same acceptance and the same arithmetic as the kernels above (fast_float / fast_int); whatever is not plain flags its row.
constexpr int kTokRowBytes = 2560;     // LDS per row of a workgroup: RB x this for the text (a row of the default shape: 2.09 kB)
constexpr int kTokMaxNtok = 2048;      // tokens per row the position table is sized for (default shape: 255)

// a byte cursor over LDS: 16 bytes (four aligned words) fetched at once into two 64-bit shift registers -- a number of the
// writer's grammar (<= 12 bytes) never refills; longer tokens fetch the next four words
struct LdsReader {
    const uint32_t* w;
    uint64_t lo, hi;
    int left;
    uint32_t pos;
    __device__ __forceinline__ void fetch() {
        const u32x4 q = *(const u32x4*)w;
        w += 4;
        lo = (uint64_t)q.x | ((uint64_t)q.y << 32);
        hi = (uint64_t)q.z | ((uint64_t)q.w << 32);
        left = 16;
    }
    __device__ __forceinline__ void init(const uint32_t* lds32, uint32_t byte) {
        w = lds32 + ((byte >> 2) & ~3u);           // 16-byte aligned: one ds_read_b128
        fetch();
        const int sk = (int)(byte & 15u);
        if (sk >= 8) { lo = hi >> (8 * (sk - 8)); hi = 0; }
        else if (sk) { lo = (lo >> (8 * sk)) | (hi << (64 - 8 * sk)); hi >>= 8 * sk; }
        left = 16 - sk;
        pos = 0;
    }
    __device__ __forceinline__ unsigned cur() const { return (unsigned)(lo & 0xffu); }
    __device__ __forceinline__ void adv() {
        lo = (lo >> 8) | (hi << 56);
        hi >>= 8;
        ++pos;
        if (--left == 0) fetch();
    }
};

template <int RB>
__global__ __launch_bounds__(256, 6) void dsp_parse_tokens_kernel(ParseArgs a) {
    constexpr int kTokCapBytes = RB * kTokRowBytes;
    DSP_DYN_LDS_T(uint32_t, lds);
    const int L = a.L, S = a.S, NTOK = 7 + 3 * L + L * S + 1;
    const int kTokCapDelims = (RB * NTOK + 64 + 7) & ~7;          // (the host sized the dynamic LDS with the same formula)
    uint32_t* buf = lds;                                          // kTokCapBytes + 48 bytes of text
    uint16_t* dpos = (uint16_t*)(lds + (kTokCapBytes + 48) / 4);   // delimiter positions (LDS byte offsets)
    uint32_t* misc = (uint32_t*)(dpos + kTokCapDelims);           // [0..3] wave sums, [8..8+RB) row flags, then 2 x RB row tables
    uint32_t* rflag = misc + 8;
    uint32_t* rstart = rflag + RB;      // LDS offset of the row's first byte
    uint32_t* rbase = rstart + RB;      // index of the row's first delimiter
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long r0 = (long long)blockIdx.x * RB;
    if (r0 >= a.n) return;
    const int nr = (int)(a.n - r0 < RB ? a.n - r0 : RB);
    const uint64_t b0 = a.row_off[r0], b1 = a.row_off[r0 + nr];
    const bool sane = b1 > b0 && b1 <= a.text_bytes && (b1 - b0) <= (uint64_t)kTokCapBytes;
    if (!sane) {                                                   // longer rows than the LDS holds: the host parser's
        if (tid < nr) flag_row(a, r0 + tid);
        return;
    }
    const uint32_t nbytes = (uint32_t)(b1 - b0);
    const uintptr_t g0 = (uintptr_t)(a.text + b0);
    const uint32_t sk = (uint32_t)(g0 & 15);
    guint4* src = (guint4*)(g0 - sk);
    const uint32_t nchunks = (sk + nbytes + 15u) / 16u;
    {   // all of a thread's loads are issued before the first store (RB x 2,560 bytes / 4 KiB per round: at most 10 rounds)
        constexpr int NR = (kTokCapBytes + 15 + 16) / 16 / 256 + 1;
        u32x4 tmp[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const uint32_t c = (uint32_t)tid + 256u * (uint32_t)i;
            if (c < nchunks) tmp[i] = src[c];
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const uint32_t c = (uint32_t)tid + 256u * (uint32_t)i;
            if (c < nchunks) ((u32x4*)buf)[c] = tmp[i];
        }
    }
    if (tid < RB) rflag[tid] = 0;
    __syncthreads();
    // bytes outside the piece (the tail of the row before it, the head of the row behind it) must not count
    if (tid < 16) {
        uint8_t* bb = (uint8_t*)buf;
        if ((uint32_t)tid < sk) bb[tid] = 0;
        const uint32_t e = sk + nbytes + (uint32_t)tid;
        if (e < nchunks * 16u) bb[e] = 0;
    }
    __syncthreads();
    // delimiters of this thread's slice of words
    const uint32_t nwords = nchunks * 4u;
    const uint32_t per = (nwords + 255u) / 256u;
    const uint32_t w0 = (uint32_t)tid * per, w1 = w0 + per < nwords ? w0 + per : nwords;
    int cnt = 0;
    for (uint32_t i = w0; i < w1; ++i) cnt += __builtin_popcount(delim_mask4(buf[i]));
    int v = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(v, d);
        if (lane >= d) v += u;
    }
    if (lane == 63) misc[wv] = (uint32_t)v;
    __syncthreads();
    int base = v - cnt;
    for (int k = 0; k < wv; ++k) base += (int)misc[k];
    const int total = (int)(misc[0] + misc[1] + misc[2] + misc[3]);
    if (total > kTokCapDelims) {                                   // not rows of numbers: the host parser's
        if (tid < nr) flag_row(a, r0 + tid);
        return;
    }
    for (uint32_t i = w0; i < w1; ++i) {
        uint32_t m = delim_mask4(buf[i]);
        while (m) {
            const int bit = __builtin_ctz(m);
            m &= m - 1;
            dpos[base++] = (uint16_t)(i * 4u + (uint32_t)(bit >> 3));
        }
    }
    __syncthreads();
    // every row's first byte and first delimiter (binary search of the row's start in the position table)
    if (tid < nr) {
        const uint32_t st = sk + (uint32_t)(a.row_off[r0 + tid] - b0);
        const uint32_t en = sk + (uint32_t)(a.row_off[r0 + tid + 1] - b0);   // one past the row's '\n'
        int lo = 0, hi = total;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (dpos[mid] < st) lo = mid + 1; else hi = mid; }
        int lo2 = lo, hi2 = total;
        while (lo2 < hi2) { const int mid = (lo2 + hi2) >> 1; if (dpos[mid] < en) lo2 = mid + 1; else hi2 = mid; }
        rstart[tid] = st;
        rbase[tid] = (uint32_t)lo;
        const uint32_t c0 = ((const uint8_t*)buf)[st];
        // exactly NTOK delimiters, the last one the row's '\n'; no blank in front (line.strip() is the host parser's)
        if (lo2 - lo != NTOK || en - st < (uint32_t)(12 + L) || is_space(c0)) rflag[tid] = 1;
    }
    __syncthreads();
    const uint8_t* bb = (const uint8_t*)buf;
    for (int r = 0; r < nr; ++r) {
        if (rflag[r]) continue;                                     // (uniform: LDS value)
        const long long row = r0 + r;
        const uint32_t st = rstart[r];
        const uint32_t db = rbase[r];
        // the first 64 tokens of a row mix five kinds (sampleinfo, k-mer, floats, integers): the wave that gets them runs five
        // code paths one after the other -- a different wave for every row ((tid + 64 r) mod 256: the waves do not wait for
        // each other between rows)
        for (int kk = 0; kk < NTOK; kk += 256) {
            const int k = kk + ((tid + 64 * r) & 255);
            if (k >= NTOK) continue;
            const uint32_t ts = k == 0 ? st : (uint32_t)dpos[db + k - 1] + 1u;
            const uint32_t te = dpos[db + k];
            const bool ok = parse_token<LdsReader>(a, row, k, NTOK, L, S, st, ts, te, bb, buf);
            if (!ok) rflag[r] = 1;                                  // (benign race: every writer writes 1)
        }
    }
    __syncthreads();
    if (tid < nr) {
        a.status[r0 + tid] = rflag[tid] ? 1 : 0;
        if (rflag[tid]) atomicAdd(a.n_flagged, 1u);
    }
}

}  // namespace

extern "C" void dsp_set_error_(const char* msg);

extern "C" int32_t dsp_parse_rows_device(void* stream, const char* text_dev, const uint64_t* row_off_dev, int64_t n, int32_t seq_len,
                                         int32_t signal_len, uint8_t* kmer, float* means, float* stds, int32_t* lens, float* signals,
                                         int32_t* labels, uint32_t* info_len, uint32_t* read_off, uint32_t* read_len,
                                         uint8_t* status_dev, uint32_t* n_flagged_dev, uint32_t* seg_dev, uint64_t text_bytes) {
    if (n < 0 || seq_len < 1 || signal_len < 1 || !n_flagged_dev ||
        (n > 0 && (!text_dev || !row_off_dev || !kmer || !means || !stds || !lens || !signals || !labels || !info_len || !read_off ||
                   !read_len || !status_dev || !seg_dev))) {
        dsp_set_error_("dsp_parse_rows_device: bad argument");
        return DSP_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(n_flagged_dev, 0, sizeof(uint32_t), s);
    if (e == hipSuccess && n > 0) {
        ParseArgs a{text_dev, row_off_dev, (long long)n, seq_len, signal_len, kmer, means, stds, lens, signals, labels,
                    info_len, read_off, read_len, status_dev, n_flagged_dev, (unsigned long long)text_bytes, seg_dev};
        // the token-parallel kernel when a row of the block's average length (+ 20 %) fits its share of the workgroup's LDS,
        // else the thread-per-row pair above (rows of several kilobytes: other k-mer lengths / signal windows)
        const int ntok = 7 + 3 * seq_len + seq_len * signal_len + 1;
        const double avg = (double)text_bytes / (double)n * 1.2;
        static const int rb_env = getenv("DSP_PARSE_RB") ? atoi(getenv("DSP_PARSE_RB")) : 4;                         // A/B switch: 4, 8, 16
        static const bool force_rows = getenv("DSP_PARSE_KERNEL") && !strcmp(getenv("DSP_PARSE_KERNEL"), "rows");   // A/B switch
        const int rb = (rb_env == 1 || rb_env == 2 || rb_env == 8 || rb_env == 16) ? rb_env : 4;
        if (avg <= kTokRowBytes && ntok <= kTokMaxNtok && !force_rows) {
            const size_t lds = (size_t)rb * kTokRowBytes + 48 + (size_t)(((rb * ntok + 64 + 7) & ~7)) * 2 + (8 + 3 * 16) * 4;
            const unsigned grid = (unsigned)((n + rb - 1) / rb);
            if (rb == 16) hipLaunchKernelGGL(dsp_parse_tokens_kernel<16>, dim3(grid), dim3(256), lds, s, a);
            else if (rb == 8) hipLaunchKernelGGL(dsp_parse_tokens_kernel<8>, dim3(grid), dim3(256), lds, s, a);
            else if (rb == 2) hipLaunchKernelGGL(dsp_parse_tokens_kernel<2>, dim3(grid), dim3(256), lds, s, a);
            else if (rb == 1) hipLaunchKernelGGL(dsp_parse_tokens_kernel<1>, dim3(grid), dim3(256), lds, s, a);
            else hipLaunchKernelGGL(dsp_parse_tokens_kernel<4>, dim3(grid), dim3(256), lds, s, a);
        } else {
            hipLaunchKernelGGL(dsp_parse_scan_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, a);
            const long long lists = (long long)n * (2 + seq_len);
            hipLaunchKernelGGL(dsp_parse_lists_kernel, dim3((unsigned)((lists + 255) / 256)), dim3(256), 0, s, a);
        }
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        dsp_set_error_(hipGetErrorString(e));
        return DSP_EHIP;
    }
    return DSP_OK;
}
