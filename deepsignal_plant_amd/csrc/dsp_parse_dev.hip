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
// A row is about 2.1 kB in 13 + 13 + 13 + 208 numbers: one lane reads its row through 16-byte loads, one ahead of the
// parse (32,768 rows = 512 waves; the kernel is latency-bound at a few hundred microseconds per block, under the
// previous block's 26 ms forward).  HBM-bound integer work: no MFMA, no LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dsp_amd.h"

namespace {

__constant__ double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                  1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// a byte cursor over global memory: 16-byte words, the next one requested while the current one is consumed
struct Reader {
    const uint4* next;     // address of the word after `ahead`
    uint4 ahead;           // the word behind the current one (already loaded)
    uint64_t lo, hi;       // the unconsumed bytes of the current word, lowest byte first in `lo`
    int left;              // bytes left in lo | hi
    uint32_t pos;          // offset of the cursor from the row start
    __device__ __forceinline__ void init(const char* p) {
        const uintptr_t a = (uintptr_t)p;
        const uint4* w = (const uint4*)(a & ~(uintptr_t)15);
        const int skip = (int)(a & 15);
        const uint4 cur = w[0];
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

// [-]digits[.digits][e[+-]digits], at most 18 digits; false = not a plain number (the row goes to the host parser)
__device__ __forceinline__ bool fast_float(Reader& r, float* dst) {
    const bool neg = r.cur() == '-';
    if (neg) r.adv();
    uint64_t m = 0;
    unsigned d;
    int nd = 0, e10 = 0;
    while ((d = r.cur() - '0') < 10u) { m = m * 10 + d; r.adv(); ++nd; }
    if (nd == 0) return false;
    if (r.cur() == '.') {
        r.adv();
        int nf = 0;
        while ((d = r.cur() - '0') < 10u) { m = m * 10 + d; r.adv(); ++nf; if (nd + nf > 19) return false; }
        e10 = -nf;
        nd += nf;
    }
    if (nd > 18) return false;
    if ((r.cur() | 0x20u) == 'e') {
        r.adv();
        const unsigned c = r.cur();
        const bool eneg = c == '-';
        if (c == '-' || c == '+') r.adv();
        int ex = 0, ne = 0;
        while ((d = r.cur() - '0') < 10u && ne < 4) { ex = ex * 10 + (int)d; r.adv(); ++ne; }
        if (ne == 0 || (r.cur() - '0') < 10u) return false;
        e10 += eneg ? -ex : ex;
    }
    if (m >= (1ull << 53) || e10 < -22 || e10 > 22) {
        if (m != 0) return false;
        e10 = 0;
    }
    double v = (double)m;                                   // exact: m < 2^53
    v = e10 < 0 ? v / kPow10[-e10] : v * kPow10[e10];       // ONE correctly rounded operation (Clinger's fast path)
    *dst = (float)(neg ? -v : v);
    return true;
}

__device__ __forceinline__ bool fast_int(Reader& r, int* out) {
    const bool neg = r.cur() == '-';
    if (neg) r.adv();
    long long v = 0;
    unsigned d;
    int nd = 0;
    while ((d = r.cur() - '0') < 10u && nd < 9) { v = v * 10 + (long long)d; r.adv(); ++nd; }   // 9 digits fit an int32
    if (nd == 0 || (r.cur() - '0') < 10u) return false;
    *out = (int)(neg ? -v : v);
    return true;
}

// base2code_dna (utils/process_utils.py:25-29): "ACGTNWSMKRYBVDHZ" -> 0..15, anything else -1
__device__ __forceinline__ int base_code(unsigned c) {
    switch (c) {
        case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case 'N': return 4; case 'W': return 5;
        case 'S': return 6; case 'M': return 7; case 'K': return 8; case 'R': return 9; case 'Y': return 10; case 'B': return 11;
        case 'V': return 12; case 'D': return 13; case 'H': return 14; case 'Z': return 15; default: return -1;
    }
}
__device__ __forceinline__ bool is_space(unsigned c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\f' || c == '\v'; }

struct ParseArgs {
    const char* text; const uint64_t* row_off; long long n; int L, S;
    uint8_t* kmer; float* means; float* stds; int* lens; float* signals; int* labels;
    uint32_t* info_len; uint32_t* read_off; uint32_t* read_len; uint8_t* status; uint32_t* n_flagged;
};

// true = the row is a plain row and every output of it has been written
__device__ bool parse_row(const ParseArgs& a, long long r) {
    const uint64_t o0 = a.row_off[r], o1 = a.row_off[r + 1];
    const uint32_t len = (uint32_t)(o1 - o0 - 1);              // without the '\n'
    const int L = a.L, S = a.S;
    if (len < (uint32_t)(12 + L)) return false;
    Reader rd;
    rd.init(a.text + o0);
    if (is_space(rd.cur())) return false;
    // the six sampleinfo fields, kept verbatim: only their tabs matter
    uint32_t tab3 = 0, tab4 = 0, tab5 = 0;
    for (int k = 0; k < 6; ++k) {
        while (rd.pos < len && rd.cur() != '\t') rd.adv();
        if (rd.pos >= len) return false;
        if (k == 3) tab3 = rd.pos; else if (k == 4) tab4 = rd.pos; else if (k == 5) tab5 = rd.pos;
        rd.adv();
    }
    if (len - rd.pos < (uint32_t)(L + 1)) return false;
    uint8_t* km = a.kmer + r * L;
    for (int i = 0; i < L; ++i) {
        const int c = base_code(rd.cur());
        if (c < 0) return false;
        km[i] = (uint8_t)c;
        rd.adv();
    }
    if (rd.cur() != '\t') return false;
    rd.adv();
    for (int which = 0; which < 2; ++which) {                  // means, stds: L numbers, ',' between, '\t' behind
        float* dst = (which ? a.stds : a.means) + r * L;
        for (int i = 0; i < L; ++i) {
            float v;
            if (!fast_float(rd, &v) || rd.cur() != (i == L - 1 ? '\t' : ',') || rd.pos >= len) return false;
            dst[i] = v;
            rd.adv();
        }
    }
    int* ln = a.lens + r * L;
    for (int i = 0; i < L; ++i) {
        int v;
        if (!fast_int(rd, &v) || rd.cur() != (i == L - 1 ? '\t' : ',') || rd.pos >= len) return false;
        ln[i] = v;
        rd.adv();
    }
    float* sg = a.signals + (size_t)r * L * S;
    for (int i = 0; i < L; ++i)
        for (int j = 0; j < S; ++j) {
            float v;
            const unsigned want = j < S - 1 ? ',' : (i == L - 1 ? '\t' : ';');
            if (!fast_float(rd, &v) || rd.cur() != want || rd.pos >= len) return false;
            sg[i * S + j] = v;
            rd.adv();
        }
    int lab;
    if (!fast_int(rd, &lab) || rd.pos > len) return false;
    // the 12th field ends at the line end (LF or CRLF) or at a tab (extra columns are ignored, as words[11] would be)
    if (!(rd.pos == len || rd.cur() == '\t' || (rd.cur() == '\r' && rd.pos + 1 == len))) return false;
    a.labels[r] = lab;
    a.info_len[r] = tab5;
    a.read_off[r] = tab3 + 1;
    a.read_len[r] = tab4 - tab3 - 1;
    return true;
}

__global__ __launch_bounds__(64) void dsp_parse_rows_kernel(ParseArgs a) {
    const long long r = (long long)blockIdx.x * 64 + threadIdx.x;
    if (r >= a.n) return;
    const bool ok = parse_row(a, r);
    a.status[r] = ok ? 0 : 1;
    if (!ok) atomicAdd(a.n_flagged, 1u);
}

}  // namespace

extern "C" void dsp_set_error_(const char* msg);

extern "C" int32_t dsp_parse_rows_device(void* stream, const char* text_dev, const uint64_t* row_off_dev, int64_t n, int32_t seq_len,
                                         int32_t signal_len, uint8_t* kmer, float* means, float* stds, int32_t* lens, float* signals,
                                         int32_t* labels, uint32_t* info_len, uint32_t* read_off, uint32_t* read_len,
                                         uint8_t* status_dev, uint32_t* n_flagged_dev) {
    if (n < 0 || seq_len < 1 || signal_len < 1 || !n_flagged_dev ||
        (n > 0 && (!text_dev || !row_off_dev || !kmer || !means || !stds || !lens || !signals || !labels || !info_len || !read_off ||
                   !read_len || !status_dev))) {
        dsp_set_error_("dsp_parse_rows_device: bad argument");
        return DSP_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(n_flagged_dev, 0, sizeof(uint32_t), s);
    if (e == hipSuccess && n > 0) {
        ParseArgs a{text_dev, row_off_dev, (long long)n, seq_len, signal_len, kmer, means, stds, lens, signals, labels,
                    info_len, read_off, read_len, status_dev, n_flagged_dev};
        hipLaunchKernelGGL(dsp_parse_rows_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, a);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        dsp_set_error_(hipGetErrorString(e));
        return DSP_EHIP;
    }
    return DSP_OK;
}
