// dsp_parse_arith.h -- the ARITHMETIC of the GPU row parser (dsp_parse_dev.hip): number and letter decoding over an abstract
// byte cursor, and the SWAR delimiter masks.  Plain C++ -- no intrinsic, no memory access of its own -- kept in a header so that
// the very same source is compiled
//   * into the gfx950 kernels (cursors over global memory / LDS), and
//   * into tests/native/parse_dev_host.cpp (a cursor over a host buffer) under AddressSanitizer + UBSan, bit-compared with the
//     host parser csrc/dsp_text.cpp on 200 k random spellings and 30 k mutated rows (GPU sanitizers are not
//     available on this pool, and a signed overflow or an out-of-range shift in here would be silent on the device).
// Same values as the host parser, by construction: the plain-number path of csrc/dsp_text.cpp (parse_row_fast) -- digits
// accumulated into an integer mantissa, value = mantissa * or / an exact power of ten in float64 (ONE correctly rounded IEEE
// operation; both translation units are built with -ffp-contract=off), then float64 -> float32 as the reference's FloatTensor
// does.  Whatever is not a plain number returns false: the row is flagged for the host parser, which owns the error messages.
// A cursor `RD` offers cur() (the byte under it, as unsigned) and adv() (one byte on); reading past the token is the cursor's
// business (the device cursors stay inside their 16-byte words / their row's staged text).
#ifndef DSP_PARSE_ARITH_H
#define DSP_PARSE_ARITH_H

#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define DSP_PA_FN __device__ __forceinline__
#define DSP_PA_TABLE static __constant__
#else
#define DSP_PA_FN inline
#define DSP_PA_TABLE static const
#endif

namespace dsp_parse_arith {

DSP_PA_TABLE double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                  1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// [-]digits[.digits][e[+-]digits], at most 18 digits; false = not a plain number (the row goes to the host parser)
template <class RD>
DSP_PA_FN bool fast_float(RD& r, float* dst) {
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

template <class RD>
DSP_PA_FN bool fast_int(RD& r, int* out) {
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
DSP_PA_FN int base_code(unsigned c) {
    switch (c) {
        case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case 'N': return 4; case 'W': return 5;
        case 'S': return 6; case 'M': return 7; case 'K': return 8; case 'R': return 9; case 'Y': return 10; case 'B': return 11;
        case 'V': return 12; case 'D': return 13; case 'H': return 14; case 'Z': return 15; default: return -1;
    }
}
DSP_PA_FN bool is_space(unsigned c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\f' || c == '\v'; }

DSP_PA_FN uint32_t eq_mask4(uint32_t w, uint32_t pat) {   // bit 7 of every byte of w that equals pat's byte
    const uint32_t x = w ^ pat;
    return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);   // exact zero-byte detector
}

DSP_PA_FN uint32_t delim_mask4(uint32_t w) {
    return eq_mask4(w, 0x2c2c2c2cu) | eq_mask4(w, 0x3b3b3b3bu) | eq_mask4(w, 0x09090909u) | eq_mask4(w, 0x0a0a0a0au);
}

// One token of a plain row: token k of NTOK = 7 + 3 L + L S + 1, the bytes [ts, te) of the staged piece `bb` (`buf` = the same
// memory as 32-bit words, what the cursor RD reads), terminated by the delimiter bb[te]; `st` = the row's first byte.  Tokens
// 0..5: the sampleinfo fields (kept verbatim, only addressed); 6: the k-mer; then L means, L stds, L lengths (integers),
// L x S signal values; the label.  Writes the token's value into the arrays of `a` (the kernel's ParseArgs / the host
// harness's twin: same member names) and returns false for anything the plain grammar excludes (the row is then flagged).
template <class RD, class ARGS, class WORDS>
DSP_PA_FN bool parse_token(const ARGS& a, long long row, int k, int NTOK, int L, int S, uint32_t st, uint32_t ts, uint32_t te, const uint8_t* bb,
                           WORDS buf) {
    const unsigned term = bb[te];
    bool ok = true;
    if (k < 6) {                                            // sampleinfo: kept verbatim, only addressed
        ok = term == '\t';
        if (k == 4) { a.read_off[row] = ts - st; a.read_len[row] = te - ts; }
        if (k == 5) a.info_len[row] = te - st;
    } else if (k == 6) {                                    // the k-mer: exactly L letters of base2code_dna
        ok = term == '\t' && te - ts == (uint32_t)L;
        if (ok) {
            uint8_t* km = a.kmer + row * L;
            for (int i = 0; i < L; ++i) {
                const int c = base_code(bb[ts + i]);
                if (c < 0) { ok = false; break; }
                km[i] = (uint8_t)c;
            }
        }
    } else if (k == NTOK - 1) {                             // the label: an integer, LF or CRLF behind it
        RD rd;
        rd.init(buf, ts);
        int lab;
        ok = term == '\n' && fast_int(rd, &lab) && (rd.pos == te - ts || (rd.cur() == '\r' && rd.pos + 1 == te - ts));
        if (ok) a.labels[row] = lab;
    } else {
        const int q = k - 7;
        const bool is_int = q >= 2 * L && q < 3 * L;
        int idx, cnt1;
        unsigned last;
        if (q < 3 * L) { idx = q % L; cnt1 = L; last = '\t'; }
        else { idx = (q - 3 * L) % S; cnt1 = S; last = (q - 3 * L) / S == L - 1 ? '\t' : ';'; }
        ok = term == (idx == cnt1 - 1 ? last : (unsigned)',');
        if (ok) {
            RD rd;
            rd.init(buf, ts);
            if (is_int) {
                int val;
                ok = fast_int(rd, &val) && rd.pos == te - ts;
                if (ok) a.lens[row * L + (q - 2 * L)] = val;
            } else {
                float val;
                ok = fast_float(rd, &val) && rd.pos == te - ts;
                if (ok) {
                    if (q < L) a.means[row * L + q] = val;
                    else if (q < 2 * L) a.stds[row * L + (q - L)] = val;
                    else a.signals[(size_t)row * L * S + (q - 3 * L)] = val;
                }
            }
        }
    }
    return ok;
}

}  // namespace dsp_parse_arith

#endif
