// parse_dev_host.cpp -- the arithmetic and the per-token grammar of the GPU row parser (csrc/dsp_parse_arith.h: the very
// source dsp_parse_dev.hip compiles for gfx950) driven on the HOST under AddressSanitizer + UBSan and compared, row by row, with
// the host parser csrc/dsp_text.cpp (dsp_parse_feature_rows).  VERDICT r5 item 4: GPU sanitizers are not available on this
// pool; a signed overflow, an out-of-range shift or an off-by-one in fast_float / fast_int / base_code / the SWAR delimiter
// masks / the token rules would be silent on the device.  TEST INFRASTRUCTURE -- the product parses on the GPU only; nothing
// here is linked into it.
//
// What runs here is the token-parallel kernel's algorithm (dsp_parse_tokens_kernel<RB>) with its data-parallel steps done in
// loops: a piece of RB rows staged at its 16-byte misalignment, the bytes outside the piece zeroed, delimiters found with
// delim_mask4 on 32-bit words and numbered, every row's delimiter range located, exactly NTOK delimiters demanded, every token
// handed to parse_token<> with a cursor that -- like the kernel's LdsReader -- fetches 16 aligned bytes at a time.  The cursor
// reads through a bounds-checked accessor: ASan sees every byte the algorithm touches.
//   (1) N rows of random float spellings (fixed / scientific, 1..17 significant digits, signs, leading zeros, exponents to +-25;
//       one row in ten holds one token outside the plain grammar: '+', blanks, inf / nan, 25 digits, 1e400)
//   (2) M blocks of 1..3 rows with 1..3 bytes overwritten from "\t,;.-+eE0123456789 \nACGTNX\r:_"
// -DPARSE_THROUGH_KERNELS (tests/test_kernel_emu.py): instead of the transcription below, the KERNELS of csrc/dsp_parse_dev.hip run
// -- compiled for the host by the SIMT interpreter tests/native/emu, all three of them (DSP_PARSE_KERNEL=rows: the thread-per-row
// pair that takes rows longer than the LDS piece), with ASan's red zones around every array they touch.
// Checks: a row the device algorithm accepts is accepted by the host parser with bit-identical arrays; a row the host parser
// rejects is flagged; plain rows are not flagged.  usage: parse_dev_host [N=200000] [M=30000]
#include <cinttypes>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dsp_amd.h"
#include "dsp_parse_arith.h"

using namespace dsp_parse_arith;

// (dsp_capi.cpp holds these in the library; the harness links dsp_text.cpp alone)
static std::string g_err;
extern "C" void dsp_set_error_(const char* msg) { g_err = msg ? msg : ""; }
extern "C" const char* dsp_last_error(void) { return g_err.c_str(); }

namespace {

constexpr int L = 13, S = 16, NTOK = 7 + 3 * L + L * S + 1;
constexpr int RB = 4, kTokRowBytes = 2560, kTokCapBytes = RB * kTokRowBytes;

struct Rng {
    uint64_t s;
    uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); }
    int below(int n) { return (int)(next() % (uint32_t)n); }
    double uniform() { return (next() + 0.5) / 2147483648.0; }
    double normal() { return std::sqrt(-2.0 * std::log(uniform())) * std::cos(6.283185307179586 * uniform()); }
};

std::string fmt(const char* f, int prec, double x) { char b[128]; snprintf(b, sizeof b, f, prec, x); return b; }
std::string spell(Rng& r, double x, int maxd, bool exotic) {
    // (mostly short tokens, so that four rows fit the token kernel's LDS piece -- longer pieces take the thread-per-row kernels,
    // whose walk is not modelled here --, one token in eight with up to `maxd` <= 17 significant digits)
    const int digs = r.below(8) == 0 ? 1 + r.below(maxd) : 1 + r.below(maxd < 6 ? maxd : 6);
    if (exotic) {
        switch (r.below(7)) {
            case 0: return fmt("%.*e", 17, x);
            case 1: return fmt("%.*e", 3, x * 1e30);
            case 2: return "+" + fmt("%.*f", 3, std::fabs(x));
            case 3: return fmt("%.*f", 3, x) + " ";
            case 4: return "1e400";
            case 5: return "nan";
            default: return fmt("%.*f", 25, x);
        }
    }
    switch (r.below(6)) {
        case 0: return fmt("%.*f", digs < 9 ? digs : 9, x);
        case 1: return fmt("%.*e", digs - 1, x);
        case 2: return fmt("%.*g", digs, x);
        case 3: return fmt("%.*E", digs - 1, x * std::pow(10.0, (double)(r.below(37) - 18)));
        case 4: { char b[32]; snprintf(b, sizeof b, "%d", (int)(x * 1000)); return b; }
        default: return std::string(x < 0 ? "-" : "") + std::string((size_t)r.below(3), '0') + fmt("%.*f", digs < 7 ? digs : 7, std::fabs(x));
    }
}
std::string random_row(Rng& r, long i) {
    const int maxd = 1 + r.below(17);
    const int ex = r.below(10) == 0 ? r.below(234) : -1;   // one row in ten: ONE token outside the plain grammar
    int cnt = 0;
    auto tok = [&](double v) { const bool e = cnt == ex; ++cnt; return spell(r, v, maxd, e); };
    std::string row = "chr" + std::to_string(i % 7) + "\t" + std::to_string(i * 3) + "\t" + "+-"[i & 1] + "\t" + std::to_string(i) + "\tread_" +
                      std::to_string(i / 50) + "\tt\t";
    for (int k = 0; k < L; ++k) row += "ACGTN"[r.below(5)];
    for (int list = 0; list < 2; ++list) { row += '\t'; for (int k = 0; k < L; ++k) { if (k) row += ','; row += tok(r.normal() * 1.5); } }
    row += '\t';
    for (int k = 0; k < L; ++k) { if (k) row += ','; row += std::to_string(1 + r.below(399)); }
    row += '\t';
    for (int g = 0; g < L; ++g) { if (g) row += ';'; for (int k = 0; k < S; ++k) { if (k) row += ','; row += tok(r.normal() * 1.5); } }
    row += '\t'; row += std::to_string(i & 1);
    return row;
}

// ---- the staged piece, every access bounds-checked (ASan would also see a stray one: the vectors are exactly sized)
struct Piece {
    std::vector<uint32_t> words;   // the kernel's `buf`: kTokCapBytes + 48 bytes of LDS
    uint8_t byte(uint32_t i) const { if (i >= words.size() * 4) { fprintf(stderr, "byte %u outside the staged piece\n", i); abort(); } return ((const uint8_t*)words.data())[i]; }
};
// the kernel's LdsReader: 16 aligned bytes at a time into two 64-bit shift registers
struct HostReader {
    const Piece* p; uint32_t w; uint64_t lo, hi; int left; uint32_t pos;
    void fetch() {
        if ((size_t)w + 4 > p->words.size()) { fprintf(stderr, "cursor fetch outside the staged piece (word %u of %zu)\n", w, p->words.size()); abort(); }
        const uint32_t* q = p->words.data() + w;
        w += 4;
        lo = (uint64_t)q[0] | ((uint64_t)q[1] << 32);
        hi = (uint64_t)q[2] | ((uint64_t)q[3] << 32);
        left = 16;
    }
    void init(const Piece* piece, uint32_t byte) {
        p = piece;
        w = (byte >> 2) & ~3u;
        fetch();
        const int sk = (int)(byte & 15u);
        if (sk >= 8) { lo = hi >> (8 * (sk - 8)); hi = 0; }
        else if (sk) { lo = (lo >> (8 * sk)) | (hi << (64 - 8 * sk)); hi >>= 8 * sk; }
        left = 16 - sk;
        pos = 0;
    }
    unsigned cur() const { return (unsigned)(lo & 0xffu); }
    void adv() { lo = (lo >> 8) | (hi << 56); hi >>= 8; ++pos; if (--left == 0) fetch(); }
};

struct Out {   // the kernel's ParseArgs members parse_token<> writes
    uint8_t* kmer; float* means; float* stds; int* lens; float* signals; int* labels; uint32_t* info_len; uint32_t* read_off; uint32_t* read_len;
};
struct Arrays {
    std::vector<uint8_t> kmer, status; std::vector<float> means, stds, signals; std::vector<int> lens, labels; std::vector<uint32_t> info_len, read_off, read_len;
    explicit Arrays(size_t n) : kmer(n * L), status(n), means(n * L), stds(n * L), signals(n * L * S), lens(n * L), labels(n), info_len(n), read_off(n), read_len(n) {}
    Out out() { return Out{kmer.data(), means.data(), stds.data(), lens.data(), signals.data(), labels.data(), info_len.data(), read_off.data(), read_len.data()}; }
};

// dsp_parse_tokens_kernel<RB> for rows [r0, r0 + nr) of the staged text (text + 64 readable bytes, row_off: n + 1 offsets)
void device_algorithm(const std::vector<char>& text, const std::vector<uint64_t>& row_off, uint64_t text_bytes, long long n, long long r0, Arrays& A) {
    const int nr = (int)(n - r0 < RB ? n - r0 : RB);
    const uint64_t b0 = row_off[(size_t)r0], b1 = row_off[(size_t)(r0 + nr)];
    const bool sane = b1 > b0 && b1 <= text_bytes && (b1 - b0) <= (uint64_t)kTokCapBytes;
    if (!sane) { for (int t = 0; t < nr; ++t) A.status[(size_t)(r0 + t)] = 1; return; }   // (the thread-per-row kernels' case: not modelled here)
    const uint32_t nbytes = (uint32_t)(b1 - b0);
    // the staging buffer starts at the 16-byte boundary below the piece: sk bytes of the row before it come along
    const uint32_t sk = (uint32_t)(b0 & 15);   // (the product's staging buffer is 16-byte aligned at offset 0)
    const uint32_t nchunks = (sk + nbytes + 15u) / 16u;
    Piece P;
    P.words.assign((kTokCapBytes + 48) / 4, 0xa5a5a5a5u);   // (LDS is not zeroed: garbage outside what the kernel writes)
    for (uint32_t c = 0; c < nchunks; ++c) {
        const size_t src = (size_t)(b0 - sk) + (size_t)c * 16;
        if (src + 16 > text.size()) { fprintf(stderr, "staging read outside the text (+ its 64 slack bytes)\n"); abort(); }
        memcpy((char*)P.words.data() + (size_t)c * 16, text.data() + src, 16);
    }
    uint8_t* bbw = (uint8_t*)P.words.data();
    for (uint32_t t = 0; t < 16; ++t) {
        if (t < sk) bbw[t] = 0;
        const uint32_t e = sk + nbytes + t;
        if (e < nchunks * 16u) bbw[e] = 0;
    }
    const uint32_t nwords = nchunks * 4u;
    const int cap_delims = (RB * NTOK + 64 + 7) & ~7;
    std::vector<uint16_t> dpos;
    for (uint32_t i = 0; i < nwords; ++i) {
        uint32_t m = delim_mask4(P.words[i]);
        while (m) { const int bit = __builtin_ctz(m); m &= m - 1; dpos.push_back((uint16_t)(i * 4u + (uint32_t)(bit >> 3))); }
    }
    const int total = (int)dpos.size();
    if (total > cap_delims) { for (int t = 0; t < nr; ++t) A.status[(size_t)(r0 + t)] = 1; return; }
    Out out = A.out();
    for (int t = 0; t < nr; ++t) {
        const uint32_t st = sk + (uint32_t)(row_off[(size_t)(r0 + t)] - b0), en = sk + (uint32_t)(row_off[(size_t)(r0 + t + 1)] - b0);
        int lo = 0, hi = total;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (dpos[(size_t)mid] < st) lo = mid + 1; else hi = mid; }
        int lo2 = lo, hi2 = total;
        while (lo2 < hi2) { const int mid = (lo2 + hi2) >> 1; if (dpos[(size_t)mid] < en) lo2 = mid + 1; else hi2 = mid; }
        bool flag = lo2 - lo != NTOK || en - st < (uint32_t)(12 + L) || is_space(P.byte(st));
        const long long row = r0 + t;
        for (int k = 0; k < NTOK && !flag; ++k) {
            const uint32_t ts = k == 0 ? st : (uint32_t)dpos[(size_t)(lo + k - 1)] + 1u, te = dpos[(size_t)(lo + k)];
            if (!parse_token<HostReader>(out, row, k, NTOK, L, S, st, ts, te, (const uint8_t*)P.words.data(), &P)) flag = true;
        }
        A.status[(size_t)row] = flag ? 1 : 0;
    }
}

struct Tally { long rows = 0, same = 0, flagged_plain = 0, host_rejected = 0; };

template <class T> bool same_bits(const T* a, const T* b, size_t n) { return memcmp(a, b, n * sizeof(T)) == 0; }

// one block of complete rows: stage it as the product does (dsp_copy_rows_index), run the device algorithm, compare every row
// with the host parser on that row alone
void check_block(const std::string& block, Tally& t, bool expect_plain_unflagged) {
    const long long cap = (long long)dsp_count_rows(block.data(), block.size());
    std::vector<char> staged(block.size() + 1 + 64, 0x5a);          // + the 64 readable bytes behind the text
    std::vector<uint64_t> row_off((size_t)cap + 1);
    const long long n = (long long)dsp_copy_rows_index(block.data(), block.size(), staged.data(), row_off.data(), cap);
    if (n < 0 || n > cap) { fprintf(stderr, "dsp_copy_rows_index: %lld\n", n); abort(); }
    if (n == 0) return;
    const uint64_t text_bytes = row_off[(size_t)n];
    Arrays dev((size_t)n), host(1);
#ifdef PARSE_THROUGH_KERNELS
    // the KERNELS themselves (csrc/dsp_parse_dev.hip compiled for the host by the test-suite's SIMT interpreter, tests/native/emu):
    // the token-parallel kernel, or -- DSP_PARSE_KERNEL=rows -- the thread-per-row pair; every array exactly sized for ASan
    {
        std::vector<uint32_t> seg((size_t)n * (L + 4)), n_flagged(1);
        const int32_t rc = dsp_parse_rows_device(nullptr, staged.data(), row_off.data(), n, L, S, dev.kmer.data(), dev.means.data(), dev.stds.data(), dev.lens.data(),
                                                 dev.signals.data(), dev.labels.data(), dev.info_len.data(), dev.read_off.data(), dev.read_len.data(),
                                                 dev.status.data(), n_flagged.data(), seg.data(), text_bytes);
        if (rc) { fprintf(stderr, "dsp_parse_rows_device: %s\n", dsp_last_error()); exit(1); }
    }
#else
    for (long long r0 = 0; r0 < n; r0 += RB) device_algorithm(staged, row_off, text_bytes, n, r0, dev);
#endif
    for (long long i = 0; i < n; ++i) {
        const char* row = staged.data() + row_off[(size_t)i];
        const size_t len = (size_t)(row_off[(size_t)i + 1] - row_off[(size_t)i]);
        uint64_t h_off[2];
        const long long hn = (long long)dsp_parse_feature_rows(row, len, L, S, 1, host.kmer.data(), host.means.data(), host.stds.data(), host.lens.data(),
                                                               host.signals.data(), host.labels.data(), h_off, host.info_len.data(), host.read_off.data(),
                                                               host.read_len.data(), 1);
        ++t.rows;
        const bool dev_ok = dev.status[(size_t)i] == 0;
        if (hn != 1) {   // the host parser rejects the row (or sees no row in it): the device algorithm must have flagged it
            if (dev_ok) { fprintf(stderr, "ACCEPTED a row the host parser rejects (%s): %.200s\n", dsp_last_error(), std::string(row, len).c_str()); exit(1); }
            ++t.host_rejected;
            continue;
        }
        if (!dev_ok) {
            if (expect_plain_unflagged) { fprintf(stderr, "flagged a plain row: %.300s\n", std::string(row, len).c_str()); exit(1); }
            ++t.flagged_plain;
            continue;
        }
        const size_t u = (size_t)i;
        const bool eq = same_bits(&dev.kmer[u * L], host.kmer.data(), L) && same_bits(&dev.means[u * L], host.means.data(), L) &&
                        same_bits(&dev.stds[u * L], host.stds.data(), L) && same_bits(&dev.lens[u * L], host.lens.data(), L) &&
                        same_bits(&dev.signals[u * L * S], host.signals.data(), (size_t)L * S) && dev.labels[u] == host.labels[0] &&
                        dev.info_len[u] == host.info_len[0] && dev.read_off[u] == host.read_off[0] && dev.read_len[u] == host.read_len[0];
        if (!eq) { fprintf(stderr, "an accepted row differs from the host parser's: %.300s\n", std::string(row, len).c_str()); exit(1); }
        ++t.same;
    }
}

}  // namespace

int main(int argc, char** argv) {
    const long N = argc > 1 ? atol(argv[1]) : 200000, M = argc > 2 ? atol(argv[2]) : 30000;
    Rng r{2024};
    Tally a;
    for (long c0 = 0; c0 < N; c0 += 2000) {
        std::string block;
        const long k = N - c0 < 2000 ? N - c0 : 2000;
        for (long i = 0; i < k; ++i) { block += random_row(r, c0 + i); block += '\n'; }
        check_block(block, a, false);
    }
    printf("random spellings: %ld rows, %ld accepted and bit-identical to the host parser, %ld flagged though the host parser takes them (longer mantissas, "
           "exponents beyond +-22, '+', blanks: left to it), %ld rejected by the host parser and flagged\n", a.rows, a.same, a.flagged_plain, a.host_rejected);
    if (a.same < a.rows / 2) { fprintf(stderr, "fewer than half of the rows were accepted: the generator or the algorithm is off\n"); return 1; }
    // plain rows (the writer's grammar: %.6f) must never be flagged
    {
        Tally p;
        std::string block;
        for (long i = 0; i < 4000; ++i) {
            std::string row = "chr1\t" + std::to_string(i) + "\t+\t" + std::to_string(i) + "\tread_" + std::to_string(i / 50) + "\tt\t";
            for (int k = 0; k < L; ++k) row += "ACGTNWSMKRYBVDHZ"[r.below(16)];
            for (int list = 0; list < 2; ++list) { row += '\t'; for (int k = 0; k < L; ++k) { if (k) row += ','; row += fmt("%.*f", 6, r.normal()); } }
            row += '\t';
            for (int k = 0; k < L; ++k) { if (k) row += ','; row += std::to_string(1 + r.below(399)); }
            row += '\t';
            for (int g = 0; g < L; ++g) { if (g) row += ';'; for (int k = 0; k < S; ++k) { if (k) row += ','; row += fmt("%.*f", 6, r.normal()); } }
            row += '\t'; row += std::to_string(i & 1); row += (i % 5 == 0) ? "\r\n" : "\n";
            block += row;
        }
        check_block(block, p, true);
        printf("writer's grammar: %ld rows (every fifth CRLF), all accepted and bit-identical\n", p.same);
        if (p.same != 4000) return 1;
    }
    std::vector<std::string> base;
    for (int i = 0; i < 64; ++i) base.push_back(random_row(r, i));
    static const char pool[] = "\t,;.-+eE0123456789 \nACGTNX\r:_";
    Tally b;
    for (long c0 = 0; c0 < M; ++c0) {
        std::string blk;
        const int k = 1 + r.below(3);
        for (int i = 0; i < k; ++i) { blk += base[(size_t)r.below(64)]; blk += '\n'; }
        const int muts = 1 + r.below(3);
        for (int i = 0; i < muts; ++i) blk[(size_t)r.below((int)blk.size())] = pool[r.below((int)sizeof pool - 1)];
        if (blk.back() != '\n') blk += '\n';
        check_block(blk, b, false);
    }
    printf("mutated blocks: %ld blocks, %ld rows: accepted and bit-identical %ld, flagged though the host parser takes them %ld, rejected by the host "
           "parser and flagged %ld; never accepted what the host rejects\n", M, b.rows, b.same, b.flagged_plain, b.host_rejected);
#ifdef PARSE_THROUGH_KERNELS
    printf("parse_dev_host: ok (through the interpreted kernels%s)\n", getenv("DSP_PARSE_KERNEL") ? ", thread-per-row pair" : ", token-parallel kernel");
#else
    printf("parse_dev_host: ok\n");
#endif
    return 0;
}
