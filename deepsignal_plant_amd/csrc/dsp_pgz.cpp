// dsp_pgz.cpp -- parallel inflate of ONE gzip stream (round 3).
//
// A feature file written by the reference's `extract --gzip` is a single gzip member (read back with gzip.open at
// call_modifications.py:66-69).  zlib inflates it at 0.37 GB/s of text on the GPU box -- a seventh of what one MI355X
// eats -- and nothing in the format says where a thread could start in the middle.  This file does what pugz /
// rapidgzip do:
//   1. cut the compressed bytes into chunks; every chunk but the first SEARCHES its range, bit by bit, for the start of
//      a dynamic-Huffman block (a header whose code lengths form complete prefix codes, followed by a trial decode that
//      yields only ASCII literals -- feature text is ASCII; a file that is not simply never finds a start and the caller
//      stays on the sequential reader);
//   2. the chunks are inflated concurrently by this file's own decoder into 16-bit symbols: a back-reference that reaches
//      before the chunk's first byte cannot be resolved yet (the 32 KiB window there is another thread's output) and is
//      kept as a MARKER 0x8000 | window position; markers are copied by later references like any other symbol;
//   3. a chunk stops at the first true block boundary at or after the next chunk's search start that begins a dynamic
//      block; if that is not the position the next chunk found, the next chunk started on a false positive: it and its
//      successors are dropped and the next round starts from the true boundary (costs time, never correctness);
//   4. the windows are resolved front to back (32 Ki symbols per chunk), then every chunk's markers in parallel, the
//      CRC-32 of every chunk in parallel and combined (crc32_combine); every member's CRC-32 and ISIZE are checked
//      against its trailer exactly as zlib does.  Nothing leaves this file unverified except through a member whose
//      trailer has not been reached yet -- the same contract as a streaming zlib reader.
// The reader thread pulls text with dsp_pgz_read like from dsp_gz_read; rounds of chunks are decoded ahead of it.
// DSP_GZ_SEQUENTIAL=1 keeps callers on the zlib reader (A/B switch).
#include "dsp_amd.h"
#include "dsp_threads.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

extern "C" void dsp_set_error_(const char* msg);

namespace {

constexpr uint32_t kWin = 32768;
// 16-bit symbols: 0..255 a byte; 0x8000 | p = the byte at position p of the unknown 32 KiB before the chunk (all 32,768
// positions are in use); kPoison = a position no valid stream can address (before the start of the member)
constexpr uint16_t kPoison = 0x4000;

// ---- bit reader over [base, base + size): LSB-first, 64-bit buffer ---------------------------------------------------
struct Bits {
    const uint8_t* base;
    size_t size;       // bytes
    size_t pos;        // next byte to load
    uint64_t buf = 0;
    int n = 0;         // valid bits in buf
    bool over = false; // tried to read past the end
    void seek(uint64_t bit) {
        pos = (size_t)(bit >> 3);
        buf = 0; n = 0; over = false;
        const int skip = (int)(bit & 7);
        if (skip) { refill(); buf >>= skip; n -= skip; }
    }
    uint64_t bitpos() const { return (uint64_t)pos * 8 - (uint64_t)n; }
    inline void refill() {
        if (pos + 8 <= size) {            // fast: unaligned 64-bit load, keep whole bytes only
            uint64_t v;
            memcpy(&v, base + pos, 8);
            buf |= v << n;
            const int take = (63 - n) >> 3;
            pos += (size_t)take;
            n += take * 8;
        } else {
            while (n <= 56 && pos < size) { buf |= (uint64_t)base[pos++] << n; n += 8; }
        }
    }
    inline uint32_t peek(int k) { return (uint32_t)(buf & ((1ull << k) - 1)); }
    inline void drop(int k) { buf >>= k; n -= k; }
    inline uint32_t get(int k) {           // k <= 32
        if (n < k) { refill(); if (n < k) { over = true; return 0; } }
        const uint32_t v = peek(k);
        drop(k);
        return v;
    }
    void align_byte() { const int r = n & 7; drop(r); }
};

// ---- canonical Huffman decoding tables: PB-bit primary table, secondary tables for longer codes ----------------------
// entry: bits 0..15 symbol (or secondary offset), 16..20 code length to drop (or primary bits for a link), 21 = link,
// 24..28 secondary index bits
struct Huff {
    static constexpr int PB = 10;
    std::vector<uint32_t> t;
    int maxlen = 0;
    bool build(const uint8_t* lens, int n, bool allow_incomplete_single) {
        int count[16] = {0};
        for (int i = 0; i < n; ++i) count[lens[i]]++;
        if (count[0] == n) { t.assign(1u << PB, 0xffffffffu); maxlen = 0; return allow_incomplete_single; }  // no codes
        maxlen = 15;
        while (maxlen > 0 && count[maxlen] == 0) --maxlen;
        int left = 1;
        for (int len = 1; len <= 15; ++len) {
            left <<= 1;
            left -= count[len];
            if (left < 0) return false;  // over-subscribed
        }
        if (left > 0) {  // incomplete: only "one code of length 1" is legal (zlib accepts it for distances)
            if (!(allow_incomplete_single && n - count[0] == 1 && count[1] == 1)) return false;
        }
        uint16_t next[16];
        uint16_t code = 0;
        count[0] = 0;
        for (int len = 1; len <= 15; ++len) { code = (uint16_t)((code + count[len - 1]) << 1); next[len] = code; }
        // secondary sizes per primary prefix
        t.assign(1u << PB, 0xffffffffu);
        std::vector<uint8_t> subbits;
        if (maxlen > PB) {
            subbits.assign(1u << PB, 0);
            uint16_t nx[16];
            memcpy(nx, next, sizeof(nx));
            for (int i = 0; i < n; ++i) {
                const int len = lens[i];
                if (len <= PB) { if (len) nx[len]++; continue; }
                const uint32_t c = nx[len]++;
                uint32_t rev = 0;
                for (int b = 0; b < len; ++b) rev |= ((c >> b) & 1u) << (len - 1 - b);
                const uint32_t pfx = rev & ((1u << PB) - 1);
                subbits[pfx] = (uint8_t)std::max<int>(subbits[pfx], len - PB);
            }
            for (uint32_t pfx = 0; pfx < (1u << PB); ++pfx)
                if (subbits[pfx]) {
                    const uint32_t off = (uint32_t)t.size();
                    if (off > 0xffff) return false;
                    t.resize(t.size() + (1u << subbits[pfx]), 0xffffffffu);
                    t[pfx] = off | ((uint32_t)PB << 16) | (1u << 21) | ((uint32_t)subbits[pfx] << 24);
                }
        }
        for (int i = 0; i < n; ++i) {
            const int len = lens[i];
            if (!len) continue;
            const uint32_t c = next[len]++;
            uint32_t rev = 0;
            for (int b = 0; b < len; ++b) rev |= ((c >> b) & 1u) << (len - 1 - b);
            if (len <= PB) {
                const uint32_t e = (uint32_t)i | ((uint32_t)len << 16);
                for (uint32_t k = rev; k < (1u << PB); k += 1u << len) t[k] = e;
            } else {
                const uint32_t pfx = rev & ((1u << PB) - 1);
                const uint32_t link = t[pfx];
                const uint32_t off = link & 0xffff, sb = link >> 24;
                const uint32_t e = (uint32_t)i | ((uint32_t)(len - PB) << 16);
                for (uint32_t k = rev >> PB; k < (1u << sb); k += 1u << (len - PB)) t[off + k] = e;
            }
        }
        return true;
    }
    // returns the symbol, or -1 (invalid code / out of input)
    inline int decode(Bits& b) const {
        if (b.n < 15) { b.refill(); }
        uint32_t e = t[b.peek(PB)];
        if (e == 0xffffffffu) return -1;
        if (e & (1u << 21)) {
            if (b.n < PB) { b.over = true; return -1; }
            b.drop(PB);
            const uint32_t sb = e >> 24;
            e = t[(e & 0xffff) + b.peek((int)sb)];
            if (e == 0xffffffffu) return -1;
        }
        const int len = (int)((e >> 16) & 31);
        if (b.n < len) { b.over = true; return -1; }
        b.drop(len);
        return (int)(e & 0xffff);
    }
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

// dynamic block header at the reader's position (after BFINAL / BTYPE): false on anything a deflate encoder cannot emit
bool read_dynamic_header(Bits& b, Huff& lit, Huff& dist) {
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const int hlit = (int)b.get(5) + 257, hdist = (int)b.get(5) + 1, hclen = (int)b.get(4) + 4;
    if (b.over || hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; ++i) cl[order[i]] = (uint8_t)b.get(3);
    if (b.over) return false;
    Huff clh;
    if (!clh.build(cl, 19, false)) return false;
    uint8_t lens[286 + 30];
    int i = 0;
    const int total = hlit + hdist;
    while (i < total) {
        const int s = clh.decode(b);
        if (s < 0) return false;
        if (s < 16) { lens[i++] = (uint8_t)s; continue; }
        int rep, val = 0;
        if (s == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + (int)b.get(2); }
        else if (s == 17) rep = 3 + (int)b.get(3);
        else rep = 11 + (int)b.get(7);
        if (b.over || i + rep > total) return false;
        while (rep--) lens[i++] = (uint8_t)val;
    }
    if (lens[256] == 0) return false;  // no end-of-block code
    if (!lit.build(lens, hlit, false)) return false;
    if (!dist.build(lens + hlit, hdist, true)) return false;
    return true;
}

struct FixedTables {
    Huff lit, dist;
    FixedTables() {
        uint8_t l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        lit.build(l, 288, false);
        uint8_t d[30];
        for (int i = 0; i < 30; ++i) d[i] = 5;
        dist.build(d, 30, true);   // 30 of 32 codes: incomplete by construction
        // (the fixed distance code is 5 bits for 32 values of which 30 are valid: build() rejects it as incomplete, so
        // fill it by hand)
        dist.t.assign(1u << Huff::PB, 0xffffffffu);
        for (uint32_t c = 0; c < 30; ++c) {
            uint32_t rev = 0;
            for (int bb = 0; bb < 5; ++bb) rev |= ((c >> bb) & 1u) << (4 - bb);
            for (uint32_t k = rev; k < (1u << Huff::PB); k += 32) dist.t[k] = c | (5u << 16);
        }
        dist.maxlen = 5;
    }
};
const FixedTables& fixed_tables() { static const FixedTables f; return f; }

// ---- one chunk: decode blocks into 16-bit symbols ---------------------------------------------------------------------
struct MemberEnd { size_t sym_pos; uint32_t crc, isize; };   // a member ended before symbol sym_pos (0-based after the window)

struct Chunk {
    uint64_t search_from = 0;     // bit position where the search for a block start begins (chunk 0 of a round: exact)
    uint64_t stop_after = 0;      // stop at the first dynamic-block boundary at or after this bit position
    bool exact_start = false;
    bool found = false;
    uint64_t start_bit = 0, end_bit = 0;
    bool hit_eof = false;         // consumed the whole file
    bool failed = false;          // structural error while decoding (corrupt or truncated stream, or a false start)
    std::string why;
    std::vector<uint16_t> sym;    // [kWin window][output...]
    std::vector<MemberEnd> ends;
    std::vector<uint8_t> out;     // resolved bytes
    uint32_t crc_tail = 0;        // crc of the bytes after the last member end inside this chunk
    std::vector<uint32_t> crc_parts;  // crc of each piece between member ends (piece k ends at ends[k])
};

inline bool is_textish(int lit) { return lit < 128; }

// gzip member header at byte position p; returns the position after it or 0
size_t skip_gzip_header(const uint8_t* base, size_t size, size_t p) {
    if (p + 10 > size || base[p] != 0x1f || base[p + 1] != 0x8b || base[p + 2] != 8) return 0;
    const uint8_t flg = base[p + 3];
    if (flg & 0xe0) return 0;
    size_t q = p + 10;
    if (flg & 4) { if (q + 2 > size) return 0; const size_t xlen = base[q] | (size_t)base[q + 1] << 8; q += 2 + xlen; }
    if (flg & 8) { while (q < size && base[q]) ++q; ++q; }
    if (flg & 16) { while (q < size && base[q]) ++q; ++q; }
    if (flg & 2) q += 2;
    return q <= size ? q : 0;
}

// The symbol loop of one Huffman block with the bit buffer in registers: one refill (>= 48 valid bits) covers the longest
// symbol (15 + 5 + 15 + 13 bits).  Works while 16 input bytes remain; returns 1 to let the careful loop finish the tail.
// 0 = end of block, 1 = continue with the careful loop, -1 invalid literal/length code, -2 invalid distance code, -3 too far
inline int fast_symbols(const uint8_t* base, size_t size, Bits& b, const Huff& L, const Huff& D, std::vector<uint16_t>& o, size_t& opos) {
    uint64_t buf = b.buf;
    int n = b.n;
    size_t pos = b.pos;
    const uint32_t* lt = L.t.data();
    const uint32_t* dt = D.t.data();
    constexpr uint32_t PM = (1u << Huff::PB) - 1;
    int rc = 1;
    for (;;) {
        if (pos + 16 > size) break;
        if (opos + 320 > o.size()) o.resize(o.size() * 2);
        uint16_t* out = o.data();
        // a burst: symbols until the output buffer must be checked again
        const size_t lim = o.size() - 300;
        while (opos < lim && pos + 16 <= size) {
            if (n < 48) {
                uint64_t v;
                memcpy(&v, base + pos, 8);
                buf |= v << n;
                const int take = (63 - n) >> 3;
                pos += (size_t)take;
                n += take * 8;
            }
            uint32_t e = lt[buf & PM];
            if (e & (1u << 21)) {
                if (e == 0xffffffffu) { rc = -1; goto done; }
                buf >>= Huff::PB; n -= Huff::PB;
                e = lt[(e & 0xffff) + (uint32_t)(buf & ((1u << (e >> 24)) - 1))];
            }
            if (e == 0xffffffffu) { rc = -1; goto done; }
            int len = (int)((e >> 16) & 31);
            buf >>= len; n -= len;
            uint32_t sym = e & 0xffff;
            if (sym < 256) { out[opos++] = (uint16_t)sym; continue; }
            if (sym == 256) { rc = 0; goto done; }
            if (sym > 285) { rc = -1; goto done; }
            const int li = (int)sym - 257;
            uint32_t mlen = kLenBase[li];
            const int xe = kLenExtra[li];
            mlen += (uint32_t)(buf & ((1u << xe) - 1));
            buf >>= xe; n -= xe;
            e = dt[buf & PM];
            if (e & (1u << 21)) {
                if (e == 0xffffffffu) { rc = -2; goto done; }
                buf >>= Huff::PB; n -= Huff::PB;
                e = dt[(e & 0xffff) + (uint32_t)(buf & ((1u << (e >> 24)) - 1))];
            }
            if (e == 0xffffffffu) { rc = -2; goto done; }
            len = (int)((e >> 16) & 31);
            buf >>= len; n -= len;
            const uint32_t ds = e & 0xffff;
            if (ds > 29) { rc = -2; goto done; }
            const int de = kDistExtra[ds];
            const uint32_t d = kDistBase[ds] + (uint32_t)(buf & ((1u << de) - 1));
            buf >>= de; n -= de;
            if (d > opos) { rc = -3; goto done; }
            uint16_t* w = out + opos;
            const uint16_t* r = w - d;
            if (d >= 4) { for (uint32_t i = 0; i < mlen; i += 4) { w[i] = r[i]; w[i + 1] = r[i + 1]; w[i + 2] = r[i + 2]; w[i + 3] = r[i + 3]; } }
            else { for (uint32_t i = 0; i < mlen; ++i) w[i] = r[i]; }
            opos += mlen;
        }
    }
done:
    b.buf = buf; b.n = n; b.pos = pos;
    return rc;
}

// Decode from c.start_bit.  markers = the 32 KiB before the chunk are unknown (sym[0..kWin) pre-filled with markers by the
// caller) -- otherwise pre-filled with real bytes.  Stops at a block boundary >= stop_after that starts a dynamic block,
// at the end of the file, or after max_out symbols at the next block boundary.
void decode_chunk(const uint8_t* base, size_t size, Chunk& c, size_t max_out) {
    Bits b{base, size, 0};
    b.seek(c.start_bit);
    Huff lit, dist;
    std::vector<uint16_t>& o = c.sym;   // sized generously; opos = symbols written (window included)
    size_t opos = kWin;
    if (o.size() < kWin + 4096) o.resize(kWin + 4096);
    auto fail = [&](const char* why) { c.failed = true; c.why = why; c.end_bit = b.bitpos(); o.resize(opos); };
    const char* kTrunc = "truncated gzip stream: Compressed file ended before the end-of-stream marker was reached";
    auto room = [&](size_t extra) { if (opos + extra + 8 > o.size()) o.resize(std::max(o.size() * 2, opos + extra + 4096)); };
    for (;;) {
        // at a block boundary
        const uint64_t here = b.bitpos();
        if (b.n < 3) b.refill();
        if (b.n < 3) return fail(kTrunc);
        const uint32_t hdr = b.peek(3);
        const int bfinal = (int)(hdr & 1), btype = (int)(hdr >> 1);
        if (here != c.start_bit && btype == 2 && (here >= c.stop_after || opos - kWin >= max_out)) { c.end_bit = here; o.resize(opos); return; }
        // (a single block can inflate without bound -- 1 GB of zeros is 1 MB of deflate -- and this decoder holds a chunk's
        // output in memory: past 2^31 symbols give up loudly; the sequential reader streams such a file in constant space)
        if (opos > (1ull << 31)) return fail("parallel inflate: a chunk of this stream inflates beyond 2 GiB; read it with DSP_GZ_SEQUENTIAL=1");
        b.drop(3);
        if (btype == 3) return fail("corrupt gzip stream: invalid block type");
        if (btype == 0) {
            b.align_byte();
            const uint32_t len = b.get(16), nlen = b.get(16);
            if (b.over || (len ^ 0xffffu) != nlen) return fail("corrupt gzip stream: invalid stored block lengths");
            room(len);
            for (uint32_t i = 0; i < len; ++i) {
                const uint32_t v = b.get(8);
                if (b.over) return fail(kTrunc);
                o[opos++] = (uint16_t)v;
            }
        } else {
            const Huff* L;
            const Huff* D;
            if (btype == 1) { L = &fixed_tables().lit; D = &fixed_tables().dist; }
            else {
                if (!read_dynamic_header(b, lit, dist)) return fail(b.over ? kTrunc : "corrupt gzip stream: invalid code lengths set");
                L = &lit; D = &dist;
            }
            for (;;) {
                const int rc = fast_symbols(base, size, b, *L, *D, o, opos);
                if (rc == 0) break;
                if (rc == -1) return fail("corrupt gzip stream: invalid literal/length code");
                if (rc == -2) return fail("corrupt gzip stream: invalid distance code");
                if (rc == -3) return fail("corrupt gzip stream: invalid distance too far back");
                // the last 16 bytes of the file: one careful symbol at a time
                room(300);
                const int s = L->decode(b);
                if (s < 0) return fail(b.over ? kTrunc : "corrupt gzip stream: invalid literal/length code");
                if (s < 256) { o[opos++] = (uint16_t)s; continue; }
                if (s == 256) break;
                if (s > 285) return fail("corrupt gzip stream: invalid literal/length code");
                const int li = s - 257;
                uint32_t len = kLenBase[li];
                if (kLenExtra[li]) len += b.get(kLenExtra[li]);
                const int ds = D->decode(b);
                if (ds < 0 || ds > 29) return fail(b.over ? kTrunc : "corrupt gzip stream: invalid distance code");
                uint32_t d = kDistBase[ds];
                if (kDistExtra[ds]) d += b.get(kDistExtra[ds]);
                if (b.over) return fail(kTrunc);
                if (d > opos) return fail("corrupt gzip stream: invalid distance too far back");
                uint16_t* w = o.data() + opos;
                const uint16_t* r = w - d;
                for (uint32_t i = 0; i < len; ++i) w[i] = r[i];   // overlapping copies replicate, as deflate requires
                opos += len;
            }
        }
        if (bfinal) {   // member trailer, then the next member (or padding / the end of the file)
            b.align_byte();
            size_t p = (size_t)(b.bitpos() >> 3);
            if (p + 8 > size) return fail(kTrunc);
            MemberEnd me;
            me.sym_pos = opos - kWin;
            me.crc = base[p] | (uint32_t)base[p + 1] << 8 | (uint32_t)base[p + 2] << 16 | (uint32_t)base[p + 3] << 24;
            me.isize = base[p + 4] | (uint32_t)base[p + 5] << 8 | (uint32_t)base[p + 6] << 16 | (uint32_t)base[p + 7] << 24;
            c.ends.push_back(me);
            p += 8;
            while (p < size && base[p] == 0) ++p;   // zero padding between / after members
            if (p >= size) { c.hit_eof = true; c.end_bit = (uint64_t)size * 8; o.resize(opos); return; }
            const size_t q = skip_gzip_header(base, size, p);
            if (!q) return fail("corrupt gzip stream: not a gzip member where one must start");
            // (a new member starts with an empty window; a reference reaching back across its start is a corrupt stream,
            // which the member's CRC-32 catches)
            b.seek((uint64_t)q * 8);
        }
    }
}

// Is there a plausible dynamic block at bit position `bit`?  Header must parse; the first symbols must be ASCII literals /
// sane matches; for a candidate in the middle of a stream BFINAL is almost always 0 but 1 is legal.
bool plausible_block_start(const uint8_t* base, size_t size, uint64_t bit) {
    Bits b{base, size, 0};
    b.seek(bit);
    if (b.n < 3) b.refill();
    if (b.n < 3) return false;
    const uint32_t hdr = b.get(3);
    if ((hdr >> 1) != 2) return false;
    Huff lit, dist;
    if (!read_dynamic_header(b, lit, dist)) return false;
    int nsym = 0, nlit = 0;
    for (; nsym < 3000; ++nsym) {
        const int s = lit.decode(b);
        if (s < 0) return b.over && nsym > 200;   // ran into the end of the file while everything looked fine
        if (s < 256) { if (!is_textish(s)) return false; ++nlit; continue; }
        if (s == 256) break;
        if (s > 285) return false;
        const int li = s - 257;
        if (kLenExtra[li]) b.get(kLenExtra[li]);
        const int ds = dist.decode(b);
        if (ds < 0 || ds > 29) return false;
        if (kDistExtra[ds]) b.get(kDistExtra[ds]);
        if (b.over) return false;
    }
    return nsym >= 16 || nlit > 0 || nsym > 0;
}

struct LibCrc {
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
    LibCrc() {
        if (getenv("DSP_GZ_ZLIB")) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (h) crc = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
    }
};
uint32_t fast_crc32(const uint8_t* p, size_t n) {
    static const LibCrc lib;
    if (lib.crc) return lib.crc(0, p, n);
    uint32_t c = (uint32_t)crc32(0L, Z_NULL, 0);
    while (n) { const uInt k = (uInt)std::min<size_t>(n, 1u << 30); c = (uint32_t)crc32(c, p, k); p += k; n -= k; }
    return c;
}

}  // namespace

struct dsp_pgz {
    int fd = -1;
    const uint8_t* map = nullptr;
    size_t size = 0;
    int nthreads = 4;
    size_t chunk_bytes = 8u << 20;
    // decoding state (owned by the decoder thread)
    uint64_t next_bit = 0;             // a true block boundary: where the next round starts
    std::vector<uint16_t> window;      // the 32 Ki symbols before next_bit, resolved (kPoison = before the stream's start)
    std::string pending_error;
    uint32_t member_crc = 0;           // running CRC-32 / size of the member being decoded
    uint64_t member_len = 0;
    bool finished = false;
    // queue of decoded text towards the reader
    std::mutex mu;
    std::condition_variable cv_put, cv_get;
    std::deque<std::vector<uint8_t>> q;
    size_t q_bytes = 0;
    bool eof = false;
    int status = 0;
    std::string error;
    bool stop = false;
    std::thread decoder;
    // reader side
    std::vector<uint8_t> cur;
    size_t cur_pos = 0;
    std::atomic<uint64_t> bytes_in{0};
    std::atomic<uint64_t> rounds{0}, dropped_chunks{0};
    int strikes = 0, solo_rounds = 0;   // rounds in which no chunk found a start -> a spell of one-chunk rounds (non-text data)
};

namespace {

void pgz_fail(dsp_pgz* z, int code, const std::string& msg) {
    std::lock_guard<std::mutex> lk(z->mu);
    if (!z->status) {   // the first error is the one that is reported
        z->status = code;
        z->error = msg;
    }
    z->eof = true;
    z->cv_get.notify_all();
}

void run_parallel(int nthreads, int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    std::atomic<int> next{0};
    // (an exception must not leave a std::thread, and a thread the system refuses is work done here: dsp_threads.h;
    // rethrown after the join)
    if (!dsp::run_indexed(std::min(nthreads, n), [&](int) { for (int i; (i = next.fetch_add(1)) < n;) fn(i); }))
        throw std::runtime_error("out of memory");
}

// The second half of a round, run by a finisher thread while the decoder threads are already inflating the next round:
// every kept chunk's markers resolved against the window before it (parallel), CRC-32 of its pieces (parallel), then -- in
// stream order -- the members' CRC-32 / ISIZE checked against their trailers and the text handed to the reader.
struct Round {
    std::vector<Chunk> ch;                       // the kept chunks
    std::vector<std::vector<uint16_t>> win;      // win[i] = the 32 Ki symbols before chunk i, resolved
    bool last = false;                           // the stream ends with this round
};

void finish_round(dsp_pgz* z, Round* r) {
    const int keep = (int)r->ch.size();
    std::atomic<int> bad{0};
    run_parallel(std::max(1, z->nthreads), keep, [&](int i) {
        Chunk& c = r->ch[(size_t)i];
        const size_t m = c.sym.size() - kWin;
        c.out.resize(m);
        const uint16_t* s = c.sym.data() + kWin;
        const uint16_t* w = r->win[(size_t)i].data();
        uint8_t* o = c.out.data();
        uint32_t any = 0;
        for (size_t k = 0; k < m; ++k) {
            uint32_t v = s[k];
            if (v & 0x8000u) v = w[v & 0x7fffu];
            any |= v;
            o[k] = (uint8_t)v;
        }
        if (any > 255) bad.store(1);   // a marker that led to a position before the start of the member, or to nothing
        std::vector<uint16_t>().swap(c.sym);
        size_t a = 0;
        for (const MemberEnd& me : c.ends) { c.crc_parts.push_back(fast_crc32(o + a, me.sym_pos - a)); a = me.sym_pos; }
        c.crc_tail = fast_crc32(o + a, m - a);
    });
    if (bad.load()) { pgz_fail(z, DSP_EPARSE, "corrupt gzip stream: invalid distance too far back"); return; }
    for (int i = 0; i < keep; ++i) {
        Chunk& c = r->ch[(size_t)i];
        size_t a = 0;
        for (size_t e = 0; e < c.ends.size(); ++e) {
            const size_t len = c.ends[e].sym_pos - a;
            z->member_crc = (uint32_t)crc32_combine(z->member_crc, c.crc_parts[e], (z_off_t)len);
            z->member_len += len;
            if (z->member_crc != c.ends[e].crc) { pgz_fail(z, DSP_EPARSE, "corrupt gzip stream: incorrect data check"); return; }
            if ((uint32_t)z->member_len != c.ends[e].isize) { pgz_fail(z, DSP_EPARSE, "corrupt gzip stream: incorrect length check"); return; }
            z->member_crc = 0; z->member_len = 0;
            a = c.ends[e].sym_pos;
        }
        const size_t tail = c.out.size() - a;
        z->member_crc = (uint32_t)crc32_combine(z->member_crc, c.crc_tail, (z_off_t)tail);
        z->member_len += tail;
        std::unique_lock<std::mutex> lk(z->mu);
        z->cv_put.wait(lk, [&] { return z->stop || z->q_bytes < (size_t)z->nthreads * z->chunk_bytes * 6; });
        if (z->stop) return;
        z->q_bytes += c.out.size();
        z->q.emplace_back(std::move(c.out));
        z->cv_get.notify_all();
    }
    if (r->last) {
        if (z->member_len != 0) { pgz_fail(z, DSP_EPARSE, "truncated gzip stream: Compressed file ended before the end-of-stream marker was reached"); return; }
        std::lock_guard<std::mutex> lk(z->mu);
        z->eof = true;
        z->cv_get.notify_all();
    }
}

// The first half of a round: up to nthreads chunks inflated concurrently from z->next_bit, the consistent prefix kept, the
// windows resolved front to back (32 Ki symbols per chunk: cheap, and the next round's first chunk needs the last one).
// Returns the round to finish, or NULL when the stream has failed.
Round* decode_round(dsp_pgz* z) {
    const uint8_t* base = z->map;
    const size_t size = z->size;
    const int T = std::max(1, z->nthreads);
    const uint64_t start_byte = z->next_bit >> 3;
    std::vector<Chunk> ch((size_t)T);
    int n = 0;
    // data that is not text never offers a block start: after two rounds in which no chunk found one, a spell of
    // one-chunk rounds (the searching threads would only make the true decoder wait for them), then another try
    const int Tn = z->solo_rounds > 0 ? 1 : T;
    if (z->solo_rounds > 0) --z->solo_rounds;
    for (; n < Tn; ++n) {
        const uint64_t from = n == 0 ? z->next_bit : (start_byte + (uint64_t)n * z->chunk_bytes) * 8;
        if (n > 0 && (from >> 3) + 64 >= size) break;
        ch[(size_t)n].search_from = from;
        ch[(size_t)n].exact_start = n == 0;
    }
    // a chunk stops at the first dynamic-block boundary at or after the end of its own share of the compressed bytes
    for (int i = 0; i < n; ++i)
        ch[(size_t)i].stop_after = i + 1 < n ? ch[(size_t)i + 1].search_from : (start_byte + (uint64_t)(i + 1) * z->chunk_bytes) * 8;
    const size_t max_out = z->chunk_bytes * 40;   // a chunk that inflates beyond 40x its share yields at the next boundary
    run_parallel(T, n, [&](int i) {
        Chunk& c = ch[(size_t)i];
        if (c.exact_start) { c.found = true; c.start_bit = c.search_from; }
        else {
            // a deflate block of text is tens of KiB (zlib: <= 32 Ki symbols): no start within 2 MiB means there is none to find
            const uint64_t limit = std::min<uint64_t>((uint64_t)size * 8, c.search_from + (uint64_t)std::min<size_t>(z->chunk_bytes, 2u << 20) * 8);
            for (uint64_t bit = c.search_from; bit + 64 < limit; ++bit)
                if (plausible_block_start(base, size, bit)) { c.found = true; c.start_bit = bit; break; }
        }
        if (!c.found) return;
        c.sym.resize(kWin + (size_t)(z->chunk_bytes * 3));   // decode_chunk trims it to what was written
        if (c.exact_start) memcpy(c.sym.data(), z->window.data(), kWin * sizeof(uint16_t));   // known: bytes (or kPoison)
        else for (uint32_t k = 0; k < kWin; ++k) c.sym[k] = (uint16_t)(0x8000u | k);
        decode_chunk(base, size, c, max_out);
    });
    // keep the consistent prefix: chunk i+1 must have started exactly where chunk i stopped
    int keep = 0;
    std::string fail_why;
    for (int i = 0; i < n; ++i) {
        Chunk& c = ch[(size_t)i];
        if (!c.found) break;
        if (i > 0 && ch[(size_t)i - 1].end_bit != c.start_bit) break;
        if (c.failed) {
            // chunk 0 decoded a verified prefix of the stream: its failure is the stream's.  A later chunk may have
            // started on a false positive: drop it, the next round re-decodes from the true boundary.
            if (i == 0) { fail_why = c.why; keep = -1; }
            break;
        }
        keep = i + 1;
        if (c.hit_eof) break;
    }
    if (getenv("DSP_PGZ_DEBUG")) {
        fprintf(stderr, "[pgz] round %llu: n=%d keep=%d", (unsigned long long)z->rounds.load(), n, keep);
        for (int i = 0; i < n; ++i)
            fprintf(stderr, " | %d: found=%d start=%llu end=%llu out=%zu failed=%d eof=%d", i, (int)ch[(size_t)i].found,
                    (unsigned long long)ch[(size_t)i].start_bit, (unsigned long long)ch[(size_t)i].end_bit,
                    ch[(size_t)i].sym.size(), (int)ch[(size_t)i].failed, (int)ch[(size_t)i].hit_eof);
        fprintf(stderr, "\n");
    }
    if (keep < 0) { z->pending_error = fail_why; return nullptr; }
    if (keep == 0) { z->pending_error = "corrupt gzip stream: no progress"; return nullptr; }
    z->dropped_chunks += (uint64_t)(n - keep);
    ++z->rounds;
    if (n > 1 && keep == 1 && !ch[0].hit_eof) { if (++z->strikes >= 2) { z->solo_rounds = 32; z->strikes = 0; } }
    else if (n > 1) z->strikes = 0;
    Round* r = new Round();
    r->win.resize((size_t)keep + 1);
    r->win[0] = z->window;
    for (int i = 1; i <= keep; ++i) {   // win[i] = the resolved tail of chunk i-1 (kPoison travels like a byte)
        const Chunk& p = ch[(size_t)i - 1];
        const std::vector<uint16_t>& pw = r->win[(size_t)i - 1];
        std::vector<uint16_t>& w = r->win[(size_t)i];
        w.resize(kWin);
        const size_t total = p.sym.size();   // window + output
        for (uint32_t k = 0; k < kWin; ++k) {
            uint16_t v = p.sym[total - kWin + k];
            if (v & 0x8000u) v = pw[v & 0x7fffu];
            w[k] = v;
        }
    }
    z->window = r->win[(size_t)keep];
    const Chunk& last = ch[(size_t)keep - 1];
    z->next_bit = last.end_bit;
    z->bytes_in.store((uint64_t)std::min<uint64_t>(z->size, (last.end_bit + 7) >> 3));
    r->last = last.hit_eof;
    r->ch.reserve((size_t)keep);
    for (int i = 0; i < keep; ++i) r->ch.emplace_back(std::move(ch[(size_t)i]));
    return r;
}

void decoder_main(dsp_pgz* z) {
    std::thread finisher;
    auto join_finisher = [&] { if (finisher.joinable()) finisher.join(); };
    try {
        for (;;) {
            {
                std::lock_guard<std::mutex> lk(z->mu);
                if (z->stop || z->status) break;
            }
            Round* r = decode_round(z);
            join_finisher();                 // rounds finish in order; at most one round waits while the next is inflated
            if (!r) {                        // what was decoded before the failure has been delivered: now the error
                pgz_fail(z, DSP_EPARSE, z->pending_error);
                break;
            }
            const bool last = r->last;
            auto finish = [z, r] {
                try { finish_round(z, r); }
                catch (const std::exception& e) { pgz_fail(z, DSP_ENOMEM, std::string("parallel inflate: ") + e.what()); }
                delete r;
            };
            try { finisher = std::thread(finish); }
            catch (const std::system_error&) { finish(); }   // (no thread to be had: the round is finished here)
            if (last) break;
        }
    } catch (const std::exception& e) {
        pgz_fail(z, DSP_ENOMEM, std::string("parallel inflate: ") + e.what());
    }
    join_finisher();
}

}  // namespace

extern "C" {

// NULL (with dsp_last_error set) when the file cannot be opened or does not start with a gzip member header
dsp_pgz* dsp_pgz_open(const char* path, int32_t nthreads, uint64_t chunk_bytes) {
    if (!path) return nullptr;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { dsp_set_error_("dsp_pgz_open: cannot open the file"); return nullptr; }
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 18) { close(fd); dsp_set_error_("dsp_pgz_open: not a gzip file"); return nullptr; }
    void* m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { close(fd); dsp_set_error_("dsp_pgz_open: cannot map the file"); return nullptr; }
    const size_t q = skip_gzip_header((const uint8_t*)m, (size_t)sb.st_size, 0);
    if (!q) { munmap(m, (size_t)sb.st_size); close(fd); dsp_set_error_("dsp_pgz_open: not a gzip file"); return nullptr; }
    dsp_pgz* z = new (std::nothrow) dsp_pgz();
    if (!z) { munmap(m, (size_t)sb.st_size); close(fd); return nullptr; }
    z->fd = fd; z->map = (const uint8_t*)m; z->size = (size_t)sb.st_size;
    z->nthreads = nthreads < 1 ? 1 : nthreads;
    if (chunk_bytes >= (1u << 16)) z->chunk_bytes = (size_t)chunk_bytes;
    z->next_bit = (uint64_t)q * 8;
    if (chunk_bytes < (1u << 16) && z->nthreads >= 8) z->chunk_bytes = 4u << 20;   // two rounds are in flight: keep their memory bounded
    try {
        z->window.assign(kWin, kPoison);
        z->decoder = std::thread(decoder_main, z);
    } catch (const std::exception&) {   // (no memory / no thread to be had: an error, not std::terminate across the C ABI)
        dsp_pgz_close(z);
        dsp_set_error_("dsp_pgz_open: cannot start the decoder thread");
        return nullptr;
    }
    return z;
}

// up to cap bytes of text; 0 at the end; DSP_EPARSE on a corrupt or truncated stream (same messages as dsp_gz_read)
int64_t dsp_pgz_read(dsp_pgz* z, uint8_t* out, size_t cap) {
    if (!z || !out) { dsp_set_error_("dsp_pgz_read: NULL argument"); return DSP_EINVAL; }
    size_t got = 0;
    while (got < cap) {
        if (z->cur_pos >= z->cur.size()) {
            std::unique_lock<std::mutex> lk(z->mu);
            z->cv_get.wait(lk, [&] { return !z->q.empty() || z->eof; });
            if (z->q.empty()) {
                if (z->status) { dsp_set_error_(z->error.c_str()); return z->status; }
                break;
            }
            z->cur = std::move(z->q.front());
            z->q.pop_front();
            z->q_bytes -= z->cur.size();
            z->cur_pos = 0;
            z->cv_put.notify_all();
            continue;
        }
        const size_t k = std::min(cap - got, z->cur.size() - z->cur_pos);
        memcpy(out + got, z->cur.data() + z->cur_pos, k);
        z->cur_pos += k;
        got += k;
    }
    return (int64_t)got;
}

uint64_t dsp_pgz_bytes_in(const dsp_pgz* z) { return z ? z->bytes_in.load() : 0; }
// rounds decoded / chunks dropped because their start was a false positive (diagnostics, tests)
void dsp_pgz_stats(const dsp_pgz* z, uint64_t* rounds, uint64_t* dropped) {
    if (rounds) *rounds = z ? z->rounds.load() : 0;
    if (dropped) *dropped = z ? z->dropped_chunks.load() : 0;
}

void dsp_pgz_close(dsp_pgz* z) {
    if (!z) return;
    {
        std::lock_guard<std::mutex> lk(z->mu);
        z->stop = true;
        z->cv_put.notify_all();
    }
    if (z->decoder.joinable()) z->decoder.join();
    if (z->map) munmap(const_cast<uint8_t*>(z->map), z->size);
    if (z->fd >= 0) close(z->fd);
    delete z;
}

}  // extern "C"
