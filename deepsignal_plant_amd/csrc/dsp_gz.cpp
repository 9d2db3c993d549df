// dsp_gz.cpp -- gzip I/O of the call_mods host pipeline (replaces the reference's `gzip.open(...)` reader / writers,
// deepsignal_plant/call_modifications.py:66-69, :264-270; extract_features.py writers).
//
// A deflate stream cannot be entered in the middle, so ONE gzip member can only be inflated by one thread (about
// 0.3-0.4 GB/s of text with zlib: a sixth of what the parser and the GPU take).  Everything this build WRITES with
// --gzip is therefore a chain of BGZF members (the blocked gzip of htslib / bgzip: ordinary gzip members of at most
// 64 KiB of text, each carrying its compressed size in a "BC" extra field, closed by an empty EOF member): any gzip
// reader -- the reference's included -- reads it as one stream, while this reader walks the member headers, deals
// contiguous member ranges to ranks and inflates a batch of members on N threads.  Foreign .gz files (one member, no
// "BC" fields) fall back to a native streaming inflate (dsp_gz_open / dsp_gz_read), single-threaded by nature.
#include "dsp_amd.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "dsp_threads.h"

extern "C" void dsp_set_error_(const char* msg);

namespace {

int gz_fail(int code, const char* fmt, long long a = 0, long long b = 0) {
    char buf[256];
    snprintf(buf, sizeof(buf), fmt, a, b);
    dsp_set_error_(buf);
    return code;
}

constexpr size_t kBgzfBlock = 0xff00;  // text bytes per member (htslib's BGZF_BLOCK_SIZE)

// libdeflate (whole-buffer deflate / inflate, 2-3x zlib's speed) is used for the BGZF members when the shared library
// is present on the box (the image ships libdeflate.so.0 without headers: bound at run time, its C ABI is stable);
// zlib otherwise, and always for the streaming reader.  DSP_GZ_ZLIB=1 forces zlib (A/B, tests).
struct LibDeflate {
    void* (*alloc_c)(int) = nullptr;
    size_t (*compress)(void*, const void*, size_t, void*, size_t) = nullptr;
    void (*free_c)(void*) = nullptr;
    void* (*alloc_d)() = nullptr;
    int (*decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*free_d)(void*) = nullptr;
    uint32_t (*crc32)(uint32_t, const void*, size_t) = nullptr;
    bool ok = false;
};
const LibDeflate& libdeflate() {
    static LibDeflate ld;
    static std::once_flag once;
    std::call_once(once, [] {
        if (getenv("DSP_GZ_ZLIB")) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        ld.alloc_c = (void* (*)(int))dlsym(h, "libdeflate_alloc_compressor");
        ld.compress = (size_t (*)(void*, const void*, size_t, void*, size_t))dlsym(h, "libdeflate_deflate_compress");
        ld.free_c = (void (*)(void*))dlsym(h, "libdeflate_free_compressor");
        ld.alloc_d = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
        ld.decompress = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_deflate_decompress");
        ld.free_d = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
        ld.crc32 = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
        ld.ok = ld.alloc_c && ld.compress && ld.free_c && ld.alloc_d && ld.decompress && ld.free_d && ld.crc32;
    });
    return ld;
}

// total size of the BGZF member starting at p (0 = not a BGZF member)
size_t bgzf_member_size(const uint8_t* p, size_t left) {
    if (left < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
    const size_t xlen = p[10] | (size_t)p[11] << 8;
    if (left < 12 + xlen) return 0;
    for (size_t o = 12; o + 4 <= 12 + xlen;) {
        const size_t slen = p[o + 2] | (size_t)p[o + 3] << 8;
        if (p[o] == 'B' && p[o + 1] == 'C' && slen == 2 && o + 6 <= 12 + xlen) {
            const size_t bs = (p[o + 4] | (size_t)p[o + 5] << 8) + 1;
            return bs >= 12 + xlen + 8 && bs <= left ? bs : 0;
        }
        o += 4 + slen;
    }
    return 0;
}

int inflate_member(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* produced) {
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (inflateInit2(&z, 15 + 16) != Z_OK) return -1;  // gzip wrapper: header + CRC32 + ISIZE are checked by zlib
    z.next_in = const_cast<Bytef*>(src); z.avail_in = (uInt)n;
    z.next_out = dst; z.avail_out = (uInt)cap;
    const int rc = inflate(&z, Z_FINISH);
    *produced = (size_t)z.total_out;
    inflateEnd(&z);
    return rc == Z_STREAM_END ? 0 : -1;
}

}  // namespace

// Streaming reader of a foreign .gz: the file is mmap'ed and run through zlib's inflate (gzip wrapper: header, CRC32
// and ISIZE of every member checked), member after member like gzip.open.  Not gzread: that reports a stream cut in
// the middle as a clean end of file (0 bytes + Z_BUF_ERROR) -- a half-copied feature file must fail, as it does under
// the reference's gzip.open (EOFError).
struct dsp_gz_stream {
    int fd = -1;
    const uint8_t* map = nullptr;
    size_t size = 0, pos = 0;       // pos = compressed bytes handed to zlib so far
    z_stream z;
    bool z_live = false;            // inside a member
    bool done = false;              // clean end of the file reached
    uint64_t bytes_out = 0;
};

extern "C" {

int64_t dsp_gz_index(const uint8_t* src, size_t len, int64_t max_members, uint64_t* member_off, uint32_t* member_isize) {
    if (!src) return gz_fail(DSP_EINVAL, "dsp_gz_index: NULL argument");
    int64_t m = 0;
    size_t pos = 0;
    while (pos < len) {
        const size_t bs = bgzf_member_size(src + pos, len - pos);
        if (!bs) return -1;  // not (entirely) BGZF: the caller streams it instead
        if (member_off) {
            if (m >= max_members) return gz_fail(DSP_ENOMEM, "dsp_gz_index: more than %lld members", (long long)max_members);
            member_off[m] = pos;
            const uint8_t* t = src + pos + bs - 4;
            member_isize[m] = t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
        }
        ++m;
        pos += bs;
    }
    if (member_off && m <= max_members) member_off[m] = pos;
    return m;
}

// newline count of every BGZF member as its writer recorded it (dsp_bgzf_compress: MTIME under XFL = 'R'); -1 for members
// of other writers (bgzip, older files of this build): the caller inflates those to count
int32_t dsp_gz_member_rows(const uint8_t* src, const uint64_t* member_off, int64_t n_members, int64_t* rows) {
    if (!src || !member_off || !rows || n_members < 0) return (int32_t)gz_fail(DSP_EINVAL, "dsp_gz_member_rows: bad argument");
    for (int64_t m = 0; m < n_members; ++m) {
        const uint8_t* p = src + member_off[m];
        const uint32_t isize_at = (uint32_t)(member_off[m + 1] - member_off[m]);
        const uint8_t* t = p + isize_at - 4;
        const uint32_t isize = t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
        const uint32_t nl = p[4] | (uint32_t)p[5] << 8 | (uint32_t)p[6] << 16 | (uint32_t)p[7] << 24;
        rows[m] = isize == 0 ? 0 : ((p[8] == 'R' && nl <= isize) ? (int64_t)nl : -1);   // an empty member (end of file) has none
    }
    return DSP_OK;
}

int64_t dsp_gz_inflate_members(const uint8_t* src, const uint64_t* member_off, const uint32_t* member_isize, int64_t m0,
                               int64_t m1, uint8_t* out, size_t out_cap, int32_t nthreads) {
    if (!src || !member_off || !member_isize || !out || m0 < 0 || m1 < m0) return gz_fail(DSP_EINVAL, "dsp_gz_inflate_members: bad argument");
    const int64_t m = m1 - m0;
    std::vector<size_t> off((size_t)m + 1, 0);
    for (int64_t i = 0; i < m; ++i) off[(size_t)i + 1] = off[(size_t)i] + member_isize[m0 + i];
    if (off[(size_t)m] > out_cap) return gz_fail(DSP_ENOMEM, "dsp_gz_inflate_members: %lld bytes do not fit %lld", (long long)off[(size_t)m], (long long)out_cap);
    int nt = nthreads < 1 ? 1 : nthreads;
    if ((int64_t)nt > m) nt = (int)std::max<int64_t>(1, m);
    std::atomic<int64_t> next(0), bad(-1);
    const LibDeflate& ld = libdeflate();
    auto work = [&]() {
        void* dec = ld.ok ? ld.alloc_d() : nullptr;
        for (;;) {
            const int64_t i = next.fetch_add(16);
            if (i >= m || bad.load() >= 0) break;
            for (int64_t j = i; j < std::min(m, i + 16); ++j) {
                size_t got = 0;
                const uint64_t a = member_off[m0 + j], b = member_off[m0 + j + 1];
                const uint32_t isz = member_isize[m0 + j];
                bool good;
                if (dec && src[a + 3] == 4) {  // BGZF member (FLG = FEXTRA only): header, raw deflate, CRC32, ISIZE
                    const uint8_t* p = src + a;
                    const size_t xlen = p[10] | (size_t)p[11] << 8, hdr = 12 + xlen, n = (size_t)(b - a);
                    good = n >= hdr + 8 && ld.decompress(dec, p + hdr, n - hdr - 8, out + off[(size_t)j], isz, &got) == 0 && got == isz;
                    if (good) {
                        const uint8_t* t = p + n - 8;
                        const uint32_t crc = t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
                        good = ld.crc32(0, out + off[(size_t)j], isz) == crc;
                    }
                } else {
                    good = inflate_member(src + a, (size_t)(b - a), out + off[(size_t)j], isz, &got) == 0 && got == isz;
                }
                if (!good) {
                    bad.store(m0 + j);
                    break;
                }
            }
        }
        if (dec) ld.free_d(dec);
    };
    if (!dsp::run_indexed(nt, [&](int) { work(); })) return gz_fail(DSP_ENOMEM, "dsp_gz_inflate_members: out of memory in a worker");
    if (bad.load() >= 0) return gz_fail(DSP_EPARSE, "corrupt gzip member %lld (inflate / CRC / size check failed)", (long long)bad.load());
    return (int64_t)off[(size_t)m];
}

// text -> BGZF members (no EOF member: dsp_bgzf_eof), `nthreads` deflate threads.  Returns the bytes written.
int64_t dsp_bgzf_compress(const uint8_t* in, size_t len, uint8_t* out, size_t out_cap, int32_t level, int32_t nthreads) {
    if ((!in && len) || !out) return gz_fail(DSP_EINVAL, "dsp_bgzf_compress: NULL argument");
    const size_t nb = (len + kBgzfBlock - 1) / kBgzfBlock;
    if (nb == 0) return 0;
    const size_t kMax = 65536;  // a member never exceeds 64 KiB (BSIZE is 16 bits)
    std::vector<uint8_t> tmp(nb * kMax);
    std::vector<uint32_t> sz(nb, 0);
    int nt = nthreads < 1 ? 1 : nthreads;
    if ((size_t)nt > nb) nt = (int)nb;
    std::atomic<size_t> next(0);
    std::atomic<int> err(0);
    const LibDeflate& ld = libdeflate();
    auto work = [&]() {
        void* comp = ld.ok ? ld.alloc_c(level < 1 ? 1 : (level > 12 ? 12 : level)) : nullptr;
        void* comp0 = nullptr;  // level 0 (stored) for incompressible blocks
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= nb || err.load()) break;
            const uint8_t* p = in + b * kBgzfBlock;
            const size_t n = std::min(kBgzfBlock, len - b * kBgzfBlock);
            uint8_t* o = tmp.data() + b * kMax;
            static const uint8_t hdr[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0, 0};
            memcpy(o, hdr, 18);
            {   // the member's newline count rides in the two informational header fields no reader interprets -- MTIME
                // (bytes 4..7) under the marker XFL = 'R' (byte 8): htslib / bgzip, zlib and Python's gzip ignore both, and a
                // rank can then learn how many rows its member range holds from the headers alone (dsp_gz_member_rows)
                uint32_t nl = 0;
                for (const uint8_t* q = p; (q = (const uint8_t*)memchr(q, '\n', (size_t)(p + n - q))) != nullptr; ++q) ++nl;
                for (int i = 0; i < 4; ++i) o[4 + i] = (uint8_t)(nl >> (8 * i));
                o[8] = 'R';
            }
            size_t produced = 0;
            uint32_t crc = 0;
            if (comp) {
                produced = ld.compress(comp, p, n, o + 18, kMax - 18 - 8);
                if (!produced) {  // did not fit 64 KiB: store it
                    if (!comp0) comp0 = ld.alloc_c(0);
                    produced = comp0 ? ld.compress(comp0, p, n, o + 18, kMax - 18 - 8) : 0;
                }
                if (!produced) { err.store(1); break; }
                crc = ld.crc32(0, p, n);
            } else {
                z_stream z;
                int lv = level;
                for (;;) {  // raw deflate; incompressible input that would overflow 64 KiB is stored (level 0)
                    memset(&z, 0, sizeof(z));
                    if (deflateInit2(&z, lv, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { err.store(1); break; }
                    z.next_in = const_cast<Bytef*>(p); z.avail_in = (uInt)n;
                    z.next_out = o + 18; z.avail_out = (uInt)(kMax - 18 - 8);
                    const int rc = deflate(&z, Z_FINISH);
                    produced = (size_t)z.total_out;
                    deflateEnd(&z);
                    if (rc == Z_STREAM_END) break;
                    produced = 0;
                    if (lv == 0) { err.store(1); break; }
                    lv = 0;
                }
                if (err.load()) break;
                crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n);
            }
            uint8_t* t = o + 18 + produced;
            for (int i = 0; i < 4; ++i) t[i] = (uint8_t)(crc >> (8 * i));
            for (int i = 0; i < 4; ++i) t[4 + i] = (uint8_t)((uint32_t)n >> (8 * i));
            const size_t total = 18 + produced + 8;
            o[16] = (uint8_t)((total - 1) & 0xff); o[17] = (uint8_t)((total - 1) >> 8);
            sz[b] = (uint32_t)total;
        }
        if (comp) ld.free_c(comp);
        if (comp0) ld.free_c(comp0);
    };
    if (!dsp::run_indexed(nt, [&](int) { work(); })) return gz_fail(DSP_ENOMEM, "dsp_bgzf_compress: out of memory in a worker");
    if (err.load()) return gz_fail(DSP_EHIP, "dsp_bgzf_compress: deflate failed");
    size_t total = 0;
    for (size_t b = 0; b < nb; ++b) total += sz[b];
    if (total > out_cap) return gz_fail(DSP_ENOMEM, "dsp_bgzf_compress: %lld bytes do not fit %lld", (long long)total, (long long)out_cap);
    size_t pos = 0;
    for (size_t b = 0; b < nb; ++b) { memcpy(out + pos, tmp.data() + b * kMax, sz[b]); pos += sz[b]; }
    return (int64_t)total;
}

// the 28-byte empty member that ends a BGZF file
int64_t dsp_bgzf_eof(uint8_t* out, size_t cap) {
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!out || cap < 28) return gz_fail(DSP_ENOMEM, "dsp_bgzf_eof: 28 bytes needed");
    memcpy(out, eof, 28);
    return 28;
}

dsp_gz_stream* dsp_gz_open(const char* path) {
    if (!path) return nullptr;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { gz_fail(DSP_EINVAL, "dsp_gz_open: cannot open the file"); return nullptr; }
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); gz_fail(DSP_EINVAL, "dsp_gz_open: cannot stat the file"); return nullptr; }
    dsp_gz_stream* s = new (std::nothrow) dsp_gz_stream();
    if (!s) { close(fd); return nullptr; }
    s->fd = fd;
    s->size = (size_t)sb.st_size;
    memset(&s->z, 0, sizeof(s->z));
    if (s->size) {
        void* m = mmap(nullptr, s->size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); delete s; gz_fail(DSP_EINVAL, "dsp_gz_open: cannot map the file"); return nullptr; }
        madvise(m, s->size, MADV_SEQUENTIAL);
        s->map = (const uint8_t*)m;
    }
    return s;
}

// up to cap bytes of text (all members of the file, like gzip.open); 0 at the end; DSP_EPARSE on a corrupt OR TRUNCATED
// stream (the reference's gzip.open raises EOFError / BadGzipFile there)
int64_t dsp_gz_read(dsp_gz_stream* s, uint8_t* out, size_t cap) {
    if (!s || s->fd < 0 || !out) return gz_fail(DSP_EINVAL, "dsp_gz_read: NULL argument");
    size_t got = 0;
    while (got < cap && !s->done) {
        if (!s->z_live) {
            // between members: zero padding is skipped (gzip.open does the same), the end of the file ends the stream,
            // anything else must be another member
            while (s->pos < s->size && s->map[s->pos] == 0) ++s->pos;
            if (s->pos >= s->size) {
                if (s->bytes_out == 0 && s->size == 0) { s->done = true; break; }  // an empty file reads as empty text
                s->done = true;
                break;
            }
            memset(&s->z, 0, sizeof(s->z));
            if (inflateInit2(&s->z, 15 + 16) != Z_OK) return gz_fail(DSP_ENOMEM, "dsp_gz_read: inflateInit2 failed");
            s->z_live = true;
        }
        const size_t in_chunk = std::min<size_t>(s->size - s->pos, 1u << 30);
        const size_t out_chunk = std::min<size_t>(cap - got, 1u << 30);
        s->z.next_in = const_cast<Bytef*>(s->map + s->pos); s->z.avail_in = (uInt)in_chunk;
        s->z.next_out = out + got; s->z.avail_out = (uInt)out_chunk;
        const int rc = inflate(&s->z, Z_NO_FLUSH);
        const size_t used = in_chunk - s->z.avail_in, made = out_chunk - s->z.avail_out;
        s->pos += used; got += made; s->bytes_out += made;
        if (rc == Z_STREAM_END) {
            inflateEnd(&s->z);
            s->z_live = false;
            continue;
        }
        if (rc == Z_OK) continue;
        if (rc == Z_BUF_ERROR && s->z.avail_out == 0) continue;  // output full: the caller comes back
        char buf[256];
        if (rc == Z_BUF_ERROR || (rc == Z_OK && used == 0 && made == 0)) {
            // no progress possible with all the input there is: the file ends inside a member
            snprintf(buf, sizeof(buf), "truncated gzip stream: Compressed file ended before the end-of-stream marker was reached "
                                       "(%llu of %llu compressed bytes read)", (unsigned long long)s->pos, (unsigned long long)s->size);
        } else {
            snprintf(buf, sizeof(buf), "corrupt gzip stream: %s", s->z.msg ? s->z.msg : (rc == Z_DATA_ERROR ? "data error" : "zlib error"));
        }
        inflateEnd(&s->z);
        s->z_live = false;
        s->done = true;
        dsp_set_error_(buf);
        return DSP_EPARSE;
    }
    return (int64_t)got;
}

// compressed bytes consumed so far (tests: N ranks of a node inflate a foreign .gz once, not N times)
uint64_t dsp_gz_bytes_in(const dsp_gz_stream* s) { return s ? (uint64_t)s->pos : 0; }

void dsp_gz_close(dsp_gz_stream* s) {
    if (!s) return;
    if (s->z_live) inflateEnd(&s->z);
    if (s->map) munmap(const_cast<uint8_t*>(s->map), s->size);
    if (s->fd >= 0) close(s->fd);
    delete s;
}

}  // extern "C"
