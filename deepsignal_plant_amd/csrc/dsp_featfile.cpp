// Binary feature container (.dspf) of the call_mods path: the parsed form of the reference's feature TSV
// (row producer extract_features.py:381-395, row consumer call_modifications.py:76-86), stored as blocks of
// ready-to-copy SoA arrays so that `call_mods` reads straight into pinned host buffers with no text parsing
// (SURVEY.md 8(f) next-2).  Values are exactly what dsp_parse_feature_rows yields for the same rows, so the
// per-read calls written from a .dspf are byte-identical to those written from the TSV.
//
//   file   := header(64 B) block* index
//   header := "DSPFEAT1" u32 version u32 seq_len u32 signal_len u32 flags u64 n_rows u64 n_blocks
//             u64 index_offset, zero padded to 64 B  (n_rows / n_blocks / index_offset patched by close())
//   block  := bhdr(64 B: u32 'DSPB', u32 n, u64 info_bytes, u64 first_row) then 64 B-aligned sections
//             kmer u8[n][L] | means f32[n][L] | stds f32[n][L] | lens i32[n][L] | signals f32[n][L][S] |
//             labels i32[n] | info_len u32[n] | read_off u32[n] | read_len u32[n] | info bytes
//             (info = the rows' six leading TSV columns, "sampleinfo" of call_modifications.py:80, concatenated)
//   index  := n_blocks x { u64 offset, u64 first_row, u64 info_bytes, u32 n, u32 0 }
//
// Little-endian, host-native float32.  Plain C++, no GPU.
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

#include "dsp_amd.h"
#include "dsp_threads.h"

extern "C" void dsp_set_error_(const char* msg);

namespace {

const char kMagic[8] = {'D', 'S', 'P', 'F', 'E', 'A', 'T', '1'};
const uint32_t kBlockMagic = 0x42505344u;  // "DSPB"
const uint32_t kVersion = 1;

int fail(int code, const char* fmt, const char* a = "", long long b = 0) {
    char buf[512];
    snprintf(buf, sizeof buf, fmt, a, b);
    dsp_set_error_(buf);
    return code;
}

struct Header {
    char magic[8];
    uint32_t version, seq_len, signal_len, flags;
    uint64_t n_rows, n_blocks, index_offset;
    uint8_t pad[16];
};
static_assert(sizeof(Header) == 64, "header is 64 bytes");

struct BlockHeader {
    uint32_t magic, n;
    uint64_t info_bytes, first_row;
    uint8_t pad[40];
};
static_assert(sizeof(BlockHeader) == 64, "block header is 64 bytes");

struct IndexEntry {
    uint64_t offset, first_row, info_bytes;
    uint32_t n, zero;
};
static_assert(sizeof(IndexEntry) == 32, "index entry is 32 bytes");

inline uint64_t rup64(uint64_t x) { return (x + 63) & ~uint64_t(63); }

// byte offsets of the sections of a block with n rows, relative to the block start
struct Layout {
    uint64_t kmer, means, stds, lens, signals, labels, info_len, read_off, read_len, info, end;
    Layout(uint64_t n, uint64_t L, uint64_t S, uint64_t info_bytes) {
        uint64_t o = sizeof(BlockHeader);
        kmer = o;      o = rup64(o + n * L);
        means = o;     o = rup64(o + n * L * 4);
        stds = o;      o = rup64(o + n * L * 4);
        lens = o;      o = rup64(o + n * L * 4);
        signals = o;   o = rup64(o + n * L * S * 4);
        labels = o;    o = rup64(o + n * 4);
        info_len = o;  o = rup64(o + n * 4);
        read_off = o;  o = rup64(o + n * 4);
        read_len = o;  o = rup64(o + n * 4);
        info = o;      o = rup64(o + info_bytes);
        end = o;
    }
};

bool pwrite_all(int fd, const void* p, uint64_t n, uint64_t off) {
    const char* c = (const char*)p;
    while (n) {
        ssize_t k = pwrite(fd, c, n > (1u << 30) ? (1u << 30) : n, (off_t)off);
        if (k < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        c += k; off += k; n -= k;
    }
    return true;
}

bool pread_all(int fd, void* p, uint64_t n, uint64_t off) {
    char* c = (char*)p;
    while (n) {
        ssize_t k = pread(fd, c, n > (1u << 30) ? (1u << 30) : n, (off_t)off);
        if (k < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        if (k == 0) return false;  // truncated
        c += k; off += k; n -= k;
    }
    return true;
}

}  // namespace

struct dsp_feat_writer {
    int fd = -1;
    uint32_t L = 0, S = 0;
    uint64_t block_rows = 0, pos = sizeof(Header), n_rows = 0;
    std::vector<IndexEntry> index;
    // pending rows (always < block_rows after add() returns)
    std::vector<uint8_t> kmer;
    std::vector<float> means, stds, signals;
    std::vector<int32_t> lens, labels;
    std::vector<uint32_t> info_len, read_off, read_len;
    std::string info;
    std::string path;

    uint64_t pending() const { return labels.size(); }

    bool flush(uint64_t n) {  // write the first n pending rows as one block
        if (n == 0) return true;
        uint64_t ib = 0;
        for (uint64_t i = 0; i < n; i++) ib += info_len[i];
        Layout lay(n, L, S, ib);
        BlockHeader bh;
        memset(&bh, 0, sizeof bh);
        bh.magic = kBlockMagic; bh.n = (uint32_t)n; bh.info_bytes = ib; bh.first_row = n_rows;
        bool ok = pwrite_all(fd, &bh, sizeof bh, pos) &&
                  pwrite_all(fd, kmer.data(), n * L, pos + lay.kmer) &&
                  pwrite_all(fd, means.data(), n * L * 4, pos + lay.means) &&
                  pwrite_all(fd, stds.data(), n * L * 4, pos + lay.stds) &&
                  pwrite_all(fd, lens.data(), n * L * 4, pos + lay.lens) &&
                  pwrite_all(fd, signals.data(), n * L * S * 4, pos + lay.signals) &&
                  pwrite_all(fd, labels.data(), n * 4, pos + lay.labels) &&
                  pwrite_all(fd, info_len.data(), n * 4, pos + lay.info_len) &&
                  pwrite_all(fd, read_off.data(), n * 4, pos + lay.read_off) &&
                  pwrite_all(fd, read_len.data(), n * 4, pos + lay.read_len) &&
                  pwrite_all(fd, info.data(), ib, pos + lay.info);
        if (!ok) return false;
        IndexEntry e = {pos, n_rows, ib, (uint32_t)n, 0};
        index.push_back(e);
        pos += lay.end;
        n_rows += n;
        kmer.erase(kmer.begin(), kmer.begin() + n * L);
        means.erase(means.begin(), means.begin() + n * L);
        stds.erase(stds.begin(), stds.begin() + n * L);
        lens.erase(lens.begin(), lens.begin() + n * L);
        signals.erase(signals.begin(), signals.begin() + n * L * S);
        labels.erase(labels.begin(), labels.begin() + n);
        info_len.erase(info_len.begin(), info_len.begin() + n);
        read_off.erase(read_off.begin(), read_off.begin() + n);
        read_len.erase(read_len.begin(), read_len.begin() + n);
        info.erase(0, ib);
        return true;
    }
};

struct dsp_feat_file {
    int fd = -1;
    Header h;
    std::vector<IndexEntry> index;
};

extern "C" {

int32_t dsp_feat_writer_create(const char* path, int32_t seq_len, int32_t signal_len, int64_t block_rows,
                               dsp_feat_writer** out) {
    if (!path || !out || seq_len <= 0 || signal_len < 0 || seq_len > 4096 || signal_len > 65536)
        return fail(DSP_EINVAL, "dsp_feat_writer_create: bad arguments%s", "");
    if (block_rows <= 0) block_rows = 32768;
    if (block_rows > (1 << 24)) return fail(DSP_EINVAL, "dsp_feat_writer_create: block_rows too large%s", "");
    int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY | O_CLOEXEC, 0644);
    if (fd < 0) return fail(DSP_EINVAL, "dsp_feat_writer_create: cannot open %s for writing", path);
    dsp_feat_writer* w = new dsp_feat_writer;
    w->fd = fd; w->L = seq_len; w->S = signal_len; w->block_rows = block_rows; w->path = path;
    *out = w;
    return DSP_OK;
}

int32_t dsp_feat_writer_add(dsp_feat_writer* w, int64_t n, const uint8_t* kmer, const float* means, const float* stds,
                            const int32_t* lens, const float* signals, const int32_t* labels, const char* text,
                            const uint64_t* row_off, const uint32_t* info_len, const uint32_t* read_off,
                            const uint32_t* read_len) {
    if (!w || n < 0) return fail(DSP_EINVAL, "dsp_feat_writer_add: bad arguments%s", "");
    if (n == 0) return DSP_OK;
    if (!kmer || !means || !stds || !lens || (!signals && w->S) || !labels || !text || !row_off || !info_len ||
        !read_off || !read_len)
        return fail(DSP_EINVAL, "dsp_feat_writer_add: NULL array%s", "");
    const uint64_t L = w->L, S = w->S;
    int64_t done = 0;
    while (done < n) {
        int64_t take = (int64_t)(w->block_rows - w->pending());
        if (take > n - done) take = n - done;
        w->kmer.insert(w->kmer.end(), kmer + done * L, kmer + (done + take) * L);
        w->means.insert(w->means.end(), means + done * L, means + (done + take) * L);
        w->stds.insert(w->stds.end(), stds + done * L, stds + (done + take) * L);
        w->lens.insert(w->lens.end(), lens + done * L, lens + (done + take) * L);
        if (S) w->signals.insert(w->signals.end(), signals + done * L * S, signals + (done + take) * L * S);
        w->labels.insert(w->labels.end(), labels + done, labels + done + take);
        w->info_len.insert(w->info_len.end(), info_len + done, info_len + done + take);
        w->read_off.insert(w->read_off.end(), read_off + done, read_off + done + take);
        w->read_len.insert(w->read_len.end(), read_len + done, read_len + done + take);
        for (int64_t i = done; i < done + take; i++) w->info.append(text + row_off[i], info_len[i]);
        done += take;
        if (w->pending() == w->block_rows && !w->flush(w->block_rows))
            return fail(DSP_EINVAL, "dsp_feat_writer_add: write to %s failed", w->path.c_str());
    }
    return DSP_OK;
}

int32_t dsp_feat_writer_close(dsp_feat_writer* w) {
    if (!w) return DSP_OK;
    bool ok = w->flush(w->pending());
    Header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, kMagic, 8);
    h.version = kVersion; h.seq_len = w->L; h.signal_len = w->S;
    h.n_rows = w->n_rows; h.n_blocks = w->index.size(); h.index_offset = w->pos;
    ok = ok && pwrite_all(w->fd, w->index.data(), w->index.size() * sizeof(IndexEntry), w->pos) &&
         pwrite_all(w->fd, &h, sizeof h, 0);
    ok = (close(w->fd) == 0) && ok;
    std::string path = w->path;
    delete w;
    return ok ? DSP_OK : fail(DSP_EINVAL, "dsp_feat_writer_close: write to %s failed", path.c_str());
}

int32_t dsp_feat_open(const char* path, dsp_feat_file** out) {
    if (!path || !out) return fail(DSP_EINVAL, "dsp_feat_open: bad arguments%s", "");
    int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return fail(DSP_EINVAL, "dsp_feat_open: cannot open %s", path);
    dsp_feat_file* f = new dsp_feat_file;
    f->fd = fd;
    struct stat st;
    bool ok = fstat(fd, &st) == 0 && (uint64_t)st.st_size >= sizeof(Header) && pread_all(fd, &f->h, sizeof(Header), 0);
    if (!ok || memcmp(f->h.magic, kMagic, 8) != 0 || f->h.version != kVersion) {
        close(fd); delete f;
        return fail(DSP_EPARSE, "dsp_feat_open: %s is not a DSPFEAT1 feature file (or it was not closed)", path);
    }
    const uint64_t nb = f->h.n_blocks;
    if (f->h.index_offset < sizeof(Header) || nb > (1ull << 40) / sizeof(IndexEntry) ||
        f->h.index_offset + nb * sizeof(IndexEntry) > (uint64_t)st.st_size) {
        close(fd); delete f;
        return fail(DSP_EPARSE, "dsp_feat_open: %s: truncated index", path);
    }
    f->index.resize(nb);
    if (nb && !pread_all(fd, f->index.data(), nb * sizeof(IndexEntry), f->h.index_offset)) {
        close(fd); delete f;
        return fail(DSP_EPARSE, "dsp_feat_open: %s: unreadable index", path);
    }
    uint64_t rows = 0;
    for (uint64_t b = 0; b < nb; b++) {
        const IndexEntry& e = f->index[b];
        Layout lay(e.n, f->h.seq_len, f->h.signal_len, e.info_bytes);
        if (e.first_row != rows || e.offset + lay.end > f->h.index_offset) {
            close(fd); delete f;
            return fail(DSP_EPARSE, "dsp_feat_open: %s: inconsistent index at block %lld", path, (long long)b);
        }
        rows += e.n;
    }
    if (rows != f->h.n_rows) {
        close(fd); delete f;
        return fail(DSP_EPARSE, "dsp_feat_open: %s: row count does not match the index", path);
    }
    *out = f;
    return DSP_OK;
}

int32_t dsp_feat_info(const dsp_feat_file* f, int32_t* seq_len, int32_t* signal_len, int64_t* n_rows, int64_t* n_blocks) {
    if (!f) return fail(DSP_EINVAL, "dsp_feat_info: NULL handle%s", "");
    if (seq_len) *seq_len = (int32_t)f->h.seq_len;
    if (signal_len) *signal_len = (int32_t)f->h.signal_len;
    if (n_rows) *n_rows = (int64_t)f->h.n_rows;
    if (n_blocks) *n_blocks = (int64_t)f->h.n_blocks;
    return DSP_OK;
}

int32_t dsp_feat_block_info(const dsp_feat_file* f, int64_t block, int64_t* n, int64_t* first_row, int64_t* info_bytes) {
    if (!f || block < 0 || (uint64_t)block >= f->index.size())
        return fail(DSP_EINVAL, "dsp_feat_block_info: block index out of range%s", "");
    const IndexEntry& e = f->index[block];
    if (n) *n = e.n;
    if (first_row) *first_row = (int64_t)e.first_row;
    if (info_bytes) *info_bytes = (int64_t)e.info_bytes;
    return DSP_OK;
}

int64_t dsp_feat_read_block(const dsp_feat_file* f, int64_t block, int64_t max_rows, uint8_t* kmer, float* means,
                            float* stds, int32_t* lens, float* signals, int32_t* labels, char* info, size_t info_cap,
                            uint64_t* row_off, uint32_t* info_len, uint32_t* read_off, uint32_t* read_len,
                            int32_t nthreads) {
    if (!f || block < 0 || (uint64_t)block >= f->index.size())
        return fail(DSP_EINVAL, "dsp_feat_read_block: block index out of range%s", "");
    const IndexEntry& e = f->index[block];
    const uint64_t n = e.n, L = f->h.seq_len, S = f->h.signal_len;
    if ((int64_t)n > max_rows || e.info_bytes > info_cap)
        return fail(DSP_ENOMEM, "dsp_feat_read_block: output buffers too small for block %s%lld", "", (long long)block);
    BlockHeader bh;
    if (!pread_all(f->fd, &bh, sizeof bh, e.offset) || bh.magic != kBlockMagic || bh.n != e.n ||
        bh.info_bytes != e.info_bytes || bh.first_row != e.first_row)
        return fail(DSP_EPARSE, "dsp_feat_read_block: corrupt header of block %s%lld", "", (long long)block);
    Layout lay(n, L, S, e.info_bytes);
    struct Job { void* dst; uint64_t bytes, off; };
    std::vector<Job> jobs;
    auto add = [&](void* dst, uint64_t bytes, uint64_t off) {
        if (dst && bytes) jobs.push_back({dst, bytes, e.offset + off});
    };
    add(kmer, n * L, lay.kmer);
    add(means, n * L * 4, lay.means);
    add(stds, n * L * 4, lay.stds);
    add(lens, n * L * 4, lay.lens);
    add(labels, n * 4, lay.labels);
    add(info_len, n * 4, lay.info_len);
    add(read_off, n * 4, lay.read_off);
    add(read_len, n * 4, lay.read_len);
    add(info, e.info_bytes, lay.info);
    // the signal section is 80 % of the block: cut it into one piece per thread
    int nt = nthreads < 1 ? 1 : (nthreads > 64 ? 64 : nthreads);
    if (signals && S) {
        const uint64_t total = n * L * S * 4, piece = rup64((total + nt - 1) / nt);
        for (uint64_t o = 0; o < total; o += piece)
            add((char*)signals + o, (total - o < piece ? total - o : piece), lay.signals + o);
    }
    std::vector<int> bad(nt, 0);
    auto work = [&](int t) {
        for (size_t j = t; j < jobs.size(); j += nt)
            if (!pread_all(f->fd, jobs[j].dst, jobs[j].bytes, jobs[j].off)) bad[t] = 1;
    };
    if (!dsp::run_indexed(nt, work)) return fail(DSP_ENOMEM, "dsp_feat_read_block: a worker failed in block %s%lld", "", (long long)block);
    for (int t = 0; t < nt; t++)
        if (bad[t]) return fail(DSP_EPARSE, "dsp_feat_read_block: short read in block %s%lld", "", (long long)block);
    if (info_len) {
        uint64_t o = 0;
        for (uint64_t i = 0; i < n; i++) {
            if (row_off) row_off[i] = o;
            o += info_len[i];
        }
        if (o != e.info_bytes)
            return fail(DSP_EPARSE, "dsp_feat_read_block: info lengths of block %s%lld do not add up", "", (long long)block);
    }
    return (int64_t)n;
}

void dsp_feat_close(dsp_feat_file* f) {
    if (!f) return;
    close(f->fd);
    delete f;
}

}  // extern "C"
