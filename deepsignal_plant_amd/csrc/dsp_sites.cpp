// Host side of the extraction stage: motif sites, strand coordinates and sampleinfo strings of a batch of reads.
// Replaces get_refloc_of_methysite_in_motif (utils/process_utils.py:97-112) and the coordinate / filter logic of
// _extract_features (extract_features.py:337-358).  Plain C++, no GPU.
//
// Two phases, both parallel over reads: (1) scan every read's bases with a rolling 2-bit code against a bitmap of
// the motif set (motifs are ACGT-only after IUPAC expansion; other sets fall back to memcmp) and count sites and
// sampleinfo bytes per read; (2) after a prefix sum, every read writes its own slice of the outputs.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "dsp_amd.h"
#include "dsp_threads.h"

extern "C" void dsp_set_error_(const char* msg);

namespace {

int fail(int code, const char* msg) {
    dsp_set_error_(msg);
    return code;
}

inline int acgt(uint8_t c) {
    switch (c) {
        case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
        default: return -1;
    }
}

inline bool in_alphabet(uint8_t c) {
    switch (c) {
        case 'A': case 'C': case 'G': case 'T': case 'N': case 'W': case 'S': case 'M':
        case 'K': case 'R': case 'Y': case 'B': case 'V': case 'D': case 'H': case 'Z': return true;
        default: return false;
    }
}

inline int digits_i64(int64_t v) {
    int n = v < 0 ? 1 : 0;
    uint64_t u = v < 0 ? (uint64_t)(-(v + 1)) + 1 : (uint64_t)v;
    do { n++; u /= 10; } while (u);
    return n;
}

inline char* put_i64(char* p, int64_t v) {
    char tmp[24];
    int n = 0;
    uint64_t u = v < 0 ? (uint64_t)(-(v + 1)) + 1 : (uint64_t)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) *p++ = '-';
    while (n) *p++ = tmp[--n];
    return p;
}

struct MotifSet {
    int len = 0, n = 0;
    const char* seqs = nullptr;
    bool use_table = false;
    std::vector<uint8_t> table;  // one byte per 2-bit-coded word
    uint32_t mask = 0;

    void build(const char* motifs, int n_motifs, int motif_len) {
        seqs = motifs; n = n_motifs; len = motif_len;
        use_table = motif_len <= 12;
        for (int i = 0; use_table && i < n * len; i++) use_table = acgt((uint8_t)motifs[i]) >= 0;
        if (!use_table) return;
        table.assign((size_t)1 << (2 * len), 0);
        mask = (uint32_t)(((uint64_t)1 << (2 * len)) - 1);
        for (int m = 0; m < n; m++) {
            uint32_t w = 0;
            for (int i = 0; i < len; i++) w = (w << 2) | (uint32_t)acgt((uint8_t)motifs[m * len + i]);
            table[w] = 1;
        }
    }
};

struct Job {
    int64_t n_reads;
    const uint8_t* ev_base;
    const int64_t* ev_off;
    const char* const* chrom;
    const char* const* readname;
    const char *read_strand, *align_strand;
    const int64_t *chrom_start, *chrom_len, *rg_lo, *rg_hi;
    const MotifSet* ms;
    int methyloc;
    int64_t nb;
};

// calls emit(loc, pos, pos_in_strand) for every site of read r in position order; returns false on a bad base
template <class Emit>
bool for_each_site(const Job& j, int64_t r, int64_t* bad_at, Emit emit) {
    const uint8_t* seq = j.ev_base + j.ev_off[r];
    const int64_t len = j.ev_off[r + 1] - j.ev_off[r];
    const MotifSet& ms = *j.ms;
    const bool minus = j.align_strand[r] == '-';
    uint32_t w = 0;
    int valid = 0;  // consecutive ACGT bases ending at the current one
    for (int64_t e = 0; e < len; e++) {  // e = last base of the candidate motif occurrence
        bool hit;
        const int64_t i = e - ms.len + 1;
        if (ms.use_table) {
            const int c = acgt(seq[e]);
            if (c < 0) { valid = 0; w = 0; continue; }
            w = ((w << 2) | (uint32_t)c) & ms.mask;
            if (++valid < ms.len) continue;
            hit = ms.table[w] != 0;
        } else {
            if (i < 0) continue;
            hit = false;
            for (int m = 0; m < ms.n && !hit; m++) hit = memcmp(seq + i, ms.seqs + (size_t)m * ms.len, ms.len) == 0;
        }
        if (!hit) continue;
        const int64_t loc = i + j.methyloc;
        if (!(j.nb <= loc && loc < len - j.nb)) continue;
        int64_t pos, pis;
        if (minus) {
            pos = j.chrom_start[r] + len - 1 - loc;
            pis = j.chrom_len[r] >= 0 ? j.chrom_len[r] - 1 - pos : -1;
        } else {
            pos = j.chrom_start[r] + loc;
            pis = j.chrom_len[r] >= 0 ? pos : -1;
        }
        if (j.rg_lo && (pos < j.rg_lo[r] || pos >= j.rg_hi[r])) continue;
        for (int64_t k = loc - j.nb; k <= loc + j.nb; k++)
            if (!in_alphabet(seq[k])) { *bad_at = k; return false; }
        emit(loc, pos, pis);
    }
    return true;
}

template <class F>
void parallel_reads(int64_t n_reads, int nthreads, F f) {
    if (nthreads <= 1 || n_reads < 2) {
        for (int64_t r = 0; r < n_reads; r++) f(r);
        return;
    }
    std::atomic<int64_t> next(0);
    auto work = [&]() {
        for (;;) {
            const int64_t r0 = next.fetch_add(8);
            if (r0 >= n_reads) return;
            for (int64_t r = r0; r < r0 + 8 && r < n_reads; r++) f(r);
        }
    };
    (void)dsp::run_indexed(nthreads, [&](int) { work(); });   // (the callers' f allocates nothing: no worker can throw)
}

}  // namespace

extern "C" int64_t dsp_extract_sites(int64_t n_reads, const uint8_t* ev_base, const int64_t* ev_off,
                                     const char* const* chrom, const char* const* readname, const char* read_strand,
                                     const char* align_strand, const int64_t* chrom_start, const int64_t* chrom_len,
                                     const int64_t* rg_lo, const int64_t* rg_hi, const char* motifs, int32_t n_motifs,
                                     int32_t motif_len, int32_t methyloc, int32_t seq_len, int64_t max_sites,
                                     int32_t* site_read, int32_t* site_loc, char* info, size_t info_cap,
                                     size_t* info_bytes, uint64_t* row_off, uint32_t* info_len, uint32_t* read_off,
                                     uint32_t* read_len, int32_t nthreads) {
    if (n_reads < 0 || (n_reads && (!ev_base || !ev_off || !chrom || !readname || !read_strand || !align_strand ||
                                    !chrom_start || !chrom_len)) ||
        !motifs || n_motifs <= 0 || motif_len <= 0 || seq_len <= 0 || !(seq_len & 1) || (rg_lo && !rg_hi))
        return fail(DSP_EINVAL, "dsp_extract_sites: bad arguments (kmer_len must be odd)");
    const bool counting = site_read == nullptr;
    if (!counting && (!site_loc || !info || !row_off || !info_len || !read_off || !read_len))
        return fail(DSP_EINVAL, "dsp_extract_sites: NULL output array");
    MotifSet ms;
    ms.build(motifs, n_motifs, motif_len);
    Job j = {n_reads, ev_base, ev_off, chrom, readname, read_strand, align_strand, chrom_start, chrom_len, rg_lo, rg_hi,
             &ms, methyloc, (seq_len - 1) / 2};
    const int nt = nthreads < 1 ? 1 : (nthreads > 64 ? 64 : nthreads);

    // phase 1: per-read site and byte counts
    std::vector<int64_t> cnt(n_reads + 1, 0);
    std::vector<uint64_t> bytes(n_reads + 1, 0);
    std::atomic<int64_t> bad_read(-1);
    std::vector<int64_t> bad_pos(n_reads > 0 ? n_reads : 1, 0);
    parallel_reads(n_reads, nt, [&](int64_t r) {
        const uint64_t fixed = strlen(chrom[r]) + strlen(readname[r]) + 7;  // 5 tabs, strand, read strand
        int64_t c = 0;
        uint64_t b = 0;
        if (!for_each_site(j, r, &bad_pos[r], [&](int64_t, int64_t pos, int64_t pis) {
                c++;
                b += fixed + digits_i64(pos) + digits_i64(pis);
            })) {
            int64_t expect = -1;
            bad_read.compare_exchange_strong(expect, r);
            return;
        }
        cnt[r + 1] = c;
        bytes[r + 1] = b;
    });
    if (bad_read.load() >= 0) {
        const int64_t r = bad_read.load();
        char buf[200];
        snprintf(buf, sizeof buf, "dsp_extract_sites: base '%c' of read %s is not in the alphabet",
                 (char)ev_base[ev_off[r] + bad_pos[r]], readname[r]);
        return fail(DSP_EPARSE, buf);
    }
    for (int64_t r = 0; r < n_reads; r++) {
        cnt[r + 1] += cnt[r];
        bytes[r + 1] += bytes[r];
    }
    const int64_t n_sites = cnt[n_reads];
    if (info_bytes) *info_bytes = bytes[n_reads];
    if (counting) return n_sites;
    if (n_sites > max_sites || bytes[n_reads] > info_cap) return fail(DSP_ENOMEM, "dsp_extract_sites: output buffers too small");

    // phase 2: every read fills its slice
    parallel_reads(n_reads, nt, [&](int64_t r) {
        int64_t s = cnt[r];
        uint64_t used = bytes[r];
        const size_t chrom_n = strlen(chrom[r]), name_n = strlen(readname[r]);
        int64_t dummy;
        for_each_site(j, r, &dummy, [&](int64_t loc, int64_t pos, int64_t pis) {
            char* const p0 = info + used;
            char* p = p0;
            memcpy(p, chrom[r], chrom_n); p += chrom_n; *p++ = '\t';
            p = put_i64(p, pos); *p++ = '\t';
            *p++ = align_strand[r]; *p++ = '\t';
            p = put_i64(p, pis); *p++ = '\t';
            read_off[s] = (uint32_t)(p - p0);
            memcpy(p, readname[r], name_n); p += name_n; *p++ = '\t';
            *p++ = read_strand[r];
            site_read[s] = (int32_t)r;
            site_loc[s] = (int32_t)loc;
            row_off[s] = used;
            info_len[s] = (uint32_t)(p - p0);
            read_len[s] = (uint32_t)name_n;
            used += (uint64_t)(p - p0);
            s++;
        });
    });
    return n_sites;
}
