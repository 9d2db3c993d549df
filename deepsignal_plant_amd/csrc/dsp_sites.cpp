// Host side of the extraction stage: motif sites, strand coordinates and sampleinfo strings of a batch of reads.
// Replaces get_refloc_of_methysite_in_motif (utils/process_utils.py:97-112) and the coordinate / filter logic of
// _extract_features (extract_features.py:337-358).  Plain C++, no GPU.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include "dsp_amd.h"

extern "C" void dsp_set_error_(const char* msg);

namespace {

int fail(int code, const char* msg) {
    dsp_set_error_(msg);
    return code;
}

bool in_alphabet(uint8_t c) { return c < 128 && strchr("ACGTNWSMKRYBVDHZ", (int)c) != nullptr && c != 0; }

int put_i64(char* p, int64_t v) { return sprintf(p, "%lld", (long long)v); }

}  // namespace

extern "C" int64_t dsp_extract_sites(int64_t n_reads, const uint8_t* ev_base, const int64_t* ev_off,
                                     const char* const* chrom, const char* const* readname, const char* read_strand,
                                     const char* align_strand, const int64_t* chrom_start, const int64_t* chrom_len,
                                     const int64_t* rg_lo, const int64_t* rg_hi, const char* motifs, int32_t n_motifs,
                                     int32_t motif_len, int32_t methyloc, int32_t seq_len, int64_t max_sites,
                                     int32_t* site_read, int32_t* site_loc, char* info, size_t info_cap,
                                     size_t* info_bytes, uint64_t* row_off, uint32_t* info_len, uint32_t* read_off,
                                     uint32_t* read_len) {
    if (n_reads < 0 || (n_reads && (!ev_base || !ev_off || !chrom || !readname || !read_strand || !align_strand ||
                                    !chrom_start || !chrom_len)) ||
        !motifs || n_motifs <= 0 || motif_len <= 0 || seq_len <= 0 || !(seq_len & 1))
        return fail(DSP_EINVAL, "dsp_extract_sites: bad arguments (kmer_len must be odd)");
    const bool counting = site_read == nullptr;
    if (!counting && (!site_loc || !info || !row_off || !info_len || !read_off || !read_len))
        return fail(DSP_EINVAL, "dsp_extract_sites: NULL output array");
    const int64_t nb = (seq_len - 1) / 2;
    int64_t n_sites = 0;
    size_t used = 0;
    char num[32];
    for (int64_t r = 0; r < n_reads; r++) {
        const uint8_t* seq = ev_base + ev_off[r];
        const int64_t len = ev_off[r + 1] - ev_off[r];
        const size_t chrom_n = strlen(chrom[r]), name_n = strlen(readname[r]);
        for (int64_t i = 0; i + motif_len <= len; i++) {
            bool hit = false;
            for (int32_t m = 0; m < n_motifs && !hit; m++) hit = memcmp(seq + i, motifs + (size_t)m * motif_len, motif_len) == 0;
            if (!hit) continue;
            const int64_t loc = i + methyloc;
            if (!(nb <= loc && loc < len - nb)) continue;
            int64_t pos, pos_in_strand;
            if (align_strand[r] == '-') {
                pos = chrom_start[r] + len - 1 - loc;
                pos_in_strand = chrom_len[r] >= 0 ? chrom_len[r] - 1 - pos : -1;
            } else {
                pos = chrom_start[r] + loc;
                pos_in_strand = chrom_len[r] >= 0 ? pos : -1;
            }
            if (rg_lo && (pos < rg_lo[r] || pos >= rg_hi[r])) continue;
            for (int64_t j = loc - nb; j <= loc + nb; j++)
                if (!in_alphabet(seq[j])) {
                    char buf[160];
                    snprintf(buf, sizeof buf, "dsp_extract_sites: base '%c' of read %s is not in the alphabet", seq[j], readname[r]);
                    return fail(DSP_EPARSE, buf);
                }
            const int np = put_i64(num, pos);
            char num2[32];
            const int np2 = put_i64(num2, pos_in_strand);
            const size_t bytes = chrom_n + 1 + np + 1 + 1 + 1 + np2 + 1 + name_n + 1 + 1;
            if (!counting) {
                if (n_sites >= max_sites || used + bytes > info_cap)
                    return fail(DSP_ENOMEM, "dsp_extract_sites: output buffers too small");
                char* p = info + used;
                memcpy(p, chrom[r], chrom_n); p += chrom_n; *p++ = '\t';
                memcpy(p, num, np); p += np; *p++ = '\t';
                *p++ = align_strand[r]; *p++ = '\t';
                memcpy(p, num2, np2); p += np2; *p++ = '\t';
                read_off[n_sites] = (uint32_t)(p - (info + used));
                memcpy(p, readname[r], name_n); p += name_n; *p++ = '\t';
                *p++ = read_strand[r];
                site_read[n_sites] = (int32_t)r;
                site_loc[n_sites] = (int32_t)loc;
                row_off[n_sites] = used;
                info_len[n_sites] = (uint32_t)bytes;
                read_len[n_sites] = (uint32_t)name_n;
            }
            used += bytes;
            n_sites++;
        }
    }
    if (info_bytes) *info_bytes = used;
    return n_sites;
}
