// CPU sanitizer harness for the host half of libdsp_amd.so (parser, formatters, feature container, site
// enumerator, call_freq aggregator): built by tests/test_host_sanitizers.py with -fsanitize=address,undefined and
// run on the committed fixtures plus mutated / truncated inputs.  It checks invariants, not values (the value
// checks live in the Python tests); what it is for is memory safety on hostile input.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <thread>

#include <string>
#include <vector>

#include "dsp_amd.h"

#include <fcntl.h>
#include <unistd.h>

static std::string g_err;
extern "C" void dsp_set_error_(const char* msg) { g_err = msg ? msg : ""; }
extern "C" const char* dsp_last_error(void) { return g_err.c_str(); }

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #c, g_err.c_str()); exit(1); } } while (0)

static std::string slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    CHECK(f != nullptr);
    std::string s;
    char buf[1 << 16];
    size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, k);
    fclose(f);
    return s;
}

struct Rows {
    int L, S;
    int64_t n = 0;
    std::vector<uint8_t> kmer;
    std::vector<float> means, stds, signals;
    std::vector<int32_t> lens, labels;
    std::vector<uint64_t> row_off;
    std::vector<uint32_t> info_len, read_off, read_len;
    Rows(int L_, int S_, int64_t cap) : L(L_), S(S_) {
        kmer.resize(cap * L); means.resize(cap * L); stds.resize(cap * L); lens.resize(cap * L);
        signals.resize(cap * L * S); labels.resize(cap); row_off.resize(cap); info_len.resize(cap);
        read_off.resize(cap); read_len.resize(cap);
    }
    int64_t parse(const std::string& text, int nthreads) {
        // an exact-size heap copy: any read past the end is an ASan report
        std::vector<char> t(text.begin(), text.end());
        n = dsp_parse_feature_rows(t.data(), t.size(), L, S, (int64_t)labels.size(), kmer.data(), means.data(), stds.data(),
                                   lens.data(), signals.data(), labels.data(), row_off.data(), info_len.data(),
                                   read_off.data(), read_len.data(), nthreads);
        return n;
    }
};

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char** argv) {
    CHECK(argc >= 3);
    const std::string golden = argv[1], tmp = argv[2];
    const std::string tsv = slurp((golden + "/f2_rows.tsv").c_str());
    const int64_t nrows = dsp_count_rows(tsv.data(), tsv.size());
    CHECK(nrows == 200);

    // ---- parser: good input, all thread counts; then truncations and byte mutations (must fail cleanly or parse)
    Rows rows(13, 16, nrows + 8);
    for (int nt : {1, 2, 5, 16}) CHECK(rows.parse(tsv, nt) == nrows);
    CHECK(rows.parse(tsv.substr(0, tsv.size() - 1), 3) == nrows);  // no trailing newline
    Rows small(13, 16, 10);
    CHECK(small.parse(tsv, 4) == DSP_ENOMEM);
    for (int it = 0; it < 300; it++) {
        std::string m = tsv.substr(0, 200 + rnd() % (tsv.size() - 200));
        for (int k = 0; k < 1 + (int)(rnd() % 4); k++) m[rnd() % m.size()] = "\t,;\n-e.0A \r\x00\xff"[rnd() % 14];
        const int64_t r = rows.parse(m, 1 + (int)(rnd() % 4));
        CHECK(r >= 0 || r == DSP_EPARSE || r == DSP_ENOMEM || r == DSP_EKEY);
    }
    // ---- the host half of the device-side row parser: one pass copy + row starts (memory: exactly len + 1 / n + 1 entries),
    // and the same through pread on a file, block by block, with and without a newline behind the last row
    for (int variant = 0; variant < 2; variant++) {
        const std::string text = variant ? tsv.substr(0, tsv.size() - 1) : tsv;
        std::vector<char> dst(text.size() + 1);
        std::vector<uint64_t> off(nrows + 1);
        CHECK(dsp_copy_rows_index(text.data(), text.size(), dst.data(), off.data(), nrows) == nrows);
        CHECK(off[nrows] == text.size() + (variant ? 1 : 0) && memcmp(dst.data(), text.data(), text.size()) == 0 && dst[off[nrows] - 1] == '\n');
        CHECK(dsp_copy_rows_index(text.data(), text.size(), dst.data(), off.data(), nrows - 1) == DSP_ENOMEM);
        const std::string path = tmp + "/rows_index.tsv";
        FILE* f = fopen(path.c_str(), "wb");
        CHECK(f && fwrite(text.data(), 1, text.size(), f) == text.size());
        fclose(f);
        const int fd = open(path.c_str(), O_RDONLY);
        CHECK(fd >= 0);
        for (uint64_t budget : {(uint64_t)700, (uint64_t)5000, (uint64_t)90000, (uint64_t)text.size() + 64}) {
            for (int64_t want : {(int64_t)1, (int64_t)7, (int64_t)1000}) {
                uint64_t pos = 0;
                int64_t seen = 0;
                while (pos < text.size()) {
                    const uint64_t cap = budget < 4096 ? 4096 : budget;   // (a budget smaller than a row: the caller falls back to its buffer)
                    std::vector<char> buf(cap + 1);
                    std::vector<uint64_t> ro((size_t)want + 1);
                    uint64_t used = 0;
                    int64_t n = dsp_read_rows_index(fd, pos, text.size() - pos, budget, want, 1, buf.data(), ro.data(), &used);
                    if (n == 0 && used == 0) n = dsp_read_rows_index(fd, pos, text.size() - pos, cap, want, 1, buf.data(), ro.data(), &used);
                    CHECK(n > 0 && n <= want && used > 0);
                    CHECK(memcmp(buf.data(), text.data() + pos, (size_t)used) == 0 && buf[ro[n] - 1] == '\n');
                    seen += n;
                    pos += used;
                }
                CHECK(seen == nrows && pos == text.size());
            }
        }
        uint64_t used = 0;
        std::vector<char> buf(128);
        std::vector<uint64_t> ro(4);
        CHECK(dsp_read_rows_index(fd, 0, text.size() + 1000, 100, 2, 1, buf.data(), ro.data(), &used) == 0 && used == 0);      // no row fits
        CHECK(dsp_read_rows_index(fd, text.size() - 10, 5000, 100, 2, 1, buf.data(), ro.data(), &used) == DSP_EPARSE);         // the file ends early
        close(fd);
    }
    Rows wrong(11, 16, nrows);
    CHECK(wrong.parse(tsv, 2) == DSP_EPARSE);
    CHECK(rows.parse(tsv, 4) == nrows);

    // ---- call formatter
    std::vector<float> probs(nrows * 2);
    std::vector<uint8_t> labels(nrows);
    for (int64_t i = 0; i < nrows; i++) {
        const float p = (float)((rnd() % 2000001) / 2000000.0);
        probs[2 * i] = p; probs[2 * i + 1] = 1.0f - p; labels[i] = p < 0.5f;
    }
    probs[0] = 0.f; probs[1] = 1.f; probs[2] = 1e-7f; probs[3] = 1.f; probs[4] = 0.5f; probs[5] = 0.5f;
    std::vector<char> out(nrows * 200);
    const int64_t fb = dsp_format_calls(tsv.data(), rows.row_off.data(), rows.info_len.data(), probs.data(), 2, labels.data(),
                                        rows.kmer.data(), 13, nrows, out.data(), out.size(), 3);
    CHECK(fb > 0 && out[fb - 1] == '\n');
    CHECK(dsp_format_calls(tsv.data(), rows.row_off.data(), rows.info_len.data(), probs.data(), 2, labels.data(),
                           rows.kmer.data(), 13, nrows, out.data(), 100, 3) == DSP_ENOMEM);

    // ---- feature-row formatter (float64)
    {
        std::vector<double> m(nrows * 13), s(nrows * 13), g(nrows * 13 * 16);
        for (auto& v : m) v = ((int64_t)(rnd() % 4000001) - 2000000) / 1e6;
        for (auto& v : s) v = (rnd() % 1000001) / 1e6;
        for (auto& v : g) v = ((int64_t)(rnd() % 8000001) - 4000000) / 1e6;
        m[0] = 1e-7; m[1] = -0.0; m[2] = 1e22; m[3] = 5e-324; g[0] = 123456789.123456;
        std::vector<char> o2(nrows * 6000);
        const int64_t k = dsp_format_feature_rows(tsv.data(), rows.row_off.data(), rows.info_len.data(), rows.kmer.data(),
                                                  m.data(), s.data(), rows.lens.data(), g.data(), rows.labels.data(), 13, 16,
                                                  nrows, o2.data(), o2.size(), 4);
        CHECK(k > 0);
        // what it writes parses back to the same number of rows
        Rows back(13, 16, nrows);
        CHECK(back.parse(std::string(o2.data(), (size_t)k), 3) == nrows);
        CHECK(dsp_format_feature_rows(tsv.data(), rows.row_off.data(), rows.info_len.data(), rows.kmer.data(), m.data(),
                                      s.data(), rows.lens.data(), g.data(), rows.labels.data(), 13, 16, nrows, o2.data(), 1000,
                                      4) == DSP_ENOMEM);
        // the parts form: worst-case values (the longest float64 / int32) in an EXACTLY sized buffer -- ASan watches the end
        for (auto& v : m) v = -2.2250738585072014e-308;
        for (auto& v : g) v = -1.7976931348623157e308;
        std::vector<int32_t> lmin(nrows * 13, INT32_MIN), labmin(nrows, INT32_MIN);
        size_t cap = (size_t)nrows * dsp_feature_row_bound(13, 16);
        for (int64_t r = 0; r < nrows; ++r) cap += rows.info_len[r];
        std::vector<char> o3(cap);
        uint64_t off[5], len[5];
        const int64_t np = dsp_format_feature_rows_parts(tsv.data(), rows.row_off.data(), rows.info_len.data(), rows.kmer.data(),
                                                         m.data(), m.data(), lmin.data(), g.data(), labmin.data(), 13, 16, nrows,
                                                         o3.data(), o3.size(), 5, off, len);
        CHECK(np >= 1 && np <= 5);
        std::string joined;
        for (int64_t t = 0; t < np; ++t) { CHECK(off[t] + len[t] <= cap); joined.append(o3.data() + off[t], len[t]); }
        std::vector<char> o4(cap);
        const int64_t k2 = dsp_format_feature_rows(tsv.data(), rows.row_off.data(), rows.info_len.data(), rows.kmer.data(), m.data(),
                                                   m.data(), lmin.data(), g.data(), labmin.data(), 13, 16, nrows, o4.data(),
                                                   o4.size(), 3);
        CHECK(k2 == (int64_t)joined.size() && !memcmp(joined.data(), o4.data(), joined.size()));
        CHECK(dsp_format_feature_rows_parts(tsv.data(), rows.row_off.data(), rows.info_len.data(), rows.kmer.data(), m.data(),
                                            m.data(), lmin.data(), g.data(), labmin.data(), 13, 16, nrows, o3.data(), cap - 1, 5,
                                            off, len) == DSP_ENOMEM);
    }

    // ---- feature container: write (several block sizes), read back, corrupt
    for (int64_t br : {7, 64, 100000}) {
        const std::string path = tmp + "/asan.dspf";
        dsp_feat_writer* w = nullptr;
        CHECK(dsp_feat_writer_create(path.c_str(), 13, 16, br, &w) == DSP_OK);
        CHECK(dsp_feat_writer_add(w, nrows, rows.kmer.data(), rows.means.data(), rows.stds.data(), rows.lens.data(),
                                  rows.signals.data(), rows.labels.data(), tsv.data(), rows.row_off.data(),
                                  rows.info_len.data(), rows.read_off.data(), rows.read_len.data()) == DSP_OK);
        CHECK(dsp_feat_writer_add(w, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == DSP_OK);
        CHECK(dsp_feat_writer_close(w) == DSP_OK);
        dsp_feat_file* f = nullptr;
        CHECK(dsp_feat_open(path.c_str(), &f) == DSP_OK);
        int32_t L, S; int64_t n, nb;
        CHECK(dsp_feat_info(f, &L, &S, &n, &nb) == DSP_OK && L == 13 && S == 16 && n == nrows);
        int64_t seen = 0;
        for (int64_t b = 0; b < nb; b++) {
            int64_t bn, first, ib;
            CHECK(dsp_feat_block_info(f, b, &bn, &first, &ib) == DSP_OK && first == seen);
            Rows r2(13, 16, bn);
            std::vector<char> info((size_t)ib + 1);
            CHECK(dsp_feat_read_block(f, b, bn, r2.kmer.data(), r2.means.data(), r2.stds.data(), r2.lens.data(),
                                      r2.signals.data(), r2.labels.data(), info.data(), (size_t)ib, r2.row_off.data(),
                                      r2.info_len.data(), r2.read_off.data(), r2.read_len.data(), 3) == bn);
            CHECK(memcmp(r2.means.data(), rows.means.data() + seen * 13, (size_t)bn * 13 * 4) == 0);
            CHECK(dsp_feat_read_block(f, b, bn - 1, r2.kmer.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                                      nullptr, nullptr, nullptr, nullptr, 1) == DSP_ENOMEM);
            seen += bn;
        }
        CHECK(seen == nrows);
        CHECK(dsp_feat_block_info(f, nb, nullptr, nullptr, nullptr) == DSP_EINVAL);
        dsp_feat_close(f);
        // random corruptions of the file: open/read must fail cleanly or succeed, never crash
        const std::string blob = slurp(path.c_str());
        for (int it = 0; it < 60; it++) {
            std::string m = blob;
            if (it % 3 == 0) m.resize(rnd() % m.size());
            else for (int k = 0; k < 8; k++) m[(it % 3 == 1 ? rnd() % 128 : m.size() - 1 - rnd() % 256) % m.size()] = (char)rnd();
            const std::string p2 = tmp + "/asan_bad.dspf";
            FILE* fo = fopen(p2.c_str(), "wb"); CHECK(fo); fwrite(m.data(), 1, m.size(), fo); fclose(fo);
            dsp_feat_file* g = nullptr;
            if (dsp_feat_open(p2.c_str(), &g) == DSP_OK) {
                int64_t n2, nb2;
                dsp_feat_info(g, nullptr, nullptr, &n2, &nb2);
                for (int64_t b = 0; b < nb2 && b < 4; b++) {
                    int64_t bn, first, ib;
                    if (dsp_feat_block_info(g, b, &bn, &first, &ib) != DSP_OK || bn > 1000000 || ib > (1 << 28)) continue;
                    Rows r2(13, 16, bn);
                    std::vector<char> info((size_t)ib + 1);
                    dsp_feat_read_block(g, b, bn, r2.kmer.data(), r2.means.data(), r2.stds.data(), r2.lens.data(),
                                        r2.signals.data(), r2.labels.data(), info.data(), (size_t)ib, r2.row_off.data(),
                                        r2.info_len.data(), r2.read_off.data(), r2.read_len.data(), 2);
                }
                dsp_feat_close(g);
            }
        }
    }

    // ---- site enumerator
    {
        const char* seqs[3] = {"ACGTTCGACGNCGAACGTACGTCGCGCGATATATCGCGACGTTTACGCGAACGCGTTACGACGCGTACGATCGCGAATT", "CG", "TTTTTTTTTTTTTTTTTTTTTTTTTTTT"};
        std::vector<uint8_t> ev;
        std::vector<int64_t> off = {0};
        for (auto s : seqs) { ev.insert(ev.end(), s, s + strlen(s)); off.push_back((int64_t)ev.size()); }
        const char* chrom[3] = {"chr1", "chrUn_random_2", ""};
        const char* names[3] = {"read-a", "b", "a-very-long-read-name-0123456789-0123456789-0123456789"};
        const int64_t cstart[3] = {100, 0, 99999999999ll}, clen[3] = {5000, -1, 100000000000ll};
        for (int k : {1, 5, 13}) for (int nt : {1, 3}) {
            size_t need = 0;
            const int64_t n = dsp_extract_sites(3, ev.data(), off.data(), chrom, names, "tct", "+-+", cstart, clen, nullptr, nullptr,
                                                "CGCAGCCGCTG", k == 5 ? 1 : 4, k == 5 ? 2 : 3 - (k == 13), 0, k, 0, nullptr, nullptr,
                                                nullptr, 0, &need, nullptr, nullptr, nullptr, nullptr, nt);
            CHECK(n >= 0);
            std::vector<int32_t> sr(n + 1), sl(n + 1);
            std::vector<char> info(need + 1);
            std::vector<uint64_t> ro(n + 1);
            std::vector<uint32_t> il(n + 1), rdo(n + 1), rdl(n + 1);
            CHECK(dsp_extract_sites(3, ev.data(), off.data(), chrom, names, "tct", "+-+", cstart, clen, nullptr, nullptr,
                                    "CGCAGCCGCTG", k == 5 ? 1 : 4, k == 5 ? 2 : 3 - (k == 13), 0, k, n, sr.data(), sl.data(),
                                    info.data(), need, nullptr, ro.data(), il.data(), rdo.data(), rdl.data(), nt) == n);
            if (n > 0) CHECK(ro[n - 1] + il[n - 1] == need);
            if (n > 1) CHECK(dsp_extract_sites(3, ev.data(), off.data(), chrom, names, "tct", "+-+", cstart, clen, nullptr, nullptr,
                                               "CGCAGCCGCTG", k == 5 ? 1 : 4, k == 5 ? 2 : 3 - (k == 13), 0, k, n - 1, sr.data(),
                                               sl.data(), info.data(), need, nullptr, ro.data(), il.data(), rdo.data(),
                                               rdl.data(), nt) == DSP_ENOMEM);
        }
        CHECK(dsp_extract_sites(3, ev.data(), off.data(), chrom, names, "tct", "+-+", cstart, clen, nullptr, nullptr, "CG", 1, 2, 0,
                                12, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1) == DSP_EINVAL);
    }

    // ---- call_freq aggregator on the formatter's own output, on mutated text, and through add_block
    {
        dsp_freq* fq = dsp_freq_create(0.1);
        CHECK(fq != nullptr);
        dsp_freq_set_threads(fq, 3);
        CHECK(dsp_freq_add_calls_text(fq, out.data(), (size_t)fb, nullptr) >= 0);
        CHECK(dsp_freq_add_block(fq, tsv.data(), rows.row_off.data(), rows.info_len.data(), probs.data(), 2, labels.data(),
                                 rows.kmer.data(), 13, nrows) >= 0);
        for (int it = 0; it < 200; it++) {
            std::string m(out.data(), (size_t)fb);
            m.resize(1 + rnd() % m.size());
            for (int k = 0; k < 3; k++) m[rnd() % m.size()] = "\t\n-e.9 \r\x00"[rnd() % 9];
            std::vector<char> t(m.begin(), m.end());
            dsp_freq_add_calls_text(fq, t.data(), t.size(), it % 5 == 0 ? "chr1" : nullptr);
        }
        int64_t c, u, s;
        dsp_freq_counts(fq, &c, &u, &s);
        CHECK(c >= u && s >= 0);
        for (int sort = 0; sort < 2; sort++) for (int bed = 0; bed < 2; bed++) {
            const int64_t need = dsp_freq_format(fq, sort, bed, nullptr, 0);
            CHECK(need >= 0);
            std::vector<char> o3((size_t)need + 1);
            CHECK(dsp_freq_format(fq, sort, bed, o3.data(), (size_t)need) == need);
        }
        dsp_freq_destroy(fq);
    }
    // ---- gzip layer: BGZF round trip on all thread counts, truncated / corrupted members fail cleanly
    {
        std::string text;
        for (int i = 0; i < 40000; i++) text += "chr1\t" + std::to_string(rnd() % 1000000) + "\t+\t0.123456\n";
        std::vector<uint8_t> comp(text.size() + text.size() / 2 + 4096);
        const int64_t cb = dsp_bgzf_compress((const uint8_t*)text.data(), text.size(), comp.data(), comp.size(), 1, 3);
        CHECK(cb > 0);
        const int64_t eb = dsp_bgzf_eof(comp.data() + cb, comp.size() - (size_t)cb);
        CHECK(eb == 28);
        const size_t total = (size_t)(cb + eb);
        std::vector<uint64_t> off(4096);
        std::vector<uint32_t> isz(4096);
        const int64_t nm = dsp_gz_index(comp.data(), total, 4095, off.data(), isz.data());
        CHECK(nm > 2);
        std::vector<uint8_t> back(text.size() + 16);
        for (int nt : {1, 4}) {
            CHECK(dsp_gz_inflate_members(comp.data(), off.data(), isz.data(), 0, nm, back.data(), back.size(), nt) == (int64_t)text.size());
            CHECK(memcmp(back.data(), text.data(), text.size()) == 0);
        }
        CHECK(dsp_gz_inflate_members(comp.data(), off.data(), isz.data(), 0, nm, back.data(), 100, 2) == DSP_ENOMEM);
        CHECK(dsp_gz_index(comp.data(), total - 5, 4095, off.data(), isz.data()) == -1);  // truncated: not a BGZF chain
        for (int it = 0; it < 40; it++) {
            std::vector<uint8_t> bad(comp.begin(), comp.begin() + (long)total);
            bad[40 + rnd() % (total - 80)] ^= (uint8_t)(1 + rnd() % 255);
            std::vector<uint64_t> o2(4096);
            std::vector<uint32_t> i2(4096);
            const int64_t n2 = dsp_gz_index(bad.data(), total, 4095, o2.data(), i2.data());
            if (n2 > 0) {
                uint64_t sum = 0;
                for (int64_t m = 0; m < n2; m++) sum += i2[(size_t)m];
                std::vector<uint8_t> b2((size_t)sum + 16);
                const int64_t r = dsp_gz_inflate_members(bad.data(), o2.data(), i2.data(), 0, n2, b2.data(), b2.size(), 2);
                CHECK(r >= 0 || r == DSP_EPARSE || r == DSP_ENOMEM);
            }
        }
    }

    // ---- streaming reader of a foreign .gz (zlib gzip members over an mmap): multi-member, zero padding, truncation at
    //      every region of the file, byte flips -- must deliver the text or fail with DSP_EPARSE, never over-read
    {
        std::string text;
        for (int i = 0; i < 30000; i++) text += "chr2\t" + std::to_string(rnd() % 1000000) + "\t-\t0.654321\tACGTA\n";
        auto gz_member = [](const std::string& t) {
            std::vector<uint8_t> o(compressBound((uLong)t.size()) + 64);
            z_stream z;
            memset(&z, 0, sizeof(z));
            deflateInit2(&z, 1, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY);
            z.next_in = (Bytef*)t.data(); z.avail_in = (uInt)t.size();
            z.next_out = o.data(); z.avail_out = (uInt)o.size();
            deflate(&z, Z_FINISH);
            o.resize(z.total_out);
            deflateEnd(&z);
            return o;
        };
        std::vector<uint8_t> file = gz_member(text.substr(0, text.size() / 3));
        const std::vector<uint8_t> m2 = gz_member(text.substr(text.size() / 3));
        file.insert(file.end(), m2.begin(), m2.end());
        file.insert(file.end(), 300, 0);  // zero padding after the last member is legal
        const std::string p = tmp + "/foreign.gz";
        auto write_file = [&](const std::vector<uint8_t>& b) { FILE* f = fopen(p.c_str(), "wb"); if (!b.empty()) fwrite(b.data(), 1, b.size(), f); fclose(f); };
        auto read_all = [&](std::string& got) -> int64_t {
            dsp_gz_stream* st = dsp_gz_open(p.c_str());
            CHECK(st != nullptr);
            std::vector<uint8_t> buf(70001);
            got.clear();
            int64_t rc;
            while ((rc = dsp_gz_read(st, buf.data(), buf.size())) > 0) got.append((const char*)buf.data(), (size_t)rc);
            CHECK(dsp_gz_bytes_in(st) <= file.size());
            dsp_gz_close(st);
            return rc;
        };
        std::string got;
        write_file(file);
        CHECK(read_all(got) == 0 && got == text);
        for (size_t cut : {(size_t)0, (size_t)5, (size_t)17, file.size() / 5, file.size() / 3, file.size() / 2, file.size() - 310, file.size() - 301}) {
            write_file(std::vector<uint8_t>(file.begin(), file.begin() + (long)cut));
            const int64_t rc = read_all(got);
            if (cut == 0) CHECK(rc == 0 && got.empty());                       // an empty file reads as empty text
            else CHECK(rc == DSP_EPARSE && g_err.find("gzip stream") != std::string::npos);
        }
        for (int it = 0; it < 30; it++) {
            std::vector<uint8_t> bad = file;
            bad[rnd() % (bad.size() - 300)] ^= (uint8_t)(1 + rnd() % 255);
            write_file(bad);
            const int64_t rc = read_all(got);
            CHECK(rc == DSP_EPARSE || (rc == 0 && got.size() == text.size()));  // (a flip in a header's MTIME / OS byte changes nothing)
        }
        dsp_gz_close(nullptr);
        CHECK(dsp_gz_open((tmp + "/absent.gz").c_str()) == nullptr);
    }
    // ---- parallel inflater of one gzip stream (its own deflate decoder): small chunks so that many chunks start in the
    //      middle of the stream with unknown windows; then truncations and byte flips -- text out or DSP_EPARSE, no over-read
    {
        std::string text;
        for (int i = 0; i < 60000; i++) text += "chr3\t" + std::to_string(rnd() % 1000000) + "\t+\t0." + std::to_string(rnd() % 1000000) + "\tACGTACGTACGTA\n";
        auto gz_member = [](const std::string& t, int level) {
            std::vector<uint8_t> o(compressBound((uLong)t.size()) + 64);
            z_stream z;
            memset(&z, 0, sizeof(z));
            deflateInit2(&z, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY);
            z.next_in = (Bytef*)t.data(); z.avail_in = (uInt)t.size();
            z.next_out = o.data(); z.avail_out = (uInt)o.size();
            deflate(&z, Z_FINISH);
            o.resize(z.total_out);
            deflateEnd(&z);
            return o;
        };
        std::vector<uint8_t> file = gz_member(text.substr(0, text.size() / 2), 6);
        const std::vector<uint8_t> m2 = gz_member(text.substr(text.size() / 2), 1);
        file.insert(file.end(), m2.begin(), m2.end());
        const std::string p = tmp + "/pgz.gz";
        auto write_file = [&](const std::vector<uint8_t>& b) { FILE* f = fopen(p.c_str(), "wb"); if (!b.empty()) fwrite(b.data(), 1, b.size(), f); fclose(f); };
        auto read_all = [&](std::string& got, int nt, uint64_t chunk) -> int64_t {
            dsp_pgz* z = dsp_pgz_open(p.c_str(), nt, chunk);
            if (!z) return DSP_EINVAL;
            std::vector<uint8_t> buf(50021);
            got.clear();
            int64_t rc;
            while ((rc = dsp_pgz_read(z, buf.data(), buf.size())) > 0) got.append((const char*)buf.data(), (size_t)rc);
            dsp_pgz_close(z);
            return rc;
        };
        std::string got;
        write_file(file);
        for (int nt : {1, 3, 6}) {
            CHECK(read_all(got, nt, 65536) == 0 && got == text);
        }
        CHECK(read_all(got, 4, 0) == 0 && got == text);
        for (size_t cut : {(size_t)10, (size_t)19, file.size() / 7, file.size() / 2, file.size() - 9, file.size() - 1}) {
            write_file(std::vector<uint8_t>(file.begin(), file.begin() + (long)cut));
            const int64_t rc = read_all(got, 4, 65536);
            CHECK(rc == DSP_EPARSE || rc == DSP_EINVAL);
        }
        for (int it = 0; it < 60; it++) {
            std::vector<uint8_t> bad = file;
            const int flips = 1 + (int)(rnd() % 3);
            for (int k = 0; k < flips; k++) bad[rnd() % bad.size()] ^= (uint8_t)(1 + rnd() % 255);
            write_file(bad);
            const int64_t rc = read_all(got, 1 + (int)(rnd() % 5), 65536 + (rnd() % 3) * 40000);
            CHECK(rc == DSP_EPARSE || rc == DSP_EINVAL || (rc == 0 && got.size() == text.size()));
        }
    }
    // ---- shared-memory ring: one producer thread, two consumer threads, more blocks than slots, then a failing producer
    {
        const std::string name = "/dsp_asan_ring_" + std::to_string((long)getpid());
        dsp_shm_ring* prod = dsp_shm_ring_create(name.c_str(), 3, 5000);
        CHECK(prod != nullptr && dsp_shm_ring_slot_bytes(prod) >= 5000);
        const int nblocks = 17;
        std::atomic<uint64_t> sum_got{0};
        auto consumer = [&](int li) {
            dsp_shm_ring* c = dsp_shm_ring_attach(name.c_str(), 10.0);
            CHECK(c != nullptr);
            for (uint64_t k = 0;; k++) {
                const uint8_t* data; uint64_t len, first, rows;
                const int32_t rc = dsp_shm_ring_wait(c, k * 2 + (uint64_t)li, 10.0, &data, &len, &first, &rows);
                if (rc == 1) break;
                CHECK(rc == 0 && len == 100 + (k * 2 + (uint64_t)li) * 10 && first == 7 * (k * 2 + (uint64_t)li) && rows == 3);
                for (uint64_t i = 0; i < len; i++) sum_got += data[i];
                dsp_shm_ring_release(c, k * 2 + (uint64_t)li);
            }
            dsp_shm_ring_close(c, 0);
        };
        std::thread c0(consumer, 0), c1(consumer, 1);
        uint64_t sum_put = 0;
        for (int i = 0; i < nblocks; i++) {
            uint8_t* slot = dsp_shm_ring_acquire(prod, (uint64_t)i, 10.0);
            CHECK(slot != nullptr);
            const uint64_t len = 100 + (uint64_t)i * 10;
            for (uint64_t j = 0; j < len; j++) { slot[j] = (uint8_t)(i + j); sum_put += slot[j]; }
            CHECK(dsp_shm_ring_publish(prod, (uint64_t)i, len, 7 * (uint64_t)i, 3) == 0);
        }
        CHECK(dsp_shm_ring_publish(prod, 99, 1u << 30, 0, 0) == DSP_EINVAL);   // larger than a slot
        dsp_shm_ring_finish(prod, (uint64_t)nblocks, 0, nullptr);
        c0.join(); c1.join();
        CHECK(sum_got.load() == sum_put);
        dsp_shm_ring_close(prod, 1);
        CHECK(dsp_shm_ring_attach(name.c_str(), 0.05) == nullptr);               // unlinked
        prod = dsp_shm_ring_create(name.c_str(), 2, 64);
        dsp_shm_ring* c = dsp_shm_ring_attach(name.c_str(), 5.0);
        CHECK(prod && c);
        dsp_shm_ring_finish(prod, 0, DSP_EPARSE, "truncated gzip stream: test");
        const uint8_t* data; uint64_t len;
        CHECK(dsp_shm_ring_wait(c, 0, 5.0, &data, &len, nullptr, nullptr) == DSP_EPARSE && g_err.find("truncated") != std::string::npos);
        dsp_shm_ring_abort(c);
        CHECK(dsp_shm_ring_acquire(prod, 2, 5.0) == nullptr);                   // a consumer gave up: the producer stops waiting
        dsp_shm_ring_close(c, 0);
        dsp_shm_ring_close(prod, 1);
    }

    // ---- fast5 reader (only when an HDF5 library is on this host): every F7 file, other groups, the region filter
    if (dsp_fast5_available()) {
        const char* files[] = {"read_00000-67ee_ch101_read0_strand.fast5", "read_00001-f97f_ch101_read1_strand.fast5",
                               "read_00002-05ce_ch101_read2_strand.fast5", "read_00009-939b_ch101_read9_strand.fast5",
                               "read_00010-d091_ch101_read10_strand.fast5", "sub/read_00011-c1cf_ch101_read11_strand.fast5",
                               "not_hdf5.fast5", "absent.fast5"};
        int ok = 0, failed = 0, skipped = 0;
        for (const char* f : files) {
            const std::string p = golden + "/fast5/reads/" + f;
            for (const char* only : {(const char*)nullptr, "chr2"}) {
                dsp_fast5_read rec;
                const int32_t rc = dsp_fast5_load(p.c_str(), "RawGenomeCorrected_000", "BaseCalled_template", only, &rec);
                if (rc == 0) {
                    CHECK(rec.n_raw > 0 && rec.n_events > 0 && rec.raw && rec.ev_start && rec.ev_len && rec.ev_base);
                    int64_t last = rec.ev_start[rec.n_events - 1] + rec.ev_len[rec.n_events - 1];
                    CHECK(last <= rec.n_raw && rec.digitisation == 8192.0);
                    ok++;
                } else if (rc == DSP_FAST5_SKIPPED) {
                    skipped++;
                } else {
                    CHECK(rc == DSP_EPARSE && !g_err.empty());
                    failed++;
                }
                dsp_fast5_free(&rec);
                dsp_fast5_free(&rec);  // idempotent
            }
            dsp_fast5_read rec;
            CHECK(dsp_fast5_load(p.c_str(), "NoSuchGroup_000", "BaseCalled_template", nullptr, &rec) != 0);
            dsp_fast5_free(&rec);
        }
        CHECK(ok >= 4 && failed >= 4 && skipped >= 4);
        // damaged copies: truncated at several lengths (chunk addresses then point past the end of the file: the reader
        // checks every size against the file before allocating) and byte flips in the chunk payloads (the zlib streams'
        // Adler-32 catches them, or the values change): a read fails or loads, the process survives
        {
            std::vector<uint8_t> good;
            {
                FILE* f = fopen((golden + "/fast5/reads/" + files[0]).c_str(), "rb");
                CHECK(f != nullptr);
                uint8_t buf[65536];
                size_t k;
                while ((k = fread(buf, 1, sizeof(buf), f)) > 0) good.insert(good.end(), buf, buf + k);
                fclose(f);
            }
            const std::string p = tmp + "/damaged.fast5";
            int n_fail = 0, n_ok = 0;
            for (int it = 0; it < 24; it++) {
                std::vector<uint8_t> bad = good;
                if (it < 8) bad.resize(good.size() * (size_t)(it + 1) / 10);
                else bad[good.size() / 2 + rnd() % (good.size() / 2)] ^= (uint8_t)(1 + rnd() % 255);
                FILE* f = fopen(p.c_str(), "wb");
                fwrite(bad.data(), 1, bad.size(), f);
                fclose(f);
                dsp_fast5_read rec;
                const int32_t rc = dsp_fast5_load(p.c_str(), "RawGenomeCorrected_000", "BaseCalled_template", nullptr, &rec);
                CHECK(rc == 0 || rc == DSP_EPARSE);
                (rc == 0 ? n_ok : n_fail)++;
                dsp_fast5_free(&rec);
            }
            CHECK(n_fail >= 8);
            printf("fast5 damaged copies: %d failed cleanly, %d still loaded\n", n_fail, n_ok);
        }
        printf("fast5: %d loaded, %d failed, %d skipped\n", ok, failed, skipped);
    }
    printf("host_asan: ok\n");
    return 0;
}
