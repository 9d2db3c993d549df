// dsp_freq.cpp -- per-site modification frequency from per-read calls (the reference's `call_freq`):
// calculate_mods_frequency (deepsignal_plant/call_mods_freq.py:29-74), ModRecord / SiteStats
// (utils/txt_formater.py:8-46) and write_sitekey2stats (call_mods_freq.py:77-122).
//
// A site is keyed by (chromosome, pos) (txt_formater.py:12); the first USED record of a site fixes its strand,
// pos_in_strand and k-mer (:52-56); a record is used when |prob_0 - prob_1| >= prob_cf in double arithmetic on
// the PRINTED probabilities (txt_formater.py:23-26); prob sums are accumulated sequentially in record order in
// double (so the "%.3f" output rounds exactly like the reference, including half-way cases).
//
// Two feeders: per-read call text (the file format call_mods writes), and parsed call_mods blocks straight
// from the GPU results (dsp_freq_add_block) -- the latter derives, with the formatter's own float32
// arithmetic, exactly the decimal values the per-read file would have carried, so skipping the text
// round-trip cannot change a single digit.
#include "dsp_amd.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

extern "C" void dsp_set_error_(const char* msg);
extern "C" int dsp_format_prob_f32_(float x, char* out);  // dsp_text.cpp: numpy str(float32)
extern "C" float dsp_np_round6_f32_(float x);             // dsp_text.cpp: numpy round(float32, 6)

namespace {

int freq_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    dsp_set_error_(buf);
    return code;
}

struct Site {
    std::string chrom;
    long long pos;
    std::string strand, kmer;
    long long pos_in_strand;
    double prob0 = 0.0, prob1 = 0.0;
    long long met = 0, unmet = 0, coverage = 0;
};

struct Key {
    std::string chrom;
    long long pos;
    bool operator==(const Key& o) const { return pos == o.pos && chrom == o.chrom; }
};
struct KeyHash {
    size_t operator()(const Key& k) const {
        return std::hash<std::string>()(k.chrom) * 1000003u ^ std::hash<long long>()(k.pos);
    }
};

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\f' || c == '\v'; }

bool parse_ll(const char* p, const char* e, long long* out) {
    while (p < e && is_space(*p)) ++p;
    while (e > p && is_space(e[-1])) --e;
    if (p >= e) return false;
    bool neg = false;
    if (*p == '+' || *p == '-') { neg = *p == '-'; ++p; }
    if (p >= e) return false;
    long long v = 0;
    for (; p < e; ++p) {
        if (*p < '0' || *p > '9' || v > (1ll << 58)) return false;
        v = v * 10 + (*p - '0');
    }
    *out = neg ? -v : v;
    return true;
}

bool parse_double(const char* p, const char* e, double* out) {  // Python float(): correctly rounded
    char tmp[96];
    while (p < e && is_space(*p)) ++p;
    while (e > p && is_space(e[-1])) --e;
    const size_t n = (size_t)(e - p);
    if (n == 0 || n >= sizeof(tmp)) return false;
    memcpy(tmp, p, n);
    tmp[n] = 0;
    char* endp = nullptr;
    *out = strtod(tmp, &endp);
    return endp == tmp + n;
}

}  // namespace

struct dsp_freq {
    double prob_cf;
    std::vector<Site> sites;  // insertion order = order of first used record (Python dict order)
    std::unordered_map<Key, size_t, KeyHash> index;
    long long count = 0, used = 0;

    void add(const char* chrom, size_t chrom_len, long long pos, const char* strand, size_t strand_len,
             long long pos_in_strand, double p0, double p1, long long label, const char* kmer, size_t kmer_len) {
        ++count;
        if (std::fabs(p0 - p1) < prob_cf) return;  // txt_formater.py:23-26
        Key k{std::string(chrom, chrom_len), pos};
        auto it = index.find(k);
        size_t idx;
        if (it == index.end()) {
            idx = sites.size();
            Site s;
            s.chrom = k.chrom; s.pos = pos;
            s.strand.assign(strand, strand_len); s.kmer.assign(kmer, kmer_len);
            s.pos_in_strand = pos_in_strand;
            sites.push_back(std::move(s));
            index.emplace(std::move(k), idx);
        } else {
            idx = it->second;
        }
        Site& s = sites[idx];
        s.prob0 += p0; s.prob1 += p1;
        s.coverage += 1;
        if (label == 1) s.met += 1; else s.unmet += 1;
        ++used;
    }
};

extern "C" {

dsp_freq* dsp_freq_create(double prob_cf) {
    dsp_freq* f = new (std::nothrow) dsp_freq();
    if (f) f->prob_cf = prob_cf;
    return f;
}

void dsp_freq_destroy(dsp_freq* f) { delete f; }

int64_t dsp_freq_add_calls_text(dsp_freq* f, const char* text, size_t len, const char* contig) {
    if (!f || (!text && len)) return freq_fail(DSP_EINVAL, "NULL argument");
    const size_t clen = contig ? strlen(contig) : 0;
    const char* p = text;
    const char* e = text + len;
    int64_t nline = 0;
    while (p < e) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
        const char* le = nl ? nl : e;
        const char* ls = p;
        p = nl ? nl + 1 : e;
        while (ls < le && is_space(*ls)) ++ls;
        while (le > ls && is_space(le[-1])) --le;
        const char* fs[10];
        const char* fe[10];
        int nf = 0;
        for (const char* q = ls; nf < 10;) {
            const char* t = (const char*)memchr(q, '\t', (size_t)(le - q));
            if (!t) t = le;
            fs[nf] = q; fe[nf] = t; ++nf;
            if (t == le) break;
            q = t + 1;
        }
        if (nf < 10) return freq_fail(DSP_EPARSE, "malformed call line %lld: need 10 tab-separated columns", (long long)nline);
        long long pos, pis, label;
        double p0, p1;
        if (!parse_ll(fs[1], fe[1], &pos) || !parse_ll(fs[3], fe[3], &pis) || !parse_double(fs[6], fe[6], &p0) ||
            !parse_double(fs[7], fe[7], &p1) || !parse_ll(fs[8], fe[8], &label))
            return freq_fail(DSP_EPARSE, "malformed call line %lld: bad number", (long long)nline);
        ++nline;
        if (contig && !((size_t)(fe[0] - fs[0]) == clen && memcmp(fs[0], contig, clen) == 0)) continue;
        f->add(fs[0], (size_t)(fe[0] - fs[0]), pos, fs[2], (size_t)(fe[2] - fs[2]), pis, p0, p1, label, fs[9],
               (size_t)(fe[9] - fs[9]));
    }
    return nline;
}

int64_t dsp_freq_add_block(dsp_freq* f, const char* text, const uint64_t* row_off, const uint32_t* info_len,
                           const float* probs, int32_t num_classes, const uint8_t* labels, const uint8_t* kmer,
                           int32_t seq_len, int64_t n) {
    if (!f || !text || !row_off || !info_len || !probs || !labels || !kmer || num_classes < 2 || seq_len < 1)
        return freq_fail(DSP_EINVAL, "NULL / bad argument");
    static const char* code2base = "ACGTNWSMKRYBVDHZ";
    const int center = seq_len / 2;
    const int k0 = center - 2 >= 0 ? center - 2 : 0;
    const int k1 = center + 3 <= seq_len ? center + 3 : seq_len;
    char k5[16], num[64];
    for (int64_t r = 0; r < n; ++r) {
        const char* ls = text + row_off[r];
        const char* le = ls + info_len[r];
        const char* fs[4];
        const char* fe[4];
        int nf = 0;
        for (const char* q = ls; nf < 4;) {
            const char* t = (const char*)memchr(q, '\t', (size_t)(le - q));
            if (!t) t = le;
            fs[nf] = q; fe[nf] = t; ++nf;
            if (t == le) break;
            q = t + 1;
        }
        long long pos, pis;
        if (nf < 4 || !parse_ll(fs[1], fe[1], &pos) || !parse_ll(fs[3], fe[3], &pis))
            return freq_fail(DSP_EPARSE, "row %lld: bad pos / pos_in_strand column", (long long)r);
        // the values the per-read file would carry (call_modifications.py:177-179, :186-187), via the same text
        const float a = probs[r * num_classes], b = probs[r * num_classes + 1];
        volatile float sum = a + b;
        volatile float q = a / sum;
        const float z0 = dsp_np_round6_f32_(q);
        volatile float om = 1.0f - z0;
        const float z1 = dsp_np_round6_f32_(om);
        num[dsp_format_prob_f32_(z0, num)] = 0;
        const double p0 = strtod(num, nullptr);
        num[dsp_format_prob_f32_(z1, num)] = 0;
        const double p1 = strtod(num, nullptr);
        for (int i = k0; i < k1; ++i) k5[i - k0] = code2base[kmer[r * seq_len + i] & 15];
        f->add(fs[0], (size_t)(fe[0] - fs[0]), pos, fs[2], (size_t)(fe[2] - fs[2]), pis, p0, p1, (long long)labels[r], k5,
               (size_t)(k1 - k0));
    }
    return n;
}

void dsp_freq_counts(const dsp_freq* f, int64_t* count, int64_t* used, int64_t* sites) {
    if (!f) return;
    if (count) *count = f->count;
    if (used) *used = f->used;
    if (sites) *sites = (int64_t)f->sites.size();
}

// write_sitekey2stats (call_mods_freq.py:77-122).  Returns bytes needed; writes at most cap bytes.
int64_t dsp_freq_format(const dsp_freq* f, int32_t is_sort, int32_t is_bed, char* out, size_t cap) {
    if (!f) return freq_fail(DSP_EINVAL, "NULL argument");
    std::vector<size_t> order(f->sites.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    if (is_sort)  // sorted(keys, key=split_key): (chrom str, pos int)
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
            const Site& x = f->sites[a];
            const Site& y = f->sites[b];
            const int c = x.chrom.compare(y.chrom);
            return c != 0 ? c < 0 : x.pos < y.pos;
        });
    std::string s;
    s.reserve(order.size() * 72);
    char buf[512];
    for (size_t i : order) {
        const Site& t = f->sites[i];
        if (t.coverage <= 0) continue;
        const double rmet = (double)t.met / (double)t.coverage;
        int k;
        if (is_bed) {
            const long long pct = (long long)std::nearbyint(rmet * 100 + 0.001);  // int(round(rmet*100+0.001, 0)), :110
            k = snprintf(buf, sizeof(buf), "%s\t%lld\t%lld\t.\t%lld\t%s\t%lld\t%lld\t0,0,0\t%lld\t%lld\n", t.chrom.c_str(),
                         t.pos, t.pos + 1, t.coverage, t.strand.c_str(), t.pos, t.pos + 1, t.coverage, pct);
        } else {
            k = snprintf(buf, sizeof(buf), "%s\t%lld\t%s\t%lld\t%.3f\t%.3f\t%lld\t%lld\t%lld\t%.4f\t%s\n", t.chrom.c_str(), t.pos,
                         t.strand.c_str(), t.pos_in_strand, t.prob0, t.prob1, t.met, t.unmet, t.coverage, rmet,
                         t.kmer.c_str());
        }
        if (k < 0 || (size_t)k >= sizeof(buf)) {  // very long contig names: format into a growing string
            std::string big(1024 + t.chrom.size() * 2 + t.kmer.size(), '\0');
            k = is_bed ? snprintf(&big[0], big.size(), "%s\t%lld\t%lld\t.\t%lld\t%s\t%lld\t%lld\t0,0,0\t%lld\t%lld\n", t.chrom.c_str(),
                                  t.pos, t.pos + 1, t.coverage, t.strand.c_str(), t.pos, t.pos + 1, t.coverage,
                                  (long long)std::nearbyint(rmet * 100 + 0.001))
                       : snprintf(&big[0], big.size(), "%s\t%lld\t%s\t%lld\t%.3f\t%.3f\t%lld\t%lld\t%lld\t%.4f\t%s\n", t.chrom.c_str(),
                                  t.pos, t.strand.c_str(), t.pos_in_strand, t.prob0, t.prob1, t.met, t.unmet, t.coverage, rmet,
                                  t.kmer.c_str());
            s.append(big.data(), (size_t)k);
        } else {
            s.append(buf, (size_t)k);
        }
    }
    if (out && cap) memcpy(out, s.data(), std::min(cap, s.size()));
    return (int64_t)s.size();
}

}  // extern "C"
