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
#include "dsp_threads.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
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

// strand and k-mer of a site are short ("+", a 5-mer): kept inline so that creating a site allocates nothing; longer
// ones (malformed or exotic input) spill into dsp_freq::long_text.
struct Site {
    uint32_t chrom;  // index into dsp_freq::chroms
    uint8_t strand_len, kmer_len;  // 255 = spilled
    char strand[6];
    char kmer[20];
    long long pos;
    long long pos_in_strand;
    double prob0 = 0.0, prob1 = 0.0;
    long long met = 0, unmet = 0, coverage = 0;
    uint64_t first_use;  // global sequence number of the record that created the site (= Python dict order)
};

// (chromosome id, pos) -> site index: open addressing, linear probing, no per-entry allocation
struct SiteIndex {
    struct Slot { long long pos; uint32_t chrom; uint32_t idx; };  // idx == UINT32_MAX: empty
    std::vector<Slot> slots;
    size_t used = 0, mask = 0;
    static uint64_t hash(uint32_t chrom, long long pos) {  // splitmix64 finaliser
        uint64_t x = (uint64_t)pos * 0x9E3779B97F4A7C15ull + ((uint64_t)chrom << 48) + chrom;
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 27; x *= 0x94D049BB133111EBull;
        x ^= x >> 31;
        return x;
    }
    void rehash(size_t cap) {
        std::vector<Slot> old;
        old.swap(slots);
        slots.assign(cap, Slot{0, 0, UINT32_MAX});
        mask = cap - 1;
        for (const Slot& s : old)
            if (s.idx != UINT32_MAX) {
                size_t i = hash(s.chrom, s.pos) & mask;
                while (slots[i].idx != UINT32_MAX) i = (i + 1) & mask;
                slots[i] = s;
            }
    }
    // returns the slot of (chrom, pos); *found tells whether it holds a site already
    Slot* find_or_slot(uint32_t chrom, long long pos, bool* found) {
        if (slots.empty() || (used + 1) * 10 > slots.size() * 6) rehash(slots.empty() ? 1024 : slots.size() * 2);
        size_t i = hash(chrom, pos) & mask;
        while (slots[i].idx != UINT32_MAX) {
            if (slots[i].pos == pos && slots[i].chrom == chrom) { *found = true; return &slots[i]; }
            i = (i + 1) & mask;
        }
        *found = false;
        return &slots[i];
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

// Python float(): correctly rounded.  Fast exact path (Clinger): up to 15 significant digits and a decimal exponent
// within +-22 -- the integer mantissa and the power of ten are both exact doubles, so one multiply or divide is
// correctly rounded; everything else goes to strtod.
bool parse_double(const char* p, const char* e, double* out) {
    static const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                      1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    while (p < e && is_space(*p)) ++p;
    while (e > p && is_space(e[-1])) --e;
    if (p >= e) return false;
    const char* q = p;
    bool neg = false;
    if (*q == '+' || *q == '-') { neg = *q == '-'; ++q; }
    uint64_t mant = 0;
    int digits = 0, frac = 0;
    bool any = false, dot = false, fast = true;
    for (; q < e; ++q) {
        if (*q >= '0' && *q <= '9') {
            any = true;
            if (mant || *q != '0') {
                if (++digits > 15) { fast = false; break; }
                mant = mant * 10 + (uint64_t)(*q - '0');
            }
            if (dot) ++frac;
        } else if (*q == '.' && !dot) {
            dot = true;
        } else {
            break;
        }
    }
    if (fast && any) {
        int ex = 0;
        if (q < e && (*q == 'e' || *q == 'E')) {
            const char* r = q + 1;
            bool eneg = false;
            if (r < e && (*r == '+' || *r == '-')) { eneg = *r == '-'; ++r; }
            if (r >= e) fast = false;
            int v = 0;
            for (; r < e && fast; ++r) {
                if (*r < '0' || *r > '9' || v > 1000) fast = false;
                else v = v * 10 + (*r - '0');
            }
            ex = eneg ? -v : v;
            q = r;
        }
        if (fast && q == e) {
            const int e10 = ex - frac;
            if (e10 >= -22 && e10 <= 22) {
                double v = (double)mant;
                v = e10 < 0 ? v / kPow10[-e10] : v * kPow10[e10];
                *out = neg ? -v : v;
                return true;
            }
        }
    }
    char tmp[96];
    size_t n = (size_t)(e - p);
    if (n == 0 || n >= sizeof(tmp)) return false;
    memcpy(tmp, p, n);
    tmp[n] = 0;
    if (memchr(tmp, '_', n)) {   // Python's float() (ModRecord, utils/txt_formater.py:8-21): one underscore between two digits
        size_t k = 0;
        for (size_t i = 0; i < n; ++i) {
            if (tmp[i] == '_') {
                if (i == 0 || i + 1 >= n || tmp[i - 1] < '0' || tmp[i - 1] > '9' || tmp[i + 1] < '0' || tmp[i + 1] > '9') return false;
                continue;
            }
            tmp[k++] = tmp[i];
        }
        n = k;
        tmp[n] = 0;
    }
    if (memchr(tmp, 'x', n) || memchr(tmp, 'X', n)) return false;  // strtod takes hex floats, Python's float() does not
    char* endp = nullptr;
    *out = strtod(tmp, &endp);
    return endp == tmp + n;
}

// one parsed per-read call line; the string fields point into the caller's text
struct Rec {
    const char *chrom, *strand, *kmer;
    uint32_t chrom_len, strand_len, kmer_len;
    long long pos, pis, label;
    double p0, p1;
    uint32_t cid;  // interned chromosome, filled by the sequential interning pass
};

// 0 = ok, 1 = fewer than 10 columns, 2 = bad number
int parse_call_line(const char* ls, const char* le, Rec* r) {
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
    if (nf < 10) return 1;
    if (!parse_ll(fs[1], fe[1], &r->pos) || !parse_ll(fs[3], fe[3], &r->pis) || !parse_double(fs[6], fe[6], &r->p0) ||
        !parse_double(fs[7], fe[7], &r->p1) || !parse_ll(fs[8], fe[8], &r->label))
        return 2;
    r->chrom = fs[0]; r->chrom_len = (uint32_t)(fe[0] - fs[0]);
    r->strand = fs[2]; r->strand_len = (uint32_t)(fe[2] - fs[2]);
    r->kmer = fs[9]; r->kmer_len = (uint32_t)(fe[9] - fs[9]);
    return 0;
}

}  // namespace

// The table is split into kParts partitions by key hash.  A record only ever touches the partition of its site, so
// partitions can be filled by different threads, each scanning the records in order: every site still sees its
// records in file order (the double sums associate exactly like a sequential pass) and remembers the sequence
// number of the record that created it, from which the global insertion order is rebuilt.  The partition count is
// fixed, so the result does not depend on the number of threads.
constexpr int kParts = 16;

struct Partition {
    std::vector<Site> sites;
    SiteIndex index;
    std::unordered_map<size_t, std::pair<std::string, std::string>> long_text;  // spilled strand / k-mer by site index
    long long used = 0;

    void apply(uint32_t cid, long long pos, const char* strand, size_t strand_len, long long pos_in_strand, double p0,
               double p1, long long label, const char* kmer, size_t kmer_len, uint64_t seq) {
        bool found;
        SiteIndex::Slot* slot = index.find_or_slot(cid, pos, &found);
        size_t idx;
        if (!found) {
            idx = sites.size();
            Site s;
            s.chrom = cid; s.pos = pos; s.pos_in_strand = pos_in_strand; s.first_use = seq;
            if (strand_len <= sizeof(s.strand) && kmer_len <= sizeof(s.kmer)) {
                s.strand_len = (uint8_t)strand_len; s.kmer_len = (uint8_t)kmer_len;
                memcpy(s.strand, strand, strand_len);
                memcpy(s.kmer, kmer, kmer_len);
            } else {
                s.strand_len = s.kmer_len = 255;
                long_text.emplace(idx, std::make_pair(std::string(strand, strand_len), std::string(kmer, kmer_len)));
            }
            sites.push_back(s);
            slot->pos = pos; slot->chrom = cid; slot->idx = (uint32_t)idx;
            ++index.used;
        } else {
            idx = slot->idx;
        }
        Site& s = sites[idx];
        s.prob0 += p0; s.prob1 += p1;
        s.coverage += 1;
        if (label == 1) s.met += 1; else s.unmet += 1;
        ++used;
    }
    std::string strand_of(size_t i) const {
        const Site& s = sites[i];
        return s.strand_len == 255 ? long_text.at(i).first : std::string(s.strand, s.strand_len);
    }
    std::string kmer_of(size_t i) const {
        const Site& s = sites[i];
        return s.kmer_len == 255 ? long_text.at(i).second : std::string(s.kmer, s.kmer_len);
    }
};

inline int part_of(uint32_t cid, long long pos) { return (int)(SiteIndex::hash(cid, pos) >> 60) & (kParts - 1); }

struct dsp_freq {
    double prob_cf;
    int nthreads = 1;
    std::vector<std::string> chroms;
    std::unordered_map<std::string, uint32_t> chrom_id;
    uint32_t last_chrom = 0;  // chromosome of the previous record (records of a read share it)
    Partition parts[kParts];
    long long count = 0;
    uint64_t seq = 0;  // records seen so far (sequence numbers of first_use)
    mutable std::string format_cache;          // text of the last dsp_freq_format call ...
    mutable long long format_cache_key = -1;   // ... for this (records seen, sort, bed) ...
    mutable size_t format_cache_sites = 0;     // ... and this many sites

    uint32_t intern(const char* chrom, size_t n) {
        if (!chroms.empty() && chroms[last_chrom].size() == n && memcmp(chroms[last_chrom].data(), chrom, n) == 0)
            return last_chrom;
        std::string key(chrom, n);
        auto it = chrom_id.find(key);
        if (it == chrom_id.end()) {
            it = chrom_id.emplace(key, (uint32_t)chroms.size()).first;
            chroms.push_back(std::move(key));
        }
        return last_chrom = it->second;
    }
    long long used() const {
        long long u = 0;
        for (const Partition& p : parts) u += p.used;
        return u;
    }
    size_t n_sites() const {
        size_t n = 0;
        for (const Partition& p : parts) n += p.sites.size();
        return n;
    }

    // one record, sequential feeders (dsp_freq_add_block)
    void add(const char* chrom, size_t chrom_len, long long pos, const char* strand, size_t strand_len,
             long long pos_in_strand, double p0, double p1, long long label, const char* kmer, size_t kmer_len) {
        ++count;
        const uint64_t my_seq = seq++;
        if (std::fabs(p0 - p1) < prob_cf) return;  // txt_formater.py:23-26
        const uint32_t cid = intern(chrom, chrom_len);
        parts[part_of(cid, pos)].apply(cid, pos, strand, strand_len, pos_in_strand, p0, p1, label, kmer, kmer_len, my_seq);
    }
};

extern "C" {

dsp_freq* dsp_freq_create(double prob_cf) {
    dsp_freq* f = new (std::nothrow) dsp_freq();
    if (f) f->prob_cf = prob_cf;
    return f;
}

void dsp_freq_destroy(dsp_freq* f) { delete f; }

int dsp_parse_double_(const char* p, size_t n, double* out) { return parse_double(p, p + n, out) ? 1 : 0; }  // test hook

void dsp_freq_set_threads(dsp_freq* f, int32_t nthreads) {
    if (f) f->nthreads = nthreads < 1 ? 1 : (nthreads > 64 ? 64 : nthreads);
}

// Lines are parsed by f->nthreads threads (byte ranges moved to line starts), then applied to the table by one
// thread in line order: the double sums are associated exactly like a sequential pass.
int64_t dsp_freq_add_calls_text(dsp_freq* f, const char* text, size_t len, const char* contig) {
    if (!f || (!text && len)) return freq_fail(DSP_EINVAL, "NULL argument");
    const size_t clen = contig ? strlen(contig) : 0;
    const char* const e = text + len;
    int nt = f->nthreads;
    if ((size_t)nt > len / 65536 + 1) nt = (int)(len / 65536 + 1);
    std::vector<const char*> cut((size_t)nt + 1);
    cut[0] = text;
    cut[nt] = e;
    for (int t = 1; t < nt; ++t) {
        const char* p = text + len / nt * t;
        if (p > text && p[-1] != '\n') {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
            p = nl ? nl + 1 : e;
        }
        cut[t] = p < cut[t - 1] ? cut[t - 1] : p;
    }
    std::vector<std::vector<Rec>> recs((size_t)nt);
    std::vector<int> bad_code((size_t)nt, 0);
    std::vector<int64_t> bad_line((size_t)nt, 0);
    auto work = [&](int t) {
        std::vector<Rec>& out = recs[t];
        out.reserve((size_t)(cut[t + 1] - cut[t]) / 56 + 16);
        const char* p = cut[t];
        const char* ce = cut[t + 1];
        while (p < ce) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(ce - p));
            const char* le = nl ? nl : ce;
            Rec r;
            const int rc = parse_call_line(p, le, &r);
            if (rc) { bad_code[t] = rc; bad_line[t] = (int64_t)out.size(); return; }
            out.push_back(r);
            p = nl ? nl + 1 : ce;
        }
    };
    if (!dsp::run_indexed(nt, work)) return freq_fail(DSP_ENOMEM, "out of memory in a worker thread");
    // sequential: how many records count (everything before the first malformed line), chromosome interning
    int64_t nline = 0;
    int bad_t = -1;
    for (int t = 0; t < nt; ++t) {
        for (Rec& r : recs[t]) r.cid = f->intern(r.chrom, r.chrom_len);
        nline += (int64_t)recs[t].size();
        if (bad_code[t]) { bad_t = t; break; }
    }
    const int t_end = bad_t >= 0 ? bad_t + 1 : nt;
    const uint32_t contig_id = contig ? f->intern(contig, clen) : 0;
    const uint64_t seq0 = f->seq;
    // parallel over partitions, every worker scanning all records in order
    const int workers = f->nthreads < kParts ? f->nthreads : kParts;
    auto apply = [&](int wk) {
        uint64_t seq = seq0;
        for (int t = 0; t < t_end; ++t)
            for (const Rec& r : recs[t]) {
                const uint64_t my_seq = seq++;
                if (contig && r.cid != contig_id) continue;
                if (std::fabs(r.p0 - r.p1) < f->prob_cf) continue;
                const int part = part_of(r.cid, r.pos);
                if (part % workers != wk) continue;
                f->parts[part].apply(r.cid, r.pos, r.strand, r.strand_len, r.pis, r.p0, r.p1, r.label, r.kmer, r.kmer_len, my_seq);
            }
    };
    if (!dsp::run_indexed(workers, apply)) return freq_fail(DSP_ENOMEM, "out of memory in a worker thread");
    f->seq += (uint64_t)nline;
    if (contig) {  // `count` counts the records of the requested contig only (call_mods_freq.py:53-56)
        for (int t = 0; t < t_end; ++t)
            for (const Rec& r : recs[t]) f->count += r.cid == contig_id;
    } else {
        f->count += nline;
    }
    if (bad_t >= 0)  // everything before the bad line has been applied, like a sequential pass that stops there
        return freq_fail(DSP_EPARSE, bad_code[bad_t] == 1 ? "malformed call line %lld: need 10 tab-separated columns"
                                                          : "malformed call line %lld: bad number", (long long)nline);
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
        double p0 = 0.0, p1 = 0.0;
        int k = dsp_format_prob_f32_(z0, num);
        parse_double(num, num + k, &p0);
        k = dsp_format_prob_f32_(z1, num);
        parse_double(num, num + k, &p1);
        for (int i = k0; i < k1; ++i) k5[i - k0] = code2base[kmer[r * seq_len + i] & 15];
        f->add(fs[0], (size_t)(fe[0] - fs[0]), pos, fs[2], (size_t)(fe[2] - fs[2]), pis, p0, p1, (long long)labels[r], k5,
               (size_t)(k1 - k0));
    }
    return n;
}

// ---- device-side / sharded aggregation (dsp_freq_dev.hip does the sums; this is its host side) ---------------
// Record encoding shared with the kernels:
//   key    = chrom_id << 40 | pos                      (int64, >= 0; INT64_MAX marks an unused record)
//   meta   = strand code (0 '+', 1 '-') | kmer5 << 2   (5 bases x 4 bits: codes of process_utils.base2code_dna)
//   packed = k0 | k1 << 20 | label==1 << 40 | meta << 41,  k = printed probability in units of 1e-6
static const char* const kCode2Base = "ACGTNWSMKRYBVDHZ";

int64_t dsp_freq_block_keys(dsp_freq* f, const char* text, const uint64_t* row_off, const uint32_t* info_len,
                            const uint8_t* kmer, int32_t seq_len, int64_t n, int64_t* key, int64_t* pis, uint32_t* meta) {
    if (!f || !text || !row_off || !info_len || !kmer || !key || !pis || !meta || seq_len < 1)
        return freq_fail(DSP_EINVAL, "NULL / bad argument");
    const int center = seq_len / 2;
    const int k0 = center - 2 >= 0 ? center - 2 : 0;
    const int k1 = center + 3 <= seq_len ? center + 3 : seq_len;
    if (k1 - k0 != 5) return freq_fail(DSP_EINVAL, "the device aggregator needs a 5-mer (seq_len >= 5)");
    struct Chrom { const char* p; uint32_t len; };
    std::vector<Chrom> chrom((size_t)n);
    int nt = f->nthreads;
    if ((int64_t)nt > n / 4096 + 1) nt = (int)(n / 4096 + 1);
    std::vector<int64_t> bad((size_t)nt, -1);
    std::vector<int> why((size_t)nt, 0);
    auto work = [&](int t) {
        const int64_t a = n * t / nt, b = n * (t + 1) / nt;
        for (int64_t r = a; r < b; ++r) {
            const char* ls = text + row_off[r];
            const char* le = ls + info_len[r];
            const char* fs[4];
            const char* fe[4];
            int nf = 0;
            for (const char* q = ls; nf < 4;) {
                const char* tb = (const char*)memchr(q, '\t', (size_t)(le - q));
                if (!tb) tb = le;
                fs[nf] = q; fe[nf] = tb; ++nf;
                if (tb == le) break;
                q = tb + 1;
            }
            long long pos, ps;
            if (nf < 4 || !parse_ll(fs[1], fe[1], &pos) || !parse_ll(fs[3], fe[3], &ps)) { bad[t] = r; why[t] = 1; return; }
            const size_t sl = (size_t)(fe[2] - fs[2]);
            if (pos < 0 || pos >= (1ll << 40) || sl != 1 || (fs[2][0] != '+' && fs[2][0] != '-')) { bad[t] = r; why[t] = 2; return; }
            uint32_t m = fs[2][0] == '-' ? 1u : 0u;
            for (int i = k0; i < k1; ++i) m |= (uint32_t)(kmer[r * seq_len + i] & 15) << (2 + 4 * (i - k0));
            chrom[r] = Chrom{fs[0], (uint32_t)(fe[0] - fs[0])};
            key[r] = pos;
            pis[r] = ps;
            meta[r] = m;
        }
    };
    if (!dsp::run_indexed(nt, work)) return freq_fail(DSP_ENOMEM, "out of memory in a worker thread");
    for (int t = 0; t < nt; ++t)
        if (bad[t] >= 0)
            return why[t] == 1 ? freq_fail(DSP_EPARSE, "row %lld: bad pos / pos_in_strand column", (long long)bad[t])
                               : freq_fail(DSP_EINVAL, "row %lld: strand / position outside the device aggregator's encoding "
                                                       "('+'/'-', 0 <= pos < 2^40): use the host aggregator", (long long)bad[t]);
    for (int64_t r = 0; r < n; ++r) {  // sequential: chromosome interning (records of a read share it: one memcmp)
        const uint32_t cid = f->intern(chrom[r].p, chrom[r].len);
        if (cid >= (1u << 22)) return freq_fail(DSP_EINVAL, "more than 2^22 chromosomes: use the host aggregator");
        key[r] |= (int64_t)cid << 40;
    }
    return n;
}

int32_t dsp_freq_chrom_count(const dsp_freq* f) { return f ? (int32_t)f->chroms.size() : 0; }

int64_t dsp_freq_chrom_name(const dsp_freq* f, int32_t id, char* out, size_t cap) {
    if (!f || id < 0 || (size_t)id >= f->chroms.size()) return freq_fail(DSP_EINVAL, "chromosome id out of range");
    const std::string& s = f->chroms[(size_t)id];
    if (out && cap) memcpy(out, s.data(), std::min(cap, s.size()));
    return (int64_t)s.size();
}

int32_t dsp_freq_intern_chrom(dsp_freq* f, const char* name, size_t len) {
    if (!f || !name) return freq_fail(DSP_EINVAL, "NULL argument");
    return (int32_t)f->intern(name, len);
}

void dsp_freq_add_counts(dsp_freq* f, int64_t count) {
    if (f) f->count += count;
}

// finished sites (sums taken on the device in record order) into the table that dsp_freq_format prints
int64_t dsp_freq_add_sites(dsp_freq* f, int64_t n, const int64_t* key, const int64_t* first_row, const int64_t* packed_first,
                           const int64_t* pis, const double* sum0, const double* sum1, const int64_t* met, const int64_t* cov) {
    if (!f || (n && (!key || !first_row || !packed_first || !pis || !sum0 || !sum1 || !met || !cov)))
        return freq_fail(DSP_EINVAL, "NULL argument");
    // partitions are independent: worker w inserts the sites of partitions w, w + workers, ... (every worker scans all
    // sites, like the record feeder above)
    const int workers = std::max(1, std::min(f->nthreads, kParts));
    std::vector<long long> bad((size_t)workers, -1);
    std::vector<int> why((size_t)workers, 0);
    const size_t nchrom = f->chroms.size();
    // pass 1 (parallel over sites): the partition of every site; pass 2 (parallel over partitions): the inserts
    std::vector<uint8_t> pid((size_t)n);
    auto hash_range = [&](int wk) {
        for (int64_t i = n * wk / workers; i < n * (wk + 1) / workers; ++i) {
            const uint32_t cid = (uint32_t)((uint64_t)key[i] >> 40);
            if (cid >= nchrom) { bad[wk] = i; why[wk] = 1; return; }
            pid[(size_t)i] = (uint8_t)part_of(cid, (long long)((uint64_t)key[i] & ((1ull << 40) - 1)));
        }
    };
    if (workers == 1 || n < 65536) {
        for (int wk = 0; wk < workers; ++wk) hash_range(wk);
    } else if (!dsp::run_indexed(workers, hash_range)) {
        return freq_fail(DSP_ENOMEM, "out of memory in a worker thread");
    }
    for (int wk = 0; wk < workers; ++wk)
        if (bad[wk] >= 0) return freq_fail(DSP_EINVAL, "site %lld: unknown chromosome id", bad[wk]);
    {   // size every partition once (no vector regrowth / rehash cascade during the inserts)
        size_t cnt[kParts] = {0};
        for (int64_t i = 0; i < n; ++i) ++cnt[pid[(size_t)i]];
        for (int p = 0; p < kParts; ++p) {
            Partition& part = f->parts[p];
            part.sites.reserve(part.sites.size() + cnt[p]);
            size_t cap = part.index.slots.empty() ? 1024 : part.index.slots.size();
            while ((part.index.used + cnt[p] + 1) * 10 > cap * 6) cap *= 2;
            if (cap != part.index.slots.size()) part.index.rehash(cap);
        }
    }
    auto work = [&](int wk) {
        for (int64_t i = 0; i < n; ++i) {
            const int pidx = pid[(size_t)i];
            if (pidx % workers != wk) continue;
            const uint32_t cid = (uint32_t)((uint64_t)key[i] >> 40);
            const long long pos = (long long)((uint64_t)key[i] & ((1ull << 40) - 1));
            const uint32_t meta = (uint32_t)((uint64_t)packed_first[i] >> 41);
            Partition& part = f->parts[pidx];
            bool found;
            SiteIndex::Slot* slot = part.index.find_or_slot(cid, pos, &found);
            if (found) { bad[wk] = i; why[wk] = 2; return; }
            Site s;
            s.chrom = cid; s.pos = pos; s.pos_in_strand = pis[i]; s.first_use = (uint64_t)first_row[i];
            s.strand_len = 1; s.strand[0] = (meta & 1) ? '-' : '+';
            s.kmer_len = 5;
            for (int b = 0; b < 5; ++b) s.kmer[b] = kCode2Base[(meta >> (2 + 4 * b)) & 15];
            s.prob0 = sum0[i]; s.prob1 = sum1[i];
            s.met = met[i]; s.coverage = cov[i]; s.unmet = cov[i] - met[i];
            slot->pos = pos; slot->chrom = cid; slot->idx = (uint32_t)part.sites.size();
            part.sites.push_back(s);
            ++part.index.used;
            part.used += cov[i];
        }
    };
    if (workers == 1 || n < 65536) {
        for (int wk = 0; wk < workers; ++wk) work(wk);
    } else if (!dsp::run_indexed(workers, work)) {
        return freq_fail(DSP_ENOMEM, "out of memory in a worker thread");
    }
    for (int wk = 0; wk < workers; ++wk)
        if (bad[wk] >= 0)
            return why[wk] == 1 ? freq_fail(DSP_EINVAL, "site %lld: unknown chromosome id", bad[wk])
                                : freq_fail(DSP_EINVAL, "site %lld: (chromosome, pos) added twice", bad[wk]);
    return n;
}

void dsp_freq_counts(const dsp_freq* f, int64_t* count, int64_t* used, int64_t* sites) {
    if (!f) return;
    if (count) *count = f->count;
    if (used) *used = f->used();
    if (sites) *sites = (int64_t)f->n_sites();
}

// write_sitekey2stats (call_mods_freq.py:77-122).  Returns bytes needed; writes at most cap bytes.
int64_t dsp_freq_format(const dsp_freq* f, int32_t is_sort, int32_t is_bed, char* out, size_t cap) {
    if (!f) return freq_fail(DSP_EINVAL, "NULL argument");
    // callers ask twice -- once for the size, once with a buffer: the second call copies the cached text
    const long long key = ((long long)f->seq << 2) | (is_sort ? 2 : 0) | (is_bed ? 1 : 0);
    if (f->format_cache_key == key && f->format_cache_sites == f->n_sites()) {
        if (out && cap) memcpy(out, f->format_cache.data(), std::min(cap, f->format_cache.size()));
        return (int64_t)f->format_cache.size();
    }
    // global insertion order: merge the partitions by the sequence number of each site's first record
    struct Ref { uint64_t first_use; uint32_t part; uint32_t idx; };
    std::vector<Ref> order;
    order.reserve(f->n_sites());
    for (int p = 0; p < kParts; ++p)
        for (size_t i = 0; i < f->parts[p].sites.size(); ++i)
            order.push_back(Ref{f->parts[p].sites[i].first_use, (uint32_t)p, (uint32_t)i});
    std::sort(order.begin(), order.end(), [](const Ref& a, const Ref& b) { return a.first_use < b.first_use; });
    if (is_sort)  // sorted(keys, key=split_key): (chrom str, pos int); stable, like Python's sorted
        std::stable_sort(order.begin(), order.end(), [&](const Ref& a, const Ref& b) {
            const Site& x = f->parts[a.part].sites[a.idx];
            const Site& y = f->parts[b.part].sites[b.idx];
            const int c = x.chrom == y.chrom ? 0 : f->chroms[x.chrom].compare(f->chroms[y.chrom]);
            return c != 0 ? c < 0 : x.pos < y.pos;
        });
    // the lines are independent: f->nthreads threads format contiguous chunks of `order`, concatenated in order
    auto format_range = [&](size_t lo, size_t hi, std::string& s) {
        s.reserve((hi - lo) * 72);
        char num[64];
        auto put_ll = [&](long long v) {  // "%lld"
            char* e = num + sizeof(num);
            char* q = e;
            unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
            do { *--q = (char)('0' + u % 10); u /= 10; } while (u);
            if (v < 0) *--q = '-';
            s.append(q, (size_t)(e - q));
        };
        auto put_f = [&](const char* fmt, double v) {  // the three "%.3f" / "%.4f" fields: glibc's exact decimal rounding
            const int k = snprintf(num, sizeof(num), fmt, v);
            if (k >= 0 && (size_t)k < sizeof(num)) { s.append(num, (size_t)k); return; }
            std::string big((size_t)(k > 0 ? k : 0) + 1, '\0');  // sums of malformed inputs can have hundreds of digits
            const int k2 = snprintf(&big[0], big.size(), fmt, v);
            if (k2 > 0) s.append(big.data(), (size_t)k2);
        };
        for (size_t oi = lo; oi < hi; ++oi) {
            const Ref& ref = order[oi];
            const Partition& part = f->parts[ref.part];
            const Site& t = part.sites[ref.idx];
            if (t.coverage <= 0) continue;
            const bool spilled = t.strand_len == 255;
            const std::string* lt = spilled ? &part.long_text.at(ref.idx).first : nullptr;
            const std::string* lk = spilled ? &part.long_text.at(ref.idx).second : nullptr;
            const char* strand = spilled ? lt->data() : t.strand;
            const size_t strand_len = spilled ? lt->size() : t.strand_len;
            const char* kmer = spilled ? lk->data() : t.kmer;
            const size_t kmer_len = spilled ? lk->size() : t.kmer_len;
            const std::string& chrom = f->chroms[t.chrom];
            const double rmet = (double)t.met / (double)t.coverage;
            s.append(chrom);
            s.push_back('\t');
            if (is_bed) {  // chrom pos pos+1 . cov strand pos pos+1 0,0,0 cov pct
                const long long pct = (long long)std::nearbyint(rmet * 100 + 0.001);  // int(round(rmet*100+0.001, 0)), :110
                put_ll(t.pos); s.push_back('\t'); put_ll(t.pos + 1); s.append("\t.\t", 3); put_ll(t.coverage); s.push_back('\t');
                s.append(strand, strand_len); s.push_back('\t'); put_ll(t.pos); s.push_back('\t'); put_ll(t.pos + 1);
                s.append("\t0,0,0\t", 7); put_ll(t.coverage); s.push_back('\t'); put_ll(pct); s.push_back('\n');
            } else {       // chrom pos strand pos_in_strand prob0 prob1 met unmet cov rmet kmer
                put_ll(t.pos); s.push_back('\t'); s.append(strand, strand_len); s.push_back('\t'); put_ll(t.pos_in_strand); s.push_back('\t');
                put_f("%.3f", t.prob0); s.push_back('\t'); put_f("%.3f", t.prob1); s.push_back('\t');
                put_ll(t.met); s.push_back('\t'); put_ll(t.unmet); s.push_back('\t'); put_ll(t.coverage); s.push_back('\t');
                put_f("%.4f", rmet); s.push_back('\t'); s.append(kmer, kmer_len); s.push_back('\n');
            }
        }
    };
    int nt = f->nthreads < 1 ? 1 : f->nthreads;
    if ((size_t)nt > order.size() / 4096 + 1) nt = (int)(order.size() / 4096 + 1);
    std::vector<std::string> chunks((size_t)nt);
    if (!dsp::run_indexed(nt, [&](int t) { format_range(order.size() * t / nt, order.size() * (t + 1) / nt, chunks[t]); }))
        return freq_fail(DSP_ENOMEM, "out of memory in a worker thread");
    size_t total = 0;
    for (const std::string& c : chunks) total += c.size();
    std::string& s = f->format_cache;
    s.clear();
    s.reserve(total);
    for (const std::string& c : chunks) s.append(c);
    f->format_cache_key = key;
    f->format_cache_sites = f->n_sites();
    if (out && cap) memcpy(out, s.data(), std::min(cap, s.size()));
    return (int64_t)s.size();
}

}  // extern "C"
