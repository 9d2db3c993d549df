// threads_refused.cpp -- the host thread pools when the system refuses threads (EAGAIN from pthread_create: RLIMIT_NPROC, a
// pids cgroup).  pthread_create is interposed here: after `g_allow` successful calls every further one fails.  Nothing may
// abort (an exception leaving a std::thread constructor inside a pool = std::terminate): every entry point gives the
// result it gives with all its threads, or an error code.  Test infrastructure only (tests/test_host_sanitizers.py).
#include <dlfcn.h>
#include <errno.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <atomic>
#include <string>
#include <vector>

#include "dsp_amd.h"
#include "dsp_threads.h"

static std::atomic<long> g_allow{1L << 40};
static std::atomic<long> g_refused{0};

extern "C" int pthread_create(pthread_t* t, const pthread_attr_t* a, void* (*fn)(void*), void* arg) {
    typedef int (*real_t)(pthread_t*, const pthread_attr_t*, void* (*)(void*), void*);
    static real_t real = (real_t)dlsym(RTLD_NEXT, "pthread_create");
    if (g_allow.fetch_sub(1) <= 0) {
        g_refused.fetch_add(1);
        return EAGAIN;
    }
    return real(t, a, fn, arg);
}

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "threads_refused: CHECK failed at line %d: %s (last error: %s)\n", __LINE__, #c, dsp_last_error()); return 1; } } while (0)

// the library's error slot lives in dsp_capi.cpp (HIP side): the host sources are linked alone here
static thread_local std::string g_err;
extern "C" void dsp_set_error_(const char* msg) { g_err = msg ? msg : ""; }
extern "C" const char* dsp_last_error(void) { return g_err.c_str(); }

int main(int argc, char** argv) {
    const std::string tmp = argc > 1 ? argv[1] : "/tmp";
    // ---- run_indexed: every index exactly once, whatever number of threads the system grants
    for (long allow : {0L, 1L, 3L, 100L}) {
        for (int nt : {1, 2, 5, 16}) {
            std::vector<std::atomic<int>> hit(nt);
            for (auto& h : hit) h.store(0);
            g_allow.store(allow);
            const bool ok = dsp::run_indexed(nt, [&](int t) { hit[(size_t)t].fetch_add(1); });
            g_allow.store(1L << 40);
            CHECK(ok);
            for (int t = 0; t < nt; ++t) CHECK(hit[(size_t)t].load() == 1);
        }
    }
    {   // a worker that throws: reported, every other index still ran, nothing escaped
        std::atomic<int> ran{0};
        g_allow.store(2);
        const bool ok = dsp::run_indexed(6, [&](int t) { ran.fetch_add(1); if (t == 1 || t == 4) throw std::bad_alloc(); });
        g_allow.store(1L << 40);
        CHECK(!ok && ran.load() == 6);
    }
    CHECK(g_refused.load() > 0);
    // ---- BGZF: compress on 6 threads / inflate on 6 with 0, 1, 2 threads granted = the bytes of the unrestricted calls
    std::string text;
    uint64_t x = 88172645463325252ull;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (int i = 0; i < 60000; i++) text += "chr2\t" + std::to_string(rnd() % 1000000) + "\t-\t0." + std::to_string(rnd() % 1000000) + "\n";
    std::vector<uint8_t> ref(text.size() + text.size() / 2 + 4096);
    const int64_t cb = dsp_bgzf_compress((const uint8_t*)text.data(), text.size(), ref.data(), ref.size(), 1, 6);
    CHECK(cb > 0);
    for (long allow : {0L, 1L, 2L}) {
        std::vector<uint8_t> comp(ref.size());
        g_allow.store(allow);
        const int64_t c2 = dsp_bgzf_compress((const uint8_t*)text.data(), text.size(), comp.data(), comp.size(), 1, 6);
        g_allow.store(1L << 40);
        CHECK(c2 == cb && memcmp(comp.data(), ref.data(), (size_t)cb) == 0);
        std::vector<uint64_t> off(4096);
        std::vector<uint32_t> isz(4096);
        const int64_t nm = dsp_gz_index(comp.data(), (size_t)cb, 4095, off.data(), isz.data());
        CHECK(nm > 2);
        std::vector<uint8_t> back(text.size() + 16);
        g_allow.store(allow);
        const int64_t got = dsp_gz_inflate_members(comp.data(), off.data(), isz.data(), 0, nm, back.data(), back.size(), 6);
        g_allow.store(1L << 40);
        CHECK(got == (int64_t)text.size() && memcmp(back.data(), text.data(), text.size()) == 0);
    }
    // ---- the parallel inflater of a foreign .gz: no decoder thread = an error from open, not an abort; a decoder thread but no
    // finisher / worker threads = the whole text all the same
    {
        std::vector<uint8_t> o(compressBound((uLong)text.size()) + 64);
        z_stream z;
        memset(&z, 0, sizeof(z));
        deflateInit2(&z, 6, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY);
        z.next_in = (Bytef*)text.data(); z.avail_in = (uInt)text.size();
        z.next_out = o.data(); z.avail_out = (uInt)o.size();
        deflate(&z, Z_FINISH);
        o.resize(z.total_out);
        deflateEnd(&z);
        const std::string p = tmp + "/refused.gz";
        FILE* f = fopen(p.c_str(), "wb");
        CHECK(f != nullptr);
        fwrite(o.data(), 1, o.size(), f);
        fclose(f);
        g_allow.store(0);
        dsp_pgz* none = dsp_pgz_open(p.c_str(), 4, 65536);
        g_allow.store(1L << 40);
        CHECK(none == nullptr && strstr(dsp_last_error(), "decoder thread") != nullptr);
        for (long allow : {1L, 2L, 4L, 1L << 40}) {
            g_allow.store(allow);
            dsp_pgz* zz = dsp_pgz_open(p.c_str(), 5, 65536);
            CHECK(zz != nullptr);
            std::vector<uint8_t> buf(70001);
            std::string got;
            int64_t rc;
            while ((rc = dsp_pgz_read(zz, buf.data(), buf.size())) > 0) got.append((const char*)buf.data(), (size_t)rc);
            dsp_pgz_close(zz);
            g_allow.store(1L << 40);
            CHECK(rc == 0 && got == text);
        }
    }
    // ---- the row parser and the call formatter on 8 threads with 0 / 2 granted
    {
        const int L = 3, S = 2;
        std::string rows;
        for (int i = 0; i < 500; i++)
            rows += "chr1\t" + std::to_string(100 + i) + "\t+\t" + std::to_string(i) + "\tread" + std::to_string(i % 7) + "\tt\tACG\t0.1,0.2,-0.3\t1.0,2.0,3.5\t4,5,6\t0.5,0.25;1e-3,2;3,4.5\t" + std::to_string(i & 1) + "\n";
        struct Out {
            std::vector<uint8_t> kmer; std::vector<float> means, stds, signals; std::vector<int32_t> lens, labels;
            std::vector<uint64_t> row_off; std::vector<uint32_t> info_len, read_off, read_len;
            explicit Out(int n) : kmer(n * L), means(n * L), stds(n * L), signals(n * L * S), lens(n * L), labels(n), row_off(n + 1), info_len(n), read_off(n), read_len(n) {}
        };
        Out a(600), b(600);
        const int64_t na = dsp_parse_feature_rows(rows.data(), rows.size(), L, S, 600, a.kmer.data(), a.means.data(), a.stds.data(), a.lens.data(),
                                                  a.signals.data(), a.labels.data(), a.row_off.data(), a.info_len.data(), a.read_off.data(), a.read_len.data(), 8);
        CHECK(na == 500);
        for (long allow : {0L, 2L}) {
            g_allow.store(allow);
            const int64_t nb = dsp_parse_feature_rows(rows.data(), rows.size(), L, S, 600, b.kmer.data(), b.means.data(), b.stds.data(), b.lens.data(),
                                                      b.signals.data(), b.labels.data(), b.row_off.data(), b.info_len.data(), b.read_off.data(), b.read_len.data(), 8);
            g_allow.store(1L << 40);
            CHECK(nb == 500 && a.kmer == b.kmer && a.lens == b.lens && a.labels == b.labels && a.info_len == b.info_len);
            CHECK(memcmp(a.means.data(), b.means.data(), 500 * L * 4) == 0 && memcmp(a.signals.data(), b.signals.data(), 500 * L * S * 4) == 0);
        }
    }
    printf("threads_refused: ok (%ld thread creations refused)\n", g_refused.load());
    return 0;
}
