// dsp_fast5.cpp -- what the reference pulls out of one tombo-resquiggled single-read fast5 before any arithmetic, read
// with the HDF5 C library itself (the library h5py wraps), loaded at run time:
//   Raw/Reads/<first read>/Signal + its read_id          extract_features.py:57-65 (_get_label_raw), :135-148
//   Analyses/<corrected group>/<subgroup>/Events         :68-89  (columns start, length, base; attribute read_start_rel_to_raw)
//   Analyses/<corrected group>/<subgroup>/Alignment      :94-131, :151-176 (mapped_chrom, mapped_strand, mapped_start)
//   UniqueGlobalKey/channel_id                           :255-270 (digitisation, range, offset)
// libhdf5 is found with dlopen (DSP_HDF5_LIB, the usual sonames, then the conda / system library directories); no HDF5
// header is needed to build.  Only plain C entry points that are stable from HDF5 1.10 to 1.14 are used; 1.8 (32-bit
// hid_t) is refused.  The library is not assumed to be thread-safe: one mutex serialises every call -- but only the
// metadata goes through it: where HDF5 >= 1.10.5 can tell where a dataset's chunks sit in the file (H5Dget_chunk_info_by_coord)
// and the filter pipeline is [shuffle,] deflate or empty, the Signal and Events chunks are read with pread, inflated
// (libdeflate if present, else zlib) and un-shuffled OUTSIDE the lock, so N loader threads decode N files at once
// (0.6 ms under the lock + 1-2 ms outside per 100 k-sample read, against 4.7 ms through H5Dread).
// VBZ-compressed signals need ONT's HDF5 filter plugin on HDF5_PLUGIN_PATH, like h5py does.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <exception>
#include <mutex>
#include <string>
#include <vector>

#include "dsp_amd.h"

extern "C" void dsp_set_error_(const char* msg);

namespace {

typedef int64_t hid_t;
typedef int herr_t;
typedef int htri_t;
typedef unsigned long long hsize_t;
typedef long long hssize_t;

struct H5 {
    void* so = nullptr;
    std::string path, why;
    herr_t (*open)() = nullptr;
    herr_t (*get_libversion)(unsigned*, unsigned*, unsigned*) = nullptr;
    herr_t (*Eset_auto2)(hid_t, void*, void*) = nullptr;
    hid_t (*Fopen)(const char*, unsigned, hid_t) = nullptr;
    herr_t (*Fclose)(hid_t) = nullptr;
    htri_t (*Lexists)(hid_t, const char*, hid_t) = nullptr;
    hid_t (*Gopen2)(hid_t, const char*, hid_t) = nullptr;
    herr_t (*Gclose)(hid_t) = nullptr;
    herr_t (*Gget_info)(hid_t, void*) = nullptr;
    long (*Lget_name_by_idx)(hid_t, const char*, int, int, hsize_t, char*, size_t, hid_t) = nullptr;
    hid_t (*Dopen2)(hid_t, const char*, hid_t) = nullptr;
    herr_t (*Dclose)(hid_t) = nullptr;
    hid_t (*Dget_space)(hid_t) = nullptr;
    hid_t (*Dget_type)(hid_t) = nullptr;
    herr_t (*Dread)(hid_t, hid_t, hid_t, hid_t, hid_t, void*) = nullptr;
    hssize_t (*Sget_simple_extent_npoints)(hid_t) = nullptr;
    int (*Sget_simple_extent_ndims)(hid_t) = nullptr;
    herr_t (*Sclose)(hid_t) = nullptr;
    int (*Tget_class)(hid_t) = nullptr;
    size_t (*Tget_size)(hid_t) = nullptr;
    int (*Tget_member_index)(hid_t, const char*) = nullptr;
    hid_t (*Tcreate)(int, size_t) = nullptr;
    herr_t (*Tinsert)(hid_t, const char*, size_t, hid_t) = nullptr;
    hid_t (*Tcopy)(hid_t) = nullptr;
    herr_t (*Tset_size)(hid_t, size_t) = nullptr;
    htri_t (*Tis_variable_str)(hid_t) = nullptr;
    int (*Tget_cset)(hid_t) = nullptr;
    herr_t (*Tset_cset)(hid_t, int) = nullptr;
    herr_t (*Tclose)(hid_t) = nullptr;
    htri_t (*Aexists)(hid_t, const char*) = nullptr;
    hid_t (*Aopen)(hid_t, const char*, hid_t) = nullptr;
    herr_t (*Aclose)(hid_t) = nullptr;
    hid_t (*Aget_type)(hid_t) = nullptr;
    herr_t (*Aread)(hid_t, hid_t, void*) = nullptr;
    herr_t (*free_memory)(void*) = nullptr;
    // optional (HDF5 >= 1.10.5): where the chunks of a dataset sit in the file, so that they can be read and inflated
    // OUTSIDE the library lock (see Direct)
    hid_t (*Dget_create_plist)(hid_t) = nullptr;
    int (*Pget_layout)(hid_t) = nullptr;
    int (*Pget_chunk)(hid_t, int, hsize_t*) = nullptr;
    int (*Pget_nfilters)(hid_t) = nullptr;
    int (*Pget_filter2)(hid_t, unsigned, unsigned*, size_t*, unsigned*, size_t, char*, unsigned*) = nullptr;
    herr_t (*Pclose)(hid_t) = nullptr;
    herr_t (*Dget_chunk_info_by_coord)(hid_t, const hsize_t*, unsigned*, uint64_t*, hsize_t*) = nullptr;
    uint64_t (*Dget_offset)(hid_t) = nullptr;
    hid_t (*Fget_create_plist)(hid_t) = nullptr;
    herr_t (*Pget_userblock)(hid_t, hsize_t*) = nullptr;
    size_t (*Tget_member_offset)(hid_t, unsigned) = nullptr;
    hid_t (*Tget_member_type)(hid_t, unsigned) = nullptr;
    int (*Tget_sign)(hid_t) = nullptr;
    int (*Tget_order)(hid_t) = nullptr;
    bool direct = false;
    hid_t t_short = -1, t_int64 = -1, t_double = -1, t_c_s1 = -1;
};

std::mutex g_mu;  // guards every HDF5 call (and the one-time load)
H5 g_h5;
bool g_tried = false;

int f5_fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    dsp_set_error_(buf);
    return code;
}

template <class F>
bool sym(void* so, const char* name, F& f) {
    f = reinterpret_cast<F>(dlsym(so, name));
    return f != nullptr;
}

bool load_locked() {
    if (g_tried) return g_h5.so != nullptr;
    g_tried = true;
    std::vector<std::string> cand;
    if (const char* e = getenv("DSP_HDF5_LIB")) cand.push_back(e);
    const bool search = getenv("DSP_HDF5_NO_SEARCH") == nullptr;  // (tests: only DSP_HDF5_LIB)
    const char* names[] = {"libhdf5.so", "libhdf5_serial.so", "libhdf5.so.310", "libhdf5.so.200", "libhdf5.so.103", "libhdf5_serial.so.103",
                           "libhdf5.so.101", "libhdf5_serial.so.100"};
    if (search)
        for (const char* n : names) cand.push_back(n);
    const char* dirs[] = {"/opt/conda/lib", "/usr/lib/x86_64-linux-gnu/hdf5/serial", "/usr/lib/x86_64-linux-gnu", "/usr/local/lib"};
    if (search) {
        if (const char* cp = getenv("CONDA_PREFIX"))
            for (const char* n : names) cand.push_back(std::string(cp) + "/lib/" + n);
        for (const char* d : dirs)
            for (const char* n : names) cand.push_back(std::string(d) + "/" + n);
    }
    std::string tried;
    for (const std::string& c : cand) {
        void* so = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!so) continue;
        H5 h;
        h.so = so; h.path = c;
        bool ok = sym(so, "H5open", h.open) && sym(so, "H5get_libversion", h.get_libversion) && sym(so, "H5Eset_auto2", h.Eset_auto2) &&
                  sym(so, "H5Fopen", h.Fopen) && sym(so, "H5Fclose", h.Fclose) && sym(so, "H5Lexists", h.Lexists) &&
                  sym(so, "H5Gopen2", h.Gopen2) && sym(so, "H5Gclose", h.Gclose) && sym(so, "H5Gget_info", h.Gget_info) &&
                  sym(so, "H5Lget_name_by_idx", h.Lget_name_by_idx) && sym(so, "H5Dopen2", h.Dopen2) && sym(so, "H5Dclose", h.Dclose) &&
                  sym(so, "H5Dget_space", h.Dget_space) && sym(so, "H5Dget_type", h.Dget_type) && sym(so, "H5Dread", h.Dread) &&
                  sym(so, "H5Sget_simple_extent_npoints", h.Sget_simple_extent_npoints) &&
                  sym(so, "H5Sget_simple_extent_ndims", h.Sget_simple_extent_ndims) && sym(so, "H5Sclose", h.Sclose) &&
                  sym(so, "H5Tget_class", h.Tget_class) && sym(so, "H5Tget_size", h.Tget_size) &&
                  sym(so, "H5Tget_member_index", h.Tget_member_index) && sym(so, "H5Tcreate", h.Tcreate) && sym(so, "H5Tinsert", h.Tinsert) &&
                  sym(so, "H5Tcopy", h.Tcopy) && sym(so, "H5Tset_size", h.Tset_size) && sym(so, "H5Tis_variable_str", h.Tis_variable_str) &&
                  sym(so, "H5Tget_cset", h.Tget_cset) && sym(so, "H5Tset_cset", h.Tset_cset) && sym(so, "H5Tclose", h.Tclose) &&
                  sym(so, "H5Aexists", h.Aexists) && sym(so, "H5Aopen", h.Aopen) && sym(so, "H5Aclose", h.Aclose) &&
                  sym(so, "H5Aget_type", h.Aget_type) && sym(so, "H5Aread", h.Aread) && sym(so, "H5free_memory", h.free_memory);
        unsigned maj = 0, min = 0, rel = 0;
        if (ok && (h.open() < 0 || h.get_libversion(&maj, &min, &rel) < 0)) ok = false;
        if (ok && (maj != 1 || min < 10)) {  // 1.8: hid_t is a 32-bit int, different calling convention for every entry point
            tried += c + " (HDF5 " + std::to_string(maj) + "." + std::to_string(min) + ": 1.10 or later is needed); ";
            ok = false;
        }
        hid_t* g;
        if (ok) {
            ok = false;
            do {
                if (!(g = (hid_t*)dlsym(so, "H5T_NATIVE_SHORT_g"))) break;
                h.t_short = *g;
                if (!(g = (hid_t*)dlsym(so, "H5T_NATIVE_INT64_g"))) break;
                h.t_int64 = *g;
                if (!(g = (hid_t*)dlsym(so, "H5T_NATIVE_DOUBLE_g"))) break;
                h.t_double = *g;
                if (!(g = (hid_t*)dlsym(so, "H5T_C_S1_g"))) break;
                h.t_c_s1 = *g;
                ok = h.t_short > 0 && h.t_int64 > 0 && h.t_double > 0 && h.t_c_s1 > 0;
            } while (0);
        }
        if (!ok) {
            if (tried.find(c) == std::string::npos) tried += c + " (missing entry points); ";
            dlclose(so);
            continue;
        }
        h.direct = sym(so, "H5Dget_create_plist", h.Dget_create_plist) && sym(so, "H5Pget_layout", h.Pget_layout) &&
                   sym(so, "H5Pget_chunk", h.Pget_chunk) && sym(so, "H5Pget_nfilters", h.Pget_nfilters) &&
                   sym(so, "H5Pget_filter2", h.Pget_filter2) && sym(so, "H5Pclose", h.Pclose) &&
                   sym(so, "H5Dget_chunk_info_by_coord", h.Dget_chunk_info_by_coord) &&
                   sym(so, "H5Dget_offset", h.Dget_offset) && sym(so, "H5Fget_create_plist", h.Fget_create_plist) &&
                   sym(so, "H5Pget_userblock", h.Pget_userblock) && sym(so, "H5Tget_member_offset", h.Tget_member_offset) &&
                   sym(so, "H5Tget_member_type", h.Tget_member_type) && sym(so, "H5Tget_sign", h.Tget_sign) &&
                   sym(so, "H5Tget_order", h.Tget_order) && getenv("DSP_FAST5_NO_DIRECT") == nullptr;
        h.Eset_auto2(0, nullptr, nullptr);  // no HDF5 error stacks on stderr: failures come back as return codes
        g_h5 = h;
        return true;
    }
    g_h5.why = tried.empty() ? "no libhdf5 found (set DSP_HDF5_LIB to the library file)" : tried;
    return false;
}

struct Closer {  // closes what a load opened, in reverse order
    std::vector<std::pair<int, hid_t>> ids;
    ~Closer() {
        for (size_t i = ids.size(); i-- > 0;) {
            const hid_t id = ids[i].second;
            switch (ids[i].first) {
                case 0: g_h5.Fclose(id); break;
                case 1: g_h5.Gclose(id); break;
                case 2: g_h5.Dclose(id); break;
                case 3: g_h5.Sclose(id); break;
                case 4: g_h5.Tclose(id); break;
                case 6: g_h5.Pclose(id); break;
                default: g_h5.Aclose(id); break;
            }
        }
    }
    hid_t add(int kind, hid_t id) { if (id >= 0) ids.push_back({kind, id}); return id; }
};

// "a/b/c" exists below loc (H5Lexists fails, not "false", when an intermediate link is missing)
bool path_exists(hid_t loc, const std::string& path) {
    size_t pos = 0;
    while (pos < path.size()) {
        size_t nxt = path.find('/', pos);
        if (nxt == std::string::npos) nxt = path.size();
        if (nxt > pos && g_h5.Lexists(loc, path.substr(0, nxt).c_str(), 0) <= 0) return false;
        pos = nxt + 1;
    }
    return true;
}

bool attr_number(hid_t obj, const char* name, bool as_int, double* d, int64_t* i, Closer& c) {
    if (g_h5.Aexists(obj, name) <= 0) return false;
    const hid_t a = c.add(5, g_h5.Aopen(obj, name, 0));
    if (a < 0) return false;
    const hid_t ft = c.add(4, g_h5.Aget_type(a));
    const int cls = ft >= 0 ? g_h5.Tget_class(ft) : -1;
    if (cls != 0 && cls != 1) return false;  // integer or float
    if (as_int && cls == 0) return g_h5.Aread(a, g_h5.t_int64, i) >= 0;
    double v = 0;
    if (g_h5.Aread(a, g_h5.t_double, &v) < 0) return false;
    if (as_int) *i = (int64_t)v; else *d = v;
    return true;
}

// fixed-length (bytes / numpy.string_) or variable-length (str) string attribute -> NUL-terminated text
bool attr_string(hid_t obj, const char* name, char* out, size_t cap, Closer& c) {
    out[0] = 0;
    if (g_h5.Aexists(obj, name) <= 0) return false;
    const hid_t a = c.add(5, g_h5.Aopen(obj, name, 0));
    if (a < 0) return false;
    const hid_t ft = c.add(4, g_h5.Aget_type(a));
    if (ft < 0 || g_h5.Tget_class(ft) != 3) return false;  // H5T_STRING
    if (g_h5.Tis_variable_str(ft) > 0) {
        const hid_t mt = c.add(4, g_h5.Tcopy(g_h5.t_c_s1));
        if (mt < 0 || g_h5.Tset_size(mt, (size_t)-1) < 0) return false;  // H5T_VARIABLE
        g_h5.Tset_cset(mt, g_h5.Tget_cset(ft));
        char* p = nullptr;
        if (g_h5.Aread(a, mt, &p) < 0 || !p) return false;
        snprintf(out, cap, "%s", p);
        g_h5.free_memory(p);
        return true;
    }
    const size_t n = g_h5.Tget_size(ft);
    std::vector<char> buf(n + 1, 0);
    if (g_h5.Aread(a, ft, buf.data()) < 0) return false;
    snprintf(out, cap, "%s", buf.data());  // stops at the first NUL (null-padded and null-terminated both end there)
    return true;
}

// Where the bytes of a one-dimensional dataset sit in the file.  libhdf5 decodes one dataset at a time per process; with
// this plan (made under the library lock, cheap) the chunks are read with pread, inflated with zlib and un-shuffled
// OUTSIDE the lock, so that N loader threads decode N files at once.  Anything unusual (other filters, a userblock,
// unallocated chunks, compact layout) leaves ok = false and the dataset goes through H5Dread under the lock.
struct Direct {
    bool ok = false;
    size_t elem = 0;        // bytes per element in the file
    hsize_t n = 0, chunk = 0;
    int f_shuffle = -1, f_deflate = -1;  // position of the filter in the pipeline (bit of the per-chunk skip mask), -1 = absent
    struct Ck { uint64_t addr, size, elem0; unsigned mask; };
    std::vector<Ck> cks;    // contiguous layout: one entry, no filters
};

void plan_direct(hid_t file, hid_t ds, size_t elem, hsize_t n, Direct& d, Closer& c) {
    d.ok = false; d.elem = elem; d.n = n;
    if (!g_h5.direct || n == 0) return;
    const hid_t fcpl = c.add(6, g_h5.Fget_create_plist(file));
    hsize_t ub = 1;
    if (fcpl < 0 || g_h5.Pget_userblock(fcpl, &ub) < 0 || ub != 0) return;
    const hid_t dcpl = c.add(6, g_h5.Dget_create_plist(ds));
    if (dcpl < 0) return;
    const int layout = g_h5.Pget_layout(dcpl);
    const int nf = g_h5.Pget_nfilters(dcpl);
    if (nf < 0 || nf > 2) return;
    for (int i = 0; i < nf; ++i) {
        unsigned flags = 0, cd[8], cfg = 0;
        size_t ncd = 8;
        char name[8];
        const int id = g_h5.Pget_filter2(dcpl, (unsigned)i, &flags, &ncd, cd, sizeof(name), name, &cfg);
        if (id == 1 && d.f_deflate < 0) d.f_deflate = i;       // H5Z_FILTER_DEFLATE
        else if (id == 2 && d.f_shuffle < 0) d.f_shuffle = i;  // H5Z_FILTER_SHUFFLE
        else return;
    }
    if (d.f_shuffle >= 0 && d.f_deflate >= 0 && d.f_shuffle > d.f_deflate) return;  // written as shuffle, then deflate
    if (layout == 1) {  // H5D_CONTIGUOUS
        const uint64_t addr = g_h5.Dget_offset(ds);
        if (nf != 0 || addr == (uint64_t)-1) return;
        d.chunk = n;
        d.cks.push_back({addr, (uint64_t)n * elem, 0, 0u});
        d.ok = true;
        return;
    }
    if (layout != 2) return;  // H5D_CHUNKED
    hsize_t chunk = 0;
    if (g_h5.Pget_chunk(dcpl, 1, &chunk) != 1 || chunk == 0) return;
    const hsize_t nchunks = (n + chunk - 1) / chunk;
    if (nchunks > (hsize_t)(1u << 24)) return;         // absurd metadata: left to the library (which will refuse it)
    // (an unallocated chunk -- read as the fill value -- has no address: left to the library)
    d.chunk = chunk;
    d.cks.resize((size_t)nchunks);
    for (hsize_t k = 0; k < nchunks; ++k) {  // by coordinate: one index lookup each (by index it is a walk from the first chunk)
        const hsize_t coord = k * chunk;
        hsize_t size = 0;
        unsigned mask = 0;
        uint64_t addr = 0;
        if (g_h5.Dget_chunk_info_by_coord(ds, &coord, &mask, &addr, &size) < 0 || addr == (uint64_t)-1 || size == 0) return;
        d.cks[(size_t)k] = {addr, (uint64_t)size, (uint64_t)coord, mask};
    }
    d.ok = true;
}

// zlib-format inflate of one chunk: libdeflate when the shared library is on the box (2-3x zlib's speed; bound at run
// time like in dsp_gz.cpp, one decompressor per thread), else zlib
struct Deflater {
    void* (*alloc_d)() = nullptr;
    void (*free_d)(void*) = nullptr;
    int (*zlib_decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    bool ok = false;
    Deflater() {
        if (getenv("DSP_GZ_ZLIB")) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc_d = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
        free_d = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
        zlib_decompress = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_zlib_decompress");
        ok = alloc_d && free_d && zlib_decompress;
    }
};
bool zlib_inflate(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, size_t* out_n) {
    static const Deflater ld;
    if (ld.ok) {
        struct PerThread {  // one decompressor per loader thread, freed when the thread ends
            void* d;
            PerThread() : d(ld.alloc_d()) {}
            ~PerThread() { if (d) ld.free_d(d); }
        };
        thread_local PerThread pt;
        if (pt.d) return ld.zlib_decompress(pt.d, src, n, dst, cap, out_n) == 0;
    }
    uLongf len = (uLongf)cap;
    if (uncompress(dst, &len, src, (uLong)n) != Z_OK) return false;
    *out_n = (size_t)len;
    return true;
}

// read + decode the planned chunks into dst (n * elem bytes); false on any I/O or zlib error
// Sizes come straight from the file's metadata: a damaged or hostile file must fail the READ (the reference counts the
// file as failed and carries on, extract_features.py:373-375), not the process -- every size is checked against the file
// itself before anything is allocated (HDF5 caps a chunk at 4 GiB; a stored chunk cannot be larger than the file).
bool read_direct(int fd, const Direct& d, uint8_t* dst) {
    std::vector<uint8_t> comp, plain;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || d.elem == 0 || d.chunk == 0 || d.chunk > (hsize_t)(1ull << 32) / d.elem) return false;
    const uint64_t fsize = (uint64_t)sb.st_size;
    const size_t cbytes = (size_t)d.chunk * d.elem;
    for (const Direct::Ck& k : d.cks) {
        if (k.elem0 >= d.n || k.size > fsize || k.addr > fsize - k.size) return false;
        const size_t want = (size_t)std::min<hsize_t>(d.chunk, d.n - k.elem0) * d.elem;
        comp.resize((size_t)k.size);
        size_t got = 0;
        while (got < comp.size()) {
            const ssize_t r = pread(fd, comp.data() + got, comp.size() - got, (off_t)(k.addr + got));
            if (r <= 0) return false;
            got += (size_t)r;
        }
        const bool defl = d.f_deflate >= 0 && !((k.mask >> d.f_deflate) & 1u);
        const bool shuf = d.f_shuffle >= 0 && !((k.mask >> d.f_shuffle) & 1u);
        const uint8_t* src = comp.data();
        size_t have = comp.size();
        if (defl) {
            plain.resize(cbytes);
            size_t out_len = 0;
            if (!zlib_inflate(comp.data(), comp.size(), plain.data(), cbytes, &out_len)) return false;
            src = plain.data();
            have = out_len;
        }
        if (have < want) return false;  // (a full chunk is stored even at the ragged end)
        uint8_t* o = dst + (size_t)k.elem0 * d.elem;
        if (shuf && d.elem > 1) {  // byte j of element i sits at src[j * count + i], count = elements in the stored chunk
            const size_t count = have / d.elem, m = want / d.elem;
            for (size_t j = 0; j < d.elem; ++j) {
                const uint8_t* col = src + j * count;
                for (size_t i = 0; i < m; ++i) o[i * d.elem + j] = col[i];
            }
        } else {
            memcpy(o, src, want);
        }
    }
    return true;
}

// an integer member of a compound record, little-endian, 1..8 bytes
struct IntField { size_t off = 0, size = 0; bool is_signed = false; bool ok = false; };
IntField int_field(hid_t ft, const char* name, Closer& c) {
    IntField f;
    const int idx = g_h5.Tget_member_index(ft, name);
    if (idx < 0) return f;
    const hid_t mt = c.add(4, g_h5.Tget_member_type(ft, (unsigned)idx));
    if (mt < 0 || g_h5.Tget_class(mt) != 0 || g_h5.Tget_order(mt) != 0) return f;
    f.size = g_h5.Tget_size(mt);
    f.off = g_h5.Tget_member_offset(ft, (unsigned)idx);
    f.is_signed = g_h5.Tget_sign(mt) == 1;
    f.ok = f.size == 1 || f.size == 2 || f.size == 4 || f.size == 8;
    return f;
}
inline int64_t get_int(const uint8_t* rec, const IntField& f) {
    uint64_t v = 0;
    memcpy(&v, rec + f.off, f.size);  // little-endian host
    if (f.is_signed && f.size < 8 && (v >> (8 * f.size - 1)) & 1u) v |= ~0ull << (8 * f.size);
    return (int64_t)v;
}

}  // namespace

extern "C" {

int32_t dsp_fast5_available(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (load_locked()) return 1;
    dsp_set_error_(("the HDF5 library could not be loaded: " + g_h5.why).c_str());
    return 0;
}

const char* dsp_fast5_library(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return load_locked() ? g_h5.path.c_str() : "";
}

void dsp_fast5_free(dsp_fast5_read* r) {
    if (!r) return;
    free(r->raw); free(r->ev_start); free(r->ev_len); free(r->ev_base);
    r->raw = nullptr; r->ev_start = r->ev_len = nullptr; r->ev_base = nullptr;
    r->n_raw = r->n_events = 0;
}

static int32_t fast5_load_impl(const char* path, const char* corrected_group, const char* basecall_subgroup,
                               const char* only_chrom, dsp_fast5_read* out);

int32_t dsp_fast5_load(const char* path, const char* corrected_group, const char* basecall_subgroup, const char* only_chrom,
                       dsp_fast5_read* out) {
    if (!path || !corrected_group || !basecall_subgroup || !out) return f5_fail(DSP_EINVAL, "dsp_fast5_load: NULL argument");
    memset(out, 0, sizeof(*out));
    try {  // no C++ exception (an allocation sized by damaged metadata) may cross the C boundary
        return fast5_load_impl(path, corrected_group, basecall_subgroup, only_chrom, out);
    } catch (const std::exception& e) {
        dsp_fast5_free(out);
        return f5_fail(DSP_EPARSE, "damaged fast5 file (%s while decoding it)", e.what());
    } catch (...) {
        dsp_fast5_free(out);
        return f5_fail(DSP_EPARSE, "damaged fast5 file");
    }
}

static int32_t fast5_load_impl(const char* path, const char* corrected_group, const char* basecall_subgroup,
                               const char* only_chrom, dsp_fast5_read* out) {
    Direct dsig, dev;                  // chunk plans of Signal and Events (decoded after the library lock is released)
    IntField f_start, f_len;
    size_t base_off = 0, rec_size = 0;
    int64_t rel = 0;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!load_locked()) return f5_fail(DSP_EINVAL, "reading fast5 files needs the HDF5 library: %s", g_h5.why.c_str());
        g_h5.Eset_auto2(0, nullptr, nullptr);  // thread-safe builds keep one error stack per thread: silence this thread's too
        Closer c;
        const hid_t f = c.add(0, g_h5.Fopen(path, 0 /* H5F_ACC_RDONLY */, 0));
        if (f < 0) {
            // the reference prints a warning, carries on with empty alignment fields (:174-176) and fails in _get_label_raw
            // (:52-55) -- unless a region of interest filters the read out first (:308-309)
            if (only_chrom) return DSP_FAST5_SKIPPED;
            return f5_fail(DSP_EPARSE, "Error opening file. Likely a corrupted file.");
        }
        const std::string sub = std::string("Analyses/") + corrected_group + "/" + basecall_subgroup;

        // ---- alignment attributes + read id (:151-176): missing Alignment group = empty fields, not an error yet
        const bool has_reads = path_exists(f, "Raw/Reads");
        hid_t reads = -1;
        char first[256] = "";
        if (has_reads) {
            reads = c.add(1, g_h5.Gopen2(f, "Raw/Reads", 0));
            if (reads >= 0 && g_h5.Lget_name_by_idx(reads, ".", 0 /* by name */, 0 /* increasing */, 0, first, sizeof(first), 0) < 0) first[0] = 0;
        }
        if (path_exists(f, sub + "/Alignment")) {
            if (!first[0]) return f5_fail(DSP_EPARSE, "no read below Raw/Reads to take the read id from");
            const hid_t rd = c.add(1, g_h5.Gopen2(reads, first, 0));
            if (rd < 0 || !attr_string(rd, "read_id", out->read_id, sizeof(out->read_id), c))
                return f5_fail(DSP_EPARSE, "no read_id attribute on Raw/Reads/%s", first);
            const hid_t al = c.add(1, g_h5.Gopen2(f, (sub + "/Alignment").c_str(), 0));
            double dummy;
            if (al < 0 || !attr_string(al, "mapped_strand", out->mapped_strand, sizeof(out->mapped_strand), c) ||
                !attr_string(al, "mapped_chrom", out->mapped_chrom, sizeof(out->mapped_chrom), c) ||
                !attr_number(al, "mapped_start", true, &dummy, &out->mapped_start, c))
                return f5_fail(DSP_EPARSE, "Alignment attributes (mapped_strand, mapped_chrom, mapped_start) are incomplete");
            out->has_alignment = 1;
        }
        if (only_chrom && strcmp(only_chrom, out->mapped_chrom) != 0) return DSP_FAST5_SKIPPED;  // :308-309

        // ---- raw signal (:57-65)
        {
            hid_t ds = -1;
            if (first[0]) ds = c.add(2, g_h5.Dopen2(reads, (std::string(first) + "/Signal").c_str(), 0));
            const hid_t ft = ds >= 0 ? c.add(4, g_h5.Dget_type(ds)) : -1;
            const hid_t sp = ds >= 0 ? c.add(3, g_h5.Dget_space(ds)) : -1;
            if (ds < 0 || ft < 0 || sp < 0)
                return f5_fail(DSP_EPARSE, "Raw data is not stored in Raw/Reads/Read_[read#] so new segments cannot be identified.");
            if (g_h5.Tget_class(ft) != 0 || g_h5.Tget_size(ft) > 2 || g_h5.Sget_simple_extent_ndims(sp) != 1)
                return f5_fail(DSP_EPARSE, "Raw/Reads/%s/Signal is not a one-dimensional array of 16-bit DAQ values", first);
            const hssize_t n = g_h5.Sget_simple_extent_npoints(sp);
            out->raw = (int16_t*)malloc(sizeof(int16_t) * (size_t)(n > 0 ? n : 1));
            if (!out->raw) return f5_fail(DSP_ENOMEM, "out of host memory");
            out->n_raw = n;
            // little-endian two's-complement int16 in the file: no conversion needed, the chunks can be decoded directly
            if (g_h5.direct && g_h5.Tget_size(ft) == 2 && g_h5.Tget_order(ft) == 0 && g_h5.Tget_sign(ft) == 1)
                plan_direct(f, ds, 2, (hsize_t)(n > 0 ? n : 0), dsig, c);
            if (!dsig.ok && n > 0 && g_h5.Dread(ds, g_h5.t_short, 0, 0, 0, out->raw) < 0) {
                dsp_fast5_free(out);
                return f5_fail(DSP_EPARSE, "Raw data is not stored in Raw/Reads/Read_[read#] so new segments cannot be identified. "
                                           "(the Signal dataset cannot be decoded: VBZ-compressed files need ONT's HDF5 plugin on HDF5_PLUGIN_PATH)");
            }
        }

        // ---- events (:68-89)
        {
            if (!path_exists(f, sub + "/Events")) { dsp_fast5_free(out); return f5_fail(DSP_EPARSE, "events not found."); }
            const hid_t ds = c.add(2, g_h5.Dopen2(f, (sub + "/Events").c_str(), 0));
            const hid_t ft = ds >= 0 ? c.add(4, g_h5.Dget_type(ds)) : -1;
            const hid_t sp = ds >= 0 ? c.add(3, g_h5.Dget_space(ds)) : -1;
            if (ds < 0 || ft < 0 || sp < 0) { dsp_fast5_free(out); return f5_fail(DSP_EPARSE, "events not found."); }
            double dummy;
            if (!attr_number(ds, "read_start_rel_to_raw", true, &dummy, &rel, c)) {
                dsp_fast5_free(out);
                return f5_fail(DSP_EPARSE, "no read_start_rel_to_raw in event attributes");
            }
            const int ib = g_h5.Tget_class(ft) == 6 ? g_h5.Tget_member_index(ft, "base") : -1;
            if (ib < 0 || g_h5.Tget_member_index(ft, "start") < 0 || g_h5.Tget_member_index(ft, "length") < 0) {
                dsp_fast5_free(out);
                return f5_fail(DSP_EPARSE, "the Events table has no start / length / base columns");
            }
            const hssize_t n = g_h5.Sget_simple_extent_npoints(sp);
            out->ev_start = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
            out->ev_len = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
            out->ev_base = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
            if (!out->ev_start || !out->ev_len || !out->ev_base) { dsp_fast5_free(out); return f5_fail(DSP_ENOMEM, "out of host memory"); }
            out->n_events = n;
            if (g_h5.direct && g_h5.Sget_simple_extent_ndims(sp) == 1) {  // records decoded by hand from the file's own layout
                f_start = int_field(ft, "start", c);
                f_len = int_field(ft, "length", c);
                const hid_t bt = c.add(4, g_h5.Tget_member_type(ft, (unsigned)ib));
                rec_size = g_h5.Tget_size(ft);
                base_off = g_h5.Tget_member_offset(ft, (unsigned)ib);
                if (f_start.ok && f_len.ok && bt >= 0 && g_h5.Tget_class(bt) == 3 && g_h5.Tis_variable_str(bt) == 0 && g_h5.Tget_size(bt) >= 1)
                    plan_direct(f, ds, rec_size, (hsize_t)(n > 0 ? n : 0), dev, c);
            }
            if (!dev.ok && n > 0) {  // through the library: one {int64, int64, char} memory type whatever the file's column widths
                struct Row { int64_t start, length; char base[8]; };
                const hid_t s1 = c.add(4, g_h5.Tcopy(g_h5.t_c_s1));
                const hid_t mt = c.add(4, g_h5.Tcreate(6 /* H5T_COMPOUND */, sizeof(Row)));
                // (a null-terminated C string of size 2 holds the one character of the file's null-padded S1)
                if (s1 < 0 || mt < 0 || g_h5.Tset_size(s1, 2) < 0 || g_h5.Tinsert(mt, "start", offsetof(Row, start), g_h5.t_int64) < 0 ||
                    g_h5.Tinsert(mt, "length", offsetof(Row, length), g_h5.t_int64) < 0 || g_h5.Tinsert(mt, "base", offsetof(Row, base), s1) < 0) {
                    dsp_fast5_free(out);
                    return f5_fail(DSP_EPARSE, "cannot build the memory type of the Events table");
                }
                std::vector<Row> rows((size_t)n);
                if (g_h5.Dread(ds, mt, 0, 0, 0, rows.data()) < 0) { dsp_fast5_free(out); return f5_fail(DSP_EPARSE, "events not found. (the table cannot be decoded)"); }
                for (hssize_t i = 0; i < n; ++i) {
                    out->ev_start[i] = rows[(size_t)i].start + rel;  // :81
                    out->ev_len[i] = rows[(size_t)i].length;
                    out->ev_base[i] = (uint8_t)rows[(size_t)i].base[0];
                }
            }
        }

        // ---- channel scaling (:255-270)
        {
            if (!path_exists(f, "UniqueGlobalKey/channel_id")) { dsp_fast5_free(out); return f5_fail(DSP_EPARSE, "no UniqueGlobalKey/channel_id group"); }
            const hid_t ch = c.add(1, g_h5.Gopen2(f, "UniqueGlobalKey/channel_id", 0));
            int64_t idummy;
            if (ch < 0 || !attr_number(ch, "digitisation", false, &out->digitisation, &idummy, c) ||
                !attr_number(ch, "range", false, &out->range, &idummy, c) || !attr_number(ch, "offset", false, &out->offset, &idummy, c)) {
                dsp_fast5_free(out);
                return f5_fail(DSP_EPARSE, "channel_id lacks digitisation / range / offset");
            }
        }
    }  // every HDF5 object is closed and the library lock released here

    // ---- the planned chunks: pread + inflate + un-shuffle, concurrently with other loader threads
    if ((dsig.ok && out->n_raw > 0) || (dev.ok && out->n_events > 0)) {
        const int fd = open(path, O_RDONLY | O_CLOEXEC);
        bool ok = fd >= 0;
        if (ok && dsig.ok && out->n_raw > 0) ok = read_direct(fd, dsig, (uint8_t*)out->raw);
        if (ok && dev.ok && out->n_events > 0) {
            std::vector<uint8_t> recs((size_t)out->n_events * rec_size);
            ok = read_direct(fd, dev, recs.data());
            for (int64_t i = 0; ok && i < out->n_events; ++i) {
                const uint8_t* r = recs.data() + (size_t)i * rec_size;
                out->ev_start[i] = get_int(r, f_start) + rel;  // :81
                out->ev_len[i] = get_int(r, f_len);
                out->ev_base[i] = r[base_off];
            }
        }
        if (fd >= 0) close(fd);
        if (!ok) {
            dsp_fast5_free(out);
            return f5_fail(DSP_EPARSE, "Raw data is not stored in Raw/Reads/Read_[read#] so new segments cannot be identified. "
                                       "(a chunk of the Signal / Events dataset could not be read or inflated: damaged file)");
        }
    }
    return 0;
}

}  // extern "C"
