// dsp_freq_dev.hip -- the device side of `call_freq` (deepsignal_plant/call_mods_freq.py:29-74,
// utils/txt_formater.py:8-46): per-read calls stay in HBM as compact records and are reduced to per-site statistics
// there.  Exactness is the design constraint: the reference adds the PRINTED probabilities of a site's records one by
// one, in file order, in double -- and because every term is a multiple of 1e-6, exact ties of the "%.3f" output are
// common (about one site in a thousand), so a different association of the sums changes output bytes.  Hence:
//   * dsp_freq_dev_encode: probabilities -> the integers k0, k1 the per-read file would print as k*1e-6
//     (call_modifications.py:177-179: round(p0/(p0+p1), 6), round(1 - that, 6) in float32; the shortest round-trip
//     decimal of such a float32 IS k*1e-6, and the reference's float() of it is the correctly rounded k/1e6 --
//     tests/test_call_freq.py checks both for every k), the |p0 - p1| >= prob_cf filter in double
//     (txt_formater.py:23-26), packed next to the label and the first-record metadata.
//   * dsp_freq_dev_sort_records: the records STABLY sorted by site key (file order inside a site survives; a radix
//     sort of (key, index) pairs by rocPRIM, then one gather of the other three columns), then
//   * dsp_freq_dev_reduce: one thread per site walks its records in order and adds k/1e6 in double, sequentially:
//     the same additions in the same order as calculate_mods_frequency.
// HBM-bound integer work: 32 B per record in, 72 B per site out; no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "dsp_amd.h"

extern "C" void dsp_set_error_(const char* msg);

namespace {

constexpr long long kUnused = 0x7fffffffffffffffll;

__global__ __launch_bounds__(256) void freq_encode_kernel(long long n, const float* __restrict__ probs, int C,
                                                          const uint8_t* __restrict__ labels,
                                                          const long long* __restrict__ key_in,
                                                          const uint32_t* __restrict__ meta_in, double prob_cf,
                                                          long long* __restrict__ key_out, long long* __restrict__ packed_out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = probs[i * C], b = probs[i * C + 1];
    // float32, one rounding per operation, as numpy evaluates it (this file is compiled with -ffp-contract=off)
    const float sum = a + b;
    const float q = a / sum;
    const float r0 = rintf(q * 1e6f);      // np.round(x, 6) = rint(x * 1e6) / 1e6
    const float z0 = r0 / 1e6f;
    const float om = 1.0f - z0;
    const float r1 = rintf(om * 1e6f);
    const long long k0 = (long long)r0, k1 = (long long)r1;
    const double p0 = (double)k0 / 1e6, p1 = (double)k1 / 1e6;  // float("0.123457"): correctly rounded
    const bool ok = k0 >= 0 && k0 <= 1000000 && k1 >= 0 && k1 <= 1000000;  // NaN / out of range: never stored
    const bool used = ok && fabs(p0 - p1) >= prob_cf;
    key_out[i] = used ? key_in[i] : kUnused;
    packed_out[i] = used ? (k0 | (k1 << 20) | ((long long)(labels[i] == 1) << 40) | ((long long)meta_in[i] << 41)) : 0;
}

__global__ __launch_bounds__(256) void freq_count_kernel(long long n, const long long* __restrict__ key, long long* n_sites) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool head = i < n && key[i] != kUnused && (i == 0 || key[i - 1] != key[i]);
    const unsigned long long m = __ballot(head);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd((unsigned long long*)n_sites, (unsigned long long)__popcll(m));
}

__global__ __launch_bounds__(256) void freq_reduce_kernel(long long n, const long long* __restrict__ key,
                                                          const long long* __restrict__ packed,
                                                          const long long* __restrict__ pis, const long long* __restrict__ row,
                                                          long long* slot_counter, long long cap, long long* site_key,
                                                          long long* site_first_row, long long* site_packed,
                                                          long long* site_pis, double* sum0, double* sum1, long long* met,
                                                          long long* cov) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long k = key[i];
    if (k == kUnused || (i > 0 && key[i - 1] == k)) return;  // not the first record of a site
    double s0 = 0.0, s1 = 0.0;
    long long m = 0, c = 0;
    for (long long j = i; j < n && key[j] == k; ++j) {  // file order inside the site (stable sort)
        const long long p = packed[j];
        s0 += (double)(p & 0xfffff) / 1e6;
        s1 += (double)((p >> 20) & 0xfffff) / 1e6;
        m += (p >> 40) & 1;
        ++c;
    }
    const long long slot = (long long)atomicAdd((unsigned long long*)slot_counter, 1ull);
    if (slot >= cap) return;  // the caller sized the outputs from freq_count_kernel: cannot happen
    site_key[slot] = k;
    site_first_row[slot] = row[i];      // the site's first used record: its metadata is the site's (txt_formater.py:52-56)
    site_packed[slot] = packed[i];
    site_pis[slot] = pis[i];
    sum0[slot] = s0; sum1[slot] = s1; met[slot] = m; cov[slot] = c;
}

int fail_hip(const char* what, hipError_t e) {
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    dsp_set_error_(buf);
    return DSP_EHIP;
}

}  // namespace

extern "C" {

int32_t dsp_freq_dev_encode(void* stream, int64_t n, const float* probs, int32_t num_classes, const uint8_t* labels,
                            const int64_t* key_in, const uint32_t* meta_in, double prob_cf, int64_t* key_out,
                            int64_t* packed_out) {
    if (n < 0 || num_classes < 2 || (n && (!probs || !labels || !key_in || !meta_in || !key_out || !packed_out))) {
        dsp_set_error_("dsp_freq_dev_encode: bad argument");
        return DSP_EINVAL;
    }
    if (n == 0) return DSP_OK;
    hipLaunchKernelGGL(freq_encode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long)n,
                       probs, (int)num_classes, labels, (const long long*)key_in, meta_in, prob_cf, (long long*)key_out,
                       (long long*)packed_out);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? DSP_OK : fail_hip("dsp_freq_dev_encode", e);
}

// gather of the three payload columns by the permutation of the sort: 8 B in + 3 x (8 B in, 8 B out) per record
__global__ __launch_bounds__(256) void freq_gather3_kernel(long long n, const unsigned* __restrict__ perm,
                                                           const long long* __restrict__ a, const long long* __restrict__ b,
                                                           const long long* __restrict__ c, long long* __restrict__ ao,
                                                           long long* __restrict__ bo, long long* __restrict__ co) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned j = perm[i];
    ao[i] = a[j]; bo[i] = b[j]; co[i] = c[j];
}
__global__ __launch_bounds__(256) void freq_iota_kernel(long long n, unsigned* __restrict__ p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (unsigned)i;
}

// Records (key, a, b, c) -> the same records STABLY sorted by key, out of place.  tmp == NULL: *tmp_bytes = the scratch
// needed for n records (device memory, any alignment of 256) and nothing runs.  Keys are non-negative int64 (site keys,
// global row indices, INT64_MAX for unused records): all 64 bits take part.
int32_t dsp_freq_dev_sort_records(void* stream, int64_t n, const int64_t* key, const int64_t* a, const int64_t* b,
                                  const int64_t* c, int64_t* key_out, int64_t* a_out, int64_t* b_out, int64_t* c_out,
                                  void* tmp, size_t* tmp_bytes) {
    if (n < 0 || n >= (1ll << 32) || !tmp_bytes) {
        dsp_set_error_("dsp_freq_dev_sort_records: bad argument (at most 2^32 - 1 records per call)");
        return DSP_EINVAL;
    }
    const size_t idx_bytes = ((size_t)n * sizeof(unsigned) + 255) / 256 * 256;
    size_t sort_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_bytes, (const long long*)nullptr, (long long*)nullptr,
                                             (const unsigned*)nullptr, (unsigned*)nullptr, (size_t)n, 0, 64, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip("dsp_freq_dev_sort_records (size query)", e);
    const size_t need = 2 * idx_bytes + (sort_bytes + 255) / 256 * 256;
    if (!tmp) { *tmp_bytes = need; return DSP_OK; }
    if (*tmp_bytes < need) {
        dsp_set_error_("dsp_freq_dev_sort_records: scratch too small");
        return DSP_ENOMEM;
    }
    if (n == 0) return DSP_OK;
    if (!key || !a || !b || !c || !key_out || !a_out || !b_out || !c_out) {
        dsp_set_error_("dsp_freq_dev_sort_records: NULL column");
        return DSP_EINVAL;
    }
    unsigned* idx_in = (unsigned*)tmp;
    unsigned* idx_out = (unsigned*)((char*)tmp + idx_bytes);
    void* sort_tmp = (char*)tmp + 2 * idx_bytes;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(freq_iota_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (long long)n, idx_in);
    e = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, (const long long*)key, (long long*)key_out, (const unsigned*)idx_in,
                                  idx_out, (size_t)n, 0, 64, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip("dsp_freq_dev_sort_records", e);
    hipLaunchKernelGGL(freq_gather3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (long long)n, idx_out,
                       (const long long*)a, (const long long*)b, (const long long*)c, (long long*)a_out, (long long*)b_out,
                       (long long*)c_out);
    e = hipGetLastError();
    return e == hipSuccess ? DSP_OK : fail_hip("dsp_freq_dev_sort_records", e);
}

int32_t dsp_freq_dev_count_sites(void* stream, int64_t n, const int64_t* key_sorted, int64_t* n_sites) {
    if (n < 0 || !n_sites || (n && !key_sorted)) {
        dsp_set_error_("dsp_freq_dev_count_sites: bad argument");
        return DSP_EINVAL;
    }
    hipError_t e = hipMemsetAsync(n_sites, 0, sizeof(int64_t), (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip("dsp_freq_dev_count_sites", e);
    if (n == 0) return DSP_OK;
    hipLaunchKernelGGL(freq_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long)n,
                       (const long long*)key_sorted, (long long*)n_sites);
    e = hipGetLastError();
    return e == hipSuccess ? DSP_OK : fail_hip("dsp_freq_dev_count_sites", e);
}

int32_t dsp_freq_dev_reduce(void* stream, int64_t n, const int64_t* key_sorted, const int64_t* packed_sorted,
                            const int64_t* pis_sorted, const int64_t* row_sorted, int64_t* slot_counter, int64_t cap,
                            int64_t* site_key, int64_t* site_first_row, int64_t* site_packed, int64_t* site_pis,
                            double* sum0, double* sum1, int64_t* met, int64_t* cov) {
    if (n < 0 || cap < 0 || !slot_counter ||
        (n && (!key_sorted || !packed_sorted || !pis_sorted || !row_sorted)) ||
        (cap && (!site_key || !site_first_row || !site_packed || !site_pis || !sum0 || !sum1 || !met || !cov))) {
        dsp_set_error_("dsp_freq_dev_reduce: bad argument");
        return DSP_EINVAL;
    }
    hipError_t e = hipMemsetAsync(slot_counter, 0, sizeof(int64_t), (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip("dsp_freq_dev_reduce", e);
    if (n == 0) return DSP_OK;
    hipLaunchKernelGGL(freq_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long)n,
                       (const long long*)key_sorted, (const long long*)packed_sorted, (const long long*)pis_sorted,
                       (const long long*)row_sorted, (long long*)slot_counter, (long long)cap, (long long*)site_key,
                       (long long*)site_first_row, (long long*)site_packed, (long long*)site_pis, sum0, sum1,
                       (long long*)met, (long long*)cov);
    e = hipGetLastError();
    return e == hipSuccess ? DSP_OK : fail_hip("dsp_freq_dev_reduce", e);
}

}  // extern "C"
