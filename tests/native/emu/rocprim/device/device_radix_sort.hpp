// rocprim/device/device_radix_sort.hpp of the test-suite's SIMT interpreter (tests/native/emu): the one rocPRIM entry point
// csrc/dsp_freq_dev.hip uses, as a stable sort on the host.  (rocPRIM's radix sort of signed keys over all their bits is a
// STABLE sort in numeric order: that is the contract the kernels around it rely on.)  TEST INFRASTRUCTURE.
#ifndef DSP_EMU_ROCPRIM_RADIX_SORT_HPP
#define DSP_EMU_ROCPRIM_RADIX_SORT_HPP

#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace rocprim {
template <class K, class V>
inline hipError_t radix_sort_pairs(void* temporary_storage, size_t& storage_size, const K* keys_in, K* keys_out, const V* values_in, V* values_out, size_t size,
                                   unsigned begin_bit = 0, unsigned end_bit = 8 * sizeof(K), hipStream_t = nullptr, bool = false) {
    if (!temporary_storage) { storage_size = 256; return hipSuccess; }
    if (begin_bit != 0 || end_bit != 8 * sizeof(K)) return hipErrorInvalidValue;
    std::vector<size_t> idx(size);
    std::iota(idx.begin(), idx.end(), (size_t)0);
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return keys_in[a] < keys_in[b]; });
    for (size_t i = 0; i < size; ++i) { keys_out[i] = keys_in[idx[i]]; values_out[i] = values_in[idx[i]]; }
    return hipSuccess;
}
}  // namespace rocprim

#endif
