// hip/hip_runtime_api.h of the test-suite's SIMT interpreter (tests/native/emu; tests/test_kernel_emu.py) -- found in front of
// the real header by -I order when libdsp_amd_emu.so is built.  TEST INFRASTRUCTURE: the host half of the HIP runtime as far as
// csrc/dsp_capi.cpp and csrc/dsp_kernels.hip use it, over host memory: "device" allocations are malloc'ed, copies are memcpy,
// streams and events do nothing (a launch runs to completion inside hipLaunchKernelGGL).  Never loadable as the product: the
// library built from it answers dsp_abi_version() with DSP_AMD_ABI_VERSION + 1000 and deepsignal_plant_amd/_native.py refuses it.
#ifndef DSP_EMU_HIP_RUNTIME_API_H
#define DSP_EMU_HIP_RUNTIME_API_H

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorNoDevice = 100 };
typedef struct emu_stream_* hipStream_t;
typedef struct emu_event_* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
typedef struct { char bytes[16]; } hipUUID;
typedef struct hipDeviceProp_t { int multiProcessorCount; hipUUID uuid; char name[64]; } hipDeviceProp_t;

#ifdef __cplusplus
extern "C" {
#endif
int emu_compute_units(void);   // EMU_CUS, default 256
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorInvalidValue ? "invalid argument" : "emulated HIP error"); }
static inline hipError_t hipGetLastError(void) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)(uintptr_t)0x51; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)(uintptr_t)0xe1; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)(uintptr_t)0xe1; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
// (device memory is not zeroed by hipMalloc: every fresh block is filled with 0xff -- quiet NaNs as floats, a set "abandoned" bit and an
// absurd count as a cluster's counters -- so that a kernel reading scratch it has not written shows in the results)
static inline hipError_t hipMalloc(void** p, size_t n) {
    const size_t m = n ? (n + 255) / 256 * 256 : 256;
    *p = aligned_alloc(256, m);
    if (*p) memset(*p, 0xff, m);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, enum hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, enum hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof *p);
    p->multiProcessorCount = emu_compute_units();
    for (int i = 0; i < 16; ++i) p->uuid.bytes[i] = (char)(0xe0 + i);
    strcpy(p->name, "SIMT interpreter (tests/native/emu)");
    return hipSuccess;
}
static inline hipError_t hipDeviceGetPCIBusId(char* buf, int cap, int) { strncpy(buf, "0000:E0:00.0", (size_t)cap); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, enum hipFuncAttribute, int) { return hipSuccess; }
#ifdef __cplusplus
}
#endif

#endif
