// hip/hip_runtime.h of the test-suite's SIMT interpreter -- the DEVICE side.  csrc/dsp_kernels.hip, compiled for the host with
// this header in front of the real one (-DDSP_EMU -I tests/native/emu -x c++), runs lane by lane: every thread of a launch is a
// fiber, 64 of them a wave, the waves of all workgroups of the launch are scheduled round robin (so that the clustered launches'
// members really wait for each other); an instruction that involves other lanes -- an MFMA, readfirstlane, a shuffle, the
// workgroup barrier -- parks the lane until its wave (its workgroup) has arrived and is then carried out for all of them at once.
// Buffer descriptors check the range as the hardware does: a load past num_records returns zeros, a store past it is dropped.
// Sequentially consistent memory, no timing: this checks the kernels' LOGIC (indexing, extents, the cluster protocol's control
// flow, the clean-up path), never their speed.  TEST INFRASTRUCTURE; see tests/native/emu/hip_emu.cpp, tests/test_kernel_emu.py.
#ifndef DSP_EMU_HIP_RUNTIME_H
#define DSP_EMU_HIP_RUNTIME_H

#include "hip_runtime_api.h"

#include <math.h>
#include <stdio.h>

#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int emu_u32x4 __attribute__((ext_vector_type(4)));
struct __amdgpu_buffer_rsrc_t { char* base; uint32_t nrec; };

namespace emu {
struct LaneCtx { dim3 tid, bid, bdim, gdim; float* lds; int lane; };
extern LaneCtx* g_cur;                                   // the lane that is running
inline LaneCtx* cur() { return g_cur; }
// sequential: one workgroup at a time (kernels whose workgroups never wait for each other; what makes `static` a faithful
// stand-in for a kernel's static __shared__ variables -- DSP_EMU_STATIC_LDS)
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body, bool sequential = false);
// instructions that involve other lanes (hip_emu.cpp)
uint32_t readfirstlane_u32(uint32_t v);
uint64_t shfl_u64(uint64_t v, int kind, int arg, int width);   // kind 0: xor mask, 1: up delta, 2: source lane (within groups of `width`)
uint64_t ballot(bool pred);                              // bit l set: lane l is active and its predicate holds
void mfma_f32(float a, float b, emu_f32x16* acc);        // 32x32x2: acc += A * B
void mfma_k16(const float a[8], const float b[8], emu_f32x16* acc);   // 32x32x16 (bf16 / f16 operands already widened)
void barrier();
void yield();
unsigned long long now();

template <class T> inline T readfirstlane(T v) {
    static_assert(sizeof(T) == 4, "readfirstlane of a 32-bit value");
    uint32_t u; memcpy(&u, &v, 4); u = readfirstlane_u32(u); memcpy(&v, &u, 4); return v;
}
template <class T> inline T shfl(T v, int kind, int arg, int width = 64) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "shuffle of a 32- or 64-bit value");
    uint64_t u = 0; memcpy(&u, &v, sizeof(T)); u = shfl_u64(u, kind, arg, width); memcpy(&v, &u, sizeof(T)); return v;
}
inline emu_f32x16 mfma_f32_32x32x2(float a, float b, emu_f32x16 c) { mfma_f32(a, b, &c); return c; }
template <class V> inline emu_f32x16 mfma_32x32x16(V a, V b, emu_f32x16 c) {
    float fa[8], fb[8];
    for (int i = 0; i < 8; ++i) { fa[i] = (float)a[i]; fb[i] = (float)b[i]; }
    mfma_k16(fa, fb, &c);
    return c;
}
// raw buffer access with the hardware's range check: the offset is voffset + soffset, added WITHOUT wrapping at 32 bits
inline emu_u32x4 buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const unsigned long long off = (unsigned long long)(uint32_t)voff + (unsigned long long)(uint32_t)soff;
    emu_u32x4 v = {0u, 0u, 0u, 0u};
    if (off + 16ull <= (unsigned long long)r.nrec) memcpy(&v, r.base + off, 16);
    return v;
}
inline void buffer_store_b128(emu_u32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const unsigned long long off = (unsigned long long)(uint32_t)voff + (unsigned long long)(uint32_t)soff;
    if (off + 16ull <= (unsigned long long)r.nrec) memcpy(r.base + off, &v, 16);
}
}  // namespace emu

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __constant__ const
#define __HIPCC_EMU__ 1
#define threadIdx (emu::cur()->tid)
#define blockIdx (emu::cur()->bid)
#define blockDim (emu::cur()->bdim)
#define gridDim (emu::cur()->gdim)
#define HIP_SYMBOL(x) (&(x))
static inline hipError_t hipMemcpyFromSymbol(void* dst, const void* sym, size_t n) { memcpy(dst, sym, n); return hipSuccess; }
static inline hipError_t hipMemcpyToSymbol(void* sym, const void* src, size_t n) { memcpy(sym, src, n); return hipSuccess; }
#ifdef DSP_EMU_STATIC_LDS   // a translation unit with static __shared__ variables (csrc/dsp_extract.hip): its workgroups run one after another
#define __shared__ static
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) emu::launch(grid, block, lds, [=]() { kern(__VA_ARGS__); }, true)
#else
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) emu::launch(grid, block, lds, [=]() { kern(__VA_ARGS__); })
#endif

// the five arch macros of csrc/dsp_kernels.hip
#define DSP_STORE_GUARD(v) ((void)(v))
#define DSP_KEEP_SGPR2(a, b) ((void)0)
#define DSP_DRAIN_STORES() ((void)0)
#define DSP_READ_XCC_ID(x) ((x) = blockIdx.x % 8u)
#define DSP_DYN_LDS(name) float* name = emu::cur()->lds
#define DSP_DYN_LDS_T(type, name) type* name = (type*)emu::cur()->lds          /* csrc/dsp_parse_dev.hip, dsp_extract.hip */
#define DSP_LOCKSTEP_SHARED(type, ptr) type ptr##_of_this_lane; ptr = &ptr##_of_this_lane
#define DSP_WAVE_LOCKSTEP() ((void)emu::ballot(true))   /* every lane of the wave that is here has executed what precedes */

// builtins of the amdgcn target
#define __builtin_amdgcn_readfirstlane(x) emu::readfirstlane(x)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu::mfma_f32_32x32x2((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu::mfma_32x32x16((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu::mfma_32x32x16((a), (b), (c))
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, nrec, flags) (__amdgpu_buffer_rsrc_t{(char*)(p), (uint32_t)(nrec)})
#define __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, aux) emu::buffer_load_b128((r), (voff), (soff))
#define __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, aux) emu::buffer_store_b128((v), (r), (voff), (soff))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) emu::yield()
#define __builtin_amdgcn_s_memtime() emu::now()
#define __builtin_amdgcn_s_getreg(x) (0u)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_sqrtf(x) sqrtf(x)
#define __builtin_amdgcn_logf(x) log2f(x)          /* v_log_f32 is a base-2 logarithm */
#define __syncthreads() emu::barrier()
#define __shfl_xor(v, m) emu::shfl((v), 0, (m))
#define __shfl_up(v, d) emu::shfl((v), 1, (d))
#define EMU_SHFL_PICK(_1, _2, _3, NAME, ...) NAME
#define EMU_SHFL2(v, src) emu::shfl((v), 2, (src))
#define EMU_SHFL3(v, src, width) emu::shfl((v), 2, (src), (width))
#define __shfl(...) EMU_SHFL_PICK(__VA_ARGS__, EMU_SHFL3, EMU_SHFL2)(__VA_ARGS__)
#define __ballot(p) emu::ballot((p))
#define __popcll(x) __builtin_popcountll((unsigned long long)(x))
#define __ffsll(x) __builtin_ffsll((long long)(x))
#define __ffs(x) __builtin_ffs((int)(x))
static inline long long __double_as_longlong(double v) { long long u; memcpy(&u, &v, 8); return u; }
static inline double __longlong_as_double(long long u) { double v; memcpy(&v, &u, 8); return v; }
static inline int atomicAdd(int* p, int v) { const int o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }
// HIP's short vector types as far as the kernels use them
struct alignas(16) int4 { int x, y, z, w; };
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
struct alignas(16) float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct alignas(16) double2 { double x, y; };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((unsigned long long)a * b) >> 32); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
static inline unsigned atomicCAS(unsigned* p, unsigned cmp, unsigned v) { const unsigned o = *p; if (o == cmp) *p = v; return o; }
// one OS thread runs every lane: the "atomics" are plain accesses (sequentially consistent by construction)
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, order, scope) (*(volatile unsigned*)(p))
#define __hip_atomic_fetch_add(p, v, order, scope) emu_fetch_add((unsigned*)(p), (v))
#define __hip_atomic_fetch_or(p, v, order, scope) emu_fetch_or((unsigned*)(p), (v))
#define __hip_atomic_compare_exchange_strong(p, exp, des, o1, o2, scope) emu_cas((unsigned*)(p), (exp), (des))
static inline unsigned emu_fetch_add(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
static inline unsigned emu_fetch_or(unsigned* p, unsigned v) { const unsigned o = *p; *p = o | v; return o; }
static inline bool emu_cas(unsigned* p, unsigned* expected, unsigned desired) {
    if (*p == *expected) { *p = desired; return true; }
    *expected = *p;
    return false;
}

#endif
