// h_exchange.hip -- what the LSTM kernels' h exchange through global memory relies on, tested in isolation on gfx950.
//
// dsp_lstm_kernel (csrc/dsp_kernels.hip) lets every wave of a workgroup store its slice of h_t with buffer_store_dwordx4,
// passes ONE s_barrier (hipcc's workgroup-scope __syncthreads(): no s_waitcnt vmcnt(0) in front of it outside
// threadgroup-split mode) and lets every wave read the whole of h_t back with buffer_load_dwordx4.  Round 2 saw a
// <2 unit tiles, 1 site tile> build of that kernel read h0 back wrong for lanes 12-15 / 28-31 of a tile -- whole 64-byte
// beats of the store data -- non-deterministically, for the second wave of a SIMD only, and only in some builds.
// Two families of explanation, both tested here:
//
//  A. memory ordering / caching of the exchange itself:  xchg<WAVES, VMCNT0, STALE>
//     every wave stores 4 x 16 B per lane (values a function of launch, workgroup, step, wave, lane), barrier, every wave
//     loads ALL waves' blocks and compares.  VMCNT0 puts s_waitcnt vmcnt(0) before the barrier; STALE makes every wave
//     first LOAD the destination rows (last launch's values, or another workgroup's) so that this CU's vector L1 holds
//     lines the stores then have to update or invalidate (the "a request that touched an h_t row early could leave
//     stale lines in this CU's L1" worry in the kernel's comments).  4-wave workgroups share CUs (two and more per CU).
//
//  B. a write-after-read hazard on the STORE DATA registers:  war<MODE, NOPS>
//     buffer_store_dwordx4 v[32:35] followed after NOPS wait states by something that overwrites v[32:35]: 4 v_mov
//     (MODE 0), one v_mfma_f32_32x32x2_f32 whose destination covers them (1), 4 v_exp_f32 (2), a buffer_load_dwordx4
//     into them (3).  Written in inline asm so that hipcc's hazard recognizer cannot pad the sequence.  The ISA manual
//     requires wait states between a VMEM store of more than 64 bits and a VALU write of its data registers; the
//     compiler inserts them -- for the instructions it models.  Memory is checked from the host afterwards.
//
// build: hipcc --offload-arch=gfx950 -O2 -o h_exchange h_exchange.hip ; run: ./h_exchange [launches]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7ffffff0, 0x00020000);
}
__device__ __forceinline__ f32x4 bld16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void bst16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, (int)soff, 0);
}

// the value lane `lane` of wave `w` of workgroup `wg` stores at (step, aa, i) in launch `launch`: exact in fp32
__device__ __host__ inline float expect(int launch, int wg, int step, int w, int aa, int lane, int i) {
    return (float)(((launch * 131 + wg) * 17 + step) % 4093) + (float)(w * 4 + aa) * 4096.0f + (float)(lane * 4 + i) * (1.0f / 256.0f);
}

// ---- A: the exchange -------------------------------------------------------------------------------------------
// GUARD: the stores are followed by `s_nop 1` with the data registers kept live (what dsp_kernels.hip does since round 3).
// Without it hipcc itself emits hazard B for this kernel (buffer_store_dwordx4 v[4:7] ... / v_pk_add_f32 v[6:7] ...): the
// unguarded rows of part A then show B's corruption, not a property of the exchange.
template <int WAVES, bool VMCNT0, bool STALE, bool GUARD = true>
__global__ __launch_bounds__(WAVES * 64) void xchg(float* scratch, int steps, int launch, unsigned* bad, unsigned* first_bad) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t voff = (uint32_t)lane * 16u;
    const uint32_t row = (uint32_t)WAVES * 4096u;                      // bytes of one step's h of this workgroup
    const __amdgpu_buffer_rsrc_t r = make_rsrc((const char*)scratch + (size_t)blockIdx.x * steps * row);
    unsigned nbad = 0;
    float sink = 0.f;
    for (int step = 0; step < steps; ++step) {
        if (STALE) {  // pull the destination rows (the previous launch's values) into this CU's L1 first
#pragma unroll
            for (int u = 0; u < WAVES; ++u) sink += bld16(r, voff, (uint32_t)step * row + (uint32_t)u * 4096u)[0];
            __builtin_amdgcn_s_waitcnt(0);   // the stale lines are now resident
        }
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) {
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // through the transcendental unit, like the cell phase that produces h_t (exact: exp2(log2(x)) is not
                // exact in general, so the value only passes through a v_rcp of a power of two)
                const float e = expect(launch, blockIdx.x, step, w, aa, lane, i);
                v[i] = e * __builtin_amdgcn_rcpf(1.0f + (float)(step & 0));
            }
            bst16(r, voff + aa * 1024u, (uint32_t)step * row + (uint32_t)w * 4096u, v);
            if (GUARD) asm volatile("s_nop 1" : : "v"(v));
        }
        if (VMCNT0) __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < WAVES; ++u)
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                const f32x4 g = bld16(r, voff + aa * 1024u, (uint32_t)step * row + (uint32_t)u * 4096u);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (g[i] != expect(launch, blockIdx.x, step, u, aa, lane, i)) {
                        if (nbad == 0 && atomicAdd(&first_bad[0], 1u) == 0) {
                            first_bad[1] = blockIdx.x; first_bad[2] = step; first_bad[3] = (unsigned)(w * 1000 + u * 100 + aa * 10 + i);
                            first_bad[4] = lane; first_bad[5] = __builtin_bit_cast(unsigned, g[i]);
                        }
                        ++nbad;
                    }
            }
    }
    if (nbad) atomicAdd(bad, nbad);
    if (sink == 12345.678f) bad[1] = 1;  // keep the stale loads
}

// ---- B: store-data write-after-read ------------------------------------------------------------------------------
#define NOP_0 ""
#define NOP_1 "s_nop 0\n"
#define NOP_2 "s_nop 1\n"
#define NOP_4 "s_nop 3\n"
template <int MODE, int NOPS, bool IMM = false>
__global__ __launch_bounds__(512) void war(float* out, const float* other, int iters) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t voff = (uint32_t)lane * 16u;
    // wave-major layout: [workgroup][wave][iter][64 lanes][4]
    // the buffer descriptor by hand (inline asm wants plain SGPRs): base, stride 0, num_records, the flags make_rsrc uses
    auto desc = [](const void* base) {
        const uint64_t b = (uint64_t)base;
        u32x4 d;
        d[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
        d[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
        d[2] = 0x7ffffff0u;
        d[3] = 0x00020000u;
        return d;
    };
    const u32x4 rs = desc((const char*)out + ((size_t)blockIdx.x * 8 + w) * iters * 1024);
    const u32x4 ro = desc(other);
    const float x = 1.0f + lane * 1e-3f, y = 0.5f;
    for (int it = 0; it < iters; ++it) {
        const float a = (float)(lane * 4) + (float)(it % 1021) * 256.0f + (float)(w + 8 * (blockIdx.x % 61)) * 262144.0f;
        // IMM: the documented hazard form -- soffset is an immediate (the iteration's offset goes into the VGPR): the ISA
        // manual asks for 1 wait state before a VALU write of the data registers THEN (an SGPR soffset costs the store an
        // extra issue cycle, which hides it).  The LSTM kernels always pass an SGPR soffset.
        const uint32_t soff = IMM ? 0u : (uint32_t)it * 1024u;
        const uint32_t voff_it = IMM ? voff + (uint32_t)it * 1024u : voff;
#define WAR_ASM(NOPSTR, OVER)                                                                             \
        asm volatile("v_mov_b32 v32, %[a]\n v_add_f32 v33, 1.0, %[a]\n v_add_f32 v34, 2.0, %[a]\n v_add_f32 v35, 3.0, %[a]\n" \
                     "v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n"                    \
                     "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n"                    \
                     "v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n"                    \
                     "s_nop 7\n"                                                                              \
                     "buffer_store_dwordx4 v[32:35], %[vst], %[rs], %[soff] offen\n" NOPSTR OVER                 \
                     "s_waitcnt vmcnt(0)\n"                                                                   \
                     :                                                                                        \
                     : [a] "v"(a), [voff] "v"(voff), [vst] "v"(voff_it), [rs] "s"(rs), [ro] "s"(ro), [soff] "s"(soff), [x] "v"(x), [y] "v"(y) \
                     : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", \
                       "v46", "v47", "memory")
#define WAR_ASM_IMM(NOPSTR, OVER)                                                                         \
        asm volatile("v_mov_b32 v32, %[a]\n v_add_f32 v33, 1.0, %[a]\n v_add_f32 v34, 2.0, %[a]\n v_add_f32 v35, 3.0, %[a]\n" \
                     "v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n"                    \
                     "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n"                    \
                     "v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n"                    \
                     "s_nop 7\n"                                                                              \
                     "buffer_store_dwordx4 v[32:35], %[vst], %[rs], 0 offen\n" NOPSTR OVER                       \
                     "s_waitcnt vmcnt(0)\n"                                                                   \
                     :                                                                                        \
                     : [a] "v"(a), [voff] "v"(voff), [vst] "v"(voff_it), [rs] "s"(rs), [ro] "s"(ro), [x] "v"(x), [y] "v"(y) \
                     : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", \
                       "v46", "v47", "memory")
#define OVER_MOV "v_mov_b32 v32, -1.0\n v_mov_b32 v33, -1.0\n v_mov_b32 v34, -1.0\n v_mov_b32 v35, -1.0\n"
#define OVER_MFMA "v_mfma_f32_32x32x2_f32 v[32:47], %[x], %[y], v[32:47]\n s_nop 15\n s_nop 3\n"
#define OVER_EXP "v_exp_f32 v32, %[x]\n v_exp_f32 v33, %[x]\n v_exp_f32 v34, %[x]\n v_exp_f32 v35, %[x]\n"
#define OVER_LOAD "buffer_load_dwordx4 v[32:35], %[voff], %[ro], 0 offen\n"
#define WAR_ASM_GLOBAL(NOPSTR, OVER)                                                                      \
        asm volatile("v_mov_b32 v32, %[a]\n v_add_f32 v33, 1.0, %[a]\n v_add_f32 v34, 2.0, %[a]\n v_add_f32 v35, 3.0, %[a]\n" \
                     "s_nop 7\n"                                                                              \
                     "global_store_dwordx4 %[gp], v[32:35], off\n" NOPSTR OVER                                   \
                     "s_waitcnt vmcnt(0)\n"                                                                   \
                     :                                                                                        \
                     : [a] "v"(a), [gp] "v"(gp)                                                               \
                     : "v32", "v33", "v34", "v35", "memory")
        if (MODE == 4) {   // a plain global store (what `*p = v` compiles to), 64-bit address in VGPRs
            float* gp = out + ((((size_t)blockIdx.x * 8 + w) * iters + it) * 64 + lane) * 4;
            if (NOPS == 0) WAR_ASM_GLOBAL(NOP_0, OVER_MOV); else if (NOPS == 1) WAR_ASM_GLOBAL(NOP_1, OVER_MOV); else WAR_ASM_GLOBAL(NOP_2, OVER_MOV);
            continue;
        }
        if (IMM) {
            if (MODE == 0) { if (NOPS == 0) WAR_ASM_IMM(NOP_0, OVER_MOV); else if (NOPS == 1) WAR_ASM_IMM(NOP_1, OVER_MOV); else WAR_ASM_IMM(NOP_2, OVER_MOV); }
            if (MODE == 1) { if (NOPS == 0) WAR_ASM_IMM(NOP_0, OVER_MFMA); else if (NOPS == 1) WAR_ASM_IMM(NOP_1, OVER_MFMA); else WAR_ASM_IMM(NOP_2, OVER_MFMA); }
            continue;
        }
        if (MODE == 0) { if (NOPS == 0) WAR_ASM(NOP_0, OVER_MOV); else if (NOPS == 1) WAR_ASM(NOP_1, OVER_MOV); else if (NOPS == 2) WAR_ASM(NOP_2, OVER_MOV); else WAR_ASM(NOP_4, OVER_MOV); }
        if (MODE == 1) { if (NOPS == 0) WAR_ASM(NOP_0, OVER_MFMA); else if (NOPS == 1) WAR_ASM(NOP_1, OVER_MFMA); else if (NOPS == 2) WAR_ASM(NOP_2, OVER_MFMA); else WAR_ASM(NOP_4, OVER_MFMA); }
        if (MODE == 2) { if (NOPS == 0) WAR_ASM(NOP_0, OVER_EXP); else if (NOPS == 1) WAR_ASM(NOP_1, OVER_EXP); else if (NOPS == 2) WAR_ASM(NOP_2, OVER_EXP); else WAR_ASM(NOP_4, OVER_EXP); }
        if (MODE == 3) { if (NOPS == 0) WAR_ASM(NOP_0, OVER_LOAD); else if (NOPS == 1) WAR_ASM(NOP_1, OVER_LOAD); else if (NOPS == 2) WAR_ASM(NOP_2, OVER_LOAD); else WAR_ASM(NOP_4, OVER_LOAD); }
    }
}

template <int WAVES, bool V, bool S, bool G = true>
static void run_xchg(const char* what, float* scratch, size_t scratch_bytes, int launches, unsigned* dbad, unsigned* dfirst) {
    const int steps = 13;
    const int wgs = (int)(scratch_bytes / ((size_t)steps * WAVES * 4096));
    const int grid = wgs < 4096 ? wgs : 4096;
    HIP_OK(hipMemset(dbad, 0, 8));
    HIP_OK(hipMemset(dfirst, 0, 32));
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((xchg<WAVES, V, S, G>), dim3(grid), dim3(WAVES * 64), 0, 0, scratch, steps, l + 1, dbad, dfirst);
    HIP_OK(hipDeviceSynchronize());
    unsigned bad[2], first[8];
    HIP_OK(hipMemcpy(bad, dbad, 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(first, dfirst, 32, hipMemcpyDeviceToHost));
    const double checked = (double)launches * grid * steps * WAVES * WAVES * 64 * 16;
    printf("A  %-58s %4d workgroups x %d launches: %u mismatches of %.3g values", what, grid, launches, bad[0], checked);
    if (bad[0]) printf("  first: workgroup %u step %u (reader wave, writer wave, aa, i) = %04u lane %u", first[1], first[2], first[3], first[4]);
    printf("\n");
}

template <int MODE, int NOPS, bool IMM = false>
static void run_war(const char* what, float* out, const float* other, int launches) {
    const int wgs = 512, iters = 64;
    std::vector<float> host((size_t)wgs * 8 * iters * 256);
    unsigned long long bad = 0, lanes_bad[64] = {0};
    for (int l = 0; l < launches; ++l) {
        HIP_OK(hipMemset(out, 0xff, host.size() * 4));
        hipLaunchKernelGGL((war<MODE, NOPS, IMM>), dim3(wgs), dim3(512), 0, 0, out, other, iters);
        HIP_OK(hipDeviceSynchronize());
        HIP_OK(hipMemcpy(host.data(), out, host.size() * 4, hipMemcpyDeviceToHost));
        for (int b = 0; b < wgs; ++b)
            for (int w = 0; w < 8; ++w)
                for (int it = 0; it < iters; ++it)
                    for (int lane = 0; lane < 64; ++lane) {
                        const float a = (float)(lane * 4) + (float)(it % 1021) * 256.0f + (float)(w + 8 * (b % 61)) * 262144.0f;
                        const float* p = &host[((((size_t)b * 8 + w) * iters + it) * 64 + lane) * 4];
                        for (int i = 0; i < 4; ++i)
                            if (p[i] != a + (float)i) { ++bad; ++lanes_bad[lane]; }
                    }
    }
    printf("B  %-58s %d wait states: %llu wrong dwords of %.3g", what, NOPS, bad, (double)launches * host.size());
    if (bad) {
        printf("  lanes:");
        for (int l = 0; l < 64; ++l) if (lanes_bad[l]) printf(" %d", l);
    }
    printf("\n");
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 20;
    const size_t scratch_bytes = (size_t)1 << 30;
    float *scratch, *out, *other;
    unsigned *dbad, *dfirst;
    HIP_OK(hipMalloc(&scratch, scratch_bytes));
    HIP_OK(hipMemset(scratch, 0, scratch_bytes));
    HIP_OK(hipMalloc(&out, (size_t)512 * 8 * 64 * 1024));
    HIP_OK(hipMalloc(&other, 4096));
    HIP_OK(hipMemset(other, 0, 4096));
    HIP_OK(hipMalloc(&dbad, 8));
    HIP_OK(hipMalloc(&dfirst, 32));
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s (%s), %d CUs\n", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    run_xchg<8, false, false, false>("8-wave, barrier only, stores AS HIPCC EMITS THEM (hazard B inside)", scratch, scratch_bytes, launches, dbad, dfirst);
    run_xchg<8, true, false, false>("8-wave, vmcnt(0) + barrier, stores as hipcc emits them", scratch, scratch_bytes, launches, dbad, dfirst);
    run_xchg<8, false, false>("8-wave workgroups, barrier only (the shipped exchange)", scratch, scratch_bytes, launches, dbad, dfirst);
    run_xchg<8, true, false>("8-wave workgroups, s_waitcnt vmcnt(0) + barrier", scratch, scratch_bytes, launches, dbad, dfirst);
    run_xchg<4, false, false>("4-wave workgroups (several per CU), barrier only", scratch, scratch_bytes, launches, dbad, dfirst);
    run_xchg<4, true, false>("4-wave workgroups, s_waitcnt vmcnt(0) + barrier", scratch, scratch_bytes, launches, dbad, dfirst);
    run_xchg<8, false, true>("8-wave, destination rows pre-loaded into L1, barrier only", scratch, scratch_bytes, launches, dbad, dfirst);
    run_xchg<4, false, true>("4-wave, destination rows pre-loaded into L1, barrier only", scratch, scratch_bytes, launches, dbad, dfirst);
    run_xchg<4, true, true>("4-wave, destination rows pre-loaded, vmcnt(0) + barrier", scratch, scratch_bytes, launches, dbad, dfirst);
    const int wl = launches < 4 ? launches : 4;
    run_war<0, 0>("store x4, then v_mov over the data registers", out, other, wl);
    run_war<0, 1>("store x4, then v_mov over the data registers", out, other, wl);
    run_war<0, 2>("store x4, then v_mov over the data registers", out, other, wl);
    run_war<1, 0>("store x4, then v_mfma_f32_32x32x2 over the data registers", out, other, wl);
    run_war<1, 1>("store x4, then v_mfma_f32_32x32x2 over the data registers", out, other, wl);
    run_war<1, 2>("store x4, then v_mfma_f32_32x32x2 over the data registers", out, other, wl);
    run_war<1, 4>("store x4, then v_mfma_f32_32x32x2 over the data registers", out, other, wl);
    run_war<2, 0>("store x4, then v_exp_f32 over the data registers", out, other, wl);
    run_war<2, 2>("store x4, then v_exp_f32 over the data registers", out, other, wl);
    run_war<3, 0>("store x4, then buffer_load_dwordx4 into the data registers", out, other, wl);
    run_war<4, 0>("global_store_dwordx4 (vaddr), then v_mov over the data registers", out, other, wl);
    run_war<4, 1>("global_store_dwordx4 (vaddr), then v_mov over the data registers", out, other, wl);
    run_war<4, 2>("global_store_dwordx4 (vaddr), then v_mov over the data registers", out, other, wl);
    run_war<0, 0, true>("IMMEDIATE soffset (documented hazard): store x4, then v_mov", out, other, wl);
    run_war<0, 1, true>("IMMEDIATE soffset (documented hazard): store x4, then v_mov", out, other, wl);
    run_war<0, 2, true>("IMMEDIATE soffset (documented hazard): store x4, then v_mov", out, other, wl);
    run_war<1, 0, true>("IMMEDIATE soffset (documented hazard): store x4, then v_mfma", out, other, wl);
    run_war<1, 1, true>("IMMEDIATE soffset (documented hazard): store x4, then v_mfma", out, other, wl);
    return 0;
}
