// hip_emu.cpp -- the scheduler of the test-suite's SIMT interpreter (see hip/hip_runtime.h next to this file).
// TEST INFRASTRUCTURE: runs the kernels of csrc/dsp_kernels.hip on the host, one fiber per GPU thread.
//
// A launch creates grid x block lanes (fibers on private stacks carved out of one lazily committed mapping), groups them into
// waves of 64 and runs the waves round robin, a slice of cross-lane instructions at a time:
//   * a lane runs until it reaches an instruction that involves other lanes (or ends);
//   * when every live lane of the wave is parked, the wave-level instruction the lowest parked lane waits for is carried out
//     for all lanes parked at that kind of instruction (an MFMA or a shuffle wants the whole wave: anything else is a bug in the
//     kernel or here and aborts; a readfirstlane inside a divergent branch takes the first ACTIVE lane, as the exec mask does);
//   * lanes parked at the workgroup barrier wait until every live wave of the workgroup is there (waves that ended are
//     forgotten, as s_barrier does);
//   * s_sleep ends the wave's slice: that is where the clustered launches' members poll for each other.
// DSP_EMU_SEED=n shuffles the order of the waves every round (an adversarial schedule for the cluster protocol).
// A full round in which no lane ran and nothing was released is a deadlock: the state of every wave is printed, abort().
#include "hip/hip_runtime.h"

#include <sys/mman.h>
#include <time.h>

#include <algorithm>
#include <vector>

extern "C" void emu_swap(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_swap
.type emu_swap,@function
emu_swap:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_swap, .-emu_swap
)");

// what the interpreter has carried out since the last call (tests: the MFMAs a forward ISSUES against the flops it is credited with)
static unsigned long long g_total[8] = {0, 0, 0, 0, 0, 0, 0, 0};
extern "C" void emu_stats(unsigned long long out[4]) {
    out[0] = g_total[1]; out[1] = g_total[2]; out[2] = g_total[0]; out[3] = g_total[3];   // fp32 MFMAs (32x32x2), k16 MFMAs (32x32x16), launches, readfirstlanes
    for (auto& t : g_total) t = 0;
}

extern "C" int emu_compute_units(void) {
    const char* v = getenv("EMU_CUS");
    const int n = v ? atoi(v) : 256;
    return n >= 16 ? n : 256;
}

namespace emu {

LaneCtx* g_cur = nullptr;

namespace {

enum Op { NONE = 0, MFMA_F32, MFMA_K16, RFL, SHFL, BARRIER, YIELD, BALLOT };

struct Lane {
    LaneCtx ctx;           // (first member: g_cur points here)
    void* sp = nullptr;
    bool done = false;
    int op = NONE;
    // operands of the parked instruction
    float a = 0, b = 0;
    const float* a8 = nullptr; const float* b8 = nullptr;
    emu_f32x16* acc = nullptr;
    uint64_t u = 0; int kind = 0, arg = 0, width = 64;
};

struct Wave { int first = 0, n = 0, wg = 0; bool finished = false, at_barrier = false; };
struct Workgroup { int first_wave = 0, n_waves = 0; std::vector<float> lds; };

constexpr size_t kStack = 96 * 1024;
void* g_sched_sp = nullptr;
std::vector<Lane> g_lanes;
const std::function<void()>* g_body = nullptr;
unsigned long long g_tick = 0;
unsigned long long g_ops[8] = {0, 0, 0, 0, 0, 0, 0, 0}, g_lane_runs = 0;
char* g_stacks = nullptr;
size_t g_stacks_bytes = 0;

void park() { Lane* l = (Lane*)g_cur; emu_swap(&l->sp, g_sched_sp); }

void fiber_main() {
    (*g_body)();
    Lane* l = (Lane*)g_cur;
    l->done = true;
    emu_swap(&l->sp, g_sched_sp);
    abort();   // (a finished lane is never resumed)
}

void run_lane(Lane& l) {
    g_cur = &l.ctx;
    emu_swap(&g_sched_sp, l.sp);
    g_cur = nullptr;
}

[[noreturn]] void die(const char* what, const std::vector<Wave>& waves) {
    fprintf(stderr, "hip_emu: %s\n", what);
    int shown = 0;
    for (size_t w = 0; w < waves.size() && shown < 40; ++w) {
        if (waves[w].finished) continue;
        int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dn = 0;
        for (int i = 0; i < waves[w].n; ++i) { const Lane& l = g_lanes[(size_t)(waves[w].first + i)]; if (l.done) ++dn; else ++cnt[l.op]; }
        fprintf(stderr, "  wave %zu (workgroup %d): done %d, running %d, mfma %d/%d, readfirstlane %d, shuffle %d, ballot %d, barrier %d, yield %d\n", w, waves[w].wg, dn,
                cnt[NONE], cnt[MFMA_F32], cnt[MFMA_K16], cnt[RFL], cnt[SHFL], cnt[BALLOT], cnt[BARRIER], cnt[YIELD]);
        ++shown;
    }
    abort();
}

// carry out the wave-level instruction of the lanes in `grp` (indices into g_lanes), all parked at the same kind
void execute(const std::vector<int>& grp, int op, int wave_n, const std::vector<Wave>& waves) {
    if (op == RFL) {
        const uint64_t v = g_lanes[(size_t)grp[0]].u;     // the first ACTIVE lane
        for (int i : grp) g_lanes[(size_t)i].u = v;
        return;
    }
    if (op == BALLOT) {                                   // over the lanes that are here: the exec mask
        const int wave0 = grp[0] - g_lanes[(size_t)grp[0]].ctx.lane;
        uint64_t m = 0;
        for (int i : grp) if (g_lanes[(size_t)i].u) m |= 1ull << (i - wave0);
        for (int i : grp) g_lanes[(size_t)i].u = m;
        return;
    }
    if (op == SHFL) {
        // (sub-wave groups in divergent control flow -- the 8-lane groups of csrc/dsp_extract.hip -- shuffle among the lanes that are
        // here; a source lane that is not gives the lane its own value back, where the hardware's answer is undefined)
        const int wave0 = grp[0] - g_lanes[(size_t)grp[0]].ctx.lane;
        uint64_t in[64]; bool here[64];
        for (int l = 0; l < 64; ++l) here[l] = false;
        for (int i : grp) { in[i - wave0] = g_lanes[(size_t)i].u; here[i - wave0] = true; }
        for (int i : grp) {
            Lane& L = g_lanes[(size_t)i];
            const int l = i - wave0, w = L.width > 0 && L.width <= 64 ? L.width : 64, g0 = l & ~(w - 1);
            int src = l;
            if (L.kind == 0) src = l ^ L.arg;
            else if (L.kind == 1) src = (l - g0) >= L.arg ? l - L.arg : l;
            else src = g0 + (L.arg & (w - 1));
            if (src >= 0 && src < 64 && (src & ~(w - 1)) == g0 && here[src]) L.u = in[src];
        }
        return;
    }
    if ((int)grp.size() != wave_n || wave_n != 64) die("an MFMA reached by a part of the wave only (or a wave that is not 64 lanes)", waves);
    const int base = grp[0];
    if (op == MFMA_F32) {
        // D[row][j] = C[row][j] + A[row][0] B[0][j] + A[row][1] B[1][j]; lane l holds column j = l % 32 of the 16 rows
        // 8 (r / 4) + 4 (l / 32) + r % 4: the rows' A values as two 16-vectors per half-wave, two vector FMAs per lane
        float A[32][2], B[2][32];
        for (int l = 0; l < 64; ++l) { const Lane& L = g_lanes[(size_t)(base + l)]; A[l & 31][l >> 5] = L.a; B[l >> 5][l & 31] = L.b; }
        emu_f32x16 ra[2][2];
        for (int half = 0; half < 2; ++half)
            for (int r = 0; r < 16; ++r) {
                const int row = 8 * (r / 4) + 4 * half + r % 4;
                ra[half][0][r] = A[row][0]; ra[half][1][r] = A[row][1];
            }
        for (int l = 0; l < 64; ++l) {
            emu_f32x16& c = *g_lanes[(size_t)(base + l)].acc;
            const int j = l & 31, half = l >> 5;
            c = __builtin_elementwise_fma(ra[half][1], (emu_f32x16)(B[1][j]), __builtin_elementwise_fma(ra[half][0], (emu_f32x16)(B[0][j]), c));
        }
        return;
    }
    // MFMA_K16: A[i][k], k = 8 (lane / 32) + e, held by lane i + 32 (k / 8) as element k % 8; B alike
    static float A[32][16], B[16][32];
    for (int l = 0; l < 64; ++l) {
        const Lane& L = g_lanes[(size_t)(base + l)];
        for (int e = 0; e < 8; ++e) { A[l & 31][8 * (l >> 5) + e] = L.a8[e]; B[8 * (l >> 5) + e][l & 31] = L.b8[e]; }
    }
    emu_f32x16 rk[2][16];
    for (int half = 0; half < 2; ++half)
        for (int k = 0; k < 16; ++k)
            for (int r = 0; r < 16; ++r) rk[half][k][r] = A[8 * (r / 4) + 4 * half + r % 4][k];
    for (int l = 0; l < 64; ++l) {
        emu_f32x16 c = *g_lanes[(size_t)(base + l)].acc;
        const int j = l & 31, half = l >> 5;
        for (int k = 0; k < 16; ++k) c = __builtin_elementwise_fma(rk[half][k], (emu_f32x16)(B[k][j]), c);
        *g_lanes[(size_t)(base + l)].acc = c;
    }
}

}  // namespace

uint32_t readfirstlane_u32(uint32_t v) { Lane* l = (Lane*)g_cur; l->u = v; l->op = RFL; park(); return (uint32_t)l->u; }
uint64_t shfl_u64(uint64_t v, int kind, int arg, int width) { Lane* l = (Lane*)g_cur; l->u = v; l->kind = kind; l->arg = arg; l->width = width; l->op = SHFL; park(); return l->u; }
uint64_t ballot(bool pred) { Lane* l = (Lane*)g_cur; l->u = pred ? 1 : 0; l->op = BALLOT; park(); return l->u; }
void mfma_f32(float a, float b, emu_f32x16* acc) { Lane* l = (Lane*)g_cur; l->a = a; l->b = b; l->acc = acc; l->op = MFMA_F32; park(); }
void mfma_k16(const float a[8], const float b[8], emu_f32x16* acc) { Lane* l = (Lane*)g_cur; l->a8 = a; l->b8 = b; l->acc = acc; l->op = MFMA_K16; park(); }
void barrier() { Lane* l = (Lane*)g_cur; l->op = BARRIER; park(); }
void yield() { Lane* l = (Lane*)g_cur; l->op = YIELD; park(); }
unsigned long long now() { return g_tick; }

static void run_workgroups(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body, size_t wg_begin, size_t nwg);

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body, bool sequential) {
    if (g_cur) { fprintf(stderr, "hip_emu: a launch from inside a kernel\n"); abort(); }
    const size_t all = (size_t)grid.x * grid.y * grid.z;
    if (!sequential) { run_workgroups(grid, block, lds_bytes, body, 0, all); return; }
    for (size_t g = 0; g < all; ++g) run_workgroups(grid, block, lds_bytes, body, g, 1);   // one workgroup at a time
}

static void run_workgroups(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body, size_t wg_begin, size_t nwg) {
    const size_t nthr = (size_t)block.x * block.y * block.z, total = nwg * nthr;
    if (total == 0) return;
    if (g_stacks_bytes < total * kStack) {
        if (g_stacks) munmap(g_stacks, g_stacks_bytes);
        g_stacks_bytes = total * kStack;
        g_stacks = (char*)mmap(nullptr, g_stacks_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == MAP_FAILED) { perror("hip_emu: mmap of the lanes' stacks"); abort(); }
    }
    g_body = &body;
    g_lanes.assign(total, Lane());
    std::vector<Workgroup> wgs(nwg);
    std::vector<Wave> waves;
    const int wpw = (int)((nthr + 63) / 64);
    for (size_t g = 0; g < nwg; ++g) {
        // LDS is not zeroed either: quiet NaNs.  Exactly the bytes the launch asked for (AddressSanitizer then sees an overrun;
        // a kernel that asks for none gets a pointer it must not follow)
        const uint32_t poison = 0x7fc0dead;
        float pf; memcpy(&pf, &poison, 4);
        wgs[g].lds.assign((lds_bytes + 3) / 4, pf);
        wgs[g].first_wave = (int)waves.size();
        wgs[g].n_waves = wpw;
        for (int w = 0; w < wpw; ++w) {
            Wave W;
            W.first = (int)(g * nthr) + w * 64;
            W.n = (int)std::min<size_t>(64, nthr - (size_t)w * 64);
            W.wg = (int)g;
            waves.push_back(W);
        }
        for (size_t t = 0; t < nthr; ++t) {
            Lane& l = g_lanes[g * nthr + t];
            l.ctx.tid = dim3((unsigned)(t % block.x), (unsigned)(t / block.x % block.y), (unsigned)(t / ((size_t)block.x * block.y)));
            const size_t gg = wg_begin + g;   // (g: the workgroup's place among the ones running now; gg: its index in the grid)
            l.ctx.bid = dim3((unsigned)(gg % grid.x), (unsigned)(gg / grid.x % grid.y), (unsigned)(gg / ((size_t)grid.x * grid.y)));
            l.ctx.bdim = block; l.ctx.gdim = grid;
            l.ctx.lds = wgs[g].lds.data();
            l.ctx.lane = (int)(t & 63);
            // a fresh stack: six callee-saved registers and the entry point, aligned as at a call
            // (staggered: 64 lanes whose frames sit at the same offset of equally spaced stacks would fight over the same cache sets)
            char* top = g_stacks + (g * nthr + t + 1) * kStack - (((g * nthr + t) * 37) % 61) * 192;
            void** sp = (void**)(((uintptr_t)top & ~(uintptr_t)15) - 8);
            *--sp = (void*)&fiber_main;
            for (int i = 0; i < 6; ++i) *--sp = nullptr;
            l.sp = sp;
        }
    }
    std::vector<int> order(waves.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    uint64_t seed = 0;
    if (const char* v = getenv("DSP_EMU_SEED")) seed = strtoull(v, nullptr, 10) * 0x9E3779B97F4A7C15ull + 1;
    const int slice = 64;
    size_t live_waves = waves.size();
    std::vector<int> grp;
    while (live_waves > 0) {
        bool progress = false;
        if (seed) for (size_t i = order.size(); i > 1; --i) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; std::swap(order[i - 1], order[(seed >> 33) % i]); }
        for (int wi : order) {
            Wave& W = waves[(size_t)wi];
            if (W.finished) continue;
            if (W.at_barrier) {   // released when every live wave of the workgroup is at the barrier
                const Workgroup& G = wgs[(size_t)W.wg];
                bool all = true;
                for (int k = 0; k < G.n_waves && all; ++k) { const Wave& O = waves[(size_t)(G.first_wave + k)]; all = O.finished || O.at_barrier; }
                if (!all) continue;
                for (int k = 0; k < G.n_waves; ++k) {
                    Wave& O = waves[(size_t)(G.first_wave + k)];
                    if (O.finished) continue;
                    O.at_barrier = false;
                    for (int i = 0; i < O.n; ++i) { Lane& l = g_lanes[(size_t)(O.first + i)]; if (!l.done && l.op == BARRIER) l.op = NONE; }
                }
                progress = true;
            }
            for (int ops = 0; ops < slice; ++ops) {
                // one pass: run every lane that can run until it parks (or ends), and take stock of where the wave stands
                int live = 0, first_wave_op = -1, n_first = 0, n_yield = 0, n_bar = 0;
                for (int i = 0; i < W.n; ++i) {
                    Lane& l = g_lanes[(size_t)(W.first + i)];
                    if (l.done) continue;
                    if (l.op == NONE) { run_lane(l); ++g_lane_runs; progress = true; if (l.done) continue; }
                    ++live;
                    if (l.op == YIELD) ++n_yield;
                    else if (l.op == BARRIER) ++n_bar;
                    else if (first_wave_op < 0) { first_wave_op = l.op; n_first = 1; }
                    else if (l.op == first_wave_op) ++n_first;
                }
                ++g_tick;
                if (live == 0) { W.finished = true; --live_waves; break; }
                if (first_wave_op >= 0) {
                    grp.clear();
                    if (n_first == W.n) { for (int i = 0; i < W.n; ++i) grp.push_back(W.first + i); }
                    else for (int i = 0; i < W.n; ++i) { const Lane& l = g_lanes[(size_t)(W.first + i)]; if (!l.done && l.op == first_wave_op) grp.push_back(W.first + i); }
                    ++g_ops[first_wave_op];
                    ++g_total[first_wave_op];
                    execute(grp, first_wave_op, W.n, waves);
                    for (int i : grp) g_lanes[(size_t)i].op = NONE;
                    continue;
                }
                if (n_yield) {   // s_sleep: the wave gives up the rest of its slice
                    for (int i = 0; i < W.n; ++i) { Lane& l = g_lanes[(size_t)(W.first + i)]; if (!l.done && l.op == YIELD) l.op = NONE; }
                    break;
                }
                W.at_barrier = n_bar == live;
                break;
            }
        }
        if (!progress) die("no lane could run and no barrier could be released: deadlock", waves);
    }
    g_body = nullptr;
    ++g_total[0];
    if (getenv("DSP_EMU_STATS")) {   // what the launch cost the interpreter
        static struct timespec last; struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
        fprintf(stderr, "hip_emu: launch of %zu x %zu lanes: lane runs %llu, mfma %llu + %llu, readfirstlane %llu, shuffle %llu; %.3f s since the previous launch ended\n", nwg, nthr,
                g_lane_runs, g_ops[MFMA_F32], g_ops[MFMA_K16], g_ops[RFL], g_ops[SHFL], last.tv_sec ? (double)(t.tv_sec - last.tv_sec) + 1e-9 * (double)(t.tv_nsec - last.tv_nsec) : 0.0);
        last = t; g_lane_runs = 0; for (auto& o : g_ops) o = 0;
    }
}

}  // namespace emu
