// dsp_capi.cpp -- host side of the C ABI declared in include/dsp_amd.h.
//
// Replaces (reference, pure Python): model construction + checkpoint load + .cuda() + .eval()
// (deepsignal_plant/call_modifications.py:214-228) and the forward call (:159-163 -> models.py:178-240).
// One-time work here: validate the state_dict, repack every weight matrix into MFMA A-fragment order
// (transposed formulation, see dsp_kernels.hip), pre-sum b_ih + b_hh, upload once.  Per call: nine
// kernel launches on the caller's stream, no allocation, no host synchronisation.
#include "dsp_amd.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "dsp_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(DSP_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline int rup(int x, int m) { return (x + m - 1) / m * m; }
// hidden sizes are zero-padded to whole unit tiles: 32 units, pairs of them (64) above 32, and multiples of 8 tiles (256)
// above 256, where every wave of the LSTM kernel computes one unit tile per pass and hidden / 256 passes per step
// (dsp_kernels.hip, NP = 0)
inline int pad_hidden(int h) { return h <= 32 ? 32 : (h <= 256 ? rup(h, 64) : rup(h, 256)); }
// one workgroup (at most 8 waves: two per SIMD, 256 registers each) holds a direction's whole hidden state, which its
// waves exchange every step through a workgroup barrier: 8 waves x NP passes x 32 units.  Up to 256 (one pass) the cell
// state sits in LDS; above, in a global scratch (round 3).  2,048 keeps every buffer offset of the kernels far inside 32
// bits; nothing else limits it.
constexpr int kMaxHidden = 2048;

struct Dims {
    int T, S, H, C, V, E, l1, l2;
    int hseq, hsig, Iseq;
    bool is_base, is_siglen;
};

int derive(const dsp_model_cfg* c, Dims* d) {
    if (!c) return fail(DSP_EINVAL, "cfg is NULL");
    if (c->module < 0 || c->module > 2) return fail(DSP_EINVAL, "--model_type is not right!");  // models.py:127-128
    if (c->seq_len < 1 || c->signal_len < 1 || c->num_layers1 < 1 || c->num_layers2 < 1 || c->num_classes < 1 ||
        c->hidden_size < 2 || c->vocab_size < 1 || c->embedding_size < 1)
        return fail(DSP_EINVAL, "non-positive model dimension");
    if (c->hidden_size > kMaxHidden)
        return fail(DSP_EINVAL, "hidden_size %d > %d is not supported by this build (one workgroup of 8 waves holds a "
                                "direction's whole hidden state, 256 units per pass)", c->hidden_size, kMaxHidden);
    if (c->num_classes > 64) return fail(DSP_EINVAL, "num_classes %d > 64 is not supported", c->num_classes);
    if (c->num_layers1 > 15 || c->num_layers2 > 15) return fail(DSP_EINVAL, "too many LSTM layers");
    d->T = c->seq_len; d->S = c->signal_len; d->H = c->hidden_size; d->C = c->num_classes;
    d->V = c->vocab_size; d->E = c->embedding_size; d->l1 = c->num_layers1; d->l2 = c->num_layers2;
    d->is_base = c->is_base != 0; d->is_siglen = c->is_signallen != 0;
    d->hseq = d->hsig = 0;
    if (c->module == DSP_MODULE_BOTH) { d->hseq = d->H / 2; d->hsig = d->H - d->hseq; }
    else if (c->module == DSP_MODULE_SEQ) d->hseq = d->H;
    else d->hsig = d->H;
    d->Iseq = (d->is_base ? d->E : 0) + (d->is_siglen ? 3 : 2);  // models.py:135-141
    return 0;
}

struct Spec { std::string name; int64_t shape[2]; int ndim; };

void lstm_spec(std::vector<Spec>& v, const char* prefix, int in, int hid, int layers) {
    for (int k = 0; k < layers; ++k) {
        const int isz = k == 0 ? in : 2 * hid;
        for (int d = 0; d < 2; ++d) {
            const char* suf = d ? "_reverse" : "";
            char nm[96];
            snprintf(nm, sizeof(nm), "%s.weight_ih_l%d%s", prefix, k, suf); v.push_back({nm, {4 * hid, isz}, 2});
            snprintf(nm, sizeof(nm), "%s.weight_hh_l%d%s", prefix, k, suf); v.push_back({nm, {4 * hid, hid}, 2});
            snprintf(nm, sizeof(nm), "%s.bias_ih_l%d%s", prefix, k, suf); v.push_back({nm, {4 * hid, 0}, 1});
            snprintf(nm, sizeof(nm), "%s.bias_hh_l%d%s", prefix, k, suf); v.push_back({nm, {4 * hid, 0}, 1});
        }
    }
}

// state_dict order of ModelBiLSTM (models.py:130-161)
std::vector<Spec> weight_spec(const Dims& d) {
    std::vector<Spec> v;
    if (d.hseq) {
        v.push_back({"embed.weight", {d.V, d.E}, 2});
        lstm_spec(v, "lstm_seq", d.Iseq, d.hseq, d.l2);
        v.push_back({"fc_seq.weight", {d.hseq, 2 * d.hseq}, 2});
        v.push_back({"fc_seq.bias", {d.hseq, 0}, 1});
    }
    if (d.hsig) {
        lstm_spec(v, "lstm_signal", d.S, d.hsig, d.l2);
        v.push_back({"fc_signal.weight", {d.hsig, 2 * d.hsig}, 2});
        v.push_back({"fc_signal.bias", {d.hsig, 0}, 1});
    }
    lstm_spec(v, "lstm_comb", d.H, d.H, d.l1);
    v.push_back({"fc1.weight", {d.H, 2 * d.H}, 2});
    v.push_back({"fc1.bias", {d.H, 0}, 1});
    v.push_back({"fc2.weight", {d.C, d.H}, 2});
    v.push_back({"fc2.bias", {d.C, 0}, 1});
    return v;
}

int64_t spec_numel(const Spec& s) { return s.ndim == 2 ? s.shape[0] * s.shape[1] : s.shape[0]; }

// ---- feature maps: padded K4 feature index -> column of the reference weight matrix (or -1 = zero pad)
std::vector<int> map_pad(int n, int padded, int off = 0) {
    std::vector<int> m(padded, -1);
    for (int i = 0; i < n; ++i) m[off + i] = i;
    return m;
}
// A front-end input of I features in a 32-wide block: the LSTM kernel wants the k-groups that carry data to be the LAST
// of the block (dsp_lstm_kernel<2, 1, XL>), so the features start at 32 - roundup(I, 8).  Other shapes: at the front.
int front_end_offset(int I, int F, int Hp) { return (F == 32 && rup(I, 8) < 32 && Hp <= 256) ? 32 - rup(I, 8) : 0; }
std::vector<int> map_bidir(int H, int Hp) {  // [fwd Hp | bwd Hp] -> [fwd H | bwd H]
    std::vector<int> m(2 * Hp, -1);
    for (int i = 0; i < H; ++i) { m[i] = i; m[Hp + i] = H + i; }
    return m;
}

struct DevLstmLayer {
    int Ipad, Ilo, Iused, H, Hp;
    float* wpk[2]; float* sbias[2];
    float* wsplit[2]; float* wsplit16[2];   // see pack_lstm_dir_split; built at the first switch to a split precision
    // what ensure_split needs (released once the pieces are on the device): the reference-layout weights and the input map
    int split_kinds = 0;                     // 0: no split form of this layer, 1: bf16 pieces, 2: + fp16 pieces
    int I = 0;
    std::vector<int> in_map;
    std::vector<float> host_wih[2], host_whh[2];
};
struct DevLinear { int Fin, ORT; float* wpk; float* bias; };
// what the launch geometry of an LSTM layer depends on (no weights): derive_geometry fills these for every layer of the
// three stacks, so that the forward's plan -- which kernel form a batch size takes, what a piece costs -- can be made, and
// tested, without a device (dsp_debug_plan)
struct LayerShape { int Ipad, Ilo, Iused, H, Hp; };
constexpr int kWsRegions = 9;   // regions of a handle's workspace (ws_layout)

// A fragments for gates^T = W * act^T :  [UT][NQ][4 gates][64 lanes][4]
//   value = Wcat[g*H + u*32 + (lane&31)][8q + 4*(lane>>5) + i],  Wcat = [W_ih(in_map) | W_hh]
void pack_lstm_dir(const float* wih, const float* whh, const float* bih, const float* bhh, int I, int H, int Hp,
                   const std::vector<int>& in_map, std::vector<float>& wpk, std::vector<float>& bias) {
    const int Ipad = (int)in_map.size();
    const int UT = Hp / 32, NQ = rup((Ipad + Hp) / 8, 4);  // padded k-groups keep zero weights
    wpk.assign((size_t)UT * NQ * 4 * 64 * 4, 0.f);
    bias.assign((size_t)4 * Hp, 0.f);
    auto pack_tile = [&](int u) {   // unit tiles are disjoint pieces of wpk: one thread each
        for (int q = 0; q < (Ipad + Hp) / 8; ++q)
            for (int g = 0; g < 4; ++g)
                for (int lane = 0; lane < 64; ++lane) {
                    const int unit = u * 32 + (lane & 31);
                    if (unit >= H) continue;
                    const size_t row = (size_t)g * H + unit;
                    float* dst = &wpk[((((size_t)u * NQ + q) * 4 + g) * 64 + lane) * 4];
                    for (int i = 0; i < 4; ++i) {
                        const int kk = 8 * q + 4 * (lane >> 5) + i;
                        if (kk < Ipad) {
                            const int col = in_map[kk];
                            if (col >= 0) dst[i] = wih[row * I + col];
                        } else {
                            const int hk = kk - Ipad;
                            if (hk < H) dst[i] = whh[row * H + hk];
                        }
                    }
                }
    };
    if (UT >= 4 && (size_t)UT * NQ >= 256) {
        std::vector<std::thread> th;
        for (int u = 1; u < UT; ++u) th.emplace_back(pack_tile, u);
        pack_tile(0);
        for (auto& t : th) t.join();
    } else {
        for (int u = 0; u < UT; ++u) pack_tile(u);
    }
    for (int g = 0; g < 4; ++g)
        for (int unit = 0; unit < H; ++unit) bias[(size_t)g * Hp + unit] = bih[g * H + unit] + bhh[g * H + unit];
}

// Split-bf16 A fragments of the same Wcat for dsp_lstm6_kernel (v_mfma_f32_32x32x16_bf16):
//   [UT][(Ipad+Hp)/16 k-stages][4 gates][3 pieces][64 lanes][8 bf16]
//   piece p of Wcat[g*H + u*32 + (lane&31)][16q + 8*(lane>>5) + j];  hi + mid + lo == the fp32 weight exactly
inline uint16_t bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    return (uint16_t)u;
}
inline float bf16_to_f32(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// fp16 = true: two fp16 pieces (hi = fp16(w), lo = fp16(w - hi)) instead of three bf16 pieces
void pack_lstm_dir_split(const float* wih, const float* whh, int I, int H, int Hp, const std::vector<int>& in_map,
                         std::vector<float>& out, bool fp16 = false) {
    const int Ipad = (int)in_map.size();
    const int UT = Hp / 32, NQ = (Ipad + Hp) / 16;
    const int NP = fp16 ? 2 : 3;
    std::vector<uint16_t> pk((size_t)UT * NQ * 4 * NP * 64 * 8, 0);
    for (int u = 0; u < UT; ++u)
        for (int q = 0; q < NQ; ++q)
            for (int g = 0; g < 4; ++g)
                for (int lane = 0; lane < 64; ++lane) {
                    const int unit = u * 32 + (lane & 31);
                    if (unit >= H) continue;
                    const size_t row = (size_t)g * H + unit;
                    for (int j = 0; j < 8; ++j) {
                        const int kk = 16 * q + 8 * (lane >> 5) + j;
                        float v = 0.f;
                        if (kk < Ipad) {
                            const int col = in_map[kk];
                            if (col >= 0) v = wih[row * I + col];
                        } else {
                            const int hk = kk - Ipad;
                            if (hk < H) v = whh[row * H + hk];
                        }
                        uint16_t piece[3];
                        if (fp16) {
                            const _Float16 h = (_Float16)v;  // round to nearest even, subnormals kept
                            const _Float16 l = (_Float16)(v - (float)h);
                            memcpy(&piece[0], &h, 2);
                            memcpy(&piece[1], &l, 2);
                        } else {
                            piece[0] = bf16_rne(v);
                            const float r1 = v - bf16_to_f32(piece[0]);
                            piece[1] = bf16_rne(r1);
                            piece[2] = bf16_rne(r1 - bf16_to_f32(piece[1]));
                        }
                        for (int p = 0; p < NP; ++p)
                            pk[((((((size_t)u * NQ + q) * 4 + g) * NP + p) * 64 + lane) * 8) + j] = piece[p];
                    }
                }
    out.resize(pk.size() / 2);
    memcpy(out.data(), pk.data(), pk.size() * 2);
}

// fp16x3 splits operands into fp16 pieces, which saturate at 65504.  The combined stack's operands are its weights,
// h in (-1, 1) and relu(fc(h)) <= max_i (sum_j |w_ij| + |b_i|): both bounds are checked against this margin.
constexpr double kFp16Safe = 3.0e4;
inline double max_abs(const float* w, size_t n) {
    double m = 0.0;
    for (size_t i = 0; i < n; ++i) { const double a = std::fabs((double)w[i]); if (!(a <= m)) m = a; }  // NaN -> kept
    return m;
}
inline double fc_out_bound(const float* w, const float* b, int O, int K) {
    double m = 0.0;
    for (int o = 0; o < O; ++o) {
        double s = std::fabs((double)b[o]);
        for (int k = 0; k < K; ++k) s += std::fabs((double)w[(size_t)o * K + k]);
        if (!(s <= m)) m = s;
    }
    return m;
}

// A fragments for out^T = W * act^T : [ORT][Fin/8][64][4]
// Opad = padded output width the kernels iterate over (row tiles Opad/32); rows O..Opad-1 keep zero weights and
// zero bias, so the padding features come out as exact zeros
void pack_linear(const float* w, const float* b, int O, int Opad, int K, const std::vector<int>& in_map,
                 std::vector<float>& wpk, std::vector<float>& bias) {
    const int Fin = (int)in_map.size();
    const int ORT = rup(Opad > O ? Opad : O, 32) / 32, NQ = Fin / 8;
    wpk.assign((size_t)ORT * NQ * 64 * 4, 0.f);
    bias.assign((size_t)ORT * 32, 0.f);
    for (int rt = 0; rt < ORT; ++rt)
        for (int q = 0; q < NQ; ++q)
            for (int lane = 0; lane < 64; ++lane) {
                const int row = rt * 32 + (lane & 31);
                if (row >= O) continue;
                float* dst = &wpk[(((size_t)rt * NQ + q) * 64 + lane) * 4];
                for (int i = 0; i < 4; ++i) {
                    const int col = in_map[8 * q + 4 * (lane >> 5) + i];
                    if (col >= 0) dst[i] = w[(size_t)row * K + col];
                }
            }
    for (int o = 0; o < O; ++o) bias[o] = b[o];
}

struct ProfEntry { const char* name; hipEvent_t a, b; int share; };   // share: launches that divide the bracket's time

}  // namespace

struct dsp_model {
    dsp_model_cfg cfg;
    Dims d;
    int device = 0;
    std::vector<void*> dev_allocs;
    float* embed = nullptr;
    std::vector<DevLstmLayer> seq, sig, comb;
    std::vector<LayerShape> seq_shape, sig_shape, comb_shape;
    DevLinear fc_seq{}, fc_sig{}, fc1{};
    float* w2 = nullptr; float* b2 = nullptr;
    int hseq_p = 0, hsig_p = 0, Hp = 0, Fseq = 0, Fsig = 0, Fcomb = 0, Fwide = 0;
    int xoff_seq = 0, xoff_sig = 0;  // first feature of the padded front-end blocks that carries data (front_end_offset)
    std::vector<int> comb_in_map;  // padded comb-input feature -> reference feature
    int trace_wave = 0;  // DSP_TRACE_WAVE: the wave of each workgroup that stamps
    int trace_launch = -1, lstm_launch_no = 0;  // DSP_TRACE_LAUNCH: index of the LSTM launch (within a forward) to stamp
    int sg_override = 0;  // DSP_LSTM_SG: site groups per LSTM workgroup (0 = default policy)
    int np8 = 1;          // DSP_LSTM_NP8: passes per step for layers of exactly 8 unit tiles (hidden 193..256): 1 = one
                          // 8-wave workgroup per 64 sites; 2 = 4-wave workgroups, two unit tiles per wave and step (A/B switch)
    bool phase_prio = true;  // s_setprio by phase in the LSTM kernel (DSP_LSTM_PRIO=0 turns it off: A/B switch)
    int tiling21 = -1;       // <2 unit tiles, 1 site tile> per wave on the dense one-pass layers (dsp_lstm21_kernel): -1 = for
                             // batches whose 32-site tiles x 2 directions fit the CUs at once (small-batch latency); DSP_LSTM_TILING=21
                             // always, =0 never (A/B switch)
    // "x ahead" (round 6, opt-in -- never timed on a GPU: DSP_LSTM_XAHEAD=1): on calls of <= xahead_tiles live site tiles the x
    // part of the clustered dense layers is summed for all T steps at once by dsp_xahead_kernel, the recurrent launch keeps one
    // ring of it (LstmArgs::xs; bit-identical results).  DSP_LSTM_XAHEAD_TILES: 1..16, default 8 (256 sites)
    int xahead = 0;
    int xahead_tiles = 8;
    int xahead_ring = 16;       // DSP_LSTM_XAHEAD_RING=8: the one-gate-per-wave form keeps 8 instead of 16 k-groups of x part (A/B switch)
    float* xacc = nullptr;      // workspace region 8: [live cluster][T][8 unit tiles][4 gates][4][64] float4
    bool wave_handoff = true;   // per-wave, deferred arrivals in the clustered launches (round 5); DSP_LSTM_HANDOFF=0: round 4's (A/B switch)
    bool small_classes = false; // the combined stack has the clustered small-batch forms (8 or 4 unit tiles, dense): batches <= 4,096 sites
                                // then cost by class (512 / 1,024 / 2,048 / 4,096) and dsp_forward plans a remainder's pieces by them
    bool forward_split = true;  // dsp_forward cuts a call into whole rounds of 8,192 sites + a small-batch remainder (round 5); DSP_FORWARD_SPLIT=0 (A/B switch)
    bool fc_small = true;    // dsp_linear1_kernel for batches <= 4,096 sites (A/B switch DSP_FC_SMALL=0)
    bool fc_fused = true;    // fc_seq + fc_signal in one launch when they have one shape (A/B switch DSP_FC_FUSED=0)
    bool local8 = true;      // dense 8-unit-tile layers of 2,049..4,096-site batches on dsp_lstmc_kernel's eight-wave workgroup-local
                             // form (two waves per SIMD: 3.43 vs 3.48 ms per forward of 4,096 sites) instead of dsp_lstm21_kernel;
                             // DSP_LSTM_LOCAL8=0 turns it off (A/B switch)
    bool head_st4 = false;   // DSP_HEAD_ST4=1: the head kernel keeps four site tiles per workgroup at every batch size (A/B switch)
    bool sync_each = false, debug_lstm = false;  // DSP_SYNC_EACH / DSP_DEBUG_LSTM: debugging aids, read when the handle is made
    bool xcc_probe_failed = false;   // dsp_model_create's XCC_ID probe did not find block b on XCD b % 8: no clustered launches
    int front_cluster = -1;  // the front ends (4 unit tiles) clustered too (round 5): -1 = whenever their grids fit the CUs at once;
                             // DSP_LSTM_FRONT_CLUSTER=0 never, =1 / 2 that many gates per wave (A/B switch)
    int cluster = -1;        // dsp_lstmc_kernel (a site tile's unit tiles spread over several CUs, small batches): -1 = whenever the
                             // whole grid fits the CUs at once; DSP_LSTM_CLUSTER=0 never, =1 / 2 / 4 that many gates per wave
    int two_streams = -1;    // the signal branch on a side stream next to the seq branch (they are independent until the combined
                             // stack): -1 = for batches that leave CUs idle (<= 4,096 sites); DSP_TWO_STREAMS=0 never, =1 always
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    float* h0buf2 = nullptr;   // the side branch's own h0 scratch
    unsigned long long cluster_timeout = 500000;  // s_memtime ticks (shader clock on gfx950: 0.2 ms) a cluster's members wait for each other to
                             // become resident before they hand the cluster to the clean-up launch; DSP_CLUSTER_TIMEOUT sets it
                             // (0: every cluster is abandoned at once -- the test of the clean-up path)
    unsigned int* cflags = nullptr;   // arrival counters of the clustered launches of ONE forward (zeroed by its pack launch)
    int n_cflag_words = 0;
    int n_cus = 256;         // compute units of the handle's device
    bool split_ready = false;  // the split-precision weight pieces are on the device (ensure_split)
    bool fp16_safe = true;         // every operand of the combined stack provably inside the fp16 range (fp16x3 only then)
    int precision = DSP_PREC_FP32; // products of the combined stack: fp32 MFMA, or split-bf16 emulation (dsp_lstm6_kernel)
    // Buffer-descriptor extents (round 6): every pointer a kernel addresses through a descriptor travels with the end of its
    // allocation.  1 (default) = the end of the workspace region / weight upload the pointer lies in; 2 = the tight logical
    // extent of THIS call (its tiles, this layer's weights: what the bounds-recording build checks against); 0 = the 2 GiB
    // windows of rounds 1-5 (DSP_RSRC_EXTENTS=wide: A/B switch and escape hatch -- the hardware range check off again)
    bool dry = false;         // dsp_debug_dry_run: a handle that never sees a device -- addresses are made up (never dereferenced),
    uintptr_t dry_next = 0x100000000ull;   // launches are checked and noted instead of made (dsp_k_set_dry)
    int extents = 1;
    size_t test_shrink = 0;   // bounds build only (DSP_BOUNDS_TEST_SHRINK): bytes taken off every LSTM launch's input extent -- the
                              // negative control of the bounds tests (an access the record must name)
    std::vector<std::pair<const char*, size_t>> uploads;   // every weight upload: base, bytes
    size_t ws_off[kWsRegions + 1] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // the workspace's regions (ws_layout) + its size
    // scratch
    void* ws = nullptr;
    int64_t ws_sites = 0;
    float *xseq = nullptr, *xsig = nullptr, *bufA = nullptr, *bufB = nullptr, *comb_in = nullptr, *h0buf = nullptr, *cbuf = nullptr;
    float* last_out = nullptr;
    bool last_split = false;   // the last dsp_forward ran as several pieces: the scratch holds the last piece only
    // profiling
    bool prof = false;
    bool prof_dominant_only = false;   // dsp_profile_enable(m, 2): only the launches of the combined stack are bracketed by events
    bool prof_serial = false;   // (reserved: per-launch timing wants the branches in sequence; DSP_TWO_STREAMS=0 gives that)
    std::vector<ProfEntry> prof_entries;
    std::vector<hipEvent_t> event_pool;
    size_t event_used = 0;
};

namespace {

// DSP_LSTM_XAHEAD / DSP_LSTM_XAHEAD_TILES (dsp_model::xahead), read when a handle -- real or dry -- is made
void read_xahead_switch(dsp_model* m) {
    if (const char* v = getenv("DSP_LSTM_XAHEAD")) m->xahead = atoi(v) != 0 ? 1 : 0;
    if (const char* v = getenv("DSP_LSTM_XAHEAD_TILES")) m->xahead_tiles = std::min(16, std::max(1, atoi(v)));
    if (const char* v = getenv("DSP_LSTM_XAHEAD_RING")) m->xahead_ring = atoi(v) == 8 ? 8 : 16;
}

// a made-up device address for a dry handle's allocation (64 KiB of nothing between neighbours)
void* dry_alloc(dsp_model* m, size_t bytes) {
    void* p = (void*)m->dry_next;
    m->dry_next += (bytes + 65535) / 65536 * 65536 + 65536;
    return p;
}

int upload(dsp_model* m, const std::vector<float>& h, float** out) {
    if (m->dry) {
        void* q = dry_alloc(m, h.size() * sizeof(float));
        m->uploads.emplace_back((const char*)q, h.size() * sizeof(float));
        *out = (float*)q;
        return 0;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, h.size() * sizeof(float));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(DSP_ENOMEM, "hipMalloc(%zu) failed: %s", h.size() * sizeof(float), hipGetErrorString(e));
    }
    m->dev_allocs.push_back(p);
    m->uploads.emplace_back((const char*)p, h.size() * sizeof(float));
    HIP_TRY(hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = (float*)p;
    return 0;
}

// The end of the allocation `p` lies in: a weight upload, or a region of the workspace (ws_layout).  NULL for a pointer this
// handle does not own -- the launch wrappers refuse a descriptor without an end, so a pointer that got lost fails its launch
// instead of reading through a window onto somebody else's memory.
const void* alloc_end(const dsp_model* m, const void* p) {
    const char* c = (const char*)p;
    if (!c) return nullptr;
    if (m->ws) {
        const char* b = (const char*)m->ws;
        if (c >= b && c < b + m->ws_off[kWsRegions])
            for (int i = 0; i < kWsRegions; ++i)
                if (c >= b + m->ws_off[i] && c < b + m->ws_off[i + 1]) return b + m->ws_off[i + 1];
    }
    for (const auto& u : m->uploads)
        if (c >= u.first && c < u.first + u.second) return u.first + u.second;
    return nullptr;
}
// ... as a descriptor's end: `logical` = the bytes this launch may touch behind p
const void* rsrc_end(const dsp_model* m, const void* p, size_t logical) {
    if (!p) return nullptr;
    if (m->extents == 0) return (const char*)p + (1ull << 40);   // clamps to the old 0x7ffffff0 window in the kernel
    const char* e = (const char*)alloc_end(m, p);
    // (what the launch is entitled to touch must lie inside the allocation: a region sized for fewer tiles than this call uses
    // would otherwise have its legitimate tail clipped by the range check -- zeros read, stores dropped, silently.  NULL makes
    // the launch wrapper refuse the launch.)
    if (e && (const char*)p + logical > e) return nullptr;
    if (e && m->extents == 2) e = (const char*)p + logical;
    return e;
}

// the shapes of a stack's layers from its input map (the same arithmetic build_stack applies to the weights' packing)
void stack_shapes(int hid, int layers, const std::vector<int>& in_map0, std::vector<LayerShape>& out) {
    const int Hp = pad_hidden(hid);
    out.clear();
    for (int k = 0; k < layers; ++k) {
        const std::vector<int> in_map = k == 0 ? in_map0 : map_bidir(hid, Hp);
        LayerShape L{(int)in_map.size(), (int)in_map.size(), 0, hid, Hp};
        for (int i = 0; i < L.Ipad; ++i)
            if (in_map[i] >= 0) { L.Iused = i + 1; L.Ilo = std::min(L.Ilo, i); }
        out.push_back(L);
    }
}

// Everything about a handle that follows from its configuration alone (padded sizes, input maps, layer shapes): no device
// call in here -- dsp_model_create runs it before it uploads anything, dsp_debug_plan runs it on a handle that never sees a GPU.
void derive_geometry(dsp_model* m) {
    const Dims& d = m->d;
    m->hseq_p = d.hseq ? pad_hidden(d.hseq) : 0;
    m->hsig_p = d.hsig ? pad_hidden(d.hsig) : 0;
    m->Hp = pad_hidden(d.H);
    // front-end inputs are padded to >= 32 features so that the first four k-groups of every step are
    // x-part groups: the LSTM kernel requests them before h_t exists (dsp_kernels.hip, SPARSE note)
    m->Fseq = d.hseq ? std::max(32, rup(d.Iseq, 8)) : 0;
    m->Fsig = d.hsig ? std::max(32, rup(d.S, 8)) : 0;
    m->xoff_seq = d.hseq ? front_end_offset(d.Iseq, m->Fseq, pad_hidden(d.hseq)) : 0;
    m->xoff_sig = d.hsig ? front_end_offset(d.S, m->Fsig, pad_hidden(d.hsig)) : 0;
    m->Fcomb = m->hseq_p + m->hsig_p;
    m->small_classes = (m->Hp == 256 && m->Fcomb % 128 == 0 && m->Fcomb >= 256) || (m->Hp == 128 && m->Fcomb % 32 == 0 && m->Fcomb >= 64);
    m->Fwide = 2 * m->Hp;
    if (2 * m->hseq_p > m->Fwide) m->Fwide = 2 * m->hseq_p;
    if (2 * m->hsig_p > m->Fwide) m->Fwide = 2 * m->hsig_p;
    m->comb_in_map.assign(m->Fcomb, -1);
    for (int i = 0; i < d.hseq; ++i) m->comb_in_map[i] = i;
    for (int i = 0; i < d.hsig; ++i) m->comb_in_map[m->hseq_p + i] = d.hseq + i;
    m->seq_shape.clear(); m->sig_shape.clear();
    if (d.hseq) stack_shapes(d.hseq, d.l2, map_pad(d.Iseq, m->Fseq, m->xoff_seq), m->seq_shape);
    if (d.hsig) stack_shapes(d.hsig, d.l2, map_pad(d.S, m->Fsig, m->xoff_sig), m->sig_shape);
    stack_shapes(d.H, d.l1, m->comb_in_map, m->comb_shape);
}

int build_stack(dsp_model* m, const float* const* w, int in, int hid, int layers, std::vector<int> in_map0,
                std::vector<DevLstmLayer>& out, int with_split = 0 /* 1: bf16 pieces, 2: + fp16 pieces */) {
    const int Hp = pad_hidden(hid);
    for (int k = 0; k < layers; ++k) {
        const int I = k == 0 ? in : 2 * hid;
        const std::vector<int> in_map = k == 0 ? in_map0 : map_bidir(hid, Hp);
        DevLstmLayer L{};
        L.Ipad = (int)in_map.size(); L.H = hid; L.Hp = Hp;
        L.Iused = 0; L.Ilo = L.Ipad;  // features [Ilo, Iused) carry data; whole k-groups outside are padding
        for (int i = 0; i < L.Ipad; ++i)
            if (in_map[i] >= 0) { L.Iused = i + 1; L.Ilo = std::min(L.Ilo, i); }
        for (int d = 0; d < 2; ++d) {
            const float* const* p = w + (k * 2 + d) * 4;
            std::vector<float> wpk, bias;
            pack_lstm_dir(p[0], p[1], p[2], p[3], I, hid, Hp, in_map, wpk, bias);
            int rc = upload(m, wpk, &L.wpk[d]);
            if (rc) return rc;
            // the kernel folds the bias into the exp2 argument of the activations: upload it pre-scaled
            std::vector<float> sb(bias.size());
            for (size_t i = 0; i < bias.size(); ++i) {  // gate-major [4][Hp]: gate 2 (g) feeds tanh
                const bool is_g = i / (size_t)Hp == 2;
                sb[i] = (float)((is_g ? -2.8853900817779268 : -1.4426950408889634) * (double)bias[i]);
            }
            rc = upload(m, sb, &L.sbias[d]);
            if (rc) return rc;
            L.wsplit[d] = L.wsplit16[d] = nullptr;
            if (with_split && L.Ipad % 16 == 0 && Hp % 16 == 0 && Hp / 32 <= 8 && L.Ipad >= 16) {
                // the split pieces (3 bf16 / 2 fp16 per weight: 2.5 x the packing work of the fp32 form) are only built when
                // a split precision is asked for (ensure_split): the default fp32 path never pays for them
                L.split_kinds = with_split >= 2 ? 2 : 1;
                L.I = I; L.in_map = in_map;
                L.host_wih[d].assign(p[0], p[0] + (size_t)4 * hid * I);
                L.host_whh[d].assign(p[1], p[1] + (size_t)4 * hid * hid);
            }
        }
        out.push_back(L);
    }
    return 0;
}

// the split-precision weight pieces of every layer that has them, built and uploaded on first use
int ensure_split(dsp_model* m) {
    if (m->split_ready) return 0;
    int prev = 0;
    if (!m->dry) {
        HIP_TRY(hipGetDevice(&prev));
        HIP_TRY(hipSetDevice(m->device));
    }
    int rc = 0;
    for (std::vector<DevLstmLayer>* stack : {&m->seq, &m->sig, &m->comb})
        for (DevLstmLayer& L : *stack) {
            if (!L.split_kinds) continue;
            std::vector<float> ws[2][2];
            std::vector<std::thread> th;
            for (int d = 0; d < 2; ++d)
                for (int k = 0; k < L.split_kinds; ++k)
                    th.emplace_back([&L, &ws, d, k] {
                        pack_lstm_dir_split(L.host_wih[d].data(), L.host_whh[d].data(), L.I, L.H, L.Hp, L.in_map, ws[d][k], k == 1);
                    });
            for (auto& t : th) t.join();
            for (int d = 0; d < 2 && !rc; ++d) {
                rc = upload(m, ws[d][0], &L.wsplit[d]);
                if (!rc && L.split_kinds >= 2) rc = upload(m, ws[d][1], &L.wsplit16[d]);
                std::vector<float>().swap(L.host_wih[d]);
                std::vector<float>().swap(L.host_whh[d]);
            }
            if (rc) break;
            L.split_kinds = 0;
        }
    if (!m->dry) hipSetDevice(prev);
    if (!rc) m->split_ready = true;
    return rc;
}

size_t ws_layout(const dsp_model* m, int64_t sites, long long* NTp_out, size_t off[kWsRegions]) {
    long long nt = (sites + 31) / 32;
    long long NTp = (nt + 15) / 16 * 16;
    if (NTp == 0) NTp = 16;
    const size_t col = (size_t)NTp * m->d.T * 32 * sizeof(float);
    size_t o = 0;
    auto take = [&](size_t feats) { size_t r = o; o += (col * feats + 255) / 256 * 256; return r; };
    off[0] = take(m->Fseq);
    off[1] = take(m->Fsig);
    off[2] = take(m->Fwide);
    off[3] = take(m->Fwide);
    off[4] = take(m->Fcomb);
    off[5] = o; o += ((size_t)NTp * m->Fwide * 32 * sizeof(float) + 255) / 256 * 256;  // h0 scratch (no T)
    // cell-state scratch of the many-pass LSTM kernel (hidden > 256): one 8-wave workgroup per tile pair and direction =
    // NTp workgroups x passes x 64 KiB
    const int hmax = std::max(m->Hp, std::max(m->hseq_p, m->hsig_p));
    // (the DSP_LSTM_NP8=2 experiment runs two passes of four waves on a hidden-256 layer: 64 KiB per workgroup as well)
    const size_t cunits = std::max<size_t>(hmax > 256 ? (size_t)(hmax / 256) : 0, m->np8 == 2 ? 1 : 0);
    off[6] = o; o += (size_t)NTp * cunits * 65536;
    off[7] = o; o += ((size_t)NTp * m->Fwide * 32 * sizeof(float) + 255) / 256 * 256;  // h0 scratch of the side branch
    // x-ahead sums (LstmArgs::xacc): one 16 KiB block of four accumulator tiles per (live cluster, step, unit tile of 8)
    off[8] = o; o += m->xahead ? (size_t)std::min<long long>(NTp, m->xahead_tiles) * 2 * m->d.T * 8 * 16384 : 0;
    if (NTp_out) *NTp_out = NTp;
    return o;
}

int ensure_ws(dsp_model* m, int64_t sites, hipStream_t stream) {
    if (sites <= m->ws_sites && m->ws) return 0;
    if (m->ws && m->dry) { m->ws = nullptr; m->ws_sites = 0; }
    if (m->ws) {
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(m->ws));
        m->ws = nullptr; m->ws_sites = 0;
        for (size_t& o : m->ws_off) o = 0;
    }
    size_t off[kWsRegions];
    long long NTp;
    const size_t bytes = ws_layout(m, sites, &NTp, off);
    hipError_t e = hipSuccess;
    if (m->dry) m->ws = dry_alloc(m, bytes);
    else e = hipMalloc(&m->ws, bytes);
    if (e != hipSuccess) {
        m->ws = nullptr;
        (void)hipGetLastError();   // the runtime remembers the failure until it is read: the next launch's check must not see it
        return fail(DSP_ENOMEM, "workspace hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    char* b = (char*)m->ws;
    for (int i = 0; i < kWsRegions; ++i) m->ws_off[i] = off[i];
    m->ws_off[kWsRegions] = bytes;
    m->xseq = (float*)(b + off[0]); m->xsig = (float*)(b + off[1]);
    m->bufA = (float*)(b + off[2]); m->bufB = (float*)(b + off[3]); m->comb_in = (float*)(b + off[4]);
    m->h0buf = (float*)(b + off[5]);
    m->cbuf = (float*)(b + off[6]);
    m->h0buf2 = (float*)(b + off[7]);
    m->xacc = (float*)(b + off[8]);
    m->ws_sites = (int64_t)NTp * 32;
    return 0;
}

struct Launcher {
    dsp_model* m;
    hipStream_t s;
    long long NTp;  // padded tile count of THIS call (launch geometry is per call, never stored in the handle)
    int rc = 0;
    bool side_by_side = false;   // the seq and signal branches of this call run on two streams
    bool counters_zeroed = false;   // this forward's pack launch zeroes the arrival / admission counters of its clustered launches
    hipEvent_t last_ev = nullptr;   // profiling: the event behind the previous launch ...
    hipStream_t last_s = nullptr;   // ... and its stream
    hipEvent_t pending_a = nullptr, pending_b = nullptr;   // dominant-only profiling: the open bracket
    int pending_n = 0;
    const char* pending_name = nullptr;
    // dominant-only mode (dsp_profile_enable(m, 2)): ONE pair of records brackets the run of consecutive dominant launches
    // (2 records per forward).  The bracket is closed in front of the next other launch; every launch of the run gets an
    // entry with an equal share of the bracket's time (gaps and clean-up launches between them included)
    void close_bracket() {
        if (!pending_b) return;
        hipEventRecord(pending_b, s);
        for (int i = 0; i < pending_n; ++i) m->prof_entries.push_back({pending_name, pending_a, pending_b, pending_n});
        pending_a = pending_b = nullptr; pending_n = 0;
    }
    bool take_events(hipEvent_t* ea, hipEvent_t* eb) {
        while (m->event_pool.size() < m->event_used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) { rc = fail(DSP_EHIP, "hipEventCreate failed"); return false; }
            m->event_pool.push_back(e);
        }
        // consecutive launches of one stream share their boundary event (9 records per forward instead of 16: an event
        // record costs the stream ~4 us, 0.06 ms per forward -- 9 % of a 512-site one)
        if (last_ev && last_s == s) *ea = last_ev;
        else { *ea = m->event_pool[m->event_used++]; hipEventRecord(*ea, s); }
        *eb = m->event_pool[m->event_used++];
        return true;
    }
    template <class F> void run(const char* name, F&& f) {
        if (rc) return;
        const bool dominant = !strcmp(name, "lstm_comb");
        const bool bracket = m->prof && m->prof_dominant_only && dominant;
        const bool timed = m->prof && !m->prof_dominant_only;
        if (!bracket) close_bracket();
        hipEvent_t ea = nullptr, eb = nullptr;
        if (bracket && !pending_b) {
            last_ev = nullptr;
            if (!take_events(&pending_a, &pending_b)) return;
            pending_name = name;
        } else if (timed) {
            if (!take_events(&ea, &eb)) return;
        }
        const int e = f();
        if (e != 0) {
            rc = fail(DSP_EHIP, "launch %s failed: %s%s", name, hipGetErrorString((hipError_t)e),
                      e == (int)hipErrorInvalidValue ? " (a pointer of the launch without the end of its allocation, or a launch that would reach past it: "
                                                       "the launch wrappers refuse a buffer descriptor without a real extent)" : "");
            return;
        }
        if (bracket) ++pending_n;
        else if (timed) { hipEventRecord(eb, s); m->prof_entries.push_back({name, ea, eb, 1}); last_ev = eb; last_s = s; }
        else last_ev = nullptr;
        if (m->sync_each) {  // DSP_SYNC_EACH (read once, in dsp_model_create): attribute an asynchronous GPU fault to its launch
            fprintf(stderr, "[launch] %s ...", name);
            const hipError_t se = hipStreamSynchronize(s);
            fprintf(stderr, " %s\n", se == hipSuccess ? "ok" : hipGetErrorString(se));
        }
    }
};

// site groups per workgroup of one LSTM layer launch (a wave owns one unit tile x two site tiles; at most 8 waves per
// workgroup).  Front ends (4 unit tiles at the default sizes): one site group = 4-wave workgroups, two of which share a
// CU independently (measured +1 % on those launches over one 8-wave workgroup).  DSP_LSTM_SG overrides (A/B switch).
int pick_site_groups(const dsp_model* m, int UT) {
    const int gmax = 8 / UT > 0 ? 8 / UT : 1;
    int g = m->sg_override;
    if (g < 1 || g > gmax || (g & (g - 1))) g = UT <= 4 ? 4 / UT : 1;
    return g;
}

// Clustered launch of a dense one-pass layer (dsp_lstmc_kernel): the 8 unit tiles x 4 gates of a (site tile, direction) spread
// over P = 2 / 4 / 8 workgroups on as many compute units, when the batch is so small that P x (site tiles x 2 directions)
// workgroups still fit the CUs at once (every member of a cluster resident from the start: the members wait for each
// other).  Returns the gates per wave (8 / P), or 0 = the layer runs unclustered.
constexpr int kClusterWordsPerLaunch = 256 * 32;   // at most 128 clusters (x 32 words apart) fit 256 CUs at P >= 2
constexpr int kClusterLaunches = 48;               // LSTM launches of one forward (3 stacks x at most 15 layers, 45)
int cluster_size(const dsp_model* m, long long NTp) {
    if (m->cluster == 0) return 0;
    const long long slots = (long long)m->n_cus;   // workgroups resident at once: one per CU (two clustered workgroups per CU
    // were measured in round 4: 1.236 vs 1.191 ms at 1,024 sites, 3.515 vs 3.464 ms at 4,096 -- slower; not kept)
    int P = 1;
    while (P < 8 && NTp * 2 * (P * 2) <= slots) P *= 2;
    if (m->cluster > 0) {   // DSP_LSTM_CLUSTER = gates per wave: that cluster size, if it fits
        const int want = m->cluster == 1 ? 8 : (m->cluster == 2 ? 4 : (m->cluster == 4 ? 2 : 0));
        P = (want && NTp * 2 * want <= slots) ? want : 1;
    }
    return P >= 2 ? P : 0;
}
// Front ends (layers of 4 unit tiles whose x part is one ring of four k-groups: hidden 128, 7 or 16 features): gates per wave
// of the clustered form (1: P = 4 workgroups per (site tile, direction), 2: P = 2), or 0 = workgroup-local.  `branches` = the
// front-end launches that run side by side (2 with the signal branch on the side stream): all their workgroups must be
// resident at once, one per CU.  Round 5: 512 sites 0.122 / 0.131 -> ~0.05 ms per front-end launch.  DSP_LSTM_FRONT_CLUSTER=0
// turns it off (A/B switch), =1 / 2 forces that many gates per wave where it fits.
int front_cluster_gates(const dsp_model* m, const LstmArgs& a, long long NTp, int branches) {
    // (4 unit tiles: the front ends of both_bilstm at hidden 128; 8: the seq front end of a seq-only model at hidden 193..256,
    // BASELINE configs[2]'s shape -- clusters of 8 / 4 workgroups)
    if (m->cluster == 0 || m->front_cluster == 0 || (a.UT != 4 && a.UT != 8) || a.NP > 1 || (a.Ipad >> 3) != 4 ||
        a.NQ != ((a.Ipad + a.Hp) >> 3) || a.NQ % 4 || a.NQ < 8 || NTp * 2 * 32 > kClusterWordsPerLaunch)
        return 0;
    const long long slots = (long long)m->n_cus / std::max(1, branches);
    int P = 1;
    while (P < a.UT && NTp * 2 * (P * 2) <= slots) P *= 2;
    if (m->front_cluster > 0) {
        const int want = m->front_cluster == 1 ? a.UT : (m->front_cluster == 2 ? a.UT / 2 : 0);
        P = (want && NTp * 2 * want <= slots) ? want : 1;
    }
    const int G = P >= 2 ? a.UT / P : 0;
    return (G == 1 || G == 2) ? G : 0;   // (the clustered front-end form exists for 1 and 2 gates per wave)
}
int pick_cluster(const dsp_model* m, const LstmArgs& a, long long NTp, bool split, int branches) {
    if (!split) {
        const int fg = front_cluster_gates(m, a, NTp, branches);
        if (fg) return fg;
    }
    // layers of 4 unit tiles (the front ends at hidden 128): one 4-wave workgroup per (site tile, direction) holds the whole
    // layer -- a wave 1 unit tile x 1 site tile, half the work per step of the 64-site tiling -- whenever those workgroups
    // fit the CUs at once (<= 4,096 sites); no exchange between workgroups, so nothing to wait for.  (On full batches this form
    // is the slower one -- 1.99 / 2.08 ms against 1.82 / 1.93 ms of dsp_lstm_kernel<2, 1, XL> per front-end launch of 65,536
    // sites, same-box A/B in round 4: a weight fragment feeds 4 MFMAs instead of 8.)
    if (m->cluster != 0 && !split && a.UT == 4 && a.NP <= 1 && NTp * 2 <= (long long)m->n_cus &&
        a.NQ == ((a.Ipad + a.Hp) >> 3) && (a.Ipad >> 3) % 4 == 0 && (a.Ipad >> 3) >= 4 && a.NQ % 4 == 0 && a.NQ >= 8) {
        // dense layers of 4 unit tiles (the combined stack of a hid_rnn-128 model: hidden 97..128) of batches that leave CUs
        // idle: clustered like those of 8 unit tiles, 4 / 2 workgroups per (site tile, direction) (round 5)
        const int P8 = cluster_size(m, NTp);
        if (P8 >= 2 && m->front_cluster != 0 && a.nqx_lo == 0 && a.nqx_used == (a.Ipad >> 3) && (a.Ipad >> 3) >= 8 &&
            NTp * 2 * 32 <= kClusterWordsPerLaunch)
            return P8 >= 4 ? 1 : 2;
        return 4;
    }
    const int P = cluster_size(m, NTp);
    if (!P && m->local8 && !split && a.UT == 8 && a.NP <= 1 && NTp * 2 <= (long long)m->n_cus && a.nqx_lo == 0 &&
        a.nqx_used == (a.Ipad >> 3) && a.NQ == ((a.Ipad + a.Hp) >> 3) && (a.Ipad >> 3) % 4 == 0 && a.NQ % 4 == 0)
        return -4;   // (negative: the workgroup-local form with eight waves)
    if (!P || split || a.UT != 8 || a.NP > 1 || NTp * 2 * 32 > kClusterWordsPerLaunch) return 0;
    const int nqx = a.Ipad >> 3, G = 8 / P, D = G == 4 ? 4 : (G == 2 ? 8 : 16);
    if (a.nqx_lo != 0 || a.nqx_used != nqx || a.NQ != ((a.Ipad + a.Hp) >> 3) || nqx % D || nqx < 2 * D || a.NQ % D) return 0;
    return G;
}

// The launch geometry of one LSTM layer at a batch of NTp tiles -- everything that does not need a pointer: dimensions, unit
// tiles / passes / site groups, the small-batch tiling flags.  One source for the forward (run_stack) and for the plan of a
// call's pieces (piece_cost_us below).
void shape_lstm(const dsp_model* m, const LayerShape& ly, int lstm_id, long long NTp, LstmArgs& a) {
    a.NTp = NTp;
    a.Ipad = ly.Ipad; a.H = ly.H; a.Hp = ly.Hp; a.T = m->d.T; a.Fout = 2 * ly.Hp;
    a.NQ = rup((ly.Ipad + ly.Hp) / 8, 4);
    a.nqx_used = (ly.Iused + 7) / 8; a.nqx_lo = ly.Ilo / 8;
    a.UT = ly.Hp / 32;
    a.NP = a.UT > 8 ? a.UT / 8 : (a.UT == 8 && lstm_id == 2 ? m->np8 : 1);
    a.SG = a.NP >= 2 ? 1 : pick_site_groups(m, a.UT);
    // small batches: 32-site workgroups of UT/2 waves put twice as many CUs to work and halve the launch's latency (a
    // batch of 512 sites: 32 workgroups of one wave per SIMD instead of 16 of two); from the size where every CU has
    // its 64-site workgroup the shipped tiling is the faster one (0.25 %: dsp_kernels.hip)
    const bool use21 = m->tiling21 > 0 || (m->tiling21 < 0 && NTp * 2 <= (long long)m->n_cus);
    // (automatic choice: one such workgroup per CU -- bit 2 -- so that forwards issued concurrently on other streams
    // spread over the idle CUs instead of doubling up on busy ones)
    a.flags = (m->phase_prio ? 1 : 0) | (use21 ? 2 : 0) | (use21 && m->tiling21 < 0 ? 4 : 0);
}
// ... and which form of the kernel it takes: a.CG (0, or gates per wave of dsp_lstmc_kernel) and the flags that go with it
void pick_form(const dsp_model* m, LstmArgs& a, long long NTp, bool split, bool side_by_side, bool counters) {
    // batches that leave most CUs idle: the layer's unit tiles spread over a cluster of workgroups (dsp_lstmc_kernel)
    a.CG = counters ? pick_cluster(m, a, NTp, split, side_by_side ? 2 : 1) : 0;
    if (a.CG < 0) { a.CG = -a.CG; a.flags |= 8; }
    if (a.CG) {
        // one workgroup per CU: not for the workgroup-local forms of two branches that run side by side on two streams (they
        // may share CUs); the CLUSTERED front ends of two branches were sized so that both grids fit the CUs together
        if (!side_by_side || (a.UT == 4 && a.CG < 4)) a.flags |= 4;
        if (m->wave_handoff) a.flags |= 64;
    }
    // x ahead: the clustered dense forms of 8 or 4 unit tiles (not the front ends' one-ring x part, not the workgroup-local forms) on a
    // call of few live tiles; the launch keeps the last ring of its x part
    a.xs = 0;
    const bool dense8 = a.UT == 8 && !(a.flags & 8), dense4 = a.UT == 4 && a.CG < 4;   // (dsp_k_lstm's forms; 4 gates per wave at 4 unit tiles = workgroup-local)
    if (m->xahead && a.CG > 0 && (dense8 || dense4) && (a.Ipad >> 3) != 4 && a.n > 0 && (a.n + 31) / 32 <= m->xahead_tiles) {
        const bool ring8 = dense8 && a.CG == 1 && m->xahead_ring == 8;
        const int D = dense4 ? 4 : (a.CG == 4 ? 4 : (a.CG == 2 || ring8 ? 8 : 16));
        if ((a.Ipad >> 3) >= 2 * D) { a.xs = (a.Ipad >> 3) - D; if (ring8) a.flags |= 128; }
    }
}

// run one BiLSTM stack; returns the buffer holding the last layer's output
float* run_stack(Launcher& L, const char* name, const std::vector<DevLstmLayer>& layers, int lstm_id, const float* x,
                 int64_t n, const dsp_init_state* init, const float* h0, const float* c0, bool side = false) {
    // side: the stack runs next to another one (one layer only): its output goes to bufB, its h0 to the second scratch
    dsp_model* m = L.m;
    const float* cur = x;
    float* dst = nullptr;
    struct Planned { LstmArgs a; bool split; int prec; };
    std::vector<Planned> plan;
    for (size_t k = 0; k < layers.size(); ++k) {
        const DevLstmLayer& ly = layers[k];
        dst = ((layers.size() - 1 - k) % 2 == 0) != side ? m->bufA : m->bufB;
        LstmArgs a{};
        a.x = cur; a.out = dst;
        a.wpk0 = ly.wpk[0]; a.wpk1 = ly.wpk[1]; a.sbias0 = ly.sbias[0]; a.sbias1 = ly.sbias[1];
        a.n = n;
        shape_lstm(m, LayerShape{ly.Ipad, ly.Ilo, ly.Iused, ly.H, ly.Hp}, lstm_id, L.NTp, a);
        a.h0buf = side ? m->h0buf2 : m->h0buf;
        a.cbuf = m->cbuf;
        a.init_mode = init ? init->mode : DSP_INIT_ZEROS;
        a.seed = init ? init->seed : 0; a.site_offset = init ? init->site_offset : 0;
        a.site_keys = (init && init->mode == DSP_INIT_PHILOX) ? (const unsigned long long*)init->site_keys : nullptr;
        a.stream_base = lstm_id * 64 + (int)k * 4;
        if (m->trace_launch >= 0 && m->trace_launch == m->lstm_launch_no) a.flags |= 256 | (m->trace_wave << 9);  // DSP_TRACE builds
        const int launch_no = m->lstm_launch_no;
        ++m->lstm_launch_no;
        if (a.init_mode == DSP_INIT_EXPLICIT) {
            a.h0 = h0 + (size_t)(2 * k) * (size_t)n * ly.H;
            a.c0 = c0 + (size_t)(2 * k) * (size_t)n * ly.H;
        }
        // fp16 pieces are only safe where the operands are bounded: the combined stack eats relu(fc(h)) and h in (-1, 1);
        // the front ends eat raw features (a signal mean of 1e6 is a legal row), so they take the bf16 variant, whose
        // pieces have fp32's range
        const int prec = (lstm_id != 2 && m->precision == DSP_PREC_FP16X3) ? DSP_PREC_BF16X6 : m->precision;
        const bool split = prec != DSP_PREC_FP32 && ly.wsplit[0] && ly.wsplit[1] && a.UT <= 8;
        if (split) {  // the split kernels run 8-wave workgroups; their own weights and k-stage count
            const bool f16 = prec == DSP_PREC_FP16X3;
            a.wpk0 = f16 ? ly.wsplit16[0] : ly.wsplit[0]; a.wpk1 = f16 ? ly.wsplit16[1] : ly.wsplit[1];
            a.NQ = (ly.Ipad + ly.Hp) / 16;
            a.SG = 8 / a.UT;
        }
        pick_form(m, a, L.NTp, split, L.side_by_side, launch_no < kClusterLaunches && m->cflags);
        if (a.CG) {
            a.cluster_timeout = m->cluster_timeout;
            a.cflags = m->cflags + (size_t)launch_no * kClusterWordsPerLaunch;
        }
        // A clustered launch on counters that were not zeroed would admit at once and find every arrival "already in" (the
        // previous forward's counts): h rows read before they exist, silently.  The pack launch zeroes them under the same
        // size condition the cluster sizes are picked by (NTp x 4 <= CUs); should the two ever drift apart, fail here.
        if (a.CG > 0 && !((a.CG == 4 && a.UT == 4) || (a.flags & 8)) && !L.counters_zeroed && !L.rc)
            L.rc = fail(DSP_EINVAL, "internal: a clustered launch (%s, layer %zu, %lld tiles) on counters this forward did not zero", name, k,
                        (long long)L.NTp);
        {   // the extents behind this launch's descriptors (after the split switch: a.wpk0 / a.NQ are the launch's own)
            const size_t tile_x = (size_t)a.T * (size_t)(a.Ipad >> 2) * 512, tile_o = (size_t)a.T * (size_t)(a.Fout >> 2) * 512;
            const size_t wbytes = split ? (size_t)a.UT * a.NQ * 4096 * (prec == DSP_PREC_FP16X3 ? 2 : 3) : (size_t)a.UT * a.NQ * 4096;
            a.x_end = rsrc_end(m, a.x, (size_t)a.NTp * tile_x);
            if (m->test_shrink && a.x_end) a.x_end = (const char*)a.x_end - m->test_shrink;
            a.out_end = rsrc_end(m, a.out, (size_t)a.NTp * tile_o);
            a.h0buf_end = rsrc_end(m, a.h0buf, (size_t)a.NTp * (size_t)(a.Fout >> 2) * 512);
            a.wpk0_end = rsrc_end(m, a.wpk0, wbytes);
            a.wpk1_end = rsrc_end(m, a.wpk1, wbytes);
            // the cell-state scratch of the many-pass kernel: one 64 KiB slice per pass and workgroup (NTp workgroups); an
            // empty region when no layer needs it (its end is then its start)
            const char* cb = (const char*)a.cbuf;
            const char* cregion = m->ws ? (const char*)m->ws + m->ws_off[7] : nullptr;
            a.cbuf_end = !cb ? nullptr : (m->extents == 0 ? cb + (1ull << 40) :
                         (m->extents == 2 && a.NP >= 2 ? std::min(cregion, cb + (size_t)a.NTp * a.NP * 65536) : cregion));
            a.cflags_end = a.cflags ? (const char*)m->cflags + (size_t)m->n_cflag_words * sizeof(unsigned int) : nullptr;
            if (a.cflags && m->extents == 2) a.cflags_end = (const char*)a.cflags + (size_t)kClusterWordsPerLaunch * sizeof(unsigned int);
            if (a.xs) {
                a.xacc = m->xacc;
                a.xacc_end = rsrc_end(m, a.xacc, (size_t)((a.n + 31) / 32) * 2 * a.T * a.UT * 16384);
            }
        }
        if (m->debug_lstm)   // DSP_DEBUG_LSTM (read once, in dsp_model_create)
            fprintf(stderr, "[lstm] %s k=%zu split=%d CG=%d Ipad=%d H=%d Hp=%d UT=%d SG=%d NQ=%d NTp=%lld n=%lld T=%d Fout=%d x=%p out=%p\n", name, k,
                    (int)split, a.CG, a.Ipad, a.H, a.Hp, a.UT, a.SG, a.NQ, a.NTp, a.n, a.T, a.Fout, (const void*)a.x, (void*)a.out);
        plan.push_back({a, split, prec});
        cur = dst;
    }
    for (const Planned& p : plan)
        L.run(name, [&] { return p.split ? dsp_k_lstm6(&p.a, p.prec, L.s) : dsp_k_lstm(&p.a, L.s); });
    return dst;
}

}  // namespace

extern "C" {

const char* dsp_last_error(void) { return g_err.c_str(); }
void dsp_set_error_(const char* msg) { g_err = msg ? msg : ""; }  // used by dsp_text.cpp
#ifdef DSP_EMU   // the test-suite's SIMT interpreter build (tests/native/emu): never loadable as the product
int32_t dsp_abi_version(void) { return DSP_AMD_ABI_VERSION + 1000; }
#else
int32_t dsp_abi_version(void) { return DSP_AMD_ABI_VERSION; }
#endif

int32_t dsp_weight_count(const dsp_model_cfg* cfg) {
    Dims d;
    int rc = derive(cfg, &d);
    if (rc) return rc;
    return (int32_t)weight_spec(d).size();
}

int32_t dsp_weight_spec(const dsp_model_cfg* cfg, int32_t idx, char* name, size_t name_cap, int64_t shape[2],
                        int32_t* ndim) {
    Dims d;
    int rc = derive(cfg, &d);
    if (rc) return rc;
    const std::vector<Spec> v = weight_spec(d);
    if (idx < 0 || (size_t)idx >= v.size()) return fail(DSP_EINVAL, "weight index %d out of range", idx);
    if (name && name_cap) { strncpy(name, v[idx].name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (shape) { shape[0] = v[idx].shape[0]; shape[1] = v[idx].shape[1]; }
    if (ndim) *ndim = v[idx].ndim;
    return 0;
}

int64_t dsp_flops_per_site(const dsp_model_cfg* cfg) {
    Dims d;
    if (derive(cfg, &d)) return -1;
    auto lstm = [&](int64_t i, int64_t h, int layers) {
        int64_t mac = 0;
        for (int k = 0; k < layers; ++k) mac += 2 * (int64_t)d.T * 4 * h * ((k == 0 ? i : 2 * h) + h);
        return mac;
    };
    int64_t mac = 0;
    if (d.hseq) mac += lstm(d.Iseq, d.hseq, d.l2) + (int64_t)d.T * d.hseq * 2 * d.hseq;
    if (d.hsig) mac += lstm(d.S, d.hsig, d.l2) + (int64_t)d.T * d.hsig * 2 * d.hsig;
    mac += lstm(d.H, d.H, d.l1) + (int64_t)d.H * 2 * d.H + (int64_t)d.C * d.H;
    return 2 * mac;
}

// the weights of a handle: repacked on the host, uploaded (or, for a dry handle, only sized)
static int build_weights(dsp_model* m, const float* const* w) {
    const Dims& d = m->d;
    int rc = 0;
    int wi = 0;
    if (d.hseq) {
        std::vector<float> emb(w[wi], w[wi] + (size_t)d.V * d.E);
        rc = upload(m, emb, &m->embed); if (rc) return rc;
        ++wi;
        rc = build_stack(m, w + wi, d.Iseq, d.hseq, d.l2, map_pad(d.Iseq, m->Fseq, m->xoff_seq), m->seq, 1); if (rc) return rc;
        wi += 8 * d.l2;
        std::vector<float> wpk, bias;
        pack_linear(w[wi], w[wi + 1], d.hseq, m->hseq_p, 2 * d.hseq, map_bidir(d.hseq, m->hseq_p), wpk, bias);
        if (!(fc_out_bound(w[wi], w[wi + 1], d.hseq, 2 * d.hseq) <= kFp16Safe)) m->fp16_safe = false;
        m->fc_seq.Fin = 2 * m->hseq_p; m->fc_seq.ORT = m->hseq_p / 32;
        rc = upload(m, wpk, &m->fc_seq.wpk); if (rc) return rc;
        rc = upload(m, bias, &m->fc_seq.bias); if (rc) return rc;
        wi += 2;
    }
    if (d.hsig) {
        rc = build_stack(m, w + wi, d.S, d.hsig, d.l2, map_pad(d.S, m->Fsig, m->xoff_sig), m->sig, 1); if (rc) return rc;
        wi += 8 * d.l2;
        std::vector<float> wpk, bias;
        pack_linear(w[wi], w[wi + 1], d.hsig, m->hsig_p, 2 * d.hsig, map_bidir(d.hsig, m->hsig_p), wpk, bias);
        if (!(fc_out_bound(w[wi], w[wi + 1], d.hsig, 2 * d.hsig) <= kFp16Safe)) m->fp16_safe = false;
        m->fc_sig.Fin = 2 * m->hsig_p; m->fc_sig.ORT = m->hsig_p / 32;
        rc = upload(m, wpk, &m->fc_sig.wpk); if (rc) return rc;
        rc = upload(m, bias, &m->fc_sig.bias); if (rc) return rc;
        wi += 2;
    }
    for (int k = 0; k < d.l1; ++k)  // weight_ih, weight_hh of both directions of every combined layer
        for (int dd = 0; dd < 2; ++dd) {
            const int I = k == 0 ? d.H : 2 * d.H;
            if (!(max_abs(w[wi + (k * 2 + dd) * 4], (size_t)4 * d.H * I) <= kFp16Safe) ||
                !(max_abs(w[wi + (k * 2 + dd) * 4 + 1], (size_t)4 * d.H * d.H) <= kFp16Safe))
                m->fp16_safe = false;
        }
    rc = build_stack(m, w + wi, d.H, d.H, d.l1, m->comb_in_map, m->comb, 2); if (rc) return rc;
    if (m->precision == DSP_PREC_FP16X3 && !m->fp16_safe) m->precision = DSP_PREC_BF16X6;  // DSP_PRECISION asked for it
    if (m->precision != DSP_PREC_FP32) { rc = ensure_split(m); if (rc) return rc; }
    wi += 8 * d.l1;
    {
        std::vector<float> wpk, bias;
        pack_linear(w[wi], w[wi + 1], d.H, m->Hp, 2 * d.H, map_bidir(d.H, m->Hp), wpk, bias);
        m->fc1.Fin = 2 * m->Hp; m->fc1.ORT = m->Hp / 32;
        rc = upload(m, wpk, &m->fc1.wpk); if (rc) return rc;
        rc = upload(m, bias, &m->fc1.bias); if (rc) return rc;
        wi += 2;
        std::vector<float> w2((size_t)d.C * m->Hp, 0.f), b2(w[wi + 1], w[wi + 1] + d.C);
        for (int c = 0; c < d.C; ++c)
            for (int k = 0; k < d.H; ++k) w2[(size_t)c * m->Hp + k] = w[wi][(size_t)c * d.H + k];
        rc = upload(m, w2, &m->w2); if (rc) return rc;
        rc = upload(m, b2, &m->b2); if (rc) return rc;
        wi += 2;
    }
    return 0;
}

int32_t dsp_model_create(const dsp_model_cfg* cfg, const float* const* host_weights, const int64_t* numels,
                         int32_t n_weights, int32_t device, dsp_model** out) {
    if (!out) return fail(DSP_EINVAL, "out is NULL");
    *out = nullptr;
    Dims d;
    int rc = derive(cfg, &d);
    if (rc) return rc;
    const std::vector<Spec> spec = weight_spec(d);
    if (!host_weights || !numels) return fail(DSP_EINVAL, "weights are NULL");
    if ((size_t)n_weights != spec.size())
        return fail(DSP_ESHAPE, "state_dict has %d tensors, model expects %zu", n_weights, spec.size());
    for (size_t i = 0; i < spec.size(); ++i) {
        if (!host_weights[i]) return fail(DSP_ESHAPE, "Missing key(s) in state_dict: \"%s\"", spec[i].name.c_str());
        if (numels[i] != spec_numel(spec[i]))
            return fail(DSP_ESHAPE, "size mismatch for %s: got %lld elements, model expects %lld", spec[i].name.c_str(),
                        (long long)numels[i], (long long)spec_numel(spec[i]));
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(DSP_EINVAL, "device %d out of range (%d visible)", device, ndev);
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(device));
    {
        const int e = dsp_k_init();
        if (e) { hipSetDevice(prev); return fail(DSP_EHIP, "kernel attribute setup failed: %s", hipGetErrorString((hipError_t)e)); }
    }

    dsp_model* m = new (std::nothrow) dsp_model();
    if (!m) { hipSetDevice(prev); return fail(DSP_ENOMEM, "out of host memory"); }
    m->cfg = *cfg; m->d = d; m->device = device;
    if (const char* v = getenv("DSP_TRACE_LAUNCH")) m->trace_launch = atoi(v);
    if (const char* v = getenv("DSP_TRACE_WAVE")) m->trace_wave = atoi(v) & 7;
    m->sync_each = getenv("DSP_SYNC_EACH") != nullptr;   // (never per launch: a 512-site forward is 9 launches in 2-4 ms)
    m->debug_lstm = getenv("DSP_DEBUG_LSTM") != nullptr;
    m->head_st4 = getenv("DSP_HEAD_ST4") != nullptr;
    if (const char* v = getenv("DSP_LSTM_LOCAL8")) m->local8 = atoi(v) != 0;
    if (const char* v = getenv("DSP_FC_FUSED")) m->fc_fused = atoi(v) != 0;
    if (const char* v = getenv("DSP_FC_SMALL")) m->fc_small = atoi(v) != 0;
    if (const char* v = getenv("DSP_FORWARD_SPLIT")) m->forward_split = atoi(v) != 0;
    if (const char* v = getenv("DSP_LSTM_HANDOFF")) m->wave_handoff = atoi(v) != 0;
    read_xahead_switch(m);
    if (const char* v = getenv("DSP_LSTM_SG")) m->sg_override = atoi(v);  // A/B switch
    if (const char* v = getenv("DSP_LSTM_NP8")) m->np8 = atoi(v) == 2 ? 2 : 1;  // A/B switch
    if (const char* v = getenv("DSP_LSTM_PRIO")) m->phase_prio = atoi(v) != 0;    // A/B switch
    if (const char* v = getenv("DSP_LSTM_TILING")) m->tiling21 = atoi(v) == 21 ? 1 : 0;   // A/B switch
    if (const char* v = getenv("DSP_LSTM_CLUSTER")) m->cluster = atoi(v);                  // A/B switch
    if (const char* v = getenv("DSP_LSTM_FRONT_CLUSTER")) m->front_cluster = atoi(v);      // A/B switch
    if (const char* v = getenv("DSP_CLUSTER_TIMEOUT")) m->cluster_timeout = strtoull(v, nullptr, 10);
    if (const char* v = getenv("DSP_TWO_STREAMS")) m->two_streams = atoi(v) != 0 ? 1 : 0;   // A/B switch
    m->extents = dsp_k_bounds_build() ? 2 : 1;   // (the bounds-recording build checks the tight extents of every call)
    if (dsp_k_bounds_build())
        if (const char* v = getenv("DSP_BOUNDS_TEST_SHRINK")) m->test_shrink = (size_t)strtoull(v, nullptr, 10);
    if (const char* v = getenv("DSP_RSRC_EXTENTS"))
        m->extents = !strcmp(v, "wide") ? 0 : (!strcmp(v, "tight") ? 2 : (!strcmp(v, "region") ? 1 : m->extents));
    if (hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        m->two_streams = 0;   // no side stream to be had: the branches run in sequence
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) m->n_cus = prop.multiProcessorCount;
    }
    {   // arrival counters of the clustered launches (zeroed by every forward's first launch; 1.5 MB)
        void* p = nullptr;
        const size_t bytes = (size_t)kClusterLaunches * kClusterWordsPerLaunch * sizeof(unsigned int);
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); hipSetDevice(prev); delete m; return fail(DSP_ENOMEM, "hipMalloc(%zu) failed", bytes); }
        m->dev_allocs.push_back(p);
        m->cflags = (unsigned int*)p;
        m->n_cflag_words = kClusterLaunches * kClusterWordsPerLaunch;
    }
    if (m->cluster != 0) {
        // The clustered launches assume block b of a launch runs on XCD b % 8 (members of a cluster = consecutive entries of
        // one XCD's dispatch list).  Observed, not promised: ask the device -- 64 blocks report their XCC_ID -- and keep to the
        // workgroup-local forms when it answers anything else (CPX / partition modes, CU masks, another XCD count).
        // DSP_LSTM_CLUSTER_PROBE=0 skips the question.  Round 6: asked ONCE per device and process,
        // not per handle; on the handle's non-blocking side stream with the answer landing in the counters' own allocation
        // (no hipMalloc / hipFree pair, no launch on the legacy stream -- illegal while another stream captures --, no
        // device-wide sync); and a device that fails the probe says so on stderr instead of only through dsp_model_query.
        static std::mutex probe_mu;
        static std::map<int, int> probe_result;   // device -> 1: blocks 8 apart share an XCD, 0: they do not
        const char* pv = getenv("DSP_LSTM_CLUSTER_PROBE");
        if (!(pv && atoi(pv) == 0)) {
            std::lock_guard<std::mutex> lock(probe_mu);
            auto it = probe_result.find(device);
            if (it == probe_result.end()) {
                unsigned host[64];
                (void)hipGetLastError();   // (whatever an earlier, unrelated call of this thread left behind -- torch's event queries leave
                // hipErrorNotReady -- is not the probe's error: the launch wrapper reports the thread's last error)
                hipStream_t ps = m->side;   // (NULL only when no side stream could be made: the legacy stream then, as until round 5)
                bool ok = dsp_k_probe_xcc(m->cflags, 64, ps) == 0 &&
                          hipMemcpyAsync(host, m->cflags, sizeof host, hipMemcpyDeviceToHost, ps) == hipSuccess &&
                          hipStreamSynchronize(ps) == hipSuccess;
                const bool read_ok = ok;
                // (the dispatcher's round robin carries on where the previous launch stopped: block 0 lands on ANY XCD, block b on the
                // b-th after it -- what the clustered launches need is that blocks 8 apart share an XCD)
                for (int b = 0; ok && b < 64; ++b) ok = host[b] < 8 && host[b] == (host[0] + (unsigned)b) % 8;
                if (m->debug_lstm && read_ok) {
                    fprintf(stderr, "[lstm] XCC probe:");
                    for (int b = 0; b < 64; ++b) fprintf(stderr, " %u", host[b]);
                    fprintf(stderr, "\n");
                }
                if (!ok) {
                    (void)hipGetLastError();
                    fprintf(stderr, "libdsp_amd: device %d: %s -- the clustered small-batch launches stay off on this device "
                                    "(batches <= 2,048 sites run the workgroup-local forms; DSP_LSTM_CLUSTER_PROBE=0 skips the question)\n", device,
                            read_ok ? "consecutive blocks of a launch do not run on consecutive XCDs here (XCC_ID probe: a partition mode, a CU mask, another XCD count?)"
                                    : "the XCC_ID probe could not run");
                }
                it = probe_result.emplace(device, ok ? 1 : 0).first;
            }
            if (!it->second) { m->cluster = 0; m->xcc_probe_failed = true; }
        }
    }
    if (const char* v = getenv("DSP_PRECISION"))
        m->precision = !strcmp(v, "bf16x6") ? DSP_PREC_BF16X6 : (!strcmp(v, "bf16x9") ? DSP_PREC_BF16X9 :
                       (!strcmp(v, "fp16x3") ? DSP_PREC_FP16X3 : DSP_PREC_FP32));
    derive_geometry(m);

    rc = build_weights(m, host_weights);
    hipSetDevice(prev);
    if (rc) { dsp_model_destroy(m); return rc; }
    *out = m;
    return 0;
}

size_t dsp_workspace_bytes(const dsp_model* m, int64_t max_sites) {
    if (!m) return 0;
    size_t off[kWsRegions];
    return ws_layout(m, max_sites, nullptr, off);
}

int32_t dsp_model_reserve(dsp_model* m, int64_t max_sites) {
    if (!m) return fail(DSP_EINVAL, "model is NULL");
    if (max_sites < 1) return fail(DSP_EINVAL, "max_sites must be >= 1");
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(m->device));
    const int rc = ensure_ws(m, max_sites, nullptr);
    hipSetDevice(prev);
    return rc;
}

static int32_t forward_chunk(dsp_model* m, void* stream, int64_t n, const void* kmer, int32_t kmer_dtype, const float* means,
                             const float* stds, const void* lens, int32_t lens_dtype, const float* signals,
                             const dsp_init_state* init, float* logits, float* probs, uint8_t* labels);

// Sites are independent and the in-kernel initial states are keyed by the site's global index, so a call may be cut anywhere
// without changing a bit.  Round 5: what a forward costs is a STEP function of its size -- the 64-site workgroups of batches
// above 4,096 sites fill the chip in rounds of 8,192 sites (one workgroup per CU; two per CU share its matrix pipe and gain
// nothing: 10,000 sites cost two rounds = 13.3 ms), and below that the small-batch forms come in classes of 512 / 1,024 /
// 2,048 / 4,096 sites (clusters of 8 / 4 / 2 CUs, one workgroup per CU) costing 0.57 / 0.98 / 1.77 / 3.38 ms whatever the
// size inside the class.  A call runs its whole rounds first and its remainder as the cheapest sequence of small-batch pieces
// (a 16-entry table below: e.g. 1,100 sites = 1,024 + 76 -> 1.55 instead of 1.77 ms, 3,000 = 2,048 + 952 -> 2.75 instead of
// 3.38, 5,000 = 4,096 + 904 -> 4.4 instead of 6.65, 10,000 = 8,192 + 1,808 -> 8.4 instead of 13.3).
// Not with explicit initial states (their layout has the site index in the middle dimension) and not in the split-precision
// modes (their kernels have no small-batch forms).  DSP_FORWARD_SPLIT=0 turns it off (A/B switch).
namespace {
// ---- what a piece costs, and how a call is cut (round 6: from the model and the device, not from a constant table) ---------
// Round 5 planned the pieces of a call with five measured costs of the DEFAULT model on a 256-CU device ({610, 1020, 1810, 3420,
// 6650} us for 512 / 1,024 / 2,048 / 4,096 sites / one round) and fixed sizes (round 8,192 = 256 CUs x 32 sites, unit 512): wrong
// ratios for every other model with small-batch forms (hidden 128: the 512 and 1,024 classes cost the same) and wrong sizes on
// another CU count.  Now: the sizes where a forward's cost steps are where the launch geometry
// changes -- asked of the same functions that pick it (shape_lstm / pick_form) -- and a piece's cost is summed over its
// launches from their geometry: an LSTM launch = rounds x T steps x (waves per SIMD x k-groups x accumulator tiles per wave x
// 256 cycles of fp32 MFMA + the step's fixed part) + the launch's fixed part.  The constants are the measured ones of
// DESIGN.md 3 / 3b (one accumulator tile x one k-group = 4 MFMAs = 256 cycles; cell phase 5.1 k cycles per 8 tiles; hand-off
// 3.1 k cycles per clustered step, + 2.4 k where no x part hides the hop; ~14 us per clustered launch with its clean-up launch);
// against profiles/r5 the sums land within 4 % (tests/test_forward_plan.py holds the table).  Only RATIOS matter: a cut never
// changes a bit of the result.
constexpr double kPlanClockGHz = 2.35;
constexpr double kPlanCutPenaltyUs = 40.0;   // a cut must pay for itself clearly

double lstm_launch_us(const dsp_model* m, const LstmArgs& a, long long NTp) {
    const int nq_live = a.NQ - a.nqx_lo;           // leading k-groups of pure padding issue no MFMAs (the front ends)
    const double tile_kgroup = 256.0;
    const long long cus = m->n_cus;
    double step = 0, fixed_us = 6.0;
    long long rounds = 1;
    if (a.CG > 0) {
        const bool local8 = a.CG == 4 && a.UT == 8 && (a.flags & 8);
        const bool local = (a.CG == 4 && a.UT == 4) || local8;
        if (local) {
            step = (local8 ? 2.0 : 1.0) * nq_live * 4 * tile_kgroup + (local8 ? 10000.0 : 4000.0);
            rounds = (NTp * 2 + cus - 1) / cus;
        } else {
            const bool xshort = (a.Ipad >> 3) == 4;
            step = (double)(nq_live - a.xs) * a.CG * tile_kgroup + 3100.0 + (xshort ? 2400.0 : 0.0);
            fixed_us = 14.0;                        // (with the clean-up launch behind it)
            if (a.xs) {
                // x ahead (an estimate, never timed): one wave per (live cluster, step, unit tile, gate), two of them sharing a
                // SIMD's matrix pipe; what is left of the hop shows (one ring of x part instead of most of a step in front of it)
                const long long waves = (a.n + 31) / 32 * 2 * a.T * a.UT * 4;
                const long long per_simd = (waves + cus * 4 - 1) / (cus * 4);
                step += 1200.0;
                fixed_us += 6.0 + (double)per_simd * a.xs * tile_kgroup / (kPlanClockGHz * 1e3);
            }
        }
    } else if ((a.flags & 2) && a.NP <= 1 && a.UT >= 2 && a.UT % 2 == 0 && a.nqx_lo == 0 && a.nqx_used == (a.Ipad >> 3) &&
               a.NQ == ((a.Ipad + a.Hp) >> 3) && (a.Ipad >> 3) >= 4) {   // dsp_lstm21_kernel (dsp_k_lstm's condition)
        const int sg = a.UT >= 8 ? 1 : 8 / a.UT;
        const long long wgs = NTp / sg * 2;
        step = (double)a.NQ * 8 * tile_kgroup + 6000.0;
        rounds = (wgs + cus - 1) / cus;
    } else {                                        // dsp_lstm_kernel: a wave = 1 unit tile x 4 gates x 2 site tiles per pass
        const int np = a.NP < 1 ? 1 : a.NP;
        const int waves = (a.UT / np) * a.SG;
        const int per_simd = (waves + 3) / 4;
        const long long wgs = NTp / (a.SG * 2) * 2;
        step = (double)np * per_simd * ((double)nq_live * 8 * tile_kgroup + 5100.0);
        // (two 4-wave workgroups share a CU and its matrix pipes: their k-loops add up)
        rounds = (wgs + cus - 1) / cus;
    }
    return (double)rounds * a.T * step / (kPlanClockGHz * 1e3) + fixed_us;
}

// microseconds of one forward of `sites` sites as ONE piece on this handle (fp32 path)
double piece_cost_us(const dsp_model* m, long long sites) {
    const Dims& d = m->d;
    const long long NTp = (((sites + 31) / 32) + 15) / 16 * 16;
    const bool two = d.hseq && d.hsig && d.l2 == 1 && m->hseq_p <= 256 && m->hsig_p <= 256 &&
                     (m->two_streams > 0 || (m->two_streams < 0 && NTp * 4 <= (long long)m->n_cus));
    auto stack = [&](const std::vector<LayerShape>& shapes, int lstm_id, bool side_by_side) {
        double us = 0;
        for (const LayerShape& ly : shapes) {
            LstmArgs a{};
            a.n = sites;
            shape_lstm(m, ly, lstm_id, NTp, a);
            pick_form(m, a, NTp, false, side_by_side, m->cluster != 0);
            us += lstm_launch_us(m, a, NTp) + 2.0;
        }
        return us;
    };
    const double k_sites = (double)NTp * 32 / 1000.0;
    const double seq = d.hseq ? stack(m->seq_shape, 0, two) : 0, sig = d.hsig ? stack(m->sig_shape, 1, two) : 0;
    double us = 10.0 + 1.0 * k_sites;                                   // pack
    us += two ? std::max(seq, sig) : seq + sig;                         // the front ends (side by side on small batches)
    const bool small_fc = NTp * 2 <= (long long)m->n_cus;
    const double fc_flops = (double)d.T * 2.0 * ((double)m->hseq_p * 2 * m->hseq_p + (double)m->hsig_p * 2 * m->hsig_p) * (double)NTp * 32;
    us += (d.hseq ? (small_fc ? 8.0 : 4.0) : 0) + (d.hsig ? (small_fc ? 8.0 : 4.0) : 0) + fc_flops / (130e6);   // fc_seq / fc_signal at ~130 TFLOP/s
    us += stack(m->comb_shape, 2, false);
    us += 25.0 + 1.6 * k_sites;                                         // head
    return us;
}

// The sizes (in sites) at which the launch geometry of a forward changes on this device: where the combined stack's cluster
// size steps down (P = 8 / 4 / 2: NTp x 16 / 8 / 4 <= CUs), where the workgroup-local forms end (NTp x 2 <= CUs), and one
// round of the 64-site workgroups (one per CU); tile counts are padded to 16, so every class is a multiple of 512 sites.
int class_caps(const dsp_model* m, long long caps[5]) {
    const long long cus = m->n_cus;
    const long long cand[5] = {cus / 16, cus / 8, cus / 4, cus / 2, cus};
    int n = 0;
    for (long long c : cand) {
        const long long tiles = c / 16 * 16;
        if (tiles >= 16 && (n == 0 || tiles * 32 > caps[n - 1])) caps[n++] = tiles * 32;
    }
    return n;
}
long long round_sites(const dsp_model* m) { const long long t = (long long)m->n_cus / 16 * 16; return (t >= 16 ? t : 16) * 32; }

// the remainder r (< one round) of a call as the cheapest sequence of pieces, largest first (the partial piece is the smallest)
struct PiecePlan { int n; long long sites[8]; bool tail_round; };   // tail_round: the one piece is of the round class
PiecePlan plan_remainder(const dsp_model* m, long long r) {
    PiecePlan pl{0, {0, 0, 0, 0, 0, 0, 0, 0}, false};
    if (r <= 0) return pl;
    long long caps[5];
    const int nc = class_caps(m, caps);
    double cost[5];
    for (int c = 0; c < nc; ++c) cost[c] = piece_cost_us(m, caps[c]) + kPlanCutPenaltyUs;
    const int kUnit = 512;
    const int u = (int)((r + kUnit - 1) / kUnit);
    std::vector<double> best((size_t)u + 1, 0.0);
    std::vector<int> pick((size_t)u + 1, -1);
    for (int k = 1; k <= u; ++k) {
        best[(size_t)k] = 1e300;
        for (int c = 0; c < nc; ++c) {
            const int cap = (int)(caps[c] / kUnit);
            const int rest = k > cap ? k - cap : 0;
            const double t = cost[c] + best[(size_t)rest];
            if (t < best[(size_t)k] - 1e-9) { best[(size_t)k] = t; pick[(size_t)k] = c; }
        }
    }
    long long got[8]; int n = 0;
    for (int k = u; k > 0 && n < 7;) {
        const int cap = (int)(caps[pick[(size_t)k]] / kUnit);
        got[n++] = caps[pick[(size_t)k]];
        k = k > cap ? k - cap : 0;
        if (n == 7 && k > 0) { n = 0; break; }   // (more pieces than the plan holds: one piece)
    }
    if (n == 0) { pl.n = 1; pl.sites[0] = r; pl.tail_round = nc < 2 || r > caps[nc - 2]; return pl; }
    pl.tail_round = n == 1 && got[0] == caps[nc - 1] && nc >= 2;
    std::sort(got, got + n, [](long long x, long long y) { return x > y; });
    long long left = r;
    for (int i = 0; i < n && left > 0; ++i) {
        const long long len = (i + 1 == n) ? left : std::min(left, got[i]);
        pl.sites[pl.n++] = len;
        left -= len;
    }
    return pl;
}
}  // namespace

static int32_t forward_pieces(dsp_model* m, void* stream, int64_t n, const void* kmer, int32_t kmer_dtype, const float* means,
                              const float* stds, const void* lens, int32_t lens_dtype, const float* signals,
                              const dsp_init_state* init, float* logits, float* probs, uint8_t* labels);

// The bounds-recording build (make bounds -> libdsp_amd_bounds.so; never the product library): every access through a buffer
// descriptor -- and the flat stores of the pack / fc launches, the counters of the clustered launches -- was compared with the
// tight extent of its operand; the first one out of range comes back here as DSP_EBOUNDS with the kernel source line, the
// operand, the workgroup, the thread, the offset and the extent.  The forward is synchronous in that build.
static int32_t bounds_verdict(dsp_model* m, void* stream) {
    if (!dsp_k_bounds_build()) return 0;
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    if (prev != m->device) HIP_TRY(hipSetDevice(m->device));
    const hipError_t se = hipStreamSynchronize((hipStream_t)stream);
    unsigned rec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int re = se == hipSuccess ? dsp_k_bounds_read(rec) : (int)se;
    if (prev != m->device) hipSetDevice(prev);
    if (re) return fail(DSP_EHIP, "bounds build: reading the record failed: %s", hipGetErrorString((hipError_t)re));
    if (!rec[0]) return 0;
    static const char* kinds[] = {"?", "weights", "K4 input", "K4 output", "h0 scratch", "cell-state scratch", "cluster counters", "flat K4 store",
                                  "x-ahead sums"};
    return fail(DSP_EBOUNDS, "%u access(es) out of range; the first: dsp_kernels.hip:%u, operand %s, workgroup %u, thread %u, byte offset %llu "
                "(+16) against an extent of %u bytes", rec[0], rec[1], kinds[rec[2] < 9 ? rec[2] : 0], rec[3], rec[4],
                (unsigned long long)rec[5] | ((unsigned long long)rec[6] << 32), rec[7]);
}

int32_t dsp_forward(dsp_model* m, void* stream, int64_t n, const void* kmer, int32_t kmer_dtype, const float* means,
                    const float* stds, const void* lens, int32_t lens_dtype, const float* signals,
                    const dsp_init_state* init, float* logits, float* probs, uint8_t* labels) {
    if (!m) return fail(DSP_EINVAL, "model is NULL");
    const int32_t rc = forward_pieces(m, stream, n, kmer, kmer_dtype, means, stds, lens, lens_dtype, signals, init, logits, probs, labels);
    if (rc) return rc;
    return bounds_verdict(m, stream);
}

static int32_t forward_pieces(dsp_model* m, void* stream, int64_t n, const void* kmer, int32_t kmer_dtype, const float* means,
                              const float* stds, const void* lens, int32_t lens_dtype, const float* signals,
                              const dsp_init_state* init, float* logits, float* probs, uint8_t* labels) {
    if (n < 0) return fail(DSP_EINVAL, "n_sites < 0");
    const int mode = init ? init->mode : DSP_INIT_ZEROS;
    const int64_t kRound = round_sites(m);
    const int64_t whole = n / kRound * kRound, r = n - whole;
    const bool may_cut = m->forward_split && mode != DSP_INIT_EXPLICIT && m->precision == DSP_PREC_FP32 && m->cluster != 0 && kmer_dtype >= 0 &&
                         kmer_dtype <= 3 && lens_dtype >= 0 && lens_dtype <= 3;
    PiecePlan pl{0, {0, 0, 0, 0, 0, 0, 0, 0}, false};
    if (may_cut && r > 0) pl = plan_remainder(m, r);
    // (nothing to cut: the call is whole rounds, or at most one round whose cheapest cover is one piece; a remainder whose one
    // piece is itself of the round class runs with the whole rounds in front of it)
    const bool cut = may_cut && (whole > 0 ? (r > 0 && !pl.tail_round) : pl.n > 1);
    m->last_split = cut;
    if (!cut) return forward_chunk(m, stream, n, kmer, kmer_dtype, means, stds, lens, lens_dtype, signals, init, logits, probs, labels);
    static const size_t dt_size[4] = {4, 1, 2, 4};   // DSP_DT_F32, U8, U16, I32
    const Dims& d = m->d;
    int64_t pieces[9];
    int np = 0;
    if (whole > 0) pieces[np++] = whole;
    for (int i = 0; i < pl.n; ++i) pieces[np++] = pl.sites[i];
    int64_t at = 0;
    for (int i = 0; i < np; ++i) {
        const int64_t len = pieces[i];
        if (len <= 0) continue;
        dsp_init_state st{};
        if (init) {
            st = *init;
            st.site_offset = init->site_offset + (uint64_t)at;
            if (init->site_keys) st.site_keys = init->site_keys + at;
        }
        const size_t row = (size_t)at * d.T;
        const int32_t rc = forward_chunk(
            m, stream, len, kmer ? (const char*)kmer + row * dt_size[kmer_dtype] : nullptr, kmer_dtype, means ? means + row : nullptr,
            stds ? stds + row : nullptr, lens ? (const char*)lens + row * dt_size[lens_dtype] : nullptr, lens_dtype,
            signals ? signals + row * d.S : nullptr, init ? &st : nullptr, logits ? logits + (size_t)at * d.C : nullptr,
            probs ? probs + (size_t)at * d.C : nullptr, labels ? labels + at : nullptr);
        if (rc) return rc;
        at += len;
    }
    return 0;
}

static int32_t forward_chunk(dsp_model* m, void* stream, int64_t n, const void* kmer, int32_t kmer_dtype, const float* means,
                             const float* stds, const void* lens, int32_t lens_dtype, const float* signals,
                             const dsp_init_state* init, float* logits, float* probs, uint8_t* labels) {
    if (n == 0) return 0;
    const Dims& d = m->d;
    if (kmer_dtype < 0 || kmer_dtype > 3 || lens_dtype < 0 || lens_dtype > 3) return fail(DSP_EINVAL, "bad dtype code");
    if (d.hseq && (!means || !stds || (d.is_base && !kmer) || (d.is_siglen && !lens)))
        return fail(DSP_EINVAL, "seq-branch input pointer is NULL");
    if (d.hsig && !signals) return fail(DSP_EINVAL, "signals pointer is NULL");
    const int mode = init ? init->mode : DSP_INIT_ZEROS;
    if (mode < 0 || mode > 2) return fail(DSP_EINVAL, "bad init-state mode %d", mode);
    if (mode == DSP_INIT_EXPLICIT) {
        if ((d.hseq && (!init->h_seq || !init->c_seq)) || (d.hsig && (!init->h_sig || !init->c_sig)) || !init->h_comb ||
            !init->c_comb)
            return fail(DSP_EINVAL, "explicit init-state pointer is NULL");
    }
    hipStream_t s = (hipStream_t)stream;
    int prev = m->device;
    if (!m->dry) {
        HIP_TRY(hipGetDevice(&prev));
        if (prev != m->device) HIP_TRY(hipSetDevice(m->device));
        (void)hipGetLastError();   // whatever an earlier, unrelated call of this thread left behind is not this forward's error
    }
    int rc = ensure_ws(m, n, s);
    if (rc) { if (prev != m->device) hipSetDevice(prev); return rc; }
    // use the tile count of THIS call (padded to 16 tiles), not the capacity
    const long long NTp = (((n + 31) / 32) + 15) / 16 * 16;

    // profiling entries accumulate across forwards until dsp_profile_read() drains them
    if (!m->prof) { m->prof_entries.clear(); m->event_used = 0; }
    Launcher L{m, s, NTp};
    m->lstm_launch_no = 0;

    PackArgs p{};
    p.kmer = kmer; p.means = means; p.stds = stds; p.lens = lens; p.signals = signals; p.embed = m->embed;
    p.xseq = d.hseq ? m->xseq : nullptr; p.xsig = d.hsig ? m->xsig : nullptr;
    p.n = n; p.NTp = NTp; p.kdt = kmer_dtype; p.ldt = lens_dtype;
    p.T = d.T; p.S = d.S; p.E = d.E; p.V = d.V; p.is_base = d.is_base; p.is_siglen = d.is_siglen;
    p.Fseq = m->Fseq; p.Fsig = m->Fsig; p.xoff_seq = m->xoff_seq; p.xoff_sig = m->xoff_sig;
    if (m->cluster != 0 && NTp * 4 <= (long long)m->n_cus) {   // this forward may run clustered LSTM launches (the combined stack
        // from P = 2: <= 2,048 sites; the front ends from P = 2 as well): their arrival counters start from zero
        const int launches = std::min(kClusterLaunches, (d.hseq ? d.l2 : 0) + (d.hsig ? d.l2 : 0) + d.l1);
        p.zero_words = m->cflags;
        p.n_zero_words = launches * kClusterWordsPerLaunch;
        L.counters_zeroed = true;
    }
    p.xseq_end = rsrc_end(m, p.xseq, (size_t)NTp * d.T * (size_t)(m->Fseq >> 2) * 512);
    p.xsig_end = rsrc_end(m, p.xsig, (size_t)NTp * d.T * (size_t)(m->Fsig >> 2) * 512);
    p.zero_words_end = p.zero_words ? (const char*)p.zero_words + (size_t)(m->extents == 2 ? p.n_zero_words : m->n_cflag_words) * sizeof(unsigned int)
                                    : nullptr;
    L.run("pack", [&] { return dsp_k_pack(&p, s); });

    // (fc2 != NULL: both branches' projections in ONE launch -- same shape, disjoint output columns)
    auto linear = [&](const char* name, const DevLinear& fc, const float* x, int out_off, const DevLinear* fc2 = nullptr,
                      const float* x2 = nullptr, int out_off2 = 0) {
        LinArgs a{};
        a.x = x; a.out = m->comb_in; a.wpk = fc.wpk; a.bias = fc.bias;
        a.ncols = NTp * d.T; a.Fin = fc.Fin; a.Fout = m->Fcomb; a.out_off = out_off; a.ORT = fc.ORT; a.relu = 1;
        if (fc2) { a.x2 = x2; a.wpk2 = fc2->wpk; a.bias2 = fc2->bias; a.out_off2 = out_off2; }
        // batches of <= 4,096 sites: one accumulator tile per wave (round 5: 512 sites 35 -> ~12 us per projection; the
        // 8-tile wave of dsp_linear_kernel runs 28 us whatever the batch).  DSP_FC_SMALL=0 turns it off (A/B switch)
        a.small = (m->fc_small && NTp * 2 <= (long long)m->n_cus) ? 1 : 0;
        const size_t xbytes = (size_t)a.ncols * (size_t)(a.Fin >> 2) * 512, wbytes = (size_t)a.ORT * (size_t)(a.Fin >> 3) * 1024;
        a.x_end = rsrc_end(m, a.x, xbytes); a.wpk_end = rsrc_end(m, a.wpk, wbytes);
        a.x2_end = rsrc_end(m, a.x2, xbytes); a.wpk2_end = rsrc_end(m, a.wpk2, wbytes);
        a.out_end = rsrc_end(m, a.out, (size_t)a.ncols * (size_t)(a.Fout >> 2) * 512);
        L.run(name, [&] { return dsp_k_linear(&a, L.s); });
    };
    // The seq and the signal branch are independent until the combined stack (models.py:181-217).  On batches that leave
    // CUs idle the signal branch runs on the handle's side stream next to the seq branch (fork / join by events: 512 sites
    // 0.81 -> 0.66 ms per forward); one layer each (the default), so that the two stacks need one output buffer each
    const bool two = (m->side || m->dry) && d.hseq && d.hsig && d.l2 == 1 && m->hseq_p <= 256 && m->hsig_p <= 256 && !m->prof_serial &&
                     (m->two_streams > 0 || (m->two_streams < 0 && NTp * 4 <= (long long)m->n_cus));   // (<= 2,048 sites: at 4,096 every CU
    // is busy with one branch already -- 3.458 vs 3.453 ms in sequence)
    L.side_by_side = two;
    if (two && !m->dry) {
        if (hipEventRecord(m->ev_fork, s) != hipSuccess || hipStreamWaitEvent(m->side, m->ev_fork, 0) != hipSuccess)
            L.rc = fail(DSP_EHIP, "fork to the side stream failed: %s", hipGetErrorString(hipGetLastError()));
    }
    // one stream: fc_seq and fc_signal of the same shape share a launch behind both stacks (the two stacks then need an
    // output buffer each, as with two streams: one layer per stack)
    const bool fc_fused = m->fc_fused && !two && d.hseq && d.hsig && d.l2 == 1 && m->hseq_p <= 256 && m->hsig_p <= 256 &&
                          m->fc_seq.Fin == m->fc_sig.Fin && m->fc_seq.ORT == m->fc_sig.ORT && !m->prof_serial;
    float* oseq = nullptr;
    if (d.hseq) {
        oseq = run_stack(L, "lstm_seq", m->seq, 0, m->xseq, n, init, init ? init->h_seq : nullptr,
                         init ? init->c_seq : nullptr);
        if (!fc_fused) linear("fc_seq", m->fc_seq, oseq, 0);
    }
    if (d.hsig) {
        if (two) L.s = m->dry ? s : m->side;
        float* o = run_stack(L, "lstm_signal", m->sig, 1, m->xsig, n, init, init ? init->h_sig : nullptr,
                             init ? init->c_sig : nullptr, two || fc_fused);
        if (fc_fused) linear("fc_seq+fc_signal", m->fc_seq, oseq, 0, &m->fc_sig, o, m->hseq_p);
        else linear("fc_signal", m->fc_sig, o, m->hseq_p);
        if (two) {
            L.s = s;
            if (!L.rc && !m->dry && (hipEventRecord(m->ev_join, m->side) != hipSuccess || hipStreamWaitEvent(s, m->ev_join, 0) != hipSuccess))
                L.rc = fail(DSP_EHIP, "join of the side stream failed: %s", hipGetErrorString(hipGetLastError()));
        }
    }
    float* o = run_stack(L, "lstm_comb", m->comb, 2, m->comb_in, n, init, init ? init->h_comb : nullptr,
                         init ? init->c_comb : nullptr);
    m->last_out = o;
    HeadArgs h{};
    h.x = o; h.w1pk = m->fc1.wpk; h.b1 = m->fc1.bias; h.w2 = m->w2; h.b2 = m->b2;
    h.logits = logits; h.probs = probs; h.labels = labels; h.n = n; h.Hp = m->Hp; h.T = d.T; h.C = d.C;
    h.flags = m->head_st4 ? 1 : 0;
    h.x_end = rsrc_end(m, h.x, (size_t)NTp * d.T * (size_t)((2 * m->Hp) >> 2) * 512);
    h.w1pk_end = rsrc_end(m, h.w1pk, (size_t)(m->Hp >> 5) * (size_t)((2 * m->Hp) >> 3) * 1024);
    L.run("head", [&] { return dsp_k_head(&h, s); });
    L.close_bracket();

    if (prev != m->device) hipSetDevice(prev);
    return L.rc;
}

int32_t dsp_debug_read_activation(dsp_model* m, void* stream, int32_t which, int64_t n, float* host_out) {
    if (!m || !host_out) return fail(DSP_EINVAL, "NULL argument");
    if (!m->ws || !m->last_out) return fail(DSP_EINVAL, "no forward has run yet");
    if (m->last_split) return fail(DSP_EINVAL, "the last forward ran in several pieces (more than 4,096 sites, not a whole number of rounds): "
                                               "its activations are not in the scratch as one batch (DSP_FORWARD_SPLIT=0)");
    if (n < 1 || n > m->ws_sites) return fail(DSP_EINVAL, "n_sites out of range");
    const Dims& d = m->d;
    const float* src; int F; std::vector<int> map; int Fref;
    if (which == 0) { src = m->comb_in; F = m->Fcomb; map = m->comb_in_map; Fref = d.H; }
    else if (which == 1) { src = m->last_out; F = 2 * m->Hp; map = map_bidir(d.H, m->Hp); Fref = 2 * d.H; }
    else return fail(DSP_EINVAL, "which must be 0 or 1");
    const long long nt = (n + 31) / 32;
    std::vector<float> tmp((size_t)nt * d.T * F * 32);
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(tmp.data(), src, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
    hipSetDevice(prev);
    for (long long site = 0; site < n; ++site)
        for (int t = 0; t < d.T; ++t)
            for (int f = 0; f < F; ++f) {
                if (map[f] < 0) continue;
                const size_t k4 = ((((size_t)(site / 32) * d.T + t) * (F / 4) + f / 4) * 32 + site % 32) * 4 + f % 4;
                host_out[((size_t)site * d.T + t) * Fref + map[f]] = tmp[k4];
            }
    return 0;
}

int32_t dsp_debug_plan(const dsp_model_cfg* cfg, int32_t n_cus, int64_t n_sites, int64_t pieces[9], double cost_us[9]) {
    Dims d;
    const int rc = derive(cfg, &d);
    if (rc) return rc;
    if (n_cus < 16 || n_sites < 0 || !pieces) return fail(DSP_EINVAL, "dsp_debug_plan: bad arguments");
    dsp_model host;                       // never sees a device: geometry and plan only
    host.cfg = *cfg; host.d = d; host.n_cus = n_cus;
    derive_geometry(&host);
    const int64_t round = round_sites(&host), whole = n_sites / round * round, r = n_sites - whole;
    PiecePlan pl{0, {0, 0, 0, 0, 0, 0, 0, 0}, false};
    if (r > 0) pl = plan_remainder(&host, r);
    int np = 0;
    const bool cut = whole > 0 ? (r > 0 && !pl.tail_round) : pl.n > 1;
    if (!cut) { if (n_sites > 0) pieces[np++] = n_sites; }
    else {
        if (whole > 0) pieces[np++] = whole;
        for (int i = 0; i < pl.n; ++i) pieces[np++] = pl.sites[i];
    }
    if (cost_us) for (int i = 0; i < np; ++i) cost_us[i] = piece_cost_us(&host, pieces[i]);
    return np;
}

int32_t dsp_debug_dry_run(const dsp_model_cfg* cfg, int32_t n_cus, int64_t n_sites, int32_t init_mode, int32_t precision, const char* extents,
                          char* log, size_t log_cap) {
    Dims d;
    int rc = derive(cfg, &d);
    if (rc) return rc;
    if (n_cus < 16 || n_sites < 0 || init_mode < 0 || init_mode > 2) return fail(DSP_EINVAL, "dsp_debug_dry_run: bad arguments");
    dsp_model* m = new (std::nothrow) dsp_model();
    if (!m) return fail(DSP_ENOMEM, "out of host memory");
    m->dry = true;
    m->cfg = *cfg; m->d = d; m->n_cus = n_cus;
    m->extents = extents && !strcmp(extents, "tight") ? 2 : (extents && !strcmp(extents, "wide") ? 0 : 1);
    read_xahead_switch(m);
    derive_geometry(m);
    m->cflags = (unsigned int*)dry_alloc(m, (size_t)kClusterLaunches * kClusterWordsPerLaunch * sizeof(unsigned int));
    m->n_cflag_words = kClusterLaunches * kClusterWordsPerLaunch;
    // weights: the spec's tensors, all zero (the geometry of a forward does not depend on their values)
    const std::vector<Spec> spec = weight_spec(d);
    std::vector<std::vector<float>> zeros;
    std::vector<const float*> ptrs;
    for (const Spec& sp : spec) { zeros.emplace_back((size_t)spec_numel(sp), 0.f); ptrs.push_back(zeros.back().data()); }
    m->precision = precision;
    rc = build_weights(m, ptrs.data());
    if (!rc && precision != DSP_PREC_FP32) rc = ensure_split(m);
    std::string notes;
    int launches = 0;
    if (!rc) {
        dsp_init_state st{};
        st.mode = init_mode; st.seed = 1; st.site_offset = 0;
        void* fake = (void*)0x1000;   // explicit states / inputs / outputs: named, never touched
        if (init_mode == DSP_INIT_EXPLICIT) { st.h_seq = st.c_seq = st.h_sig = st.c_sig = st.h_comb = st.c_comb = (const float*)fake; }
        dsp_k_set_dry(1, &notes);
        rc = forward_pieces(m, nullptr, n_sites, fake, DSP_DT_F32, (const float*)fake, (const float*)fake, fake, DSP_DT_F32, (const float*)fake,
                            &st, (float*)fake, (float*)fake, (uint8_t*)fake);
        dsp_k_set_dry(0, nullptr);
        for (char c : notes) launches += c == '\n';
    }
    if (log && log_cap) {
        const size_t k = notes.size() < log_cap - 1 ? notes.size() : log_cap - 1;
        memcpy(log, notes.data(), k);
        log[k] = 0;
    }
    delete m;   // (nothing of a dry handle lives on a device)
    return rc ? rc : launches;
}

void dsp_debug_split_bf16(const float* x, int64_t n, uint16_t* hi, uint16_t* mid, uint16_t* lo) {
    for (int64_t i = 0; i < n; ++i) {   // (exactly the statements of pack_lstm_dir_split's bf16 branch)
        hi[i] = bf16_rne(x[i]);
        const float r1 = x[i] - bf16_to_f32(hi[i]);
        mid[i] = bf16_rne(r1);
        lo[i] = bf16_rne(r1 - bf16_to_f32(mid[i]));
    }
}

double dsp_debug_piece_cost(const dsp_model_cfg* cfg, int32_t n_cus, int64_t n_sites) {
    Dims d;
    if (derive(cfg, &d) || n_cus < 16 || n_sites < 1) return -1.0;
    dsp_model host;
    host.cfg = *cfg; host.d = d; host.n_cus = n_cus;
    read_xahead_switch(&host);
    derive_geometry(&host);
    return piece_cost_us(&host, n_sites);
}

int32_t dsp_debug_range_probe(int32_t device, int32_t out[4]) {
    if (!out) return fail(DSP_EINVAL, "out is NULL");
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(device));
    float* buf = nullptr;
    unsigned* cnt = nullptr;
    std::vector<float> ones(1024, 1.0f), back(1024, 0.f);
    unsigned host[3] = {0, 0, 0};
    hipError_t e = hipMalloc((void**)&buf, 4096);
    if (e == hipSuccess) e = hipMalloc((void**)&cnt, sizeof host);
    if (e == hipSuccess) e = hipMemcpy(buf, ones.data(), 4096, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(cnt, 0, sizeof host);
    if (e == hipSuccess) e = (hipError_t)dsp_k_range_probe(buf, cnt, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(host, cnt, sizeof host, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(back.data(), buf, 4096, hipMemcpyDeviceToHost);
    if (buf) hipFree(buf);
    if (cnt) hipFree(cnt);
    hipSetDevice(prev);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(DSP_EHIP, "dsp_debug_range_probe: %s", hipGetErrorString(e)); }
    int untouched = 0;
    for (float v : back) untouched += v == 1.0f ? 1 : 0;
    out[0] = (int32_t)host[0]; out[1] = (int32_t)host[1]; out[2] = (int32_t)host[2]; out[3] = untouched;
    return 0;
}

int32_t dsp_profile_enable(dsp_model* m, int32_t on) {
    if (!m) return fail(DSP_EINVAL, "model is NULL");
    m->prof = on != 0;
    m->prof_dominant_only = on == 2;
    return 0;
}

int32_t dsp_profile_read(dsp_model* m, char* names, size_t names_cap, float* ms, int32_t cap) {
    if (!m) return fail(DSP_EINVAL, "model is NULL");
    size_t pos = 0;
    int32_t k = 0;
    for (const ProfEntry& e : m->prof_entries) {
        if (k >= cap) break;
        float t = 0.f;
        hipError_t err = hipEventElapsedTime(&t, e.a, e.b);
        if (err != hipSuccess) return fail(DSP_EHIP, "hipEventElapsedTime: %s (synchronise the stream first)", hipGetErrorString(err));
        ms[k] = t / (float)(e.share > 0 ? e.share : 1);
        const size_t len = strlen(e.name) + 1;
        if (names && pos + len <= names_cap) { memcpy(names + pos, e.name, len); pos += len; }
        ++k;
    }
    m->prof_entries.clear();
    m->event_used = 0;
    return k;
}

int32_t dsp_model_set_precision(dsp_model* m, int32_t precision) {
    if (!m || (precision != DSP_PREC_FP32 && precision != DSP_PREC_BF16X6 && precision != DSP_PREC_BF16X9 &&
               precision != DSP_PREC_FP16X3))
        return fail(DSP_EINVAL, "dsp_model_set_precision: bad arguments");
    if (precision == DSP_PREC_FP16X3 && !m->fp16_safe)
        return fail(DSP_EINVAL, "dsp_model_set_precision: this checkpoint's combined-stack operands are not provably inside the "
                                "fp16 range (weights or fc outputs above 3e4): use bf16x6");
    if (precision != DSP_PREC_FP32) {
        const int rc = ensure_split(m);
        if (rc) return rc;
    }
    m->precision = precision;
    return DSP_OK;
}

int32_t dsp_model_query(const dsp_model* m, int32_t what) {
    if (!m) return fail(DSP_EINVAL, "model is NULL");
    switch (what) {
        case DSP_QUERY_CLUSTERING: return m->cluster != 0 ? 1 : 0;
        case DSP_QUERY_XCC_PROBE_FAILED: return m->xcc_probe_failed ? 1 : 0;
        case DSP_QUERY_COMPUTE_UNITS: return m->n_cus;
        default: return fail(DSP_EINVAL, "dsp_model_query: unknown item %d", (int)what);
    }
}

int64_t dsp_device_pci_bdf(int32_t device, char* out, size_t cap) {
    char buf[64] = {0};
    if (hipDeviceGetPCIBusId(buf, (int)sizeof buf, device) != hipSuccess) {
        (void)hipGetLastError();
        return fail(DSP_EHIP, "dsp_device_pci_bdf: the HIP runtime does not know device %d", (int)device);
    }
    size_t k = strlen(buf);
    for (size_t i = 0; i < k; ++i) buf[i] = (char)tolower((unsigned char)buf[i]);
    if (out && cap) {
        const size_t c = k < cap - 1 ? k : cap - 1;
        memcpy(out, buf, c);
        out[c] = 0;
    }
    return (int64_t)k;
}

int64_t dsp_device_uuid(int32_t device, char* out, size_t cap) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        (void)hipGetLastError();
        return fail(DSP_EHIP, "dsp_device_uuid: the HIP runtime does not know device %d", (int)device);
    }
    char buf[33];
    static const char* hex = "0123456789abcdef";
    for (int i = 0; i < 16; ++i) {
        const unsigned char b = (unsigned char)prop.uuid.bytes[i];
        buf[2 * i] = hex[b >> 4];
        buf[2 * i + 1] = hex[b & 15];
    }
    buf[32] = 0;
    if (out && cap) {
        const size_t c = 32 < cap - 1 ? 32 : cap - 1;
        memcpy(out, buf, c);
        out[c] = 0;
    }
    return 32;
}

void dsp_model_destroy(dsp_model* m) {
    if (!m) return;
    int prev = 0;
    hipGetDevice(&prev);
    hipSetDevice(m->device);
    for (void* p : m->dev_allocs) hipFree(p);
    if (m->ws) hipFree(m->ws);
    for (hipEvent_t e : m->event_pool) hipEventDestroy(e);
    if (m->ev_fork) hipEventDestroy(m->ev_fork);
    if (m->ev_join) hipEventDestroy(m->ev_join);
    if (m->side) hipStreamDestroy(m->side);
    hipSetDevice(prev);
    delete m;
}

}  // extern "C"
