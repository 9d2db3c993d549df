// emu_asan_driver.cpp -- the kernels of csrc/dsp_kernels.hip under AddressSanitizer + UBSan (round 6).
// GPU sanitizers are not available on this pool; the test-suite's SIMT interpreter (tests/native/emu) is a HOST build of the
// kernels, every "device" allocation of which is a malloc block with ASan's red zones around it: an access of any kernel past
// the end of the workspace, of a weight upload, of the caller's input or output arrays -- through a buffer descriptor or through
// a plain pointer -- is reported with the kernel's source line.  TEST INFRASTRUCTURE (linked from the sources, never shipped).
//
// To take the descriptors' own range check out of the way (it would drop exactly the accesses ASan should see), the forwards run
// with DSP_RSRC_EXTENTS=wide -- the 2 GiB windows of rounds 1-5 -- first, then with the default extents; both must give the same
// bytes.  Shapes: every kernel form of the forward (front ends of 1 / 2 / 4 unit tiles, combined stacks of 2 / 4 / 8 / 10 unit
// tiles, split precision), batch sizes with tile tails, one call cut into pieces, explicit / Philox / zero states, the
// small-batch switch matrix incl. abandoned clusters, the opt-in "x ahead" forms.  usage: emu_asan_driver [quick]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dsp_amd.h"

namespace {

struct Rng {
    uint64_t s;
    float uni() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xffffff) / 16777216.0f; }
    float sym(float a) { return (2.f * uni() - 1.f) * a; }
};

struct Case { const char* label; dsp_model_cfg cfg; std::vector<long> sizes; };

dsp_model_cfg mk(int T, int S, int l1, int l2, int C, int H, int isb, int isl, int module) {
    dsp_model_cfg c;
    c.seq_len = T; c.signal_len = S; c.num_layers1 = l1; c.num_layers2 = l2; c.num_classes = C; c.hidden_size = H; c.vocab_size = 16; c.embedding_size = 4;
    c.is_base = isb; c.is_signallen = isl; c.module = module;
    return c;
}

long g_forwards = 0;

// one model: create, run every size in three state modes, return the concatenated probabilities
std::vector<float> run_model(const dsp_model_cfg& cfg, const std::vector<long>& sizes, int precision, uint64_t seed) {
    Rng r{seed};
    const int nW = dsp_weight_count(&cfg);
    if (nW <= 0) { fprintf(stderr, "dsp_weight_count: %s\n", dsp_last_error()); exit(1); }
    std::vector<std::vector<float>> w((size_t)nW);
    std::vector<const float*> wp((size_t)nW);
    std::vector<int64_t> numel((size_t)nW);
    for (int i = 0; i < nW; ++i) {
        char name[128]; int64_t shp[2]; int32_t nd;
        dsp_weight_spec(&cfg, i, name, sizeof name, shp, &nd);
        numel[(size_t)i] = nd == 2 ? shp[0] * shp[1] : shp[0];
        w[(size_t)i].resize((size_t)numel[(size_t)i]);     // (exactly sized: ASan sees the packer read past a tensor)
        for (float& v : w[(size_t)i]) v = r.sym(0.15f);
        wp[(size_t)i] = w[(size_t)i].data();
    }
    dsp_model* m = nullptr;
    if (dsp_model_create(&cfg, wp.data(), numel.data(), nW, 0, &m)) { fprintf(stderr, "dsp_model_create: %s\n", dsp_last_error()); exit(1); }
    if (precision && dsp_model_set_precision(m, precision)) { fprintf(stderr, "set_precision: %s\n", dsp_last_error()); exit(1); }
    const int T = cfg.seq_len, S = cfg.signal_len, C = cfg.num_classes, H = cfg.hidden_size;
    const int hseq = cfg.module == DSP_MODULE_BOTH ? H / 2 : (cfg.module == DSP_MODULE_SEQ ? H : 0), hsig = cfg.module == DSP_MODULE_BOTH ? H - H / 2 : (cfg.module == DSP_MODULE_SIGNAL ? H : 0);
    std::vector<float> all;
    for (long n : sizes) {
        std::vector<float> kmer((size_t)n * T), means((size_t)n * T), stds((size_t)n * T), lens((size_t)n * T), sig((size_t)n * T * S);
        for (float& v : kmer) v = (float)(int)(r.uni() * 15.99f);
        for (float& v : means) v = r.sym(2.f);
        for (float& v : stds) v = r.uni();
        for (float& v : lens) v = (float)(2 + (int)(r.uni() * 30));
        for (float& v : sig) v = r.sym(2.f);
        for (int mode = 0; mode < 3; ++mode) {
            dsp_init_state st;
            memset(&st, 0, sizeof st);
            st.mode = mode; st.seed = 5; st.site_offset = 17;
            std::vector<float> hs((size_t)2 * cfg.num_layers2 * n * (hseq ? hseq : 1)), cs(hs.size()), hg((size_t)2 * cfg.num_layers2 * n * (hsig ? hsig : 1)), cg(hg.size()),
                hc((size_t)2 * cfg.num_layers1 * n * H), cc(hc.size());
            if (mode == DSP_INIT_EXPLICIT) {
                for (auto* v : {&hs, &cs, &hg, &cg, &hc, &cc}) for (float& x : *v) x = r.sym(1.f);
                st.h_seq = hs.data(); st.c_seq = cs.data(); st.h_sig = hg.data(); st.c_sig = cg.data(); st.h_comb = hc.data(); st.c_comb = cc.data();
            }
            std::vector<float> logits((size_t)n * C), probs((size_t)n * C);
            std::vector<uint8_t> labels((size_t)n);
            if (dsp_forward(m, nullptr, n, kmer.data(), DSP_DT_F32, means.data(), stds.data(), lens.data(), DSP_DT_F32, sig.data(), &st, logits.data(), probs.data(),
                            labels.data())) {
                fprintf(stderr, "dsp_forward(n = %ld, mode %d): %s\n", n, mode, dsp_last_error());
                exit(1);
            }
            ++g_forwards;
            all.insert(all.end(), probs.begin(), probs.end());
        }
    }
    dsp_model_destroy(m);
    return all;
}

void set_env(const std::vector<std::pair<const char*, const char*>>& kv) {
    static const char* all[] = {"DSP_LSTM_CLUSTER", "DSP_LSTM_TILING", "DSP_LSTM_LOCAL8", "DSP_TWO_STREAMS", "DSP_HEAD_ST4", "DSP_FC_FUSED", "DSP_FC_SMALL", "DSP_LSTM_FRONT_CLUSTER",
                                "DSP_LSTM_HANDOFF", "DSP_CLUSTER_TIMEOUT", "DSP_FORWARD_SPLIT", "DSP_RSRC_EXTENTS", "DSP_EMU_SEED", "EMU_CUS", "DSP_LSTM_XAHEAD",
                                "DSP_LSTM_XAHEAD_RING", "DSP_LSTM_XAHEAD_TILES"};
    for (const char* k : all) unsetenv(k);
    for (const auto& p : kv) setenv(p.first, p.second, 1);
}

}  // namespace

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    std::vector<Case> cases = {
        {"hidden 256: front ends of 4 unit tiles, a combined stack of 8 (every clustered form)", mk(2, 8, 1, 1, 2, 256, 1, 1, DSP_MODULE_BOTH), {1, 45, 513}},
        {"hidden 128 x 2 layers: a combined stack of 4 unit tiles, front ends of 2", mk(2, 8, 2, 1, 2, 128, 1, 1, DSP_MODULE_BOTH), {33, 100}},
        {"hidden 64, two front-end layers, no k-mer / no lengths, three classes", mk(3, 12, 1, 2, 3, 64, 0, 0, DSP_MODULE_BOTH), {31, 64}},
        {"seq only, hidden 256 (configs[2]'s shape): a front end of 8 unit tiles", mk(2, 8, 2, 1, 2, 256, 1, 1, DSP_MODULE_SEQ), {40}},
        {"signal only, a 40-wide window", mk(2, 40, 1, 1, 2, 96, 1, 1, DSP_MODULE_SIGNAL), {50}},
        {"hidden 320: the many-pass kernel", mk(2, 8, 1, 1, 2, 320, 1, 1, DSP_MODULE_BOTH), {37}},
    };
    if (quick) cases.resize(3);
    struct Mode { const char* label; std::vector<std::pair<const char*, const char*>> env; };
    std::vector<Mode> modes = {
        {"auto", {}},
        {"full-batch kernels", {{"DSP_LSTM_CLUSTER", "0"}, {"DSP_LSTM_TILING", "0"}, {"DSP_LSTM_LOCAL8", "0"}, {"DSP_TWO_STREAMS", "0"}, {"DSP_HEAD_ST4", "1"}, {"DSP_FC_SMALL", "0"}}},
        {"every cluster abandoned, adversarial schedule", {{"DSP_CLUSTER_TIMEOUT", "0"}, {"DSP_EMU_SEED", "3"}}},
        {"clusters of 4, round-4 hand-off, 64 compute units", {{"DSP_LSTM_CLUSTER", "2"}, {"DSP_LSTM_HANDOFF", "0"}, {"EMU_CUS", "64"}}},
        // x ahead (dsp_xahead_kernel + the XA form of the clustered launches: calls of <= 8 live tiles, the dense layers of 8 unit tiles)
        {"x ahead", {{"DSP_LSTM_XAHEAD", "1"}}},
        {"x ahead, rings 8 deep, round-4 hand-off, adversarial schedule", {{"DSP_LSTM_XAHEAD", "1"}, {"DSP_LSTM_XAHEAD_RING", "8"}, {"DSP_LSTM_HANDOFF", "0"}, {"DSP_EMU_SEED", "5"}}},
        {"x ahead, clusters of 4, up to 16 live tiles", {{"DSP_LSTM_XAHEAD", "1"}, {"DSP_LSTM_CLUSTER", "2"}, {"DSP_LSTM_XAHEAD_TILES", "16"}}},
    };
    if (quick) { modes.erase(modes.begin() + 3); modes.resize(4); }
    for (const Case& c : cases) {
        std::vector<float> ref;
        for (const Mode& md : modes) {
            for (const char* extents : {"wide", "region"}) {   // wide first: nothing between the kernels' offsets and ASan
                auto kv = md.env;
                kv.push_back({"DSP_RSRC_EXTENTS", extents});
                set_env(kv);
                const std::vector<float> got = run_model(c.cfg, c.sizes, 0, 99);
                if (ref.empty()) ref = got;
                else if (got.size() != ref.size() || memcmp(got.data(), ref.data(), got.size() * 4)) {
                    fprintf(stderr, "%s: mode '%s', extents %s: the probabilities differ from the first run's\n", c.label, md.label, extents);
                    return 1;
                }
            }
        }
        printf("%s: %zu modes x 2 extents modes, bit-identical\n", c.label, modes.size());
        fflush(stdout);
    }
    // split precision (its own kernels), and a call cut into pieces on a small device
    set_env({{"DSP_RSRC_EXTENTS", "wide"}});
    run_model(mk(2, 16, 1, 1, 2, 256, 1, 1, DSP_MODULE_BOTH), {40}, DSP_PREC_BF16X9, 7);
    if (!quick) run_model(mk(2, 16, 1, 1, 2, 256, 1, 1, DSP_MODULE_BOTH), {40}, DSP_PREC_FP16X3, 7);
    set_env({{"DSP_RSRC_EXTENTS", "wide"}, {"EMU_CUS", "32"}});
    run_model(mk(2, 8, 1, 1, 2, 64, 1, 1, DSP_MODULE_BOTH), {1100}, 0, 8);
    printf("emu_asan_driver: ok (%ld forwards)\n", g_forwards);
    return 0;
}
