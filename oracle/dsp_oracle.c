/*
 * ORACLE (test infrastructure, NOT product code) -- plain-C fp32 restatement of the reference's
 * per-site classifier, used (a) as the checker in tests/, __graft_entry__.smoke() and (b) as
 * bench.py's cpu_baseline ("port").  The shipped library never links or calls this file.
 *
 * Restates (paths relative to /root/reference):
 *   ModelBiLSTM.forward            deepsignal_plant/models.py:178-240
 *   ModelBiLSTM.init_hidden        deepsignal_plant/models.py:169-176   (shape/order of h0,c0)
 *   torch.nn.LSTM / Linear / Embedding / Softmax (third-party: torch, pinned by the reference at
 *   torch>=1.2.0,<=1.11.0, requirements.txt:5) -- published semantics: gates packed i,f,g,o;
 *   gates = x W_ih^T + b_ih + h W_hh^T + b_hh; c' = s(f) c + s(i) tanh(g); h' = s(o) tanh(c');
 *   reverse direction runs t=L-1..0 and writes out[:,t]; layer out = [fwd|bwd]; h0[2*layer+dir].
 *
 * Parity pinning: the reference has no tests/golden vectors for this path; this file is pinned by
 * tests/golden/f1_*.npz, generated in the build container by importing the reference
 * (tests/golden/make_golden.py) with the LSTM initial states pinned.
 *
 * Init-state modes: 0 zeros, 1 explicit buffers in the reference layout (2*layers, N, H),
 * 2 counter-based N(0,1): Philox4x32-10 (Salmon et al., SC'11) + Box-Muller, keyed by
 * (seed; global site index, stream id, unit/4) -- the build's documented stand-in for torch.randn
 * (DESIGN.md "initial-state policy").
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int32_t seq_len, signal_len, num_layers1, num_layers2, num_classes, hidden_size, vocab_size,
        embedding_size, is_base, is_signallen, module; /* 0 both, 1 seq, 2 signal */
} orc_cfg;

#define SB 16 /* sites per block (weights are reused across the block from cache) */

/* ---------------- Philox4x32-10 ---------------- */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

/* 4 N(0,1) floats for (site, stream, group) */
static void philox_normal4(uint64_t seed, uint64_t site, uint32_t stream, uint32_t group, float out[4]) {
    uint32_t c[4] = {(uint32_t)site, (uint32_t)(site >> 32), stream, group};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    for (int p = 0; p < 2; ++p) {
        /* 23 random bits + 0.5: exactly representable in fp32 too, so a float implementation sees the same u */
        double u1 = ((double)(c[2 * p] >> 9) + 0.5) * (1.0 / 8388608.0);
        double u2 = ((double)(c[2 * p + 1] >> 9) + 0.5) * (1.0 / 8388608.0);
        double r = sqrt(-2.0 * log(u1));
        double th = 6.283185307179586476925 * u2;
        out[2 * p] = (float)(r * cos(th));
        out[2 * p + 1] = (float)(r * sin(th));
    }
}

/* exported for tests: raw Philox4x32-10 block (known-answer vectors of the Random123 distribution) */
void orc_philox_raw(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    memcpy(out, c, sizeof(c));
}

/* exported for tests: fill n*4 normals for consecutive groups of one (site, stream) */
void orc_philox_normal(uint64_t seed, uint64_t site, uint32_t stream, uint32_t ngroups, float* out) {
    for (uint32_t g = 0; g < ngroups; ++g) philox_normal4(seed, site, stream, g, out + 4 * g);
}

/* exported for tests: the initial states of n sites of one LSTM (0 seq, 1 signal, 2 comb) exactly as init mode 2 draws them
 * inside the forward (init_block below), written in the EXPLICIT layout (2 * layers, n, H) -- so that a test can perturb
 * them (wrong keying, wrong variance) and run the forward on the perturbed states */
void orc_philox_states(int lstm, int layers, int H, int64_t n, uint64_t seed, uint64_t site_offset, const uint64_t* site_keys,
                       float* h, float* c) {
    for (int layer = 0; layer < layers; ++layer)
        for (int dir = 0; dir < 2; ++dir)
            for (int64_t s = 0; s < n; ++s) {
                const uint64_t site = site_keys ? site_keys[s] : site_offset + (uint64_t)s;
                const size_t off = ((size_t)(2 * layer + dir) * (size_t)n + (size_t)s) * (size_t)H;
                float v[4];
                for (int g = 0; g < (H + 3) / 4; ++g) {
                    philox_normal4(seed, site, (uint32_t)(lstm * 64 + (layer * 2 + dir) * 2 + 0), (uint32_t)g, v);
                    for (int j = 0; j < 4 && 4 * g + j < H; ++j) h[off + 4 * g + j] = v[j];
                    philox_normal4(seed, site, (uint32_t)(lstm * 64 + (layer * 2 + dir) * 2 + 1), (uint32_t)g, v);
                    for (int j = 0; j < 4 && 4 * g + j < H; ++j) c[off + 4 * g + j] = v[j];
                }
            }
}

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* stream id: lstm (0 seq, 1 signal, 2 comb), layer, dir, which (0 h, 1 c) */
static inline uint32_t stream_id(int lstm, int layer, int dir, int which) {
    return (uint32_t)(lstm * 64 + (layer * 2 + dir) * 2 + which);
}

typedef struct {
    const float *wih, *whh, *bih, *bhh;
} lstm_w;

/*
 * One direction of one LSTM layer over a block of nb (<=SB) sites.
 *   x   [L][SB][I]   (block-local, time-major)
 *   out [L][SB][2H]  (this direction writes columns dir*H .. dir*H+H)
 *   h, c [SB][H]     initial state in, final state out
 *   wt  [(I+H)][4H]  transposed, concatenated weights (k-major) ; b [4H] = b_ih + b_hh
 */
#if defined(__GNUC__) && defined(__x86_64__)
__attribute__((target_clones("avx512f", "avx2", "default")))  /* resolved at load time */
#endif
static void lstm_dir_block(int L, int I, int H, int dir, int nb, const float* x, float* out, float* h,
                           float* c, const float* wt, const float* b, float* gates) {
    const int G = 4 * H;
    for (int step = 0; step < L; ++step) {
        const int t = dir ? (L - 1 - step) : step;
        /* gates[s][j] = b[j] + sum_k x[s][k] wt[k][j] + sum_k h[s][k] wt[I+k][j]  (k-ordered fp32 sums);
         * j is blocked so a [SB][JB] gate panel stays in L1 while each weight row is reused by all sites */
        enum { JB = 256 };
        for (int jb = 0; jb < G; jb += JB) {
            const int jn = (G - jb) < JB ? (G - jb) : JB;
            for (int s = 0; s < nb; ++s) {
                float* g = gates + (size_t)s * G + jb;
                for (int j = 0; j < jn; ++j) g[j] = b[jb + j];
            }
            for (int k = 0; k < I; ++k) {
                const float* w = wt + (size_t)k * G + jb;
                for (int s = 0; s < nb; ++s) {
                    const float xv = x[((size_t)t * SB + s) * I + k];
                    float* g = gates + (size_t)s * G + jb;
                    for (int j = 0; j < jn; ++j) g[j] += xv * w[j];
                }
            }
            for (int k = 0; k < H; ++k) {
                const float* w = wt + (size_t)(I + k) * G + jb;
                for (int s = 0; s < nb; ++s) {
                    const float hv = h[(size_t)s * H + k];
                    float* g = gates + (size_t)s * G + jb;
                    for (int j = 0; j < jn; ++j) g[j] += hv * w[j];
                }
            }
        }
        for (int s = 0; s < nb; ++s) {
            const float* g = gates + (size_t)s * G;
            float* hs = h + (size_t)s * H;
            float* cs = c + (size_t)s * H;
            float* o = out + ((size_t)t * SB + s) * (2 * H) + (size_t)dir * H;
            for (int u = 0; u < H; ++u) {
                const float ig = sigmoidf_(g[u]), fg = sigmoidf_(g[H + u]);
                const float gg = tanhf(g[2 * H + u]), og = sigmoidf_(g[3 * H + u]);
                const float cn = fg * cs[u] + ig * gg;
                const float hn = og * tanhf(cn);
                cs[u] = cn; hs[u] = hn; o[u] = hn;
            }
        }
    }
}

typedef struct {
    int I, H, layers;
    float** wt; /* [layers*2] transposed concat weights */
    float** b;  /* [layers*2] summed bias */
} lstm_pack;

static int pack_lstm(lstm_pack* p, int I, int H, int layers, const float* const* w /* 8 per layer */) {
    p->I = I; p->H = H; p->layers = layers;
    p->wt = (float**)calloc((size_t)layers * 2, sizeof(float*));
    p->b = (float**)calloc((size_t)layers * 2, sizeof(float*));
    if (!p->wt || !p->b) return -1;
    for (int k = 0; k < layers; ++k) {
        const int Ik = k == 0 ? I : 2 * H;
        for (int d = 0; d < 2; ++d) {
            const float* wih = w[(k * 2 + d) * 4 + 0];
            const float* whh = w[(k * 2 + d) * 4 + 1];
            const float* bih = w[(k * 2 + d) * 4 + 2];
            const float* bhh = w[(k * 2 + d) * 4 + 3];
            const int G = 4 * H;
            float* wt = (float*)malloc(sizeof(float) * (size_t)(Ik + H) * G);
            float* b = (float*)malloc(sizeof(float) * G);
            if (!wt || !b) return -1;
            for (int j = 0; j < G; ++j) {
                for (int kk = 0; kk < Ik; ++kk) wt[(size_t)kk * G + j] = wih[(size_t)j * Ik + kk];
                for (int kk = 0; kk < H; ++kk) wt[(size_t)(Ik + kk) * G + j] = whh[(size_t)j * H + kk];
                b[j] = bih[j] + bhh[j];
            }
            p->wt[k * 2 + d] = wt; p->b[k * 2 + d] = b;
        }
    }
    return 0;
}

static void free_lstm(lstm_pack* p) {
    if (!p->wt) return;
    for (int i = 0; i < p->layers * 2; ++i) { free(p->wt[i]); free(p->b[i]); }
    free(p->wt); free(p->b);
    p->wt = NULL; p->b = NULL;
}

/* initial state for a block: h0,c0 [SB][H] for (lstm,layer,dir) */
static void fill_state(int mode, const float* hsrc, const float* csrc, int64_t n, int64_t site0, int nb, int H,
                       int lstm, int layer, int dir, uint64_t seed, uint64_t site_offset, const uint64_t* site_keys,
                       float* h, float* c) {
    for (int s = 0; s < nb; ++s) {
        float* hs = h + (size_t)s * H;
        float* cs = c + (size_t)s * H;
        if (mode == 0) {
            memset(hs, 0, sizeof(float) * H); memset(cs, 0, sizeof(float) * H);
        } else if (mode == 1) {
            const size_t off = ((size_t)(2 * layer + dir) * (size_t)n + (size_t)(site0 + s)) * H;
            memcpy(hs, hsrc + off, sizeof(float) * H); memcpy(cs, csrc + off, sizeof(float) * H);
        } else {
            /* Philox counter of the site: its own 64-bit key when the caller names the sites (include/dsp_amd.h,
             * dsp_init_state.site_keys), else its global index */
            const uint64_t site = site_keys ? site_keys[site0 + s] : site_offset + (uint64_t)(site0 + s);
            float v[4];
            for (int g = 0; g < (H + 3) / 4; ++g) {
                philox_normal4(seed, site, stream_id(lstm, layer, dir, 0), (uint32_t)g, v);
                for (int j = 0; j < 4 && 4 * g + j < H; ++j) hs[4 * g + j] = v[j];
                philox_normal4(seed, site, stream_id(lstm, layer, dir, 1), (uint32_t)g, v);
                for (int j = 0; j < 4 && 4 * g + j < H; ++j) cs[4 * g + j] = v[j];
            }
        }
    }
}

/* run a (multi-layer) BiLSTM over a block; in [L][SB][I] -> out [L][SB][2H]; tmp same size as out */
static void bilstm_block(const lstm_pack* p, int L, int nb, int lstm_idx, int mode, const float* hsrc,
                         const float* csrc, int64_t n, int64_t site0, uint64_t seed, uint64_t site_offset,
                         const uint64_t* site_keys, const float* in, float* out, float* tmp, float* h, float* c, float* gates) {
    const float* cur = in;
    int I = p->I;
    for (int k = 0; k < p->layers; ++k) {
        float* dst = ((p->layers - 1 - k) % 2 == 0) ? out : tmp; /* last layer lands in out */
        for (int d = 0; d < 2; ++d) {
            fill_state(mode, hsrc, csrc, n, site0, nb, p->H, lstm_idx, k, d, seed, site_offset, site_keys, h, c);
            lstm_dir_block(L, I, p->H, d, nb, cur, dst, h, c, p->wt[k * 2 + d], p->b[k * 2 + d], gates);
        }
        cur = dst; I = 2 * p->H;
    }
}

/* y[L][SB][O] (+ column offset into a wider row) = relu(x[L][SB][K] W^T + b) */
static void linear_relu_block(int rows_t, int nb, int K, int O, const float* x, const float* w, const float* b,
                              float* y, int ystride, int yoff) {
    for (int t = 0; t < rows_t; ++t)
        for (int s = 0; s < nb; ++s) {
            const float* xs = x + ((size_t)t * SB + s) * K;
            float* ys = y + ((size_t)t * SB + s) * ystride + yoff;
            for (int o = 0; o < O; ++o) {
                const float* wr = w + (size_t)o * K;
                float acc = 0.f;
                for (int k = 0; k < K; ++k) acc += xs[k] * wr[k];
                acc += b[o];
                ys[o] = acc > 0.f ? acc : 0.f;
            }
        }
}

/*
 * weights: pointers in the reference's state_dict order (see oracle/forward_np.py:state_dict_spec).
 * states (mode 1): h_seq,c_seq,h_sig,c_sig,h_comb,c_comb in the reference layout (absent ones NULL).
 * returns 0, or <0 on bad arguments / allocation failure.
 */
int orc_forward_keys(const orc_cfg* cfg, const float* const* weights, int n_weights, int64_t n, const float* kmer,
                     const float* means, const float* stds, const float* lens, const float* signals, int init_mode,
                     const float* const* states, uint64_t seed, uint64_t site_offset, const uint64_t* site_keys,
                     float* logits, float* probs, int nthreads) {
    const int L = cfg->seq_len, S = cfg->signal_len, H = cfg->hidden_size, C = cfg->num_classes;
    int hseq = 0, hsig = 0;
    if (cfg->module == 0) { hseq = H / 2; hsig = H - hseq; }
    else if (cfg->module == 1) hseq = H;
    else if (cfg->module == 2) hsig = H;
    else return -2;
    const int sigfea = cfg->is_signallen ? 3 : 2;
    const int E = cfg->embedding_size;
    const int Iseq = (cfg->is_base ? E : 0) + sigfea;
    const int l1 = cfg->num_layers1, l2 = cfg->num_layers2;
    int expect = 0;
    if (hseq) expect += 1 + 8 * l2 + 2;
    if (hsig) expect += 8 * l2 + 2;
    expect += 8 * l1 + 4;
    if (n_weights != expect) return -3;
    if (init_mode == 1 && !states) return -4;

    int wi = 0;
    const float* embed = NULL; const float *fcseq_w = NULL, *fcseq_b = NULL, *fcsig_w = NULL, *fcsig_b = NULL;
    lstm_pack pseq = {0}, psig = {0}, pcomb = {0};
    int rc = 0;
    if (hseq) {
        embed = weights[wi++];
        rc |= pack_lstm(&pseq, Iseq, hseq, l2, weights + wi); wi += 8 * l2;
        fcseq_w = weights[wi++]; fcseq_b = weights[wi++];
    }
    if (hsig) {
        rc |= pack_lstm(&psig, S, hsig, l2, weights + wi); wi += 8 * l2;
        fcsig_w = weights[wi++]; fcsig_b = weights[wi++];
    }
    rc |= pack_lstm(&pcomb, H, H, l1, weights + wi); wi += 8 * l1;
    const float* fc1_w = weights[wi++]; const float* fc1_b = weights[wi++];
    const float* fc2_w = weights[wi++]; const float* fc2_b = weights[wi++];
    if (rc) { free_lstm(&pseq); free_lstm(&psig); free_lstm(&pcomb); return -5; }

    const int64_t nblocks = (n + SB - 1) / SB;
    int hmax = H; if (hseq > hmax) hmax = hseq; if (hsig > hmax) hmax = hsig;
    int fail = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    (void)nthreads;
#pragma omp parallel
    {
        const size_t wide = (size_t)L * SB * 2 * hmax;
        float* xin = (float*)malloc(sizeof(float) * (size_t)L * SB * (Iseq > S ? Iseq : S));
        float* a = (float*)malloc(sizeof(float) * wide);
        float* bb = (float*)malloc(sizeof(float) * wide);
        float* comb = (float*)malloc(sizeof(float) * (size_t)L * SB * H);
        float* hbuf = (float*)malloc(sizeof(float) * (size_t)SB * hmax);
        float* cbuf = (float*)malloc(sizeof(float) * (size_t)SB * hmax);
        float* gates = (float*)malloc(sizeof(float) * (size_t)SB * 4 * hmax);
        float* feat = (float*)malloc(sizeof(float) * (size_t)SB * 2 * H);
        float* hid = (float*)malloc(sizeof(float) * (size_t)SB * H);
        if (!xin || !a || !bb || !comb || !hbuf || !cbuf || !gates || !feat || !hid) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int64_t blk = 0; blk < nblocks; ++blk) {
                const int64_t site0 = blk * SB;
                const int nb = (int)((n - site0) < SB ? (n - site0) : SB);
                if (hseq) { /* models.py:181-201 */
                    for (int t = 0; t < L; ++t)
                        for (int s = 0; s < nb; ++s) {
                            float* xs = xin + ((size_t)t * SB + s) * Iseq;
                            const size_t r = (size_t)(site0 + s) * L + t;
                            int o = 0;
                            if (cfg->is_base) {
                                const long code = (long)kmer[r]; /* kmer.long(): truncation toward zero */
                                for (int e = 0; e < E; ++e) xs[o++] = embed[(size_t)code * E + e];
                            }
                            xs[o++] = means[r]; xs[o++] = stds[r];
                            if (cfg->is_signallen) xs[o++] = lens[r];
                        }
                    bilstm_block(&pseq, L, nb, 0, init_mode, states ? states[0] : NULL, states ? states[1] : NULL, n,
                                 site0, seed, site_offset, site_keys, xin, a, bb, hbuf, cbuf, gates);
                    linear_relu_block(L, nb, 2 * hseq, hseq, a, fcseq_w, fcseq_b, comb, H, 0);
                }
                if (hsig) { /* models.py:205-217 */
                    for (int t = 0; t < L; ++t)
                        for (int s = 0; s < nb; ++s)
                            memcpy(xin + ((size_t)t * SB + s) * S, signals + ((size_t)(site0 + s) * L + t) * S,
                                   sizeof(float) * S);
                    bilstm_block(&psig, L, nb, 1, init_mode, states ? states[2] : NULL, states ? states[3] : NULL, n,
                                 site0, seed, site_offset, site_keys, xin, a, bb, hbuf, cbuf, gates);
                    linear_relu_block(L, nb, 2 * hsig, hsig, a, fcsig_w, fcsig_b, comb, H, hseq);
                }
                /* models.py:219-231 */
                bilstm_block(&pcomb, L, nb, 2, init_mode, states ? states[4] : NULL, states ? states[5] : NULL, n, site0,
                             seed, site_offset, site_keys, comb, a, bb, hbuf, cbuf, gates);
                for (int s = 0; s < nb; ++s) {
                    memcpy(feat + (size_t)s * 2 * H, a + ((size_t)(L - 1) * SB + s) * 2 * H, sizeof(float) * H);
                    memcpy(feat + (size_t)s * 2 * H + H, a + ((size_t)0 * SB + s) * 2 * H + H, sizeof(float) * H);
                }
                /* models.py:234-240 (dropouts are identity in eval) */
                for (int s = 0; s < nb; ++s) {
                    const float* f = feat + (size_t)s * 2 * H;
                    float* hd = hid + (size_t)s * H;
                    for (int o = 0; o < H; ++o) {
                        const float* wr = fc1_w + (size_t)o * 2 * H;
                        float acc = 0.f;
                        for (int k = 0; k < 2 * H; ++k) acc += f[k] * wr[k];
                        acc += fc1_b[o];
                        hd[o] = acc > 0.f ? acc : 0.f;
                    }
                    float* lg = logits + (size_t)(site0 + s) * C;
                    float* pr = probs + (size_t)(site0 + s) * C;
                    float mx = -INFINITY;
                    for (int o = 0; o < C; ++o) {
                        const float* wr = fc2_w + (size_t)o * H;
                        float acc = 0.f;
                        for (int k = 0; k < H; ++k) acc += hd[k] * wr[k];
                        acc += fc2_b[o];
                        lg[o] = acc;
                        if (acc > mx) mx = acc;
                    }
                    float sum = 0.f;
                    for (int o = 0; o < C; ++o) { pr[o] = expf(lg[o] - mx); sum += pr[o]; }
                    for (int o = 0; o < C; ++o) pr[o] /= sum;
                }
            }
        }
        free(xin); free(a); free(bb); free(comb); free(hbuf); free(cbuf); free(gates); free(feat); free(hid);
    }
    free_lstm(&pseq); free_lstm(&psig); free_lstm(&pcomb);
    return fail ? -6 : 0;
}

/* site_keys == NULL: the Philox counter of site i is site_offset + i */
int orc_forward(const orc_cfg* cfg, const float* const* weights, int n_weights, int64_t n, const float* kmer,
                const float* means, const float* stds, const float* lens, const float* signals, int init_mode,
                const float* const* states, uint64_t seed, uint64_t site_offset, float* logits, float* probs,
                int nthreads) {
    return orc_forward_keys(cfg, weights, n_weights, n, kmer, means, stds, lens, signals, init_mode, states, seed,
                            site_offset, NULL, logits, probs, nthreads);
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
