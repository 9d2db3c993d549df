// dsp_kernels.h -- argument blocks shared by the gfx950 kernels (dsp_kernels.hip) and the host side of the
// C ABI (dsp_capi.cpp).  All activation pointers use the K4 layout  act[tile][t][F/4][32][4]  (fp32).
#ifndef DSP_KERNELS_H
#define DSP_KERNELS_H

#include <hip/hip_runtime_api.h>
#include <stdint.h>

struct PackArgs {
    const void* kmer;      // [n, T] codes (kdt: DSP_DT_*)
    const float* means;    // [n, T]
    const float* stds;     // [n, T]
    const void* lens;      // [n, T] (ldt)
    const float* signals;  // [n, T, S]
    const float* embed;    // [V, E] device copy of embed.weight
    float* xseq;           // K4, Fseq features (NULL when the module has no seq branch)
    float* xsig;           // K4, Fsig features (NULL when the module has no signal branch)
    long long n;           // live sites
    long long NTp;         // padded tile count (all of it is written; dead sites get zeros)
    int kdt, ldt;
    int T, S, E, V;
    int is_base, is_siglen;
    int Fseq, Fsig;
    int xoff_seq, xoff_sig;  // first padded feature that carries data (the features sit at the end of a padded block)
    const void* xseq_end;      // ends of the allocations behind xseq / xsig / zero_words (flat stores: read by the bounds build)
    const void* xsig_end;
    const void* zero_words_end;
    unsigned int* zero_words;  // arrival counters of this forward's clustered LSTM launches (dsp_lstmc_kernel): zeroed here, by
    int n_zero_words;          // the first launch of the forward (a kernel boundary orders it before every later launch)
};

struct LstmArgs {
    const float* x;        // K4 input, Ipad features
    float* out;            // K4 output, Fout features; this layer writes [dir*Hp, dir*Hp + Hp)
    const float* wpk0;     // forward  direction A fragments [UT][NQ][4 gates][64 lanes][4]
    const float* wpk1;     // backward direction
    const float* sbias0;   // forward  pre-scaled biases: -log2e*(b_ih+b_hh) (gates i,f,o), -2*log2e*(..) (gate g), [4][Hp]
    const float* sbias1;   // backward
    const float* h0;       // EXPLICIT: reference layout, already offset to this layer: [2 dirs][n][H]
    const float* c0;
    float* h0buf;          // K4 scratch [NTp][Fout/4][32][4] holding h0 (read by step 0 as "h_{-1}")
    const unsigned long long* site_keys;  // PHILOX: per-site counter keys [n], or NULL = site_offset + site
    float* cbuf;           // NP > 2 only: cell-state scratch, [workgroup][pass][2 site tiles][4 groups][threads] float4
    long long n;
    long long NTp;
    unsigned long long seed, site_offset;
    int Ipad, H, Hp, T, Fout;
    int nqx_lo, nqx_used;  // x-part k-groups [nqx_lo, nqx_used) carry real features (the rest of Ipad/8 is zero padding)
    int NQ;                // k-groups per unit tile in wpk: (Ipad+Hp)/8 padded to a multiple of 4 (zero weights)
    int UT, SG;            // unit tiles (Hp/32), site groups (of two 32-site tiles) per workgroup
    int NP;                // passes over the unit tiles per time step (UT / 8 above 8 unit tiles, else 1); block = 64*(UT/NP)*SG threads
    int init_mode;         // DSP_INIT_*
    int stream_base;       // philox stream of (lstm, layer, dir=0, h): lstm*64 + layer*4
    unsigned int* cflags;  // clustered launches (CG > 0): arrival counters of this launch, one per (site tile, direction), 32 words apart
    unsigned long long cluster_timeout;  // s_memtime ticks a member of a cluster waits for the others to become resident before it
                           // abandons the cluster to the clean-up launch
    // Round 6: the END of the allocation behind every pointer the kernels address through a buffer descriptor (the region of
    // the handle's workspace, the weight upload): a descriptor's num_records = end - base, so the hardware range check is live on
    // every operand -- a load past the end returns 0, a store past it is dropped, a wild offset is a parity failure with a test
    // name instead of a memory fault that ends the process.  The launch wrappers refuse a NULL or inverted end.
    const void* x_end;
    const void* out_end;
    const void* wpk0_end;
    const void* wpk1_end;
    const void* h0buf_end;
    const void* cbuf_end;   // (NULL with cbuf)
    const void* cflags_end; // (NULL with cflags; the counters are addressed flat: the end is what the bounds build checks)
    int CG;                // 0, or gates per wave of dsp_lstmc_kernel: 4 / 2 / 1 = a (site tile, direction) spread over UT/CG workgroups
    int flags;             // bit 0: issue priority by phase (k-loop 2, cell 0); bit 1: <2 unit tiles, 1 site tile> per wave; bit 2: ... and one such workgroup per CU; bit 3: the workgroup-local form of dsp_lstmc_kernel with eight waves; bit 4: ... as the clean-up launch behind a clustered one (abandoned clusters only); bit 6: the per-wave hand-off of the clustered launches (round 5: arrivals per wave, deferred into the next step's x part; poll one block early); bit 7: x ahead at one gate per wave with rings 8 deep (xs = nqx - 8); bit 8: stamp this launch, bits 9..11: its stamping wave (DSP_TRACE builds only)
    // Round 6, "x ahead" (DSP_LSTM_XAHEAD=1; clustered dense layers of batches <= 512 sites): the x-part k-groups [0, xs) of every
    // step do not depend on the recurrence -- dsp_xahead_kernel computes their partial sums for ALL T steps at once on the idle
    // compute units (same MFMAs in the same order, from zero) into xacc, [cluster][t][unit tile][gate][4][64 lanes] float4; the
    // recurrent launch starts each step's accumulators from them and runs the k-groups [xs, NQ) only.  xs = 0: off.
    float* xacc;
    const void* xacc_end;
    int xs;
};

struct LinArgs {
    const float* x;        // K4 input, Fin features
    float* out;            // K4 output, Fout features, rows written at feature offset out_off
    const float* wpk;      // [ORT][Fin/8][64][4]
    const float* bias;     // [ORT*32]
    long long ncols;       // NTp * T column blocks of 32 sites
    int Fin, Fout, out_off, ORT, relu;
    // a second problem of the same shape in the same launch (fc_seq + fc_signal): workgroups [nbx, 2 nbx)
    const float* x2;       // NULL: one problem
    const float* wpk2;
    const float* bias2;
    int out_off2;
    // ends of the allocations behind x / x2 / wpk / wpk2 / out (see LstmArgs)
    const void* x_end;
    const void* x2_end;
    const void* wpk_end;
    const void* wpk2_end;
    const void* out_end;
    unsigned nbx;          // workgroups (of 8 column blocks) per problem (set by dsp_k_linear)
    int small;             // 1: dsp_linear1_kernel (one accumulator tile per wave: batches that leave CUs idle)
};

struct HeadArgs {
    const float* x;        // K4 output of the last combined layer, 2*Hp features
    const float* w1pk;     // fc1 A fragments [Hp/32][2Hp/8][64][4]
    const float* b1;       // [Hp]
    const float* w2;       // [C][Hp] (padded columns zero)
    const float* b2;       // [C]
    float* logits;         // [n, C] or NULL
    float* probs;          // [n, C] or NULL
    uint8_t* labels;       // [n] or NULL
    const void* x_end;     // ends of the allocations behind x / w1pk (see LstmArgs)
    const void* w1pk_end;
    long long n;
    int Hp, T, C;
    int flags;             // bit 0: keep four site tiles per workgroup whatever the batch (A/B switch DSP_HEAD_ST4=1)
};

/* operand kinds in a bounds record */
enum { DSP_BND_W = 1, DSP_BND_X = 2, DSP_BND_OUT = 3, DSP_BND_H0 = 4, DSP_BND_C = 5, DSP_BND_FLAGS = 6, DSP_BND_FLAT_OUT = 7, DSP_BND_XACC = 8 };

#ifdef __cplusplus
extern "C" {
#endif
int dsp_k_init(void);
int dsp_k_pack(const PackArgs* a, hipStream_t s);
int dsp_k_lstm(const LstmArgs* a, hipStream_t s);
int dsp_k_lstm6(const LstmArgs* a, int nprod, hipStream_t s);
int dsp_k_linear(const LinArgs* a, hipStream_t s);
int dsp_k_head(const HeadArgs* a, hipStream_t s);
int dsp_k_probe_xcc(unsigned* dev_out, int blocks, hipStream_t s);   /* out[b] = XCC_ID block b ran on */
int dsp_k_range_probe(float* dev_buf4k, unsigned* dev_out, hipStream_t s);   /* see dsp_debug_range_probe */
/* dry runs (dsp_debug_dry_run): on != 0 -> the wrappers above check and NOTE their launches (one text line each appended to the
   std::string behind log_string, may be NULL) instead of making them; per thread */
void dsp_k_set_dry(int on, void* log_string);
/* 1 in the bounds-recording build (make bounds -> libdsp_amd_bounds.so), else 0 */
int dsp_k_bounds_build(void);
/* bounds build: the first out-of-range access recorded since the last read (and clears it); rec[0] = accesses out of range
   (0: none), rec[1] = source line of the access, rec[2] = which operand (DSP_BND_*), rec[3] = workgroup, rec[4] = thread,
   rec[5] / rec[6] = byte offset from the descriptor's base (low / high word), rec[7] = the descriptor's extent in bytes.
   Returns a hipError_t; all zeros in the product build. */
int dsp_k_bounds_read(unsigned rec[8]);
#ifdef __cplusplus
}
#endif

#endif
