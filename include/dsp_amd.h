/*
 * dsp_amd.h -- C ABI of libdsp_amd.so: the MI355X (gfx950) implementation of the
 * `deepsignal_plant call_mods` hot path (feature rows -> ModelBiLSTM forward -> per-site calls).
 *
 * The reference (PengNi/deepsignal-plant, pure Python) exposes no FFI; the seam this library sits
 * behind is the Python call at deepsignal_plant/call_modifications.py:159-163 plus the checkpoint
 * contract at :214-228.  Each entry point cites the reference interface it
 * replaces.  Plain pointers and sizes only; no torch types.  All device pointers are caller-owned
 * (e.g. torch.Tensor.data_ptr() on PyTorch-ROCm); the handle owns only repacked weights + scratch.
 *
 * Return convention: 0 on success, negative dsp_status on error; dsp_last_error() returns a
 * thread-local message.  The Python mirror maps DSP_EINVAL / DSP_EPARSE -> ValueError (a short row's DSP_EPARSE -> IndexError), DSP_EKEY -> KeyError, the rest -> RuntimeError,
 * matching the exceptions the reference raises (models.py:127-128, call_modifications.py:219-223).
 */
#ifndef DSP_AMD_H
#define DSP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSP_AMD_ABI_VERSION 3   /* 2: dsp_init_state.site_keys; 3: dsp_model_query, dsp_device_pci_bdf / _uuid, dsp_profile_enable
                                   mode 2 (round 5's additions, numbered in round 6), DSP_EBOUNDS */

typedef enum dsp_status {
    DSP_OK = 0,
    DSP_EINVAL = -1,      /* bad argument / unsupported configuration (reference: ValueError) */
    DSP_ESHAPE = -2,      /* weight count/shape mismatch (reference: strict load_state_dict RuntimeError) */
    DSP_EHIP = -3,        /* HIP runtime error (message carries hipGetErrorString) */
    DSP_ENOMEM = -4,      /* host or device allocation failure */
    DSP_EPARSE = -5,      /* malformed feature row (reference: ValueError in the row parser) */
    DSP_EKEY = -6,        /* a base letter outside base2code_dna (reference: KeyError, call_modifications.py:84); the message
                             starts with the quoted letter */
    DSP_EBOUNDS = -7      /* only from the bounds-recording debug build (libdsp_amd_bounds.so, `make bounds`): a kernel of the
                             forward addressed memory outside the extent of its operand; the message names the source line, the
                             operand, the workgroup, the thread, the offset and the extent.  The product library cannot return it:
                             there the hardware range check of the buffer descriptors drops such an access (loads return 0). */
} dsp_status;

/* module codes: ModelBiLSTM(module=...) at deepsignal_plant/models.py:120-128 */
enum { DSP_MODULE_BOTH = 0, DSP_MODULE_SEQ = 1, DSP_MODULE_SIGNAL = 2 };

/* element types accepted for the kmer-code and signal-length inputs.  The reference feeds both as
 * float32 and casts inside forward (models.py:182-186); compact types are the fast path. */
enum { DSP_DT_F32 = 0, DSP_DT_U8 = 1, DSP_DT_U16 = 2, DSP_DT_I32 = 3 };

/* LSTM initial-state policy.  The reference draws h0,c0 ~ N(0,1) with torch.randn on every forward
 * (models.py:169-176).  EXPLICIT pins them (parity runs); PHILOX is the in-kernel counter-based
 * stand-in (Philox4x32-10 + Box-Muller keyed by seed, a 64-bit site key, stream, unit/4).  The site key is the site's
 * global index (site_offset + i) or, when the caller names its sites, site_keys[i]: results then depend on neither the
 * batching nor the number of ranks nor the order in which sites arrive (feature files: the global row index; the
 * reads-directory branch of call_mods, call_modifications.py:285-358: read uid << 24 | base index in the read). */
enum { DSP_INIT_ZEROS = 0, DSP_INIT_EXPLICIT = 1, DSP_INIT_PHILOX = 2 };

/* Constructor arguments of ModelBiLSTM, deepsignal_plant/models.py:103-106 (dropout_rate and device
 * excluded: dropout is identity at inference, call_modifications.py:228/:677; device is an argument
 * of dsp_model_create). */
typedef struct dsp_model_cfg {
    int32_t seq_len;        /* --seq_len, default 13 */
    int32_t signal_len;     /* --signal_len, default 16 */
    int32_t num_layers1;    /* --layernum1: combined BiLSTM layers, default 3 */
    int32_t num_layers2;    /* --layernum2: seq / signal BiLSTM layers, default 1 */
    int32_t num_classes;    /* --class_num, default 2 */
    int32_t hidden_size;    /* --hid_rnn, default 256 (this build: <= 2048) */
    int32_t vocab_size;     /* --n_vocab, default 16 */
    int32_t embedding_size; /* --n_embed, default 4 */
    int32_t is_base;        /* --is_base */
    int32_t is_signallen;   /* --is_signallen */
    int32_t module;         /* DSP_MODULE_* */
} dsp_model_cfg;

typedef struct dsp_init_state {
    int32_t mode;           /* DSP_INIT_* */
    uint64_t seed;          /* PHILOX key */
    uint64_t site_offset;   /* PHILOX: global index of site 0 of this call (range-sharding keeps results
                               independent of how sites are split over GPUs / batches) */
    /* EXPLICIT: device pointers in the reference layout (2*num_layers, n_sites, H) fp32, i.e. what
     * init_hidden returns at models.py:196-198 (seq), :212-214 (signal), :226-228 (comb).  Unused
     * branches may be NULL. */
    const float* h_seq;  const float* c_seq;
    const float* h_sig;  const float* c_sig;
    const float* h_comb; const float* c_comb;
    /* PHILOX, optional: DEVICE pointer to n_sites 64-bit site keys; NULL = site_offset + site index (ABI >= 2) */
    const uint64_t* site_keys;
} dsp_init_state;

typedef struct dsp_model dsp_model; /* opaque handle: repacked weights + scratch, bound to one device */

/* Number of weight tensors / name and shape of tensor `idx` in the reference's state_dict order
 * (49 tensors for the default both_bilstm: SURVEY.md 8(b)).  Lets a binding validate a checkpoint
 * the way strict load_state_dict does (call_modifications.py:219-223). */
int32_t dsp_weight_count(const dsp_model_cfg* cfg);
int32_t dsp_weight_spec(const dsp_model_cfg* cfg, int32_t idx, char* name, size_t name_cap, int64_t shape[2],
                        int32_t* ndim);

/* Algorithmic FLOPs per site (2 x MAC; SURVEY.md 8(d): 118,447,104 for the default model). */
int64_t dsp_flops_per_site(const dsp_model_cfg* cfg);

/* Replaces: ModelBiLSTM(...) + load_state_dict + .cuda(device) + .eval()
 * (call_modifications.py:214-228).  host_weights[i] / numels[i]: fp32 tensors in state_dict order,
 * contiguous, HOST memory.  Repacks once (transpose into MFMA fragment order, pre-summed biases)
 * and uploads once. */
int32_t dsp_model_create(const dsp_model_cfg* cfg, const float* const* host_weights, const int64_t* numels,
                         int32_t n_weights, int32_t device, dsp_model** out);

/* Pre-size the scratch for batches up to max_sites (no allocation happens inside dsp_forward unless a
 * larger batch arrives, in which case the scratch grows once, synchronously). */
int32_t dsp_model_reserve(dsp_model* m, int64_t max_sites);
size_t dsp_workspace_bytes(const dsp_model* m, int64_t max_sites);

/* Replaces: logits, probs = model(kmer, base_means, base_stds, base_signal_lens, signals)
 * (call_modifications.py:159-162 -> models.py:178-240) and the argmax at :163.
 *   kmer    [n, seq_len]              codes 0..vocab-1, dtype kmer_dtype (DSP_DT_*); a code outside the table is the
 *                                     caller's error (nn.Embedding raises IndexError, models.py:186): the kernel clamps the
 *                                     index for memory safety and does not report it -- validate on the host, as this
 *                                     build's call_mods does
 *   means, stds [n, seq_len]          fp32
 *   lens    [n, seq_len]              dtype lens_dtype
 *   signals [n, seq_len, signal_len]  fp32
 *   logits, probs [n, num_classes]    fp32 (either may be NULL);  labels [n] uint8 argmax (may be NULL)
 * All pointers are DEVICE pointers on the handle's device.  Asynchronous on `stream` (hipStream_t);
 * unused inputs of a branch that the module does not have may be NULL.
 * Threading / ownership: a handle owns ONE scratch workspace (activations of the forward in flight), so forwards on
 * the same handle must be issued by one host thread at a time and are ordered by the streams they are given; use one
 * handle per stream for concurrent forwards (the repacked weights are 19 MB).  Launch geometry is derived per call and
 * never stored in the handle: batches of up to 4,096 sites (on 256 CUs) run the combined stack with 32-site workgroups,
 * one per CU -- 3.7 ms per forward instead of the 6.6 ms of one round of 64-site workgroups, the same bytes out
 * (DSP_LSTM_TILING=0 / =21 force either tiling); the rate is flat from 16,384 sites on (INTEGRATION.md "Batch size").
 * A call is run as its whole 8,192-site rounds + its remainder as the cheapest sequence of small-batch pieces (classes of 512 /
 * 1,024 / 2,048 / 4,096 sites; round 5: 3,000 sites 3.3 -> 2.75 ms, 9,000 sites 13.8 -> 7.7 ms, the same bytes out; not with DSP_INIT_EXPLICIT states, whose layout
 * has the site index in the middle dimension, nor in the split-precision modes; DSP_FORWARD_SPLIT=0 turns it off);
 * dsp_debug_read_activation refuses after such a call (the scratch holds its last piece only).
 * hidden_size <= 2048 is the one model-shape limit of this build (one workgroup of 8 waves
 * holds a direction's whole hidden state, 256 units per pass over the step; above 256 the cell state moves from LDS to a
 * global scratch); the split-precision modes cover hidden_size <= 256 and fall back to the fp32 kernels above it. */
int32_t dsp_forward(dsp_model* m, void* stream, int64_t n_sites, const void* kmer, int32_t kmer_dtype,
                    const float* means, const float* stds, const void* lens, int32_t lens_dtype,
                    const float* signals, const dsp_init_state* init, float* logits, float* probs,
                    uint8_t* labels);

/* Bring-up / test hook: copy an intermediate activation of the LAST dsp_forward into HOST memory in the
 * reference layout [n_sites, seq_len, features].  which: 0 = combined-LSTM input (relu(fc_seq) |
 * relu(fc_signal), models.py:225 input), 1 = last combined-LSTM layer output [.., 2*hidden]
 * (models.py:226-228 output).  Synchronises the stream. */
int32_t dsp_debug_read_activation(dsp_model* m, void* stream, int32_t which, int64_t n_sites, float* host_out);

/* Per-kernel timing measured with HIP events on the launch stream (used by bench.py for the roofline
 * line).  While enabled, every launch of every dsp_forward is bracketed by two events (consecutive launches of
 * one stream share the event between them: a launch's time then includes the gap before it; an event record costs
 * the stream a few microseconds -- ~0.03 ms per forward); entries accumulate until dsp_profile_read() drains them
 * (call it after synchronising the stream).  on = 2: ONE pair of records per forward brackets the run of consecutive launches of
 * the DOMINANT kernel (the combined stack, "lstm_comb") -- what bench.py keeps on through its timed steps, 2 event records per
 * forward instead of 9 (round 5: at 512 sites the full set cost 5 % of the forward it timed); every launch of the run is
 * reported with an equal share of the bracket's time, the gaps and clean-up launches between them included.
 * names: NUL-separated list written into `names`; ms[i] per launch. Returns the launch count. */
int32_t dsp_profile_enable(dsp_model* m, int32_t on);
int32_t dsp_profile_read(dsp_model* m, char* names, size_t names_cap, float* ms, int32_t cap);

/* How the fp32 products of the combined BiLSTM stack (92 % of the FLOPs) are evaluated.  DSP_PREC_FP32 (default):
 * v_mfma_f32_32x32x2_f32.  DSP_PREC_BF16X9 / DSP_PREC_BF16X6: every operand is split into three bf16 pieces
 * (hi + mid + lo == x exactly) and a product becomes 9 (all: exact) or 6 (without the three smallest: ~2^-24
 * relative, below the rounding of an fp32 accumulation) piece products on v_mfma_f32_32x32x16_bf16, accumulated in
 * fp32.  DSP_PREC_FP16X3: two fp16 pieces per operand (11 + 11 mantissa bits) and the three products lh, hl, hh on
 * v_mfma_f32_32x32x16_f16: ~2^-22 relative, half the matrix work of BF16X6; operands beyond the fp16 range (6.5e4)
 * would saturate: the mode is therefore used for the combined stack only (the front ends, which eat raw features,
 * take bf16 pieces) and dsp_model_set_precision returns DSP_EINVAL for a checkpoint whose combined-stack weights or fc
 * output bounds exceed 3e4.  Inputs, outputs, layouts and every other kernel are unchanged.  Also settable at creation through the
 * environment variable DSP_PRECISION = fp32 | bf16x6 | bf16x9 | fp16x3.  (No reference counterpart: torch.nn.LSTM on CPU is
 * fp32 throughout, models.py:137-157.) */
enum { DSP_PREC_FP32 = 0, DSP_PREC_FP16X3 = 3, DSP_PREC_BF16X6 = 6, DSP_PREC_BF16X9 = 9 };
int32_t dsp_model_set_precision(dsp_model* m, int32_t precision);

void dsp_model_destroy(dsp_model* m);

/* ---- host-side text I/O of the path (plain C++, multi-threaded; no GPU involved) ------------------------
 *
 * dsp_parse_feature_rows replaces the row grammar of _read_features_file
 * (call_modifications.py:76-86, :111-117; same grammar as dataloader.py:14-31): `text` holds complete
 * lines of 12 tab-separated fields; outputs are SoA host buffers (ideally pinned) with room for max_rows
 * rows: kmer codes u8 [r][L] (base2code_dna), means/stds f32 [r][L], lens i32 [r][L], signals f32
 * [r][L][S], labels i32 [r]; row_off/info_len = byte range of the first six fields (the `sampleinfo`
 * string kept verbatim, :80) inside `text`; read_off/read_len = the readname field (:77) relative to
 * row_off, for read-boundary batching (:94-109).  Tokens go decimal -> correctly rounded double -> float32,
 * as Python float() + torch.tensor(dtype=float) do.  Returns the row count, or DSP_EKEY (unknown base
 * letter = the reference's KeyError), DSP_EPARSE (malformed number = ValueError) / DSP_EINVAL. */
int64_t dsp_count_rows(const char* text, size_t len);
/* The host half of the DEVICE-side row parser (below): one pass that copies a block of rows into a (page-locked) staging
 * buffer of len + 1 bytes and notes where every row starts (row_off: n + 1 entries; an unterminated last row gets its
 * '\n').  Returns the number of rows, DSP_ENOMEM when there are more than max_rows. */
int64_t dsp_copy_rows_index(const char* text, size_t len, char* dst, uint64_t* row_off, int64_t max_rows);
/* ... and for a plain file: the block is read (pread, 1 MiB pieces scanned for row ends while hot) until want_rows rows are in
 * or range_bytes / cap_bytes are used up; *consumed = bytes of the file the rows took; at_eof: an unterminated last row
 * counts.  dst needs cap_bytes + 1 bytes, row_off want_rows + 1 entries.  0 rows with *consumed == 0: no row fits cap_bytes. */
int64_t dsp_read_rows_index(int32_t fd, uint64_t file_off, uint64_t range_bytes, uint64_t cap_bytes, int64_t want_rows,
                            int32_t at_eof, char* dst, uint64_t* row_off, uint64_t* consumed);
/* The row grammar of _read_features_file (call_modifications.py:76-86) parsed ON THE GPU (csrc/dsp_parse_dev.hip) over the
 * raw text in HBM -- rows that fit the LDS: a workgroup stages four rows, finds their delimiters and parses ONE TOKEN PER
 * THREAD; longer rows: one thread per row + one per float list.  text_dev: the bytes dsp_copy_rows_index /
 * dsp_read_rows_index staged, + 64 readable bytes behind them; row_off_dev: its n + 1 offsets; text_bytes: the bytes staged
 * (no offset beyond them is followed); outputs: the arrays of dsp_parse_feature_rows (all DEVICE pointers; lens i32);
 * seg_dev: scratch of n x (seq_len + 4) words.
 * A row is either parsed completely by the plain-row rules of the host's one-pass parser -- decimal -> integer mantissa
 * (<= 18 digits, < 2^53) times / divided by an exact power of ten (|e| <= 22) in float64, ONE correctly rounded
 * operation, then float32: bit-identical to the host parser -- or, for anything else (blanks, '+', inf / nan, long
 * mantissas, a wrong field count, an unknown base, ...), left alone with status_dev[row] = 1 and counted in *n_flagged_dev
 * (flag events: zero = every row was plain).  The caller hands blocks with flagged rows to dsp_parse_feature_rows, which
 * also owns the error messages.  Asynchronous on `stream`; *n_flagged_dev is zeroed by the launch. */
int32_t dsp_parse_rows_device(void* stream, const char* text_dev, const uint64_t* row_off_dev, int64_t n, int32_t seq_len,
                              int32_t signal_len, uint8_t* kmer, float* means, float* stds, int32_t* lens, float* signals,
                              int32_t* labels, uint32_t* info_len, uint32_t* read_off, uint32_t* read_len,
                              uint8_t* status_dev, uint32_t* n_flagged_dev, uint32_t* seg_dev, uint64_t text_bytes);
int64_t dsp_find_row_end(const char* text, size_t len, int64_t n_rows);   /* bytes of the first n_rows rows (len if there are fewer) */
int64_t dsp_parse_feature_rows(const char* text, size_t len, int32_t seq_len, int32_t signal_len, int64_t max_rows,
                               uint8_t* kmer, float* means, float* stds, int32_t* lens, float* signals,
                               int32_t* labels, uint64_t* row_off, uint32_t* info_len, uint32_t* read_off,
                               uint32_t* read_len, int32_t nthreads);

/* dsp_format_calls replaces the per-row string building of _call_mods (call_modifications.py:175-188) and
 * the line writing of _write_predstr_to_file (:262-282): for each row
 *   sampleinfo \t round(p0/(p0+p1),6) \t round(1-that,6) \t label \t centre-5-mer \n
 * with the reference's numpy-float32 arithmetic and str() formatting reproduced byte for byte.
 * Returns bytes written to `out`, or a negative dsp_status (DSP_ENOMEM if out_cap is too small). */
int64_t dsp_format_calls(const char* text, const uint64_t* row_off, const uint32_t* info_len, const float* probs,
                         int32_t num_classes, const uint8_t* labels, const uint8_t* kmer, int32_t seq_len, int64_t n,
                         char* out, size_t out_cap, int32_t nthreads);

/* ---- binary feature container (.dspf) ----------------------------------------------
 *
 * The parsed form of the feature TSV (rows written by extract_features.py:381-395, read by
 * call_modifications.py:76-86) as blocks of ready-to-copy SoA arrays + the rows' sampleinfo strings, so that
 * call_mods reads straight into pinned buffers.  Array meanings and dtypes are those of
 * dsp_parse_feature_rows; values are bit-identical to what it yields for the same rows (layout:
 * csrc/dsp_featfile.cpp).  The writer coalesces added rows into blocks of `block_rows` (<=0: 32768);
 * dsp_feat_writer_close writes the block index and frees the writer.  Readers are thread-safe (pread);
 * dsp_feat_read_block fills the buffers (NULL = skip that array), `info` receives the concatenated
 * sampleinfo bytes with row_off/info_len/read_off/read_len addressing it, and returns the block's row count
 * (DSP_ENOMEM if max_rows / info_cap are too small, DSP_EPARSE for a corrupt or truncated file). */
typedef struct dsp_feat_writer dsp_feat_writer;
typedef struct dsp_feat_file dsp_feat_file;
int32_t dsp_feat_writer_create(const char* path, int32_t seq_len, int32_t signal_len, int64_t block_rows,
                               dsp_feat_writer** out);
int32_t dsp_feat_writer_add(dsp_feat_writer* w, int64_t n, const uint8_t* kmer, const float* means, const float* stds,
                            const int32_t* lens, const float* signals, const int32_t* labels, const char* text,
                            const uint64_t* row_off, const uint32_t* info_len, const uint32_t* read_off,
                            const uint32_t* read_len);
int32_t dsp_feat_writer_close(dsp_feat_writer* w);
int32_t dsp_feat_open(const char* path, dsp_feat_file** out);
int32_t dsp_feat_info(const dsp_feat_file* f, int32_t* seq_len, int32_t* signal_len, int64_t* n_rows, int64_t* n_blocks);
int32_t dsp_feat_block_info(const dsp_feat_file* f, int64_t block, int64_t* n, int64_t* first_row, int64_t* info_bytes);
int64_t dsp_feat_read_block(const dsp_feat_file* f, int64_t block, int64_t max_rows, uint8_t* kmer, float* means,
                            float* stds, int32_t* lens, float* signals, int32_t* labels, char* info, size_t info_cap,
                            uint64_t* row_off, uint32_t* info_len, uint32_t* read_off, uint32_t* read_len,
                            int32_t nthreads);
void dsp_feat_close(dsp_feat_file* f);

/* ---- feature extraction from resquiggled reads -------------------------------------
 *
 * Replaces the arithmetic of extract_features.py after the HDF5 reads: _rescale_signals (:273-274),
 * _normalize_signals (:179-190), event slicing + per-base np.mean / np.std (:331-335, :363-365), the k-mer
 * window and _get_signals_rect (:360-368, :232-251), producing the tensors dsp_forward consumes directly in
 * HBM (the reference's fast5 route of call_mods, call_modifications.py:285-325) -- or, with round_stats=1,
 * exactly the values the feature TSV would carry (_features_to_str, :381-395).  float64 arithmetic in numpy's
 * evaluation order: results are bit-identical to the reference's for every base that is not subsampled.
 *
 * A batch of reads is SoA with CSR offsets; ALL pointers are DEVICE pointers:
 *   raw [n_samples] int16 DAQ values of the reads back to back, raw_off [n_reads+1];
 *   scaling, offset [n_reads]   pA = scaling * (raw + offset)  (_get_scaling_of_a_read, :255-270);
 *   ev_start, ev_len [n_events] event start (relative to the read's raw, read_start_rel_to_raw added, :81) and
 *   length; ev_base [n_events] ASCII base; ev_off [n_reads+1]   (_get_label_raw, :44-91). */
typedef struct dsp_read_batch {
    int64_t n_reads, n_samples, n_events;
    const int16_t* raw;
    const int64_t* raw_off;
    const double* scaling;
    const double* offset;
    const int64_t* ev_start;
    const int64_t* ev_len;
    const uint8_t* ev_base;
    const int64_t* ev_off;
} dsp_read_batch;

enum { DSP_NORM_MAD = 0, DSP_NORM_ZSCORE = 1 }; /* --normalize_method, extract_features.py:179-185 */

/* per-read shift / scale [n_reads] float64: mad -> (np.median, statsmodels robust.mad), zscore -> (np.mean, np.std) */
int32_t dsp_extract_normalize(void* stream, const dsp_read_batch* b, int32_t method, double* shift, double* scale);
/* per-base float64 mean / std of the normalised, 6-decimal-rounded samples, clamped length and first sample
 * index [n_events]; blk_off [n_reads+1] int64 is device workspace */
int32_t dsp_extract_base_stats(void* stream, const dsp_read_batch* b, const double* shift, const double* scale,
                               int64_t* blk_off, double* base_mean, double* base_std, int32_t* base_len,
                               int64_t* base_lo);
/* window gather for n_sites sites (site_read = read index in the batch, site_loc = base index of the targeted
 * base in the read; the caller guarantees seq_len/2 <= loc < n_bases - seq_len/2): kmer u8 [n][L] codes
 * (base2code_dna), means/stds f32 [n][L] (rounded to 6 decimals first when round_stats), lens i32 [n][L],
 * signals f32 [n][L][S] (centred zero padding; bases longer than S keep S samples in time order chosen by a
 * counter-based sampler keyed by (seed, read_uid[read], base index) -- the reference draws them from the
 * unseeded process-global random.sample, :247-249). */
int32_t dsp_extract_gather(void* stream, const dsp_read_batch* b, const double* shift, const double* scale,
                           const double* base_mean, const double* base_std, const int32_t* base_len,
                           const int64_t* base_lo, int64_t n_sites, const int32_t* site_read, const int32_t* site_loc,
                           int32_t seq_len, int32_t signal_len, int32_t round_stats, uint64_t seed,
                           const uint64_t* read_uid, uint8_t* kmer, float* means, float* stds, int32_t* lens,
                           float* signals);
/* The same gather with float64 outputs: the values the feature TSV prints (_features_to_str, :381-395, prints
 * doubles; use round_stats = 1).  Feeds dsp_format_feature_rows. */
int32_t dsp_extract_gather_f64(void* stream, const dsp_read_batch* b, const double* shift, const double* scale,
                               const double* base_mean, const double* base_std, const int32_t* base_len,
                               const int64_t* base_lo, int64_t n_sites, const int32_t* site_read,
                               const int32_t* site_loc, int32_t seq_len, int32_t signal_len, int32_t round_stats,
                               uint64_t seed, const uint64_t* read_uid, uint8_t* kmer, double* means, double* stds,
                               int32_t* lens, double* signals);
/* HOST: the feature-TSV rows of _features_to_str (extract_features.py:381-395) for n sites:
 *   sampleinfo \t k-mer \t means(csv) \t stds(csv) \t lens(csv) \t signals(';'-separated csv groups) \t label \n
 * with every float64 printed like str(numpy.float64) (shortest round-trip digits, positional for
 * 1e-4 <= |x| < 1e16, else scientific with a two-digit exponent).  Returns the bytes written, DSP_ENOMEM if
 * out_cap is too small. */
int64_t dsp_format_feature_rows(const char* text, const uint64_t* row_off, const uint32_t* info_len, const uint8_t* kmer,
                                const double* means, const double* stds, const int32_t* lens, const double* signals,
                                const int32_t* labels, int32_t seq_len, int32_t signal_len, int64_t n, char* out,
                                size_t out_cap, int32_t nthreads);
/* The same rows, left where the formatting threads wrote them (no compaction, no allocation: `extract` keeps `out`
 * across batches).  Thread t's rows start at out + part_off[t] and take part_len[t] bytes; the text is the parts in
 * order.  out_cap >= sum(info_len) + n * dsp_feature_row_bound(seq_len, signal_len); part_off / part_len hold nthreads
 * entries.  Returns the number of parts. */
int64_t dsp_format_feature_rows_parts(const char* text, const uint64_t* row_off, const uint32_t* info_len,
                                      const uint8_t* kmer, const double* means, const double* stds, const int32_t* lens,
                                      const double* signals, const int32_t* labels, int32_t seq_len, int32_t signal_len,
                                      int64_t n, char* out, size_t out_cap, int32_t nthreads, uint64_t* part_off,
                                      uint64_t* part_len);
uint64_t dsp_feature_row_bound(int32_t seq_len, int32_t signal_len);
/* HOST side of the same stage: motif sites of every read (get_refloc_of_methysite_in_motif,
 * utils/process_utils.py:97-112), the +/- strand coordinates and window bounds of _extract_features
 * (:346-358) and the per-read region bounds [rg_lo, rg_hi) (NULL = no region).  All pointers are HOST pointers.
 * motifs = n_motifs strings of motif_len bases back to back.  Writes site_read / site_loc and the sites'
 * sampleinfo strings (chrom \t pos \t strand \t pos_in_strand \t readname \t read_strand) into `info` with the
 * addressing arrays of dsp_parse_feature_rows.  Returns the number of sites; with site_read == NULL only counts
 * (and stores the info bytes needed in *info_bytes).  Reads are processed by `nthreads` host threads.  DSP_ENOMEM when max_sites / info_cap are too small,
 * DSP_EPARSE for a base outside the alphabet inside a window (the reference's KeyError). */
int64_t dsp_extract_sites(int64_t n_reads, const uint8_t* ev_base, const int64_t* ev_off, const char* const* chrom,
                          const char* const* readname, const char* read_strand, const char* align_strand,
                          const int64_t* chrom_start, const int64_t* chrom_len, const int64_t* rg_lo,
                          const int64_t* rg_hi, const char* motifs, int32_t n_motifs, int32_t motif_len,
                          int32_t methyloc, int32_t seq_len, int64_t max_sites, int32_t* site_read, int32_t* site_loc,
                          char* info, size_t info_cap, size_t* info_bytes, uint64_t* row_off, uint32_t* info_len,
                          uint32_t* read_off, uint32_t* read_len, int32_t nthreads);

/* ---- per-site modification frequency (the reference's `call_freq`) ----------------
 *
 * dsp_freq replaces calculate_mods_frequency + SiteStats/ModRecord (call_mods_freq.py:29-74,
 * utils/txt_formater.py:8-46): sites keyed by (chromosome, pos); a record is used when
 * |prob_0 - prob_1| >= prob_cf; prob sums, met/unmet/coverage per site; first used record fixes
 * strand / pos_in_strand / k-mer.  dsp_freq_add_calls_text consumes per-read call lines (the file call_mods
 * writes; `contig` non-NULL keeps only that chromosome, :53-54); dsp_freq_add_block consumes parsed call_mods
 * blocks + GPU results directly (no text round-trip, bit-identical sums).  dsp_freq_format replaces
 * write_sitekey2stats (:77-122): tsv or bedMethyl lines, insertion order or sorted by (chrom, pos); returns
 * the byte count (call with out=NULL to size the buffer). */
typedef struct dsp_freq dsp_freq;
dsp_freq* dsp_freq_create(double prob_cf);
void dsp_freq_destroy(dsp_freq* f);
/* host threads that parse the lines of dsp_freq_add_calls_text (default 1); the table is always updated by one
 * thread in line order, so the result does not depend on the thread count */
void dsp_freq_set_threads(dsp_freq* f, int32_t nthreads);
int64_t dsp_freq_add_calls_text(dsp_freq* f, const char* text, size_t len, const char* contig);
int64_t dsp_freq_add_block(dsp_freq* f, const char* text, const uint64_t* row_off, const uint32_t* info_len,
                           const float* probs, int32_t num_classes, const uint8_t* labels, const uint8_t* kmer,
                           int32_t seq_len, int64_t n);
void dsp_freq_counts(const dsp_freq* f, int64_t* count, int64_t* used, int64_t* sites);
int64_t dsp_freq_format(const dsp_freq* f, int32_t is_sort, int32_t is_bed, char* out, size_t cap);

/* ---- `call_freq` on the device and over several ranks (profiles/LAB_NOTEBOOK_r1_r3.md section 6c) ----------
 * Records stay in HBM as (key, packed, pos_in_strand, global row) and are reduced there.  Encoding:
 *   key    = chromosome id << 40 | pos   (ids from the handle's dictionary; INT64_MAX = record not used)
 *   meta   = strand code (0 '+', 1 '-') | 5-mer << 2 (4 bits per base, codes of process_utils.base2code_dna)
 *   packed = k0 | k1 << 20 | (label == 1) << 40 | meta << 41,  k = the printed probability in units of 1e-6
 * dsp_freq_block_keys (HOST): key / pos_in_strand / meta of the rows of a parsed call_mods block (multi-threaded).
 *   DSP_EINVAL when a row cannot be encoded (strand other than +/-, pos outside [0, 2^40), > 2^22 chromosomes):
 *   the caller must fall back to the host aggregator, loudly.
 * dsp_freq_dev_encode (DEVICE): probabilities + labels of the block (still in HBM after dsp_forward) -> key_out
 *   (INT64_MAX where |p0 - p1| < prob_cf, txt_formater.py:23-26) and packed_out.
 * dsp_freq_dev_sort_records (DEVICE): the four record columns STABLY sorted by `key`, out of place (a radix sort of
 *   (key, index) pairs, then one gather of the other columns).  Two calls: with tmp == NULL it only stores the scratch
 *   size for n records in *tmp_bytes; n < 2^32.  Sorting by the global row first and by the site key second restores
 *   file order inside every site whatever order the records arrived in.
 * dsp_freq_dev_count_sites / dsp_freq_dev_reduce (DEVICE): on records STABLY sorted by key: the number of sites, then
 *   per site (in no particular order; slot_counter is device scratch) its key, the global row / packed word /
 *   pos_in_strand of its first record, the double sums of p0 and p1 taken sequentially in record order, the
 *   methylated count and the coverage.
 * dsp_freq_add_sites (HOST): finished sites into the table dsp_freq_format prints; dsp_freq_add_counts adds to the
 *   number of records seen; dsp_freq_chrom_count / _name / dsp_freq_intern_chrom expose the chromosome dictionary
 *   so that ranks can agree on global ids. */
int64_t dsp_freq_block_keys(dsp_freq* f, const char* text, const uint64_t* row_off, const uint32_t* info_len,
                            const uint8_t* kmer, int32_t seq_len, int64_t n, int64_t* key, int64_t* pis, uint32_t* meta);
int32_t dsp_freq_dev_encode(void* stream, int64_t n, const float* probs, int32_t num_classes, const uint8_t* labels,
                            const int64_t* key_in, const uint32_t* meta_in, double prob_cf, int64_t* key_out,
                            int64_t* packed_out);
int32_t dsp_freq_dev_sort_records(void* stream, int64_t n, const int64_t* key, const int64_t* a, const int64_t* b,
                                  const int64_t* c, int64_t* key_out, int64_t* a_out, int64_t* b_out, int64_t* c_out,
                                  void* tmp, size_t* tmp_bytes);
int32_t dsp_freq_dev_count_sites(void* stream, int64_t n, const int64_t* key_sorted, int64_t* n_sites);
int32_t dsp_freq_dev_reduce(void* stream, int64_t n, const int64_t* key_sorted, const int64_t* packed_sorted,
                            const int64_t* pis_sorted, const int64_t* row_sorted, int64_t* slot_counter, int64_t cap,
                            int64_t* site_key, int64_t* site_first_row, int64_t* site_packed, int64_t* site_pis,
                            double* sum0, double* sum1, int64_t* met, int64_t* cov);
int64_t dsp_freq_add_sites(dsp_freq* f, int64_t n, const int64_t* key, const int64_t* first_row, const int64_t* packed_first,
                           const int64_t* pis, const double* sum0, const double* sum1, const int64_t* met, const int64_t* cov);
void dsp_freq_add_counts(dsp_freq* f, int64_t count);
int32_t dsp_freq_chrom_count(const dsp_freq* f);
int64_t dsp_freq_chrom_name(const dsp_freq* f, int32_t id, char* out, size_t cap);
int32_t dsp_freq_intern_chrom(dsp_freq* f, const char* name, size_t len);

/* ---- gzip I/O of the host pipeline (csrc/dsp_gz.cpp; replaces gzip.open at call_modifications.py:66-69, :264-270) ----
 * Everything this build writes with --gzip is a chain of BGZF members (ordinary gzip members of <= 64 KiB of text that
 * carry their compressed size, as written by htslib / bgzip; any gzip reader reads the chain as one stream) closed by
 * the empty EOF member: such files are indexed by a header walk, dealt to ranks by member ranges and inflated on N
 * threads.  Other .gz files are streamed through dsp_gz_open / dsp_gz_read (one thread: a deflate stream cannot be
 * entered in the middle).
 * dsp_gz_index: number of members if [src, src+len) is entirely BGZF (member_off[m+1] / member_isize[m] filled when
 *   given), -1 if it is not.  dsp_gz_inflate_members: members [m0, m1) -> out back to back; zlib checks every CRC. */
typedef struct dsp_gz_stream dsp_gz_stream;
int64_t dsp_gz_index(const uint8_t* src, size_t len, int64_t max_members, uint64_t* member_off, uint32_t* member_isize);
int64_t dsp_gz_inflate_members(const uint8_t* src, const uint64_t* member_off, const uint32_t* member_isize, int64_t m0,
                               int64_t m1, uint8_t* out, size_t out_cap, int32_t nthreads);
int32_t dsp_gz_member_rows(const uint8_t* src, const uint64_t* member_off, int64_t n_members, int64_t* rows);  /* newlines per member as recorded by dsp_bgzf_compress (gzip MTIME under XFL = 'R'); -1 = not recorded */
int64_t dsp_bgzf_compress(const uint8_t* in, size_t len, uint8_t* out, size_t out_cap, int32_t level, int32_t nthreads);
int64_t dsp_bgzf_eof(uint8_t* out, size_t cap);
dsp_gz_stream* dsp_gz_open(const char* path);
int64_t dsp_gz_read(dsp_gz_stream* s, uint8_t* out, size_t cap);   /* 0 = end; DSP_EPARSE: corrupt or TRUNCATED stream */
uint64_t dsp_gz_bytes_in(const dsp_gz_stream* s);                   /* compressed bytes consumed so far */
void dsp_gz_close(dsp_gz_stream* s);

/* ---- parallel inflate of ONE gzip stream (csrc/dsp_pgz.cpp): chunks of the compressed bytes are inflated concurrently
 * from block starts found by search, back-references into the unknown 32 KiB before a chunk kept as markers and resolved
 * front to back; every member's CRC-32 and ISIZE verified like zlib does.  Same reading contract as dsp_gz_read (0 = end,
 * DSP_EPARSE with the same messages on a corrupt or truncated stream).  open: NULL when the file is not a gzip file. */
typedef struct dsp_pgz dsp_pgz;
dsp_pgz* dsp_pgz_open(const char* path, int32_t nthreads, uint64_t chunk_bytes /* 0 = default */);
int64_t dsp_pgz_read(dsp_pgz* z, uint8_t* out, size_t cap);
uint64_t dsp_pgz_bytes_in(const dsp_pgz* z);
void dsp_pgz_stats(const dsp_pgz* z, uint64_t* rounds, uint64_t* dropped_chunks);
void dsp_pgz_close(dsp_pgz* z);

/* ---- node-local ring of text blocks in POSIX shared memory (csrc/dsp_shmring.cpp) ---------------------------------
 * A feature file written by the reference's `extract --gzip` is ONE gzip stream (read back with gzip.open at
 * call_modifications.py:66-69): it cannot be range-split, so the first rank of a node inflates it once into this ring
 * as blocks of complete rows and every rank of the node copies ITS blocks out (block i -> rank i % world).  One
 * producer, any number of consumers, each block consumed by exactly one of them.  Sequence numbers are node-local and
 * dense.  Names are shm_open names ("/..."); timeouts in seconds.
 * create: reserves the memory up front (DSP_ENOMEM via dsp_last_error when /dev/shm is too small: the caller falls back
 *   to per-rank inflation); attach: waits for the creator.
 * acquire (producer): payload of block seq once the consumer of block seq - n_slots released it; publish: makes it
 *   visible with its length, the global index of its first row and its row count; finish: end of stream after n_blocks
 *   (status 0) or the producer's failure (status < 0, message).
 * wait (consumer): 0 = block there (*data points into the ring: copy out, then release), 1 = stream ended before seq,
 *   < 0 = producer failure (its message in dsp_last_error) or timeout.  abort: a failing consumer unblocks the producer. */
typedef struct dsp_shm_ring dsp_shm_ring;
dsp_shm_ring* dsp_shm_ring_create(const char* name, int32_t n_slots, uint64_t slot_bytes);
dsp_shm_ring* dsp_shm_ring_attach(const char* name, double timeout_s);
uint64_t dsp_shm_ring_slot_bytes(const dsp_shm_ring* r);
uint8_t* dsp_shm_ring_acquire(dsp_shm_ring* r, uint64_t seq, double timeout_s);
int32_t dsp_shm_ring_publish(dsp_shm_ring* r, uint64_t seq, uint64_t len, uint64_t first_row, uint64_t n_rows);
int32_t dsp_shm_ring_finish(dsp_shm_ring* r, uint64_t n_blocks, int32_t status, const char* message);
int32_t dsp_shm_ring_wait(dsp_shm_ring* r, uint64_t seq, double timeout_s, const uint8_t** data, uint64_t* len,
                          uint64_t* first_row, uint64_t* n_rows);
int32_t dsp_shm_ring_release(dsp_shm_ring* r, uint64_t seq);
void dsp_shm_ring_abort(dsp_shm_ring* r);
void dsp_shm_ring_unlink(dsp_shm_ring* r);   /* creator: drop the name once everybody has attached (a killed run leaves nothing) */
void dsp_shm_ring_close(dsp_shm_ring* r, int32_t unlink_it);

/* ---- fast5 ingestion (csrc/dsp_fast5.cpp): what extract_features.py:44-91 (_get_label_raw), :94-176
 * (_get_alignment_info_from_fast5) and :255-270 (_get_scaling_of_a_read) read from one tombo-resquiggled single-read
 * fast5 through h5py, read here through the HDF5 C library itself (found with dlopen at run time: DSP_HDF5_LIB, the usual
 * sonames, conda / system library directories; HDF5 >= 1.10).
 * dsp_fast5_load fills `out` (arrays are malloc'ed, release with dsp_fast5_free) and returns 0; DSP_FAST5_SKIPPED when
 * `only_chrom` is given and the read maps elsewhere, has no Alignment group or cannot be opened (the reference's region
 * filter at :308-309 runs before anything else is read); a negative dsp_status whose message is the reference's
 * exception text otherwise (the caller counts the file as failed, :373-375).  A file without an Alignment group loads with
 * has_alignment = 0 (the reference carries on with empty fields and fails later, at :327).
 * ev_start already includes the Events attribute read_start_rel_to_raw (:81).  Thread-safe (one internal lock). */
#define DSP_FAST5_SKIPPED 1
typedef struct dsp_fast5_read {
    int64_t n_raw;
    int16_t* raw;              /* Raw/Reads/<first read>/Signal */
    int64_t n_events;
    int64_t* ev_start;         /* Events 'start' + read_start_rel_to_raw */
    int64_t* ev_len;           /* Events 'length' */
    uint8_t* ev_base;          /* Events 'base' (ASCII) */
    double digitisation, range, offset; /* UniqueGlobalKey/channel_id */
    int64_t mapped_start;
    int32_t has_alignment;
    int32_t reserved;
    char read_id[256];
    char mapped_chrom[256];
    char mapped_strand[8];
} dsp_fast5_read;
int32_t dsp_fast5_available(void);       /* 1 if an HDF5 library was found (else dsp_last_error says what was tried) */
const char* dsp_fast5_library(void);     /* path of the library in use, "" if none */
int32_t dsp_fast5_load(const char* path, const char* corrected_group, const char* basecall_subgroup, const char* only_chrom,
                       dsp_fast5_read* out);
void dsp_fast5_free(dsp_fast5_read* r);

/* What a handle decided about its device when it was made (diagnostics; the tests assert that the clustered small-batch
 * launches are really on where they are expected): DSP_QUERY_CLUSTERING 1 = batches <= 2,048 sites may run clustered launches
 * (0: DSP_LSTM_CLUSTER=0, or the XCC_ID probe of dsp_model_create did not find consecutive blocks on consecutive XCDs);
 * DSP_QUERY_XCC_PROBE_FAILED; DSP_QUERY_COMPUTE_UNITS. */
#define DSP_QUERY_CLUSTERING 0
#define DSP_QUERY_XCC_PROBE_FAILED 1
#define DSP_QUERY_COMPUTE_UNITS 2
int32_t dsp_model_query(const dsp_model* m, int32_t what);

/* ---- which GPU is this? (multi-GPU runs: rank placement and the bench line's proof of N distinct devices) -----------
 * The reference maps its model processes to devices by index only (call_modifications.py:523-529 _get_gpus, :613-621);
 * one process per GPU over RCCL additionally wants (a) the CPUs next to each GPU (dist.pin_rank) and (b) evidence in the
 * output that N ranks ran on N different devices (bench.py).  dsp_device_pci_bdf writes the sysfs name of HIP device
 * `device` ("0000:c1:00.0", lower case: /sys/bus/pci/devices/<that>/numa_node) and returns its length, DSP_EHIP when
 * the runtime does not know the device; dsp_device_uuid writes the 16 bytes of hipDeviceProp_t::uuid as 32 hex digits. */
int64_t dsp_device_pci_bdf(int32_t device, char* out, size_t cap);
int64_t dsp_device_uuid(int32_t device, char* out, size_t cap);

/* Test hook (round 6; no device involved): a DRY RUN of dsp_forward for model `cfg` on a device of n_cus compute units -- a
 * handle whose allocations are made-up addresses runs the whole host half of a forward of n_sites sites (the cut into pieces, the
 * kernel form of every launch, the extents behind every buffer descriptor, the checks of the launch wrappers), and every launch
 * is NOTED -- "(kernel<...>) grid x,y block b lds n", one line each in `log` -- instead of made.  init_mode: DSP_INIT_*;
 * precision: DSP_PREC_*; extents: "region" (default) / "tight" / "wide" (DSP_RSRC_EXTENTS).  Returns the number of launches, or
 * the dsp_status a real forward of that shape would have failed with before reaching the device (a refused launch: a pointer
 * without the end of its allocation, an extent past it, a clustered launch on counters that were not zeroed, ...). */
int32_t dsp_debug_dry_run(const dsp_model_cfg* cfg, int32_t n_cus, int64_t n_sites, int32_t init_mode, int32_t precision,
                          const char* extents, char* log, size_t log_cap);

/* Test hook (round 6; no device involved): the three bf16 pieces the split-precision modes cut an fp32 weight into
 * (csrc/dsp_capi.cpp pack_lstm_dir_split: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), round to nearest even);
 * tests/test_split_exact.py holds hi + mid + lo == x bit for bit over 1e7 random values and every exponent edge, and states
 * what happens where it cannot hold (|x| < 2^-109: the low piece underflows bf16's subnormals -- an absolute error below
 * 2^-133; |x| > the largest bf16: hi is infinite -- visible, never silent). */
void dsp_debug_split_bf16(const float* x, int64_t n, uint16_t* hi, uint16_t* mid, uint16_t* lo);

/* Test hook (round 6; no device involved): how dsp_forward cuts a call of n_sites sites of model `cfg` on a device of n_cus
 * compute units -- whole rounds of n_cus x 32 sites first, the remainder as the cheapest sequence of small-batch pieces by the
 * launch geometry's cost model (csrc/dsp_capi.cpp piece_cost_us) -- and what it estimates for each piece (microseconds; may be
 * NULL).  pieces / cost_us: room for 9 entries.  Returns the number of pieces (1: the call runs as one). */
int32_t dsp_debug_plan(const dsp_model_cfg* cfg, int32_t n_cus, int64_t n_sites, int64_t pieces[9], double cost_us[9]);
/* ... and the model's estimate for n_sites sites run as ONE piece (microseconds; < 0: bad arguments) */
double dsp_debug_piece_cost(const dsp_model_cfg* cfg, int32_t n_cus, int64_t n_sites);

/* Test hook (round 6): what this device does with an access past a buffer descriptor's num_records -- the hardware range
 * check the forward's descriptors rely on since they carry real extents.  One wave reads 1,024 bytes of ones through a
 * descriptor of 256 bytes: out[0] = lanes inside the extent that read their data (expect 16), out[1] = lanes past it by VGPR
 * offset that read zeros (48), out[2] = lanes past it by SGPR offset that read zeros (64: the kernels carry almost all of an
 * address there), out[3] = floats of the 4 KiB buffer still 1.0 after 64 out-of-range stores (1,024: dropped). */
int32_t dsp_debug_range_probe(int32_t device, int32_t out[4]);

const char* dsp_last_error(void);
int32_t dsp_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DSP_AMD_H */
