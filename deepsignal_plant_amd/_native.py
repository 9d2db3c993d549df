"""ctypes binding of libdsp_amd.so (include/dsp_amd.h).  Fails loudly when the library is absent."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DSP_AMD_LIB") or os.path.join(_HERE, "libdsp_amd.so")
ABI_VERSION = 3   # include/dsp_amd.h DSP_AMD_ABI_VERSION

DSP_OK, DSP_EINVAL, DSP_ESHAPE, DSP_EHIP, DSP_ENOMEM, DSP_EPARSE, DSP_EKEY, DSP_EBOUNDS = 0, -1, -2, -3, -4, -5, -6, -7
MODULE_CODE = {"both_bilstm": 0, "seq_bilstm": 1, "signal_bilstm": 2}
DT_F32, DT_U8, DT_U16, DT_I32 = 0, 1, 2, 3
INIT_ZEROS, INIT_EXPLICIT, INIT_PHILOX = 0, 1, 2
NORM_MAD, NORM_ZSCORE = 0, 1
PRECISION = {"fp32": 0, "fp16x3": 3, "bf16x6": 6, "bf16x9": 9}


class ModelCfg(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in (
        "seq_len", "signal_len", "num_layers1", "num_layers2", "num_classes", "hidden_size", "vocab_size",
        "embedding_size", "is_base", "is_signallen", "module")]


class InitState(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int32), ("seed", ctypes.c_uint64), ("site_offset", ctypes.c_uint64),
                ("h_seq", ctypes.c_void_p), ("c_seq", ctypes.c_void_p), ("h_sig", ctypes.c_void_p),
                ("c_sig", ctypes.c_void_p), ("h_comb", ctypes.c_void_p), ("c_comb", ctypes.c_void_p),
                ("site_keys", ctypes.c_void_p)]


_lib = None


class Fast5Read(ctypes.Structure):  # dsp_fast5_read, include/dsp_amd.h
    _fields_ = [("n_raw", ctypes.c_int64), ("raw", ctypes.POINTER(ctypes.c_int16)),
                ("n_events", ctypes.c_int64), ("ev_start", ctypes.POINTER(ctypes.c_int64)),
                ("ev_len", ctypes.POINTER(ctypes.c_int64)), ("ev_base", ctypes.POINTER(ctypes.c_uint8)),
                ("digitisation", ctypes.c_double), ("range", ctypes.c_double), ("offset", ctypes.c_double),
                ("mapped_start", ctypes.c_int64), ("has_alignment", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("read_id", ctypes.c_char * 256), ("mapped_chrom", ctypes.c_char * 256), ("mapped_strand", ctypes.c_char * 8)]


# Set by the fast5 reader's worker processes (reads.ReadBatches(procs=...)) before their first lib() call: they only use
# the host-side entry points, never the GPU, so they skip the 1.5 s torch import.  Never set it in a process that computes.
NO_TORCH = False


def lib():
    """Load libdsp_amd.so or raise: the product path never falls back to a CPU implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libdsp_amd.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  Import torch FIRST so that
    # libdsp_amd.so's NEEDED libamdhip64.so.7 resolves to the already-loaded runtime; loading ours first
    # would leave two HIP runtimes in one process (the second one reports "no ROCm-capable device").
    if not NO_TORCH:
        import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    L.dsp_last_error.restype = ctypes.c_char_p
    L.dsp_abi_version.restype = ctypes.c_int32
    # an older or variant library (DSP_AMD_LIB is how the A/B scripts and the bounds / trace builds are loaded) must say so
    # itself, not fail later with a bare AttributeError on the first symbol it lacks
    if L.dsp_abi_version() != ABI_VERSION:
        raise RuntimeError("%s implements C-ABI version %d, this package binds version %d (include/dsp_amd.h "
                           "DSP_AMD_ABI_VERSION): rebuild it with `make -C deepsignal_plant_amd/csrc`"
                           % (LIB_PATH, L.dsp_abi_version(), ABI_VERSION))
    L.dsp_weight_count.restype = ctypes.c_int32
    L.dsp_weight_count.argtypes = [ctypes.POINTER(ModelCfg)]
    L.dsp_weight_spec.restype = ctypes.c_int32
    L.dsp_weight_spec.argtypes = [ctypes.POINTER(ModelCfg), ctypes.c_int32, ctypes.c_char_p, ctypes.c_size_t,
                                  ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32)]
    L.dsp_flops_per_site.restype = ctypes.c_int64
    L.dsp_flops_per_site.argtypes = [ctypes.POINTER(ModelCfg)]
    L.dsp_model_create.restype = ctypes.c_int32
    L.dsp_model_create.argtypes = [ctypes.POINTER(ModelCfg), ctypes.POINTER(ctypes.c_void_p),
                                   ctypes.POINTER(ctypes.c_int64), ctypes.c_int32, ctypes.c_int32,
                                   ctypes.POINTER(ctypes.c_void_p)]
    L.dsp_model_reserve.restype = ctypes.c_int32
    L.dsp_model_reserve.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    L.dsp_workspace_bytes.restype = ctypes.c_size_t
    L.dsp_workspace_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    L.dsp_forward.restype = ctypes.c_int32
    L.dsp_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32,
                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                              ctypes.POINTER(InitState), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.dsp_debug_read_activation.restype = ctypes.c_int32
    L.dsp_debug_read_activation.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                            ctypes.c_void_p]
    L.dsp_profile_enable.restype = ctypes.c_int32
    L.dsp_profile_enable.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    L.dsp_profile_read.restype = ctypes.c_int32
    L.dsp_profile_read.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t,
                                   ctypes.POINTER(ctypes.c_float), ctypes.c_int32]
    L.dsp_model_set_precision.restype = ctypes.c_int32
    L.dsp_model_set_precision.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    L.dsp_model_destroy.restype = None
    L.dsp_model_destroy.argtypes = [ctypes.c_void_p]
    L.dsp_find_row_end.restype = ctypes.c_int64
    L.dsp_find_row_end.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int64]
    L.dsp_count_rows.restype = ctypes.c_int64
    L.dsp_count_rows.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    L.dsp_copy_rows_index.restype = ctypes.c_int64
    L.dsp_copy_rows_index.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    L.dsp_read_rows_index.restype = ctypes.c_int64
    L.dsp_read_rows_index.argtypes = [ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64,
                                      ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.dsp_parse_rows_device.restype = ctypes.c_int32
    L.dsp_parse_rows_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                        ctypes.c_int32] + [ctypes.c_void_p] * 12 + [ctypes.c_uint64]
    L.dsp_parse_feature_rows.restype = ctypes.c_int64
    L.dsp_parse_feature_rows.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32,
                                         ctypes.c_int64] + [ctypes.c_void_p] * 10 + [ctypes.c_int32]
    L.dsp_format_calls.restype = ctypes.c_int64
    L.dsp_format_calls.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p,
                                   ctypes.c_size_t, ctypes.c_int32]
    L.dsp_freq_create.restype = ctypes.c_void_p
    L.dsp_freq_create.argtypes = [ctypes.c_double]
    L.dsp_freq_set_threads.restype = None
    L.dsp_freq_set_threads.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    L.dsp_freq_destroy.restype = None
    L.dsp_freq_destroy.argtypes = [ctypes.c_void_p]
    L.dsp_freq_add_calls_text.restype = ctypes.c_int64
    L.dsp_freq_add_calls_text.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p]
    L.dsp_freq_add_block.restype = ctypes.c_int64
    L.dsp_freq_add_block.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64]
    L.dsp_freq_counts.restype = None
    L.dsp_freq_counts.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int64)] * 3
    L.dsp_freq_format.restype = ctypes.c_int64
    L.dsp_freq_format.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_size_t]
    L.dsp_freq_block_keys.restype = ctypes.c_int64
    L.dsp_freq_block_keys.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int32, ctypes.c_int64] + [ctypes.c_void_p] * 3
    L.dsp_freq_dev_encode.restype = ctypes.c_int32
    L.dsp_freq_dev_encode.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    L.dsp_freq_dev_sort_records.restype = ctypes.c_int32
    L.dsp_freq_dev_sort_records.argtypes = [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 9 + [ctypes.POINTER(ctypes.c_size_t)]
    L.dsp_freq_dev_count_sites.restype = ctypes.c_int32
    L.dsp_freq_dev_count_sites.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    L.dsp_freq_dev_reduce.restype = ctypes.c_int32
    L.dsp_freq_dev_reduce.argtypes = [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_int64] + [ctypes.c_void_p] * 8
    L.dsp_freq_add_sites.restype = ctypes.c_int64
    L.dsp_freq_add_sites.argtypes = [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 8
    L.dsp_freq_add_counts.restype = None
    L.dsp_freq_add_counts.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    L.dsp_freq_chrom_count.restype = ctypes.c_int32
    L.dsp_freq_chrom_count.argtypes = [ctypes.c_void_p]
    L.dsp_freq_chrom_name.restype = ctypes.c_int64
    L.dsp_freq_chrom_name.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_size_t]
    L.dsp_freq_intern_chrom.restype = ctypes.c_int32
    L.dsp_freq_intern_chrom.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    L.dsp_gz_member_rows.restype = ctypes.c_int32
    L.dsp_gz_member_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    L.dsp_gz_index.restype = ctypes.c_int64
    L.dsp_gz_index.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    L.dsp_gz_inflate_members.restype = ctypes.c_int64
    L.dsp_gz_inflate_members.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                         ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32]
    L.dsp_bgzf_compress.restype = ctypes.c_int64
    L.dsp_bgzf_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32]
    L.dsp_bgzf_eof.restype = ctypes.c_int64
    L.dsp_bgzf_eof.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    L.dsp_fast5_available.restype = ctypes.c_int32
    L.dsp_fast5_available.argtypes = []
    L.dsp_fast5_library.restype = ctypes.c_char_p
    L.dsp_fast5_library.argtypes = []
    L.dsp_fast5_load.restype = ctypes.c_int32
    L.dsp_fast5_load.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(Fast5Read)]
    L.dsp_fast5_free.restype = None
    L.dsp_fast5_free.argtypes = [ctypes.POINTER(Fast5Read)]
    L.dsp_gz_open.restype = ctypes.c_void_p
    L.dsp_gz_open.argtypes = [ctypes.c_char_p]
    L.dsp_gz_read.restype = ctypes.c_int64
    L.dsp_gz_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.dsp_gz_close.restype = None
    L.dsp_gz_close.argtypes = [ctypes.c_void_p]
    L.dsp_gz_bytes_in.restype = ctypes.c_uint64
    L.dsp_gz_bytes_in.argtypes = [ctypes.c_void_p]
    L.dsp_pgz_open.restype = ctypes.c_void_p
    L.dsp_pgz_open.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_uint64]
    L.dsp_pgz_read.restype = ctypes.c_int64
    L.dsp_pgz_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.dsp_pgz_bytes_in.restype = ctypes.c_uint64
    L.dsp_pgz_bytes_in.argtypes = [ctypes.c_void_p]
    L.dsp_pgz_stats.restype = None
    L.dsp_pgz_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    L.dsp_pgz_close.restype = None
    L.dsp_pgz_close.argtypes = [ctypes.c_void_p]
    L.dsp_shm_ring_create.restype = ctypes.c_void_p
    L.dsp_shm_ring_create.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_uint64]
    L.dsp_shm_ring_attach.restype = ctypes.c_void_p
    L.dsp_shm_ring_attach.argtypes = [ctypes.c_char_p, ctypes.c_double]
    L.dsp_shm_ring_slot_bytes.restype = ctypes.c_uint64
    L.dsp_shm_ring_slot_bytes.argtypes = [ctypes.c_void_p]
    L.dsp_shm_ring_acquire.restype = ctypes.c_void_p
    L.dsp_shm_ring_acquire.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_double]
    L.dsp_shm_ring_publish.restype = ctypes.c_int32
    L.dsp_shm_ring_publish.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]
    L.dsp_shm_ring_finish.restype = ctypes.c_int32
    L.dsp_shm_ring_finish.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32, ctypes.c_char_p]
    L.dsp_shm_ring_wait.restype = ctypes.c_int32
    L.dsp_shm_ring_wait.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_double, ctypes.POINTER(ctypes.c_void_p),
                                    ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                    ctypes.POINTER(ctypes.c_uint64)]
    L.dsp_shm_ring_release.restype = ctypes.c_int32
    L.dsp_shm_ring_release.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    L.dsp_shm_ring_abort.restype = None
    L.dsp_shm_ring_abort.argtypes = [ctypes.c_void_p]
    L.dsp_shm_ring_unlink.restype = None
    L.dsp_shm_ring_unlink.argtypes = [ctypes.c_void_p]
    L.dsp_shm_ring_close.restype = None
    L.dsp_shm_ring_close.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    L.dsp_feat_writer_create.restype = ctypes.c_int32
    L.dsp_feat_writer_create.argtypes = [ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                         ctypes.POINTER(ctypes.c_void_p)]
    L.dsp_feat_writer_add.restype = ctypes.c_int32
    L.dsp_feat_writer_add.argtypes = [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 11
    L.dsp_feat_writer_close.restype = ctypes.c_int32
    L.dsp_feat_writer_close.argtypes = [ctypes.c_void_p]
    L.dsp_feat_open.restype = ctypes.c_int32
    L.dsp_feat_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
    L.dsp_feat_info.restype = ctypes.c_int32
    L.dsp_feat_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    L.dsp_feat_block_info.restype = ctypes.c_int32
    L.dsp_feat_block_info.argtypes = [ctypes.c_void_p, ctypes.c_int64] + [ctypes.POINTER(ctypes.c_int64)] * 3
    L.dsp_feat_read_block.restype = ctypes.c_int64
    L.dsp_feat_read_block.argtypes = ([ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64] + [ctypes.c_void_p] * 7 +
                                      [ctypes.c_size_t] + [ctypes.c_void_p] * 4 + [ctypes.c_int32])
    L.dsp_feat_close.restype = None
    L.dsp_feat_close.argtypes = [ctypes.c_void_p]
    L.dsp_extract_normalize.restype = ctypes.c_int32
    L.dsp_extract_normalize.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    L.dsp_extract_base_stats.restype = ctypes.c_int32
    L.dsp_extract_base_stats.argtypes = [ctypes.c_void_p] * 9
    L.dsp_extract_gather.restype = ctypes.c_int32
    L.dsp_extract_gather.argtypes = ([ctypes.c_void_p] * 8 + [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p] +
                                     [ctypes.c_int32] * 3 + [ctypes.c_uint64] + [ctypes.c_void_p] * 6)
    L.dsp_extract_sites.restype = ctypes.c_int64
    L.dsp_extract_sites.argtypes = ([ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_char_p, ctypes.c_char_p] + [ctypes.c_void_p] * 4 +
                                    [ctypes.c_char_p] + [ctypes.c_int32] * 4 + [ctypes.c_int64, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)] +
                                    [ctypes.c_void_p] * 4 + [ctypes.c_int32])
    L.dsp_extract_gather_f64.restype = ctypes.c_int32
    L.dsp_extract_gather_f64.argtypes = L.dsp_extract_gather.argtypes
    L.dsp_format_feature_rows.restype = ctypes.c_int64
    L.dsp_format_feature_rows.argtypes = ([ctypes.c_void_p] * 9 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                          ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32])
    L.dsp_format_feature_rows_parts.restype = ctypes.c_int64
    L.dsp_format_feature_rows_parts.argtypes = L.dsp_format_feature_rows.argtypes + [ctypes.c_void_p, ctypes.c_void_p]
    L.dsp_feature_row_bound.restype = ctypes.c_uint64
    L.dsp_feature_row_bound.argtypes = [ctypes.c_int32, ctypes.c_int32]
    L.dsp_model_query.restype = ctypes.c_int32
    L.dsp_model_query.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    for fn in (L.dsp_device_pci_bdf, L.dsp_device_uuid):
        fn.restype = ctypes.c_int64
        fn.argtypes = [ctypes.c_int32, ctypes.c_char_p, ctypes.c_size_t]
    L.dsp_debug_plan.restype = ctypes.c_int32
    L.dsp_debug_plan.argtypes = [ctypes.POINTER(ModelCfg), ctypes.c_int32, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64),
                                 ctypes.POINTER(ctypes.c_double)]
    L.dsp_debug_dry_run.restype = ctypes.c_int32
    L.dsp_debug_dry_run.argtypes = [ctypes.POINTER(ModelCfg), ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_char_p,
                                    ctypes.c_char_p, ctypes.c_size_t]
    L.dsp_debug_split_bf16.restype = None
    L.dsp_debug_split_bf16.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.dsp_debug_piece_cost.restype = ctypes.c_double
    L.dsp_debug_piece_cost.argtypes = [ctypes.POINTER(ModelCfg), ctypes.c_int32, ctypes.c_int64]
    L.dsp_debug_range_probe.restype = ctypes.c_int32
    L.dsp_debug_range_probe.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]
    _lib = L
    return L


def range_probe(device: int = 0):
    """(lanes in range that read their data, lanes past the extent by VGPR offset that read zeros, ... by SGPR offset,
    floats untouched by 64 out-of-range stores): expect (16, 48, 64, 1024) -- include/dsp_amd.h dsp_debug_range_probe"""
    out = (ctypes.c_int32 * 4)()
    check(int(lib().dsp_debug_range_probe(int(device), out)))
    return tuple(int(v) for v in out)


def device_pci_bdf(device: int) -> str:
    """sysfs name of HIP device `device` ("0000:c1:00.0"): /sys/bus/pci/devices/<that>/numa_node, bench.py's proof of N GPUs"""
    buf = ctypes.create_string_buffer(64)
    check(int(lib().dsp_device_pci_bdf(int(device), buf, 64)))
    return buf.value.decode()


def device_uuid(device: int) -> str:
    buf = ctypes.create_string_buffer(40)
    check(int(lib().dsp_device_uuid(int(device), buf, 40)))
    return buf.value.decode()


def last_error() -> str:
    return (lib().dsp_last_error() or b"").decode("utf-8", "replace")


def check(rc: int):
    """Map C status codes onto the exceptions the reference raises for the same conditions."""
    if rc >= 0:
        return rc
    msg = last_error()
    if rc == DSP_EINVAL:
        raise ValueError(msg)
    if rc == DSP_EPARSE:
        if "bad field count" in msg:   # a row with fewer than 12 fields (a blank line too): words[k] of the reference's reader
            raise IndexError("list index out of range (%s)" % msg)   # (call_modifications.py:84-86, :117) is an IndexError
        raise ValueError(msg)
    if rc == DSP_EKEY:   # base2code_dna[x] of the reference's reader (call_modifications.py:84): KeyError(x)
        err = KeyError(msg.split("'")[1] if msg.count("'") >= 2 else msg)
        err.detail = msg
        raise err
    raise RuntimeError(msg)
