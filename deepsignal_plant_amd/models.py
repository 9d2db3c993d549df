"""ModelBiLSTM -- host-side mirror of deepsignal_plant/models.py:99-240 over the gfx950 kernels.

Same constructor arguments, same state_dict keys/shapes, same call signature and return value
(logits, softmax) as the reference class; the arithmetic runs in libdsp_amd.so (hand-written HIP, fp32
MFMA) through the C ABI in include/dsp_amd.h.  torch is used only for device memory, streams and
checkpoint I/O.  There is no CPU path: forward() on a non-GPU tensor or without the library raises.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict

import torch

from . import _native as nat

_DT = {torch.float32: nat.DT_F32, torch.uint8: nat.DT_U8, torch.uint16: nat.DT_U16, torch.int32: nat.DT_I32}


class ModelBiLSTM(object):
    """BiLSTM per-site 5mC classifier (reference: models.py:99-164).

    Extra, build-only keyword: ``init_state`` -- how the LSTM initial states are produced.  The reference
    draws h0,c0 ~ N(0,1) with torch.randn on every forward (models.py:169-176); here
    ``"randn"`` (default) = in-kernel Philox N(0,1) keyed by (seed, global site index), ``"zeros"``, or pass
    explicit tensors per call with ``forward(..., init_states=dict(h_seq=..., c_seq=..., ...))``.
    """

    def __init__(self, seq_len=13, signal_len=16, num_layers1=3, num_layers2=1, num_classes=2,
                 dropout_rate=0.5, hidden_size=256, vocab_size=16, embedding_size=4, is_base=True,
                 is_signallen=True, module="both_bilstm", device=0, init_state="randn", seed=0):
        if module not in nat.MODULE_CODE:
            raise ValueError("--model_type is not right!")  # models.py:127-128
        if init_state not in ("randn", "zeros"):
            raise ValueError("init_state must be 'randn' or 'zeros'")
        self.model_type = 'BiLSTM'
        self.module = module
        self.device = device
        self.seq_len, self.signal_len = int(seq_len), int(signal_len)
        self.num_layers1, self.num_layers2 = int(num_layers1), int(num_layers2)
        self.num_classes, self.hidden_size = int(num_classes), int(hidden_size)
        self.vocab_size, self.embedding_size = int(vocab_size), int(embedding_size)
        self.is_base, self.is_signallen = bool(is_base), bool(is_signallen)
        self.dropout_rate = dropout_rate  # identity at inference (call_modifications.py:228)
        self.init_state, self.seed = init_state, int(seed)
        self.site_offset = 0  # advanced by the caller so Philox draws depend on the global site index
        self._cfg = nat.ModelCfg(self.seq_len, self.signal_len, self.num_layers1, self.num_layers2,
                                 self.num_classes, self.hidden_size, self.vocab_size, self.embedding_size,
                                 int(self.is_base), int(self.is_signallen), nat.MODULE_CODE[module])
        self._spec = self._query_spec()
        # parameters start as zeros; real values come from load_state_dict (call_modifications.py:219-223)
        self._params = OrderedDict((k, torch.zeros(shp, dtype=torch.float32)) for k, shp in self._spec)
        self._handle = None
        self._precision = None
        self._training = True

    # ---- reference-compatible surface -------------------------------------------------------------
    def get_model_type(self):
        return self.model_type

    def _query_spec(self):
        L = nat.lib()
        n = nat.check(L.dsp_weight_count(ctypes.byref(self._cfg)))
        spec = []
        name = ctypes.create_string_buffer(128)
        shape = (ctypes.c_int64 * 2)()
        nd = ctypes.c_int32()
        for i in range(n):
            nat.check(L.dsp_weight_spec(ctypes.byref(self._cfg), i, name, 128, shape, ctypes.byref(nd)))
            spec.append((name.value.decode(), tuple(int(shape[j]) for j in range(nd.value))))
        return spec

    def state_dict(self):
        return OrderedDict((k, v.clone()) for k, v in self._params.items())

    def load_state_dict(self, state_dict, strict=True):
        """Strict key/shape validation with torch's error wording (RuntimeError), as the reference's
        load_state_dict at call_modifications.py:223 would raise."""
        missing = [k for k in self._params if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._params]
        errs = []
        if strict and unexpected:
            errs.append('Unexpected key(s) in state_dict: {}. '.format(', '.join('"%s"' % k for k in unexpected)))
        if strict and missing:
            errs.append('Missing key(s) in state_dict: {}. '.format(', '.join('"%s"' % k for k in missing)))
        for k, v in self._params.items():
            if k in state_dict and tuple(state_dict[k].shape) != tuple(v.shape):
                errs.append('size mismatch for {}: copying a param with shape {} from checkpoint, the shape in '
                            'current model is {}.'.format(k, tuple(state_dict[k].shape), tuple(v.shape)))
        if errs:
            raise RuntimeError('Error(s) in loading state_dict for ModelBiLSTM:\n\t' + "\n\t".join(errs))
        for k in self._params:
            if k in state_dict:
                self._params[k] = state_dict[k].detach().to(device="cpu", dtype=torch.float32).contiguous().clone()
        self._release()
        return self

    def cuda(self, device=None):
        if device is not None:
            self.device = device
        self._ensure_handle()
        return self

    def eval(self):
        self._training = False
        return self

    def parameters_count(self):
        return sum(v.numel() for v in self._params.values())

    def flops_per_site(self):
        return int(nat.lib().dsp_flops_per_site(ctypes.byref(self._cfg)))

    # ---- native handle ------------------------------------------------------------------------------
    def _device_index(self):
        d = self.device
        if isinstance(d, torch.device):
            return d.index or 0
        if isinstance(d, str):
            return torch.device(d).index or 0
        return int(d)

    def _ensure_handle(self):
        if self._handle is not None:
            return
        L = nat.lib()
        tensors = [self._params[k] for k, _ in self._spec]
        ptrs = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        numels = (ctypes.c_int64 * len(tensors))(*[t.numel() for t in tensors])
        h = ctypes.c_void_p()
        nat.check(L.dsp_model_create(ctypes.byref(self._cfg), ptrs, numels, len(tensors), self._device_index(),
                                     ctypes.byref(h)))
        self._handle = h
        if self._precision is not None:
            nat.check(L.dsp_model_set_precision(h, nat.PRECISION[self._precision]))

    def set_precision(self, precision):
        """How the fp32 products of the combined BiLSTM stack are evaluated: "fp32" (fp32 MFMA, the default),
        "bf16x9" / "bf16x6" (split-bf16 emulation on the bf16 matrix cores, include/dsp_amd.h).  None = whatever
        DSP_PRECISION says (default fp32)."""
        if precision is not None and precision not in nat.PRECISION:
            raise ValueError("precision must be one of %s" % sorted(nat.PRECISION))
        self._precision = precision
        if self._handle is not None and precision is not None:
            nat.check(nat.lib().dsp_model_set_precision(self._handle, nat.PRECISION[precision]))
        return self

    def _release(self):
        if self._handle is not None:
            nat.lib().dsp_model_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def reserve(self, max_sites):
        self._ensure_handle()
        nat.check(nat.lib().dsp_model_reserve(self._handle, int(max_sites)))

    # ---- forward ------------------------------------------------------------------------------------
    def forward(self, kmer, base_means, base_stds, base_signal_lens, signals, init_states=None, want_labels=False,
                site_keys=None):
        """(logits[B,C], softmax[B,C]) exactly like models.py:178-240.  Inputs are torch tensors on this
        model's GPU; kmer / base_signal_lens may be float32 (reference convention) or uint8/uint16/int32.
        site_keys (init_state="randn" only): int64 / uint64 tensor [B] on the GPU naming each site for the in-kernel
        initial-state generator (dsp_init_state.site_keys); default = self.site_offset + row index."""
        self._ensure_handle()
        dev = torch.device("cuda", self._device_index())
        has_seq = self.module != "signal_bilstm"
        has_sig = self.module != "seq_bilstm"
        ref = signals if has_sig else base_means
        n = int(ref.shape[0])

        def prep(t, allow_int):
            if t is None:
                return None, nat.DT_F32
            if not t.is_cuda or t.device != dev:
                raise RuntimeError("ModelBiLSTM.forward: input tensors must live on %s (no CPU path)" % dev)
            if t.dtype not in _DT or (t.dtype != torch.float32 and not allow_int):
                t = t.float()
            return t.contiguous(), _DT[t.dtype]
        kmer_t, kdt = prep(kmer if has_seq else None, True)
        means_t, _ = prep(base_means if has_seq else None, False)
        stds_t, _ = prep(base_stds if has_seq else None, False)
        lens_t, ldt = prep(base_signal_lens if has_seq else None, True)
        sig_t, _ = prep(signals if has_sig else None, False)
        if has_seq:
            for t in (kmer_t, means_t, stds_t, lens_t):
                if t.numel() != n * self.seq_len:
                    raise RuntimeError("seq feature tensor has %d elements, expected %d" % (t.numel(), n * self.seq_len))
        if has_sig and sig_t.numel() != n * self.seq_len * self.signal_len:
            raise RuntimeError("signals has %d elements, expected %d" % (sig_t.numel(), n * self.seq_len * self.signal_len))

        init = nat.InitState()
        keep = []
        if init_states is not None:
            init.mode = nat.INIT_EXPLICIT
            for k in ("h_seq", "c_seq", "h_sig", "c_sig", "h_comb", "c_comb"):
                t = init_states.get(k)
                if t is not None:
                    t = t.to(device=dev, dtype=torch.float32).contiguous()
                    keep.append(t)
                    setattr(init, k, t.data_ptr())
        elif self.init_state == "zeros":
            init.mode = nat.INIT_ZEROS
        else:
            init.mode = nat.INIT_PHILOX
            init.seed = self.seed
            init.site_offset = self.site_offset
            if site_keys is not None:
                if not site_keys.is_cuda or site_keys.device != dev or site_keys.dtype not in (torch.int64, torch.uint64):
                    raise RuntimeError("site_keys must be an int64 / uint64 tensor on %s" % dev)
                if site_keys.numel() != n:
                    raise RuntimeError("site_keys has %d elements, expected %d" % (site_keys.numel(), n))
                site_keys = site_keys.contiguous()
                keep.append(site_keys)
                init.site_keys = site_keys.data_ptr()

        logits = torch.empty((n, self.num_classes), dtype=torch.float32, device=dev)
        probs = torch.empty((n, self.num_classes), dtype=torch.float32, device=dev)
        labels = torch.empty((n,), dtype=torch.uint8, device=dev) if want_labels else None
        stream = torch.cuda.current_stream(dev).cuda_stream

        def p(t):
            return ctypes.c_void_p(t.data_ptr()) if t is not None else None
        nat.check(nat.lib().dsp_forward(self._handle, ctypes.c_void_p(stream), n, p(kmer_t), kdt, p(means_t), p(stds_t),
                                        p(lens_t), ldt, p(sig_t), ctypes.byref(init), p(logits), p(probs), p(labels)))
        # tensors in `keep`/inputs were used asynchronously on `stream`; torch's caching allocator is
        # stream-ordered on the same stream, so releasing them here is safe.
        if want_labels:
            return logits, probs, labels
        return logits, probs

    __call__ = forward

    # ---- bring-up / profiling hooks -------------------------------------------------------------------
    def debug_activation(self, which, n):
        import numpy as np
        feats = self.hidden_size if which == 0 else 2 * self.hidden_size
        out = np.zeros((n, self.seq_len, feats), np.float32)
        dev = torch.device("cuda", self._device_index())
        stream = torch.cuda.current_stream(dev).cuda_stream
        nat.check(nat.lib().dsp_debug_read_activation(self._handle, ctypes.c_void_p(stream), which, n,
                                                      out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def query(self, what):
        """what the handle decided about its device (include/dsp_amd.h DSP_QUERY_*): "clustering", "xcc_probe_failed",
        "compute_units" """
        self._ensure_handle()
        return nat.check(int(nat.lib().dsp_model_query(self._handle, {"clustering": 0, "xcc_probe_failed": 1, "compute_units": 2}[what])))

    def profile(self, on=True):
        """HIP events around every launch (True), around the dominant kernel's launches only ("dominant"), or off"""
        self._ensure_handle()
        nat.check(nat.lib().dsp_profile_enable(self._handle, 2 if on == "dominant" else int(bool(on))))

    def profile_read(self):
        """[(launch name, ms)] of every launch since profiling was enabled / last read (HIP events on the
        launch stream; synchronise first)."""
        cap = 1 << 16
        names = ctypes.create_string_buffer(cap * 12)
        ms = (ctypes.c_float * cap)()
        k = nat.check(nat.lib().dsp_profile_read(self._handle, names, cap * 12, ms, cap))
        raw = names.raw.split(b"\0")
        return [(raw[i].decode(), float(ms[i])) for i in range(k)]
