"""ORACLE (test infrastructure, NOT product code): ctypes wrapper around oracle/dsp_oracle.c.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

from . import forward_np as onp

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libdsp_oracle.so")
_MODULE_CODE = {"both_bilstm": 0, "seq_bilstm": 1, "signal_bilstm": 2}


class _Cfg(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in (
        "seq_len", "signal_len", "num_layers1", "num_layers2", "num_classes", "hidden_size", "vocab_size",
        "embedding_size", "is_base", "is_signallen", "module")]


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "dsp_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
        _lib.orc_forward.restype = ctypes.c_int
        _lib.orc_forward_keys.restype = ctypes.c_int
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def forward(cfg: onp.OracleConfig, w, kmer, means, stds, lens, signals, states=None, init_mode="explicit",
            seed=0, site_offset=0, nthreads=0, site_keys=None):
    """fp32 C oracle forward. init_mode: 'zeros' | 'explicit' (states dict) | 'philox' (seed, and either site_offset:
    the Philox counter of site i is site_offset + i, or site_keys: a uint64 per site)."""
    L = lib()
    c = _Cfg(cfg.seq_len, cfg.signal_len, cfg.num_layers1, cfg.num_layers2, cfg.num_classes, cfg.hidden_size,
             cfg.vocab_size, cfg.embedding_size, int(cfg.is_base), int(cfg.is_signallen), _MODULE_CODE[cfg.module])
    spec = onp.state_dict_spec(cfg)
    arrs = [np.ascontiguousarray(w[k], np.float32) for k, _ in spec]
    for a, (k, shp) in zip(arrs, spec):
        assert tuple(a.shape) == tuple(shp), (k, a.shape, shp)
    wptr = (ctypes.POINTER(ctypes.c_float) * len(arrs))(*[_fp(a) for a in arrs])
    n = int(kmer.shape[0])
    ins = [np.ascontiguousarray(a, np.float32) for a in (kmer, means, stds, lens, signals)]
    mode = {"zeros": 0, "explicit": 1, "philox": 2}[init_mode]
    sptr = None
    keep = []
    if mode == 1:
        ptrs = []
        for k in ("h_seq", "c_seq", "h_sig", "c_sig", "h_comb", "c_comb"):
            if states is not None and k in states:
                a = np.ascontiguousarray(states[k], np.float32)
                keep.append(a)
                ptrs.append(_fp(a))
            else:
                ptrs.append(ctypes.POINTER(ctypes.c_float)())
        sptr = (ctypes.POINTER(ctypes.c_float) * 6)(*ptrs)
    logits = np.empty((n, cfg.num_classes), np.float32)
    probs = np.empty((n, cfg.num_classes), np.float32)
    kptr = None
    if site_keys is not None:
        site_keys = np.ascontiguousarray(site_keys, np.uint64)
        assert site_keys.shape == (n,)
        kptr = site_keys.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
    rc = L.orc_forward_keys(ctypes.byref(c), wptr, len(arrs), ctypes.c_int64(n), *[_fp(a) for a in ins], mode, sptr,
                            ctypes.c_uint64(seed), ctypes.c_uint64(site_offset), kptr, _fp(logits), _fp(probs),
                            int(nthreads))
    if rc != 0:
        raise RuntimeError("orc_forward failed: %d" % rc)
    return logits, probs


def philox_normal(seed, site, stream, ngroups):
    out = np.empty(4 * ngroups, np.float32)
    lib().orc_philox_normal(ctypes.c_uint64(seed), ctypes.c_uint64(site), ctypes.c_uint32(stream),
                            ctypes.c_uint32(ngroups), _fp(out))
    return out


def philox_states(cfg: onp.OracleConfig, n, seed, site_offset=0, site_keys=None):
    """the initial states init_mode='philox' draws inside the forward, as the dict init_mode='explicit' takes
    (forward_np.init_state_shapes): forward(..., 'explicit', philox_states(...)) == forward(..., 'philox') bit for bit"""
    kptr = None
    if site_keys is not None:
        site_keys = np.ascontiguousarray(site_keys, np.uint64)
        kptr = site_keys.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
    out = {}
    for lstm, (name, layers, hid) in enumerate((("seq", cfg.num_layers2, cfg.nhid_seq), ("sig", cfg.num_layers2, cfg.nhid_signal),
                                                ("comb", cfg.num_layers1, cfg.hidden_size))):
        if (name == "seq" and cfg.module == "signal_bilstm") or (name == "sig" and cfg.module == "seq_bilstm"):
            continue
        h = np.empty((2 * layers, n, hid), np.float32)
        c = np.empty((2 * layers, n, hid), np.float32)
        lib().orc_philox_states(lstm, layers, hid, ctypes.c_int64(n), ctypes.c_uint64(seed), ctypes.c_uint64(site_offset), kptr,
                                _fp(h), _fp(c))
        out["h_" + name], out["c_" + name] = h, c
    return out


def num_threads():
    """Host threads worth using: min(OpenMP default, CPU affinity, cgroup CPU quota) -- a container with a
    16-CPU quota on a 256-thread host must not spawn 256 OpenMP threads."""
    n = lib().orc_num_threads()
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)
