"""Feature-TSV parsing and per-read-call formatting through the native text functions of libdsp_amd.so
(csrc/dsp_text.cpp).  Replaces the per-row Python of _read_features_file (call_modifications.py:76-86,
:111-117) and of _call_mods (:175-188)."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _native as nat


class ParsedRows(object):
    """SoA view of a block of feature rows.  `text` keeps the raw bytes alive: sampleinfo (the first six
    columns, kept verbatim like call_modifications.py:80) is addressed by (row_off, info_len)."""
    __slots__ = ("text", "n", "kmer", "means", "stds", "lens", "signals", "labels", "row_off", "info_len",
                 "read_off", "read_len", "seq_len", "signal_len")

    def sampleinfo(self, i):
        o = int(self.row_off[i])
        return bytes(self.text[o:o + int(self.info_len[i])]).decode()

    def readname(self, i):
        o = int(self.row_off[i]) + int(self.read_off[i])
        return bytes(self.text[o:o + int(self.read_len[i])]).decode()


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def _buf_ptr(text):
    if isinstance(text, np.ndarray):
        return ctypes.c_void_p(text.ctypes.data), text.nbytes, text
    mv = memoryview(text)
    arr = np.frombuffer(mv, dtype=np.uint8)
    return ctypes.c_void_p(arr.ctypes.data), arr.nbytes, arr


def count_rows(text):
    p, n, _keep = _buf_ptr(text)
    return int(nat.lib().dsp_count_rows(p, n))


def find_row_end(arr, n_rows):
    """bytes of the first n_rows rows of a uint8 array (len(arr) when it holds fewer)"""
    if len(arr) == 0 or n_rows <= 0:
        return 0
    return int(nat.lib().dsp_find_row_end(_ptr(arr), len(arr), int(n_rows)))


def count_newlines(arr):
    """newline bytes in a uint8 array (dsp_count_rows counts an unterminated tail as a row)"""
    n = len(arr)
    if n == 0:
        return 0
    return count_rows(arr) - (1 if arr[n - 1] != 10 else 0)


def parse_rows(text, seq_len=13, signal_len=16, nthreads=4, out=None):
    """text: bytes-like holding complete lines.  out: optional dict of preallocated (e.g. pinned) numpy arrays
    with capacity >= the row count (keys as ParsedRows slots).  Raises ValueError on malformed rows (the
    reference raises KeyError for an unknown base and ValueError for a malformed number)."""
    L = nat.lib()
    p, nbytes, keep = _buf_ptr(text)
    cap = int(L.dsp_count_rows(p, nbytes)) if out is None else int(out["labels"].shape[0])
    if out is None:
        out = alloc_rows(cap, seq_len, signal_len)
    r = ParsedRows()
    r.text, r.seq_len, r.signal_len = keep, seq_len, signal_len
    n = L.dsp_parse_feature_rows(p, nbytes, seq_len, signal_len, cap, _ptr(out["kmer"]), _ptr(out["means"]),
                                 _ptr(out["stds"]), _ptr(out["lens"]), _ptr(out["signals"]), _ptr(out["labels"]),
                                 _ptr(out["row_off"]), _ptr(out["info_len"]), _ptr(out["read_off"]),
                                 _ptr(out["read_len"]), int(nthreads))
    n = nat.check(int(n))
    r.n = n
    for k in ("kmer", "means", "stds", "lens", "signals", "labels", "row_off", "info_len", "read_off", "read_len"):
        setattr(r, k, out[k][:n])
    return r


def alloc_rows(cap, seq_len=13, signal_len=16, pinned=False):
    """Host SoA buffers for `cap` rows; pinned=True allocates page-locked memory through torch (plumbing)."""
    shapes = dict(kmer=((cap, seq_len), np.uint8), means=((cap, seq_len), np.float32),
                  stds=((cap, seq_len), np.float32), lens=((cap, seq_len), np.int32),
                  signals=((cap, seq_len, signal_len), np.float32), labels=((cap,), np.int32),
                  row_off=((cap,), np.uint64), info_len=((cap,), np.uint32), read_off=((cap,), np.uint32),
                  read_len=((cap,), np.uint32))
    out = {}
    if pinned:
        import torch
        tmap = {np.uint8: torch.uint8, np.float32: torch.float32, np.int32: torch.int32}
        out["_torch"] = {}
    for k, (shp, dt) in shapes.items():
        if pinned and dt in tmap:
            t = torch.empty(shp, dtype=tmap[dt], pin_memory=True)
            out["_torch"][k] = t
            out[k] = t.numpy()
        else:
            out[k] = np.empty(shp, dt)
    return out


def format_calls(rows, probs, labels, nthreads=4, start=0, stop=None):
    """bytes of the per-read-call lines for rows[start:stop] (call_modifications.py:175-188 + :279-280)."""
    stop = rows.n if stop is None else stop
    n = stop - start
    if n <= 0:
        return b""
    probs = np.ascontiguousarray(probs, np.float32)
    labels = np.ascontiguousarray(labels, np.uint8)
    assert probs.shape[0] == n and labels.shape[0] == n
    cap = int(rows.info_len[start:stop].sum()) + n * (rows.seq_len + 64)
    out = np.empty(cap, np.uint8)
    tp, _, _keep = _buf_ptr(rows.text)
    k = nat.lib().dsp_format_calls(tp, _ptr(rows.row_off[start:stop]), _ptr(rows.info_len[start:stop]), _ptr(probs),
                                   probs.shape[1], _ptr(labels), _ptr(rows.kmer[start:stop]), rows.seq_len, n,
                                   _ptr(out), cap, int(nthreads))
    k = nat.check(int(k))
    return out[:k].tobytes()


def format_feature_rows(rows, means, stds, signals, nthreads=4, as_view=False):
    """bytes of the feature-TSV rows (extract_features.py:381-395) for `rows` (sampleinfo, kmer, lens, labels from
    the ParsedRows; means / stds / signals as the float64 values the row prints).  as_view: a memoryview of the
    formatter's own buffer instead of a bytes copy (gigabytes per batch in `extract`)."""
    n = rows.n
    if n <= 0:
        return b""
    L, S = rows.seq_len, rows.signal_len
    means = np.ascontiguousarray(means, np.float64)
    stds = np.ascontiguousarray(stds, np.float64)
    signals = np.ascontiguousarray(signals, np.float64)
    kmer = np.ascontiguousarray(rows.kmer, np.uint8)
    lens = np.ascontiguousarray(rows.lens, np.int32)
    labels = np.ascontiguousarray(rows.labels, np.int32)
    assert means.shape == (n, L) and stds.shape == (n, L) and signals.shape == (n, L, S)
    cap = int(rows.info_len[:n].sum()) + n * (L + 16 + L * 2 * 26 + L * 12 + L * S * 26)
    out = np.empty(cap, np.uint8)
    tp, _, _keep = _buf_ptr(rows.text)
    k = nat.lib().dsp_format_feature_rows(tp, _ptr(rows.row_off[:n]), _ptr(rows.info_len[:n]), _ptr(kmer), _ptr(means),
                                          _ptr(stds), _ptr(lens), _ptr(signals), _ptr(labels), L, S, n, _ptr(out), cap,
                                          int(nthreads))
    k = nat.check(int(k))
    return memoryview(out)[:k] if as_view else out[:k].tobytes()


def feature_rows_capacity(rows):
    """bytes format_feature_rows_parts needs for `rows` (worst case: every number at its longest)"""
    n = rows.n
    return int(rows.info_len[:n].sum()) + n * int(nat.lib().dsp_feature_row_bound(rows.seq_len, rows.signal_len))


def format_feature_rows_parts(rows, means, stds, signals, nthreads=4, out=None):
    """The same text as format_feature_rows, as a list of memoryviews into `out` (a uint8 array of at least
    feature_rows_capacity(rows) bytes; allocated when None): every formatting thread leaves its rows where it wrote them,
    so nothing is copied and a caller that reuses `out` touches no fresh memory.  -> (parts, out)"""
    n = rows.n
    if n <= 0:
        return [], out
    L, S = rows.seq_len, rows.signal_len
    means = np.ascontiguousarray(means, np.float64)
    stds = np.ascontiguousarray(stds, np.float64)
    signals = np.ascontiguousarray(signals, np.float64)
    kmer = np.ascontiguousarray(rows.kmer, np.uint8)
    lens = np.ascontiguousarray(rows.lens, np.int32)
    labels = np.ascontiguousarray(rows.labels, np.int32)
    assert means.shape == (n, L) and stds.shape == (n, L) and signals.shape == (n, L, S)
    cap = feature_rows_capacity(rows)
    if out is None or out.nbytes < cap:
        out = np.empty(cap, np.uint8)
    nthreads = max(1, int(nthreads))
    off = np.zeros(nthreads, np.uint64)
    ln = np.zeros(nthreads, np.uint64)
    tp, _, _keep = _buf_ptr(rows.text)
    k = nat.lib().dsp_format_feature_rows_parts(tp, _ptr(rows.row_off[:n]), _ptr(rows.info_len[:n]), _ptr(kmer), _ptr(means),
                                                _ptr(stds), _ptr(lens), _ptr(signals), _ptr(labels), L, S, n, _ptr(out),
                                                out.nbytes, nthreads, _ptr(off), _ptr(ln))
    k = nat.check(int(k))
    mv = memoryview(out)
    return [mv[int(off[t]):int(off[t]) + int(ln[t])] for t in range(k)], out

