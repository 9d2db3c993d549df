"""The feature-row parser on the GPU: host half (block staging) and device half (csrc/dsp_parse_dev.hip) of one block.

Replaces the per-row work of _read_features_file (deepsignal_plant/call_modifications.py:76-86) for plain rows: the host
copies a block of rows into a page-locked buffer and notes where the rows start -- one pass, dsp_copy_rows_index -- the raw
text crosses PCIe (2.08 kB per row), one GPU thread per row parses it into the arrays the forward takes.  Rows the plain
grammar does not cover are flagged; a block with flagged rows goes through the host parser (textio.parse_rows), which
gives the same values for the rows that are fine and owns the error messages for the ones that are not."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _native as nat
from . import textio

import os
PAD = 64          # readable bytes behind the staged text (the kernel's 16-byte cursor runs two words ahead)


def alloc_stage(cap_rows, cap_bytes, seq_len, pinned=True):
    """host staging of one block: text + row starts (in), the small per-row arrays the writer needs (out)"""
    import torch
    mk = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=pinned)
    t = {"text": mk((cap_bytes + PAD,), torch.uint8), "row_off": mk((cap_rows + 1,), torch.int64),
         "kmer": mk((cap_rows, seq_len), torch.uint8), "labels": mk((cap_rows,), torch.int32),
         "info_len": mk((cap_rows,), torch.int32), "read_off": mk((cap_rows,), torch.int32),
         "read_len": mk((cap_rows,), torch.int32), "n_flagged": mk((1,), torch.int32)}
    out = {"_torch": t, "_stage": True, "cap_rows": cap_rows, "cap_bytes": cap_bytes}
    out["text"] = t["text"].numpy()
    out["row_off"] = t["row_off"].numpy().view(np.uint64)
    for k in ("info_len", "read_off", "read_len"):
        out[k] = t[k].numpy().view(np.uint32)
    out["kmer"], out["labels"] = t["kmer"].numpy(), t["labels"].numpy()
    return out


def stage_rows(data, stage, seq_len, signal_len):
    """one pass: data (uint8 array of complete rows) -> stage["text"], row starts -> stage["row_off"].
    Returns a ParsedRows whose small arrays are views of the stage (filled by DeviceRowParser.submit's copies back)."""
    n_bytes = len(data)
    src, _, keep = textio._buf_ptr(data)
    n = nat.check(int(nat.lib().dsp_copy_rows_index(src, n_bytes, textio._ptr(stage["text"]), textio._ptr(stage["row_off"]),
                                                  stage["cap_rows"])))
    r = textio.ParsedRows()
    r.text, r.n, r.seq_len, r.signal_len = stage["text"], n, seq_len, signal_len
    r.row_off = stage["row_off"][:n]
    for k in ("info_len", "read_off", "read_len", "kmer", "labels"):
        setattr(r, k, stage[k][:n])
    r.means = r.stds = r.lens = r.signals = None    # these exist on the device only
    return r, int(stage["row_off"][n])               # (bytes staged, the '\n' given to an unterminated last row included)


def read_rows(fd, file_off, range_bytes, want_rows, at_eof, stage, seq_len, signal_len, budget_bytes=None):
    """a plain file's next block straight into the staging buffer (pread + row starts, dsp_read_rows_index): rows until
    want_rows are in or budget_bytes (default: the buffer) / range_bytes (what is left of the rank's range) are used up.
    -> (ParsedRows, bytes staged, bytes of the file consumed); rows.n == 0 and consumed == 0: the budget is too small for a row"""
    consumed = ctypes.c_uint64(0)
    want = min(int(want_rows), stage["cap_rows"])
    budget = stage["cap_bytes"] if budget_bytes is None else min(int(budget_bytes), stage["cap_bytes"])
    n = nat.check(int(nat.lib().dsp_read_rows_index(int(fd), int(file_off), int(range_bytes), budget, want, int(bool(at_eof)),
                                                  textio._ptr(stage["text"]), textio._ptr(stage["row_off"]), ctypes.byref(consumed))))
    r = textio.ParsedRows()
    r.text, r.n, r.seq_len, r.signal_len = stage["text"], n, seq_len, signal_len
    r.row_off = stage["row_off"][:n]
    for k in ("info_len", "read_off", "read_len", "kmer", "labels"):
        setattr(r, k, stage[k][:n])
    r.means = r.stds = r.lens = r.signals = None
    return r, int(stage["row_off"][n]), int(consumed.value)


class DeviceRowParser(object):
    """device buffers of the reader's slots + the launches of one block"""

    def __init__(self, dev, seq_len, signal_len):
        import torch
        self.torch, self.dev, self.L, self.S = torch, dev, seq_len, signal_len
        self.sync = bool(os.environ.get("DSP_PARSE_SYNC"))

    def _bufs(self, stage, stream):
        """the device buffers of a staging slot.  Allocated UNDER `stream` (the stream the parse runs on): torch's caching
        allocator hands a block freed on one stream to the next allocation on that stream while kernels queued on it may still
        be writing it -- the forward's logits, dropped by Python right after the launch -- which is only safe for work on the
        SAME stream.  (Allocated under the compute stream and used on the copy stream, a new slot's row offsets and segment
        tables were overwritten by the head kernel of a forward still in the queue: garbage offsets, a GPU fault.)"""
        torch = self.torch
        # The device buffers live IN the slot (stage["_dev"]): when the reader replaces a slot by a larger one
        # (feed._emit_staged, _run_plain_staged) the old slot's device arrays go with it.  (Until round 4 they sat in a
        # dict keyed by id() of the slot's pinned tensor: a grown slot leaked ~100 MB of device arrays, and a recycled id()
        # could alias another slot's entry -- ADVICE r4.)  A slot is only replaced after the writer released it, i.e. after
        # the forward that read these arrays has completed.
        b = stage.get("_dev")
        if b is None or b["cap_rows"] < stage["cap_rows"] or b["cap_bytes"] < stage["cap_bytes"]:
            cr, cb, L, S = stage["cap_rows"], stage["cap_bytes"], self.L, self.S
            stage["_dev"] = None
            with torch.cuda.stream(stream):
                stage["_dev"] = b = self._alloc(cr, cb, L, S)
        return b

    def _alloc(self, cr, cb, L, S):
        torch = self.torch
        mk = lambda shape, dt: torch.empty(shape, dtype=dt, device=self.dev)
        b = dict(cap_rows=cr, cap_bytes=cb, text=mk((cb + PAD,), torch.uint8), row_off=mk((cr + 1,), torch.int64),
                 kmer=mk((cr, L), torch.uint8), means=mk((cr, L), torch.float32), stds=mk((cr, L), torch.float32),
                 lens=mk((cr, L), torch.int32), signals=mk((cr, L, S), torch.float32), labels=mk((cr,), torch.int32),
                 info_len=mk((cr,), torch.int32), read_off=mk((cr,), torch.int32), read_len=mk((cr,), torch.int32),
                 status=mk((cr,), torch.uint8), n_flagged=mk((1,), torch.int32), seg=mk((cr, L + 4), torch.int32))
        return b

    def submit(self, rows, n_bytes, stage, stream):
        """text up, parse, the writer's small arrays back: all asynchronous on `stream`.  Returns (device arrays, event)."""
        torch = self.torch
        b = self._bufs(stage, stream)
        t, n = stage["_torch"], rows.n
        with torch.cuda.stream(stream):
            b["text"][:n_bytes + PAD].copy_(t["text"][:n_bytes + PAD], non_blocking=True)
            b["row_off"][:n + 1].copy_(t["row_off"][:n + 1], non_blocking=True)
            p = lambda x: ctypes.c_void_p(x.data_ptr())
            nat.check(int(nat.lib().dsp_parse_rows_device(
                ctypes.c_void_p(stream.cuda_stream), p(b["text"]), p(b["row_off"]), n, self.L, self.S, p(b["kmer"]), p(b["means"]),
                p(b["stds"]), p(b["lens"]), p(b["signals"]), p(b["labels"]), p(b["info_len"]), p(b["read_off"]), p(b["read_len"]),
                p(b["status"]), p(b["n_flagged"]), p(b["seg"]), int(n_bytes))))
            for k in ("kmer", "labels", "info_len", "read_off", "read_len"):
                t[k][:n].copy_(b[k][:n], non_blocking=True)
            t["n_flagged"].copy_(b["n_flagged"], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(stream)
        if self.sync:   # DSP_PARSE_SYNC=1 (debugging aid): attribute an asynchronous GPU fault to this block's parse
            import sys
            sys.stderr.write("[parse_dev] block of %d rows, %d bytes ..." % (n, n_bytes))
            stream.synchronize()
            sys.stderr.write(" ok, %d flag events\n" % int(t["n_flagged"][0]))
        return b, ev

    def host_fallback(self, rows, n_bytes, stage, b, nthreads, stream):
        """a block with rows outside the plain grammar: the host parser decides (and raises what the reference would for a
        malformed row); its arrays replace the device parser's -- identical for every row that one had accepted"""
        torch = self.torch
        full = textio.parse_rows(stage["text"][:n_bytes], self.L, self.S, nthreads=nthreads)
        assert full.n == rows.n
        with torch.cuda.stream(stream):
            for k in ("kmer", "means", "stds", "lens", "signals", "labels"):
                b[k][:full.n].copy_(torch.from_numpy(np.ascontiguousarray(getattr(full, k))), non_blocking=False)
        for k in ("info_len", "read_off", "read_len", "kmer", "labels"):
            stage[k][:full.n] = getattr(full, k)
        stage["row_off"][:full.n] = full.row_off    # (a row with leading blanks starts behind them)
        return full
