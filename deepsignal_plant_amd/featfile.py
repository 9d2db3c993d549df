"""Binary feature container (.dspf): the parsed form of the feature TSV, block-structured so that call_mods
copies blocks straight into pinned buffers (csrc/dsp_featfile.cpp).

The reference moves features between `extract` and `call_mods` as ~2.08 kB text rows
(extract_features.py:381-395 -> call_modifications.py:76-86); this container holds the same rows as 1.03 kB of
arrays + the sampleinfo string, with values bit-identical to what the TSV parser yields, so the per-read calls
do not depend on which of the two formats fed them."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _native as nat
from . import textio

MAGIC = b"DSPFEAT1"


def is_feature_file(path):
    try:
        with open(path, "rb") as f:
            return f.read(8) == MAGIC
    except (IOError, OSError):
        return False


class FeatureFileWriter(object):
    """Append parsed rows (textio.ParsedRows); rows are coalesced into blocks of `block_rows`."""

    def __init__(self, path, seq_len=13, signal_len=16, block_rows=32768):
        self._h = ctypes.c_void_p()
        nat.check(nat.lib().dsp_feat_writer_create(path.encode(), seq_len, signal_len, block_rows, ctypes.byref(self._h)))
        self.rows = 0

    def add(self, rows):
        if self._h is None:
            raise ValueError("FeatureFileWriter is closed")
        if rows.n == 0:
            return
        tp, _, _keep = textio._buf_ptr(rows.text)
        p = textio._ptr
        c = np.ascontiguousarray
        arrs = [c(rows.kmer, np.uint8), c(rows.means, np.float32), c(rows.stds, np.float32), c(rows.lens, np.int32),
                c(rows.signals, np.float32), c(rows.labels, np.int32)]
        offs = [c(rows.row_off, np.uint64), c(rows.info_len, np.uint32), c(rows.read_off, np.uint32),
                c(rows.read_len, np.uint32)]
        nat.check(nat.lib().dsp_feat_writer_add(self._h, rows.n, *[p(a) for a in arrs], tp, *[p(a) for a in offs]))
        self.rows += rows.n

    def close(self):
        if self._h is not None:
            h, self._h = self._h, None
            nat.check(nat.lib().dsp_feat_writer_close(h))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class FeatureFile(object):
    """Random access to the blocks of a .dspf file (thread-safe reads)."""

    def __init__(self, path):
        self._h = ctypes.c_void_p()
        nat.check(nat.lib().dsp_feat_open(path.encode(), ctypes.byref(self._h)))
        L, S = ctypes.c_int32(), ctypes.c_int32()
        n, nb = ctypes.c_int64(), ctypes.c_int64()
        nat.check(nat.lib().dsp_feat_info(self._h, ctypes.byref(L), ctypes.byref(S), ctypes.byref(n), ctypes.byref(nb)))
        self.seq_len, self.signal_len, self.n_rows, self.n_blocks = L.value, S.value, n.value, nb.value
        self.block_n = np.zeros(self.n_blocks, np.int64)
        self.block_first_row = np.zeros(self.n_blocks, np.int64)
        self.block_info_bytes = np.zeros(self.n_blocks, np.int64)
        a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        for i in range(self.n_blocks):
            nat.check(nat.lib().dsp_feat_block_info(self._h, i, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
            self.block_n[i], self.block_first_row[i], self.block_info_bytes[i] = a.value, b.value, c.value

    def max_block_rows(self):
        return int(self.block_n.max()) if self.n_blocks else 0

    def blocks_for_rank(self, world, rank):
        """Contiguous block range [b0, b1) of `rank`: blocks are dealt so that every rank's first global row is
        the smallest block boundary >= rank * n_rows / world (no collective needed: the index has the counts)."""
        if self.n_blocks == 0:
            return 0, 0

        def cut(r):
            if r <= 0:
                return 0
            if r >= world:
                return self.n_blocks
            target = (self.n_rows * r + world - 1) // world
            return int(np.searchsorted(self.block_first_row, target, side="left"))
        return cut(rank), cut(rank + 1)

    def read_block(self, b, out=None, info=None, nthreads=4):
        """-> textio.ParsedRows of block b.  `out`: dict of preallocated arrays (textio.alloc_rows), `info`: uint8
        array for the sampleinfo bytes; both are allocated when absent or too small."""
        n, ib = int(self.block_n[b]), int(self.block_info_bytes[b])
        if out is None or out["labels"].shape[0] < n:
            out = textio.alloc_rows(max(n, 1), self.seq_len, self.signal_len)
        if info is None or info.nbytes < ib:
            info = np.empty(max(ib, 1), np.uint8)
        p = textio._ptr
        k = nat.lib().dsp_feat_read_block(self._h, b, out["labels"].shape[0], p(out["kmer"]), p(out["means"]),
                                          p(out["stds"]), p(out["lens"]), p(out["signals"]), p(out["labels"]),
                                          p(info), info.nbytes, p(out["row_off"]), p(out["info_len"]),
                                          p(out["read_off"]), p(out["read_len"]), int(nthreads))
        k = nat.check(int(k))
        r = textio.ParsedRows()
        r.text, r.seq_len, r.signal_len, r.n = info, self.seq_len, self.signal_len, k
        for key in ("kmer", "means", "stds", "lens", "signals", "labels", "row_off", "info_len", "read_off", "read_len"):
            setattr(r, key, out[key][:k])
        return r, out, info

    def close(self):
        if self._h is not None:
            h, self._h = self._h, None
            nat.lib().dsp_feat_close(h)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_features(tsv_path, out_path, seq_len=13, signal_len=16, block_rows=32768, nthreads=8, chunk_bytes=64 << 20):
    """Feature TSV (plain, BGZF or any .gz) -> .dspf.  Returns the row count.  The rows come from the reader call_mods
    itself uses (feed.FeatureReader: the file mapped, complete-row blocks parsed on `nthreads` threads into rotating
    buffers, a foreign .gz through the parallel inflater) and are appended by this thread while the next block is parsed."""
    from . import feed
    reader = feed.FeatureReader(tsv_path, seq_len, signal_len, nthreads=nthreads, nbuf=3, block_bytes=chunk_bytes, pinned=False)
    with FeatureFileWriter(out_path, seq_len, signal_len, block_rows) as w:
        reader.start()
        for block in reader:
            w.add(block.rows)
            reader.release(block)
        return w.rows


def add_pack_features_args(p):
    p.add_argument("--input_path", "-i", type=str, required=True, help="feature file written by `extract` (plain or .gz)")
    p.add_argument("--result_file", "-o", type=str, required=True, help="binary feature file to write (.dspf)")
    p.add_argument("--seq_len", type=int, default=13)
    p.add_argument("--signal_len", type=int, default=16)
    p.add_argument("--block_rows", type=int, default=32768, help="rows per block (one block = one GPU batch)")
    p.add_argument("--nproc", "-p", type=int, default=10, help="parser threads")
    return p
