"""Block reader + pinned, multi-buffered feed of the call_mods path.

Replaces the reference's reader process and pickled-list queue (_read_features_file,
call_modifications.py:55-127, Queue at :40-44): a reader thread cuts the input into blocks of complete rows
(plain files: this rank's byte range via mmap; .gz: sequential inflate; .dspf binary containers: this rank's
block range, read with no parsing at all -- featfile.py), the native parser
(csrc/dsp_text.cpp, multi-threaded, GIL released) writes straight into PINNED SoA buffers, and the consumer
issues async H2D copies on a HIP stream.  NBUF buffer sets rotate, so parse(k+1), H2D/compute(k) and
format/write(k-1) overlap."""
from __future__ import annotations

import gzip
import mmap
import os
import queue
import threading

import numpy as np

from . import dist as dsp_dist
from . import featfile, textio

BLOCK_BYTES = int(os.environ.get("DSP_BLOCK_BYTES", 0) or (48 << 20))  # first block: ~23k rows of ~2.08 kB
# Later blocks are sized to TARGET_ROWS rows from the bytes/row seen so far: 32,768 sites are exactly four full
# rounds of the LSTM kernels' workgroups on 256 CUs (64 sites x 2 directions per workgroup pair); aiming 2 % low
# keeps a block from spilling into a fifth round.  DSP_BLOCK_BYTES pins the size instead (tests).
TARGET_ROWS = 0 if os.environ.get("DSP_BLOCK_BYTES") else int(32768 * 0.98)


class Block(object):
    __slots__ = ("rows", "first_row", "slot")


def count_rows_in_range(path, a, b):
    """Rows in bytes [a, b) of a plain file whose ends are row boundaries (native memchr scan over mmap'd
    chunks, GIL released): newlines + 1 if the range does not end with one."""
    if b <= a:
        return 0
    newlines = 0
    with open(path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
        pos = a
        while pos < b:
            end = min(b, pos + (256 << 20))
            view = np.frombuffer(mm, dtype=np.uint8, count=end - pos, offset=pos)
            k = textio.count_rows(view)  # = newlines, +1 when the chunk has an unterminated tail
            if view[-1] != 10:
                k -= 1
            del view
            newlines += k
            pos = end
        tail_open = mm[b - 1:b] != b"\n"
    return newlines + (1 if tail_open else 0)


def count_rows_bgzf(path, world, rank, nthreads=8, block_bytes=BLOCK_BYTES):
    """Rows owned by `rank` of a BGZF feature file (= newlines inside its member range, see FeatureReader._run_bgzf), or
    None when the file is not BGZF.  One inflate pass on `nthreads` threads; the ranks then all_gather these counts to
    learn the global index of their first row."""
    from . import gzio
    bz = gzio.BgzfFile(path)
    if not bz.ok:
        return None
    m0, m1 = bz.members_for_rank(world, rank)
    mine, m = 0, m0
    while m < m1:
        e = min(m1, max(m + 1, int(np.searchsorted(bz.text_off, bz.text_off[m] + block_bytes, side="left"))))
        buf, n = bz.inflate(m, e, nthreads=nthreads)
        mine += textio.count_newlines(buf[:n])
        m = e
    return mine


class FeatureReader(threading.Thread):
    """Producer thread: yields parsed blocks of this rank's rows, in file order, through a bounded queue."""

    def __init__(self, path, seq_len, signal_len, rank=0, world=1, nthreads=4, nbuf=3, block_bytes=BLOCK_BYTES,
                 first_row=0, byte_range=None, pinned=True, max_rows_per_block=None):
        super().__init__(daemon=True)
        self.path, self.L, self.S = path, seq_len, signal_len
        self.ff = None
        if featfile.is_feature_file(path):
            self.ff = featfile.FeatureFile(path)
            if (self.ff.seq_len, self.ff.signal_len) != (seq_len, signal_len):
                raise ValueError("%s holds seq_len=%d signal_len=%d features, the model expects %d / %d" % (
                    path, self.ff.seq_len, self.ff.signal_len, seq_len, signal_len))
            max_rows_per_block = max(1, self.ff.max_block_rows())
        self.rank, self.world, self.nthreads = rank, world, max(1, nthreads)
        self.block_bytes = block_bytes
        self.first_row = first_row
        self.byte_range = byte_range
        self.q = queue.Queue(maxsize=max(1, nbuf - 1))
        self.free = queue.Queue()
        cap = max_rows_per_block or max(1024, block_bytes // 600)
        self.cap = cap
        for s in range(nbuf):
            self.free.put(textio.alloc_rows(cap, seq_len, signal_len, pinned=pinned))
        self.error = None

    # -- consumer side
    def __iter__(self):
        while True:
            item = self.q.get()
            if item is None:
                if self.error is not None:
                    raise self.error
                return
            yield item

    def release(self, block):
        self.free.put(block.slot)

    # -- producer side
    def _emit(self, data, row0):
        slot = self.free.get()
        try:
            rows = textio.parse_rows(data, self.L, self.S, nthreads=self.nthreads, out=slot)
        except RuntimeError as e:  # rows much shorter than expected: this block gets its own (unpinned) buffers
            if "capacity" not in str(e):
                raise
            self.free.put(slot)
            slot = textio.alloc_rows(textio.count_rows(data), self.L, self.S, pinned=False)
            rows = textio.parse_rows(data, self.L, self.S, nthreads=self.nthreads, out=slot)
        b = Block()
        b.rows, b.first_row, b.slot = rows, row0, slot
        self.q.put(b)
        return rows.n

    def run(self):
        try:
            row = self.first_row
            if self.ff is not None:
                row = self._run_dspf()
            elif self.path.endswith(".gz"):
                row = self._run_gz(row)
            else:
                row = self._run_plain(row)
        except BaseException as e:  # surfaced in the consumer
            self.error = e
        finally:
            self.q.put(None)

    def _run_dspf(self):
        b0, b1 = self.ff.blocks_for_rank(self.world, self.rank)
        row = 0
        for bi in range(b0, b1):
            slot = self.free.get()
            rows, _out, info = self.ff.read_block(bi, out=slot, info=slot.get("_info"), nthreads=self.nthreads)
            slot["_info"] = info
            b = Block()
            b.rows, b.first_row, b.slot = rows, int(self.ff.block_first_row[bi]), slot
            self.q.put(b)
            row = b.first_row + rows.n
        return row

    def _run_plain(self, row):
        size = os.path.getsize(self.path)
        if size == 0:
            return row
        # the mapping is parsed in place (no copy out of the page cache; the parser threads take the page faults) and
        # stays referenced by the blocks' sampleinfo views, so it is left to the garbage collector
        with open(self.path, "rb") as f:
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        a, b = self.byte_range if self.byte_range is not None else dsp_dist.byte_range_for_rank(mm, size, self.world, self.rank)
        pos = a
        while pos < b:
            end = min(b, pos + self.block_bytes)
            if end < b:
                nl = mm.rfind(b"\n", pos, end)
                if nl < 0:
                    nl = mm.find(b"\n", end)
                    nl = b - 1 if nl < 0 else nl
                end = min(b, nl + 1)
            data = np.frombuffer(mm, dtype=np.uint8, count=end - pos, offset=pos)
            n = self._emit(data, row)
            row += n
            if TARGET_ROWS and n > 256:
                self.block_bytes = int(max(1 << 20, (end - pos) / n * min(TARGET_ROWS, 0.95 * self.cap)))
            pos = end
        return row

    def _run_gz(self, row):
        from . import gzio
        bz = gzio.BgzfFile(self.path)
        if bz.ok:
            return self._run_bgzf(bz, row)
        return self._run_gz_stream(row)

    def _run_bgzf(self, bz, row):
        """BGZF (what this build writes with --gzip, and what bgzip writes): this rank's contiguous member range, inflated
        in batches on all parser threads.  A rank owns the rows that END inside its members; the start of its first row
        is the tail of the previous rank's last members."""
        m0, m1 = bz.members_for_rank(self.world, self.rank)
        carry = np.zeros(0, np.uint8)
        if m0 > 0:  # bytes after the last newline before member m0
            k = m0
            while k > 0:
                k -= 1
                tail, n = bz.inflate(k, k + 1, nthreads=1)
                nl = np.flatnonzero(tail[:n] == 10)
                if len(nl):
                    carry = np.concatenate((tail[nl[-1] + 1:n], carry))
                    break
                carry = np.concatenate((tail[:n], carry))
        m = m0
        last_rank = self.rank == self.world - 1
        while m < m1:
            e = self._batch_end(bz, m, m1)
            need = int(bz.text_off[e] - bz.text_off[m])
            buf = np.empty(len(carry) + need, np.uint8)
            buf[:len(carry)] = carry
            bz.inflate(m, e, out=buf, out_offset=len(carry), nthreads=self.nthreads)
            m = e
            nl = -1
            if len(buf):
                tailpos = np.flatnonzero(buf[max(0, len(buf) - (1 << 16)):] == 10)
                if len(tailpos):
                    nl = max(0, len(buf) - (1 << 16)) + int(tailpos[-1])
                else:
                    allpos = np.flatnonzero(buf == 10)
                    nl = int(allpos[-1]) if len(allpos) else -1
            if nl < 0:
                carry = buf
                continue
            carry = buf[nl + 1:].copy()
            n = self._emit(buf[:nl + 1], row)
            row += n
        if last_rank and len(carry) and carry.tobytes().strip():
            row += self._emit(carry, row)  # an unterminated last row
        return row

    def _batch_end(self, bz, m, m1):
        e = int(np.searchsorted(bz.text_off, bz.text_off[m] + self.block_bytes, side="left"))
        return min(m1, max(m + 1, e))

    def _run_gz_stream(self, row):
        """A foreign .gz (one deflate stream): it cannot be range-split, so every rank inflates it -- natively, the GIL
        released -- and keeps the blocks it owns (block i -> rank i % world), counting the rows of foreign blocks to
        keep global row indices."""
        from . import gzio
        st = gzio.GzStream(self.path)
        carry = np.zeros(0, np.uint8)
        i = 0
        try:
            while True:
                buf = np.empty(len(carry) + self.block_bytes, np.uint8)
                buf[:len(carry)] = carry
                got = st.readinto(buf, len(carry))
                if got == 0:
                    break
                data = buf[:len(carry) + got]
                pos = np.flatnonzero(data[max(0, len(data) - (1 << 16)):] == 10)
                if len(pos):
                    nl = max(0, len(data) - (1 << 16)) + int(pos[-1])
                else:
                    allpos = np.flatnonzero(data == 10)
                    nl = int(allpos[-1]) if len(allpos) else -1
                if nl < 0:
                    carry = data.copy()
                    continue
                carry = data[nl + 1:].copy()
                data = data[:nl + 1]
                if i % self.world == self.rank:
                    row += self._emit(data, row)
                else:
                    row += textio.count_rows(data)
                i += 1
        finally:
            st.close()
        if len(carry) and carry.tobytes().strip():
            if i % self.world == self.rank:
                row += self._emit(carry, row)
            else:
                row += 1
        return row
