"""Block reader + pinned, multi-buffered feed of the call_mods path.

Replaces the reference's reader process and pickled-list queue (_read_features_file,
call_modifications.py:55-127, Queue at :40-44): a reader thread cuts the input into blocks of complete rows
(plain files: this rank's byte range via mmap; .gz: sequential inflate; .dspf binary containers: this rank's
block range, read with no parsing at all -- featfile.py), the native parser
(csrc/dsp_text.cpp, multi-threaded, GIL released) writes straight into PINNED SoA buffers, and the consumer
issues async H2D copies on a HIP stream.  NBUF buffer sets rotate, so parse(k+1), H2D/compute(k) and
format/write(k-1) overlap."""
from __future__ import annotations

import mmap
import os
import queue
import threading

import numpy as np

from . import dist as dsp_dist
from . import canary, featfile, textio

BLOCK_BYTES = int(os.environ.get("DSP_BLOCK_BYTES", 0) or (48 << 20))  # first block: ~23k rows of ~2.08 kB
# blocks of a foreign .gz are inflated into fixed buffers (private, or slots of the node's shared-memory ring): room for
# EXACT_ROWS rows of the default k-mer / signal window (2.1 kB each) and some
GZ_BLOCK_BYTES = int(os.environ.get("DSP_BLOCK_BYTES", 0) or (76 << 20))
# Later blocks are sized to TARGET_ROWS rows from the bytes/row seen so far: 32,768 sites are exactly four full
# rounds of the LSTM kernels' workgroups on 256 CUs (64 sites x 2 directions per workgroup pair); aiming 2 % low
# keeps a block from spilling into a fifth round.  DSP_BLOCK_BYTES pins the size instead (tests).
TARGET_ROWS = 0 if os.environ.get("DSP_BLOCK_BYTES") else int(32768 * 0.98)
# Plain text is cut at EXACTLY this many rows (the byte offset behind the n-th newline, textio.find_row_end): no 2 % margin,
# every forward but a rank's last runs whole rounds (round 3: 97.7 % -> 99 % of the forward's own rate while streaming)
EXACT_ROWS = 0 if os.environ.get("DSP_BLOCK_BYTES") else 32768


def refresh_env():
    """re-read DSP_BLOCK_BYTES (a second call_mods in one process -- the tests' job runner -- may have changed it)"""
    global BLOCK_BYTES, GZ_BLOCK_BYTES, TARGET_ROWS, EXACT_ROWS
    BLOCK_BYTES = int(os.environ.get("DSP_BLOCK_BYTES", 0) or (48 << 20))
    GZ_BLOCK_BYTES = int(os.environ.get("DSP_BLOCK_BYTES", 0) or (76 << 20))
    TARGET_ROWS = 0 if os.environ.get("DSP_BLOCK_BYTES") else int(32768 * 0.98)
    EXACT_ROWS = 0 if os.environ.get("DSP_BLOCK_BYTES") else 32768


class Block(object):
    # n_bytes: device-parsed blocks only -- the bytes staged in slot["text"] (rows.means / stds / lens / signals are None: they
    # will exist on the GPU only, parse_dev.DeviceRowParser)
    __slots__ = ("rows", "first_row", "slot", "n_bytes")


def count_rows_in_range(path, a, b, nthreads=1):
    """Rows in bytes [a, b) of a plain file whose ends are row boundaries: newlines + 1 if the range does not end with
    one.  Native memchr scans over the mmap (GIL released) on `nthreads` threads, each with its own piece of the range:
    this pass runs before a rank's first forward (the ranks need each other's counts for the global row indices), so at
    config 5's scale -- 260 GB of text per rank -- one thread at 5 GB/s would hold the GPUs back for a minute."""
    if b <= a:
        return 0
    nthreads = max(1, min(int(nthreads), (b - a) // (64 << 20) + 1))
    with open(path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
        def piece(t):
            lo = a + (b - a) * t // nthreads
            hi = a + (b - a) * (t + 1) // nthreads
            k, pos = 0, lo
            while pos < hi:
                end = min(hi, pos + (64 << 20))
                view = np.frombuffer(mm, dtype=np.uint8, count=end - pos, offset=pos)
                k += textio.count_newlines(view)
                del view
                pos = end
            return k
        if nthreads == 1:
            newlines = piece(0)
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(nthreads) as ex:
                newlines = sum(ex.map(piece, range(nthreads)))
        tail_open = mm[b - 1:b] != b"\n"
    return newlines + (1 if tail_open else 0)


def count_rows_bgzf(path, world, rank, nthreads=8, block_bytes=None):
    """Rows owned by `rank` of a BGZF feature file (= newlines inside its member range, see FeatureReader._run_bgzf), or
    None when the file is not BGZF.  One inflate pass on `nthreads` threads; the ranks then all_gather these counts to
    learn the global index of their first row."""
    from . import gzio
    bz = gzio.BgzfFile(path)
    if not bz.ok:
        return None
    m0, m1 = bz.members_for_rank(world, rank)
    block_bytes = BLOCK_BYTES if block_bytes is None else block_bytes
    if m1 > m0 and not os.environ.get("DSP_BGZF_COUNT_BY_INFLATE") and bool((bz.rows[m0:m1] >= 0).all()):
        # written by this build: every member's header carries its newline count (csrc/dsp_gz.cpp) -- no inflate pass
        # before the first forward (at config 5's scale it would be a minute of every rank's parser threads); the caller
        # checks the sum against the rows it then parses
        return int(bz.rows[m0:m1].sum())
    mine, m = 0, m0
    while m < m1:
        e = min(m1, max(m + 1, int(np.searchsorted(bz.text_off, bz.text_off[m] + block_bytes, side="left"))))
        buf, n = bz.inflate(m, e, nthreads=nthreads)
        mine += textio.count_newlines(buf[:n])
        m = e
    return mine


class FeatureReader(threading.Thread):
    """Producer thread: yields parsed blocks of this rank's rows, in file order, through a bounded queue."""

    def __init__(self, path, seq_len, signal_len, rank=0, world=1, nthreads=4, nbuf=3, block_bytes=None,
                 first_row=0, byte_range=None, pinned=True, max_rows_per_block=None, gz_ring=None, device_parse=False):
        """device_parse: blocks of text rows are NOT parsed here -- they are copied into page-locked staging with their row
        starts (one pass, parse_dev.stage_rows) and parsed on the GPU by the consumer (round 4); .dspf containers hold
        parsed rows already and come out as before."""
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
        block_bytes = BLOCK_BYTES if block_bytes is None else block_bytes
        self.block_bytes = block_bytes
        self.gz_block_bytes = GZ_BLOCK_BYTES if block_bytes == BLOCK_BYTES else block_bytes   # an explicit size is kept
        self.first_row = first_row
        self.byte_range = byte_range
        self.gz_ring = gz_ring       # foreign .gz read by several ranks: the node's shared-memory ring (open_gz_ring)
        self.gz_bytes_in = 0         # compressed bytes this rank inflated from a foreign .gz
        self.gz_parallel = False     # ... through the parallel inflater (csrc/dsp_pgz.cpp) rather than zlib
        self.q = queue.Queue(maxsize=max(1, nbuf - 1))
        self.free = queue.Queue()
        cap = max_rows_per_block or max(1024, block_bytes // 600)
        self.cap = cap
        self.device_parse = bool(device_parse) and self.ff is None
        self.pinned = pinned
        self.stage_seconds = 0.0     # device_parse: time this reader spent copying blocks into staging (tools/bench_feed.py)
        self.canary = canary.on()    # DSP_SLOT_CANARY=1: slots are poisoned on release and verified on take (canary.py)
        for s in range(nbuf):
            if self.device_parse:
                from . import parse_dev
                slot = parse_dev.alloc_stage(cap, int(max(block_bytes, self.gz_block_bytes) * 1.25) + (1 << 20), seq_len, pinned=pinned)
            else:
                slot = textio.alloc_rows(cap, seq_len, signal_len, pinned=pinned)
            if self.canary:
                canary.poison(slot)
            self.free.put(slot)
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
        if self.canary:
            canary.poison(block.slot)        # the consumer is done with it: whoever still reads it reads 0xFF
        self.free.put(block.slot)

    def _take(self):
        """the producer's next free slot"""
        slot = self.free.get()
        if self.canary:
            canary.expect_poisoned(slot, "reader takes an input slot")
        return slot

    def _fresh(self, slot):
        """a slot allocated to replace one that was too small"""
        if self.canary:
            canary.poison(slot)
        return slot

    # -- producer side
    def _emit_staged(self, data, row0):
        """device_parse: the block's text into a staging slot, its row starts noted on the way (no parsing here)"""
        from . import parse_dev
        slot = self._take()
        if len(data) + 1 > slot["cap_bytes"]:      # longer rows than the slots were sized for: this slot grows
            slot = self._fresh(parse_dev.alloc_stage(slot["cap_rows"], int(len(data) * 1.25) + (1 << 20), self.L, pinned=self.pinned))
        import time as _time
        t0 = _time.time()
        try:
            rows, n_bytes = parse_dev.stage_rows(data, slot, self.L, self.S)
        except RuntimeError as e:                  # shorter rows than expected: more of them than the slot has room for
            if "more than" not in str(e):
                raise
            slot = self._fresh(parse_dev.alloc_stage(textio.count_rows(data) + 1, slot["cap_bytes"], self.L, pinned=self.pinned))
            rows, n_bytes = parse_dev.stage_rows(data, slot, self.L, self.S)
        self.stage_seconds += _time.time() - t0
        b = Block()
        b.rows, b.first_row, b.slot, b.n_bytes = rows, row0, slot, n_bytes
        self.q.put(b)
        return rows.n

    def _emit(self, data, row0):
        if self.device_parse:
            return self._emit_staged(data, row0)
        slot = self._take()
        try:
            rows = textio.parse_rows(data, self.L, self.S, nthreads=self.nthreads, out=slot)
        except RuntimeError as e:  # rows much shorter than expected: this block gets its own (unpinned) buffers
            if "capacity" not in str(e):
                raise
            if self.canary:
                canary.poison(slot)          # (the failed parse had begun to fill it)
            self.free.put(slot)
            slot = self._fresh(textio.alloc_rows(textio.count_rows(data), self.L, self.S, pinned=False))
            rows = textio.parse_rows(data, self.L, self.S, nthreads=self.nthreads, out=slot)
        b = Block()
        b.rows, b.first_row, b.slot, b.n_bytes = rows, row0, slot, None
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
            slot = self._take()
            rows, _out, info = self.ff.read_block(bi, out=slot, info=slot.get("_info"), nthreads=self.nthreads)
            slot["_info"] = info
            b = Block()
            b.rows, b.first_row, b.slot, b.n_bytes = rows, int(self.ff.block_first_row[bi]), slot, None
            self.q.put(b)
            row = b.first_row + rows.n
        return row

    def _run_plain_staged(self, row):
        """device_parse, plain text: every block is read straight into a staging slot (parse_dev.read_rows: pread + row
        starts in one pass; exactly EXACT_ROWS rows per block, or block_bytes when that is pinned)"""
        import time as _time
        from . import parse_dev
        size = os.path.getsize(self.path)
        if size == 0:
            return row
        if self.byte_range is not None:
            a, b = self.byte_range
        else:
            with open(self.path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
                a, b = dsp_dist.byte_range_for_rank(mm, size, self.world, self.rank)
        fd = os.open(self.path, os.O_RDONLY)
        try:
            pos = a
            while pos < b:
                slot = self._take()
                exact = EXACT_ROWS if (EXACT_ROWS and self.cap >= EXACT_ROWS) else 0
                t0 = _time.time()
                budget = None if exact else max(self.block_bytes, 1)     # (a pinned block size: tests)
                while True:
                    rows, n_bytes, used = parse_dev.read_rows(fd, pos, b - pos, exact or slot["cap_rows"], b == size, slot, self.L, self.S,
                                                              budget_bytes=budget)
                    if rows.n > 0 or used > 0:
                        break
                    if budget is not None and budget < slot["cap_bytes"]:
                        budget = None         # a pinned block size smaller than a row: the block is what the slot holds
                        continue
                    # not even one row fits the slot: it grows
                    slot = self._fresh(parse_dev.alloc_stage(slot["cap_rows"], slot["cap_bytes"] * 2, self.L, pinned=self.pinned))
                self.stage_seconds += _time.time() - t0
                blk = Block()
                blk.rows, blk.first_row, blk.slot, blk.n_bytes = rows, row, slot, n_bytes
                self.q.put(blk)
                row += rows.n
                pos += used
        finally:
            os.close(fd)
        return row

    def _run_plain(self, row):
        if self.device_parse:
            return self._run_plain_staged(row)
        size = os.path.getsize(self.path)
        if size == 0:
            return row
        # the mapping is parsed in place (no copy out of the page cache; the parser threads take the page faults) and
        # stays referenced by the blocks' sampleinfo views, so it is left to the garbage collector
        with open(self.path, "rb") as f:
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        a, b = self.byte_range if self.byte_range is not None else dsp_dist.byte_range_for_rank(mm, size, self.world, self.rank)
        pos = a
        exact_bytes = 0   # bytes the previous block of EXACT_ROWS rows took
        while pos < b:
            if EXACT_ROWS and self.cap >= EXACT_ROWS:
                # the window holds EXACT_ROWS rows at the bytes/row of the PREVIOUS block (+5 %; the first block guesses from
                # block_bytes); when it falls short the scan continues behind it with the rows still missing -- no byte is
                # scanned twice (a fixed 64 MB window was too small for the default 68 MB blocks, every block
                # was searched twice)
                win = int(exact_bytes * 1.05) + (1 << 16) if exact_bytes else int(self.block_bytes * 1.25) + (1 << 20)
                lo, left = pos, EXACT_ROWS
                while True:
                    top = min(b, lo + win)
                    view = np.frombuffer(mm, dtype=np.uint8, count=top - lo, offset=lo)
                    k = textio.find_row_end(view, left)
                    if k < top - lo or top == b:
                        del view
                        break
                    left -= textio.count_newlines(view)          # rows that end inside the window scanned so far
                    del view
                    lo = top
                    if left <= 0:                                # (the window ended exactly behind the last wanted row)
                        k = 0
                        break
                k += lo - pos
                end = pos + k
            else:
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
            if n == EXACT_ROWS:
                exact_bytes = end - pos
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
        exact = EXACT_ROWS if self.cap >= EXACT_ROWS else 0
        want = self.block_bytes   # text to have in hand per block; with `exact`: a little more than EXACT_ROWS rows of it
        while m < m1:
            e = self._batch_end(bz, m, m1, max(1, want - len(carry)))
            need = int(bz.text_off[e] - bz.text_off[m])
            buf = np.empty(len(carry) + need, np.uint8)
            buf[:len(carry)] = carry
            bz.inflate(m, e, out=buf, out_offset=len(carry), nthreads=self.nthreads)
            m = e
            if exact:
                # cut at exactly EXACT_ROWS rows (whole rounds of workgroups in the forward); what is behind them -- a few
                # hundred rows and the head of a split one -- is carried into the next block
                k = textio.find_row_end(buf, exact)
                if k < len(buf) or (k and buf[k - 1] == 10 and textio.count_newlines(buf) == exact):
                    carry = buf[k:].copy()
                    row += self._emit(buf[:k], row)
                    want = int(k / exact * exact * 1.02) + (1 << 16)
                    continue
            nl = self._gz_cut(buf, len(buf)) if len(buf) else -1
            if m >= m1 and last_rank and len(buf) and nl < len(buf) - 1 and buf[nl + 1:].tobytes().strip():
                nl = len(buf) - 1   # the file's last block: an unterminated last row goes with it
            if nl < 0:
                carry = buf
                continue
            carry = buf[nl + 1:].copy()
            n = self._emit(buf[:nl + 1], row)
            row += n
            if exact and n > 256:   # fewer rows than wanted (the first block's guess): size the next ones by bytes per row
                want = int((nl + 1) / n * exact * 1.02) + (1 << 16)
        while len(carry):   # rows behind the last cut: their ends lie inside this rank's members
            k = textio.find_row_end(carry, exact) if exact else len(carry)
            if exact and k < len(carry):              # still more than a block's worth
                row += self._emit(carry[:k], row)
                carry = carry[k:].copy()
                continue
            if last_rank:                             # the file's end: an unterminated last row goes with the rest
                if carry.tobytes().strip():
                    row += self._emit(carry, row)
            else:                                     # the head of a row that ends in the next rank's members stays behind
                nl = self._gz_cut(carry, len(carry))
                if nl >= 0:
                    row += self._emit(carry[:nl + 1], row)
            break
        return row

    def _batch_end(self, bz, m, m1, nbytes=None):
        e = int(np.searchsorted(bz.text_off, bz.text_off[m] + (self.block_bytes if nbytes is None else nbytes), side="left"))
        return min(m1, max(m + 1, e))

    # ---- foreign .gz (one deflate stream; what the reference's `extract --gzip` writes) --------------------------------
    def _gz_cut(self, buf, n):
        """index of the last newline of buf[:n], or -1"""
        lo = max(0, n - (1 << 16))
        pos = np.flatnonzero(buf[lo:n] == 10)
        if len(pos):
            return lo + int(pos[-1])
        allpos = np.flatnonzero(buf[:n] == 10)
        return int(allpos[-1]) if len(allpos) else -1

    def _gz_inflate_blocks(self, get_buf, put_block):
        """The one inflater of a foreign .gz: blocks of complete rows, inflated straight into the buffers get_buf(i)
        hands out (private arrays, or slots of the node's shared-memory ring); put_block(i, buf, nbytes, first_row,
        n_rows) passes block i on.  Returns (number of blocks, global index after the last row); the compressed bytes read
        are left in self.gz_bytes_in."""
        from . import gzio
        # several host threads and a big file: the parallel inflater.  The node's ONE inflater feeds every rank of the
        # node, so with a ring it takes the CPUs the ranks' parsers leave idle while they wait for it (one stays per rank)
        nt = self.nthreads
        if self.gz_ring is not None and self.gz_ring.get("producer"):
            nt = max(nt, dsp_dist.spare_cpus(self.gz_ring["local_world"]))
        st = gzio.open_gz_stream(self.path, nt)
        self.gz_parallel = isinstance(st, gzio.PgzStream)
        carry = np.zeros(0, np.uint8)
        i, row = 0, self.first_row
        try:
            while True:
                buf = get_buf(i)
                room = min(len(buf), len(carry) + self.gz_block_bytes)
                if len(carry) >= room:
                    raise ValueError("a row of %s is longer than a reader block (%d bytes)" % (self.path, room))
                buf[:len(carry)] = carry
                got = st.readinto(buf[:room], len(carry))
                n = len(carry) + got
                exact = EXACT_ROWS if self.cap >= EXACT_ROWS else 0
                if got == 0:       # end of the stream: the carry holds the rows behind the last cut (with exact cutting
                    # usually thousands of complete ones) and possibly an unterminated last row: the same cut loop, then
                    # the remainder with its true row count (it went out as "1 row")
                    first = True   # (the carry already sits at the head of this block's buffer)
                    while len(carry) and carry.tobytes().strip():
                        cut = textio.find_row_end(carry, exact) if exact else len(carry)
                        k = exact if (exact and cut < len(carry)) else textio.count_rows(carry[:cut])
                        if not first:
                            buf = get_buf(i)
                            buf[:cut] = carry[:cut]
                        first = False
                        put_block(i, buf, cut, row, k)
                        i, row = i + 1, row + k
                        carry = carry[cut:].copy()
                    break
                exact = EXACT_ROWS if self.cap >= EXACT_ROWS else 0
                cut = textio.find_row_end(buf[:n], exact) if exact else n
                if exact and cut < n:                     # exactly EXACT_ROWS rows: whole rounds of workgroups in the forward
                    carry = buf[cut:n].copy()
                    put_block(i, buf, cut, row, exact)
                    i, row = i + 1, row + exact
                    continue
                nl = self._gz_cut(buf, n)
                if nl < 0:
                    carry = buf[:n].copy()
                    continue
                carry = buf[nl + 1:n].copy()
                k = textio.count_rows(buf[:nl + 1])
                put_block(i, buf, nl + 1, row, k)
                i, row = i + 1, row + k
            self.gz_bytes_in = st.bytes_in()
        finally:
            st.close()
        return i, row

    def _run_gz_stream(self, row):
        """A foreign .gz cannot be range-split.  One rank: the inflater runs in its own thread, two blocks ahead of the
        parser.  Several ranks: the first rank of the node inflates ONCE into a shared-memory ring (gz_ring, set up by
        open_gz_ring) and every rank copies its blocks out (block i -> rank i % world), with the global index of the
        block's first row attached; without a ring (no shared memory to be had) every rank inflates the stream itself
        and keeps its blocks, as before round 3."""
        if self.world > 1 and self.gz_ring is not None:
            return self._run_gz_ring(row)
        mine = lambda i: i % self.world == self.rank
        bq = queue.Queue(maxsize=2)
        slack = 1 << 20
        stop = threading.Event()   # the consumer failed (a malformed row, ...): the inflater must not wait on a full queue

        def hand_over(item):
            while not stop.is_set():
                try:
                    bq.put(item, timeout=0.2)
                    return
                except queue.Full:
                    pass
            raise RuntimeError("reader stopped")

        def get_buf(i):
            return np.empty(self.gz_block_bytes + slack, np.uint8)

        def put_block(i, buf, n, first_row, n_rows):
            if mine(i):
                hand_over((buf[:n], first_row))

        def produce():
            try:
                _, end_row = self._gz_inflate_blocks(get_buf, put_block)
                hand_over(("end", end_row))
            except BaseException as e:
                if not stop.is_set():
                    hand_over(("error", e))
        t = threading.Thread(target=produce, daemon=True)
        t.start()
        try:
            while True:
                data, info = bq.get()
                if isinstance(data, str):
                    if data == "error":
                        raise info
                    row = info
                    break
                self._emit(data, info)
        finally:
            stop.set()
            t.join()
        return row

    def _run_gz_ring(self, row):
        ring, li, lw = self.gz_ring["ring"], self.gz_ring["local_index"], self.gz_ring["local_world"]
        first = self.rank - li            # global rank of this node's first rank
        producer = None
        if self.gz_ring.get("producer"):
            def local_seq(i):             # dense numbering of the blocks this node's ranks own, or -1
                o = i % self.world - first
                return (i // self.world) * lw + o if 0 <= o < lw else -1
            scratch = []

            def get_buf(i):
                s = local_seq(i)
                if s >= 0:
                    return ring.acquire(s)
                if not scratch:
                    scratch.append(np.empty(ring.slot_bytes, np.uint8))
                return scratch[0]
            published = [0]

            def put_block(i, buf, n, first_row, n_rows):
                s = local_seq(i)
                if s >= 0:
                    ring.publish(s, n, first_row, n_rows)
                    published[0] = s + 1

            def produce():
                try:
                    self._gz_inflate_blocks(get_buf, put_block)
                    ring.finish(published[0], 0)
                except BaseException as e:
                    ring.finish(published[0], -5, "%s: %s" % (type(e).__name__, e))
            producer = threading.Thread(target=produce, daemon=True)
            producer.start()
        k = 0
        try:
            while True:
                got = ring.take(k * lw + li)
                if got is None:
                    break
                data, first_row, n_rows = got
                self._emit(data, first_row)
                row = first_row + n_rows
                k += 1
        except BaseException:
            ring.abort()
            raise
        finally:
            if producer is not None:
                producer.join()
        return row


def open_gz_ring(path, rank, world, local_rank, local_world, all_gather_object, block_bytes=GZ_BLOCK_BYTES):
    """Collective over all ranks (call it on every rank, before the readers start): for a foreign single-stream .gz read
    by several ranks, the first rank of every node creates a shared-memory ring; returns the `gz_ring` argument of
    FeatureReader, or None when the file needs no ring (one rank) or a node could not reserve the shared memory (every
    rank then inflates for itself).  all_gather_object(obj) -> list of every rank's obj (torch.distributed's, or a test's)."""
    if world <= 1:
        return None
    import secrets
    from . import gzio
    ring, name = None, None
    if local_rank == 0:
        name = "/dsp_gz_%d_%s" % (os.getpid(), secrets.token_hex(6))
        try:
            ring = gzio.ShmRing.create(name, local_world + 2, block_bytes + (1 << 20))
        except (MemoryError, OSError) as e:
            import sys
            sys.stderr.write("[feed] no shared-memory ring for %s (%s): every rank inflates the stream itself\n" % (path, e))
            name = None
    names = all_gather_object(name)
    if any(names[r - lr] is None for r, lr in enumerate(all_gather_object(local_rank))):
        if ring is not None:
            ring.close()
        return None
    producer = ring is not None
    if ring is None:
        ring = gzio.ShmRing.attach(names[rank - local_rank])
    all_gather_object(True)   # everybody has attached: the names can go, so that even a killed run leaves nothing in /dev/shm
    if producer:
        ring.unlink()
    return dict(ring=ring, producer=producer, local_index=local_rank, local_world=local_world)
