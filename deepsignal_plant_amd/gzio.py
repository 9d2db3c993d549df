"""gzip I/O of the host pipeline over the native layer in libdsp_amd.so (csrc/dsp_gz.cpp).

Writers emit BGZF -- a chain of ordinary gzip members of <= 64 KiB of text, each carrying its compressed size, ended
by an empty member -- which `gzip.open` / zcat / the reference read like any .gz (call_modifications.py:66-69 reads the
feature file with gzip.open when it ends in .gz), and which this build's reader can index, deal to ranks by member
ranges and inflate on many threads.  Other .gz files are streamed through one native inflate thread."""
from __future__ import annotations

import ctypes
import mmap
import os

import numpy as np

from . import _native as nat


class BgzfWriter(object):
    """file-like, write-only: bytes in -> BGZF members out.  write() only queues the data: a background thread deflates
    it (on `nthreads` threads, inside the library, GIL released) and appends to the file, so that the caller formats its
    next block meanwhile; errors surface at the next write() / close()."""

    def __init__(self, path, level=4, nthreads=8, chunk=8 << 20):
        import queue
        import threading
        self.f = open(path, "wb")
        self.level, self.nthreads, self.chunk = int(level), max(1, int(nthreads)), int(chunk)
        self.buf = bytearray()
        self.q = queue.Queue(maxsize=3)
        self.error = None
        self.pos = 0            # compressed bytes written so far (worker thread)
        self.block_ends = []    # file offsets recorded by end_block(), in order
        self.worker = threading.Thread(target=self._run, daemon=True)
        self.worker.start()

    def _run(self):
        L = nat.lib()
        out = None
        while True:
            data = self.q.get()
            if data is None:
                return
            if callable(data):   # write(..., on_done=): everything queued before it has been dealt with
                data()
                continue
            if self.error is not None:
                continue  # drain
            if data is BgzfWriter._MARK:
                self.block_ends.append(self.pos)
                continue
            try:
                n = len(data)
                need = n + (n // 0xff00 + 2) * 64 + 65536
                if out is None or out.nbytes < need:
                    out = np.empty(need, np.uint8)
                src = np.frombuffer(data, np.uint8)
                k = nat.check(int(L.dsp_bgzf_compress(ctypes.c_void_p(src.ctypes.data), n, ctypes.c_void_p(out.ctypes.data),
                                                      out.nbytes, self.level, self.nthreads)))
                self.f.write(memoryview(out)[:k])
                self.pos += k
            except BaseException as e:  # surfaced by the producer
                self.error = e

    def _submit(self, final=False):
        while len(self.buf) >= self.chunk or (final and self.buf):
            n = min(len(self.buf), max(self.chunk, 0xff00))
            if not final or n < len(self.buf):
                n -= n % 0xff00   # whole members only, so that the chain stays maximally packed
            self.q.put(bytes(self.buf[:n]))
            del self.buf[:n]
            if self.error is not None:
                raise self.error

    def write(self, data, on_done=None):
        """data: a buffer, or a list of buffers that go to the deflater as they are (no copy through the carry buffer; every
        one ends its own member); on_done() runs on the writer's thread once they have been compressed and written --
        the caller's signal that their memory may be reused (without it the caller must not touch a listed buffer again)"""
        if self.error is not None:
            raise self.error
        if isinstance(data, (list, tuple)):
            self._submit(final=True)
            for piece in data:
                if len(piece):
                    self.q.put(piece)
            if on_done is not None:
                self.q.put(on_done)
            if self.error is not None:
                raise self.error
            return sum(len(x) for x in data)
        if on_done is not None:
            raise ValueError("on_done needs a list of buffers")
        if not self.buf and len(data) >= self.chunk and len(data) % 0xff00 == 0:
            self.q.put(bytes(data))  # already member-aligned: no copy through the carry buffer
        else:
            self.buf += data
            self._submit()
        return len(data)

    _MARK = object()

    def end_block(self):
        """close the current member here and remember the file offset (block_ends, complete after close()): the caller
        can later cut the file at these offsets into pieces that are whole BGZF members"""
        self._submit(final=True)
        self.q.put(BgzfWriter._MARK)

    def close(self):
        if self.f is None:
            return
        try:
            self._submit(final=True)
        finally:
            self.q.put(None)
            self.worker.join()
        if self.error is not None:
            self.f.close()
            self.f = None
            raise self.error
        eof = np.empty(28, np.uint8)
        nat.check(int(nat.lib().dsp_bgzf_eof(ctypes.c_void_p(eof.ctypes.data), 28)))
        self.f.write(eof.tobytes())
        self.f.close()
        self.f = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class BgzfFile(object):
    """A .gz that is entirely BGZF: member table from a header walk over the mmap; members inflate independently."""

    def __init__(self, path):
        self.path = path
        self.size = os.path.getsize(path)
        self.f = open(path, "rb")
        self.mm = mmap.mmap(self.f.fileno(), 0, access=mmap.ACCESS_READ) if self.size else None
        self.src = np.frombuffer(self.mm, np.uint8) if self.mm is not None else np.zeros(0, np.uint8)
        L = nat.lib()
        m = int(L.dsp_gz_index(ctypes.c_void_p(self.src.ctypes.data), self.size, 0, None, None)) if self.size else 0
        self.ok = m >= 0
        self.n_members = max(m, 0)
        if self.ok:
            self.off = np.zeros(self.n_members + 1, np.uint64)
            self.isize = np.zeros(max(self.n_members, 1), np.uint32)
            if self.n_members:
                nat.check(int(L.dsp_gz_index(ctypes.c_void_p(self.src.ctypes.data), self.size, self.n_members,
                                             ctypes.c_void_p(self.off.ctypes.data), ctypes.c_void_p(self.isize.ctypes.data))))
            self.text_off = np.r_[0, np.cumsum(self.isize[:self.n_members].astype(np.int64))]
            # newlines per member as the writer recorded them in the header (this build's BgzfWriter; -1: not recorded)
            self.rows = np.full(max(self.n_members, 1), -1, np.int64)
            if self.n_members:
                nat.check(int(L.dsp_gz_member_rows(ctypes.c_void_p(self.src.ctypes.data), ctypes.c_void_p(self.off.ctypes.data),
                                                   self.n_members, ctypes.c_void_p(self.rows.ctypes.data))))

    def inflate(self, m0, m1, out=None, out_offset=0, nthreads=8):
        """members [m0, m1) -> out[out_offset:...]; returns (array, bytes written)"""
        need = int(self.text_off[m1] - self.text_off[m0])
        if out is None:
            out = np.empty(out_offset + need, np.uint8)
        if need:
            k = nat.check(int(nat.lib().dsp_gz_inflate_members(
                ctypes.c_void_p(self.src.ctypes.data), ctypes.c_void_p(self.off.ctypes.data), ctypes.c_void_p(self.isize.ctypes.data),
                int(m0), int(m1), ctypes.c_void_p(out.ctypes.data + out_offset), out.nbytes - out_offset, int(nthreads))))
            assert k == need
        return out, need

    def members_for_rank(self, world, rank):
        """contiguous member range of `rank`: equal shares of the compressed bytes"""
        if self.n_members == 0:
            return 0, 0
        cuts = [int(np.searchsorted(self.off[:self.n_members], self.size * r // world, side="left")) for r in range(world)] + [self.n_members]
        return cuts[rank], cuts[rank + 1]


class GzStream(object):
    """any .gz, sequentially (all members), through zlib in C with the GIL released"""

    def __init__(self, path):
        self.h = ctypes.c_void_p(nat.lib().dsp_gz_open(os.fsencode(path)))
        if not self.h:
            raise ValueError("cannot open %s" % path)

    def readinto(self, arr, offset=0):
        return nat.check(int(nat.lib().dsp_gz_read(self.h, ctypes.c_void_p(arr.ctypes.data + offset), arr.nbytes - offset)))

    def bytes_in(self):
        """compressed bytes consumed so far"""
        return int(nat.lib().dsp_gz_bytes_in(self.h)) if self.h else 0

    def close(self):
        if self.h:
            nat.lib().dsp_gz_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PgzStream(object):
    """one gzip stream inflated on `nthreads` threads (csrc/dsp_pgz.cpp): same reading contract as GzStream"""

    def __init__(self, path, nthreads=4, chunk_bytes=0):
        self.h = ctypes.c_void_p(nat.lib().dsp_pgz_open(os.fsencode(path), int(nthreads), int(chunk_bytes)))
        if not self.h:
            raise ValueError("cannot open %s: %s" % (path, nat.lib().dsp_last_error().decode()))

    def readinto(self, arr, offset=0):
        return nat.check(int(nat.lib().dsp_pgz_read(self.h, ctypes.c_void_p(arr.ctypes.data + offset), arr.nbytes - offset)))

    def bytes_in(self):
        return int(nat.lib().dsp_pgz_bytes_in(self.h)) if self.h else 0

    def stats(self):
        r, d = ctypes.c_uint64(), ctypes.c_uint64()
        nat.lib().dsp_pgz_stats(self.h, ctypes.byref(r), ctypes.byref(d))
        return int(r.value), int(d.value)

    def close(self):
        if self.h:
            nat.lib().dsp_pgz_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


PGZ_MIN_BYTES = 16 << 20   # smaller files are done before the second chunk would have found its start
PGZ_MAX_THREADS = 32        # a round holds one chunk per thread in memory (4 MB compressed -> ~30 MB each), two rounds in flight


def open_gz_stream(path, nthreads=1):
    """the reader of a foreign .gz: the parallel inflater when there are threads to use and the file is big enough to
    hold more than a chunk or two, else the sequential zlib reader (DSP_GZ_SEQUENTIAL=1 forces that one)"""
    if nthreads >= 2 and not os.environ.get("DSP_GZ_SEQUENTIAL") and os.path.getsize(path) >= PGZ_MIN_BYTES:
        return PgzStream(path, min(nthreads, PGZ_MAX_THREADS))
    return GzStream(path)


class ShmRing(object):
    """node-local ring of text blocks in POSIX shared memory (csrc/dsp_shmring.cpp): the node's first rank inflates a
    foreign single-stream .gz ONCE into it, every rank of the node copies its own blocks out."""

    TIMEOUT = float(os.environ.get("DSP_RING_TIMEOUT_S", 900.0))

    def __init__(self, handle, name, owner):
        self.h, self.name, self.owner = ctypes.c_void_p(handle), name, owner
        self.slot_bytes = int(nat.lib().dsp_shm_ring_slot_bytes(self.h))

    @classmethod
    def create(cls, name, n_slots, slot_bytes):
        h = nat.lib().dsp_shm_ring_create(name.encode(), int(n_slots), int(slot_bytes))
        if not h:
            raise MemoryError(nat.lib().dsp_last_error().decode())
        return cls(h, name, True)

    @classmethod
    def attach(cls, name, timeout=60.0):
        h = nat.lib().dsp_shm_ring_attach(name.encode(), float(timeout))
        if not h:
            raise RuntimeError(nat.lib().dsp_last_error().decode())
        return cls(h, name, False)

    # -- producer
    def acquire(self, seq):
        """writable uint8 view of the slot of block `seq` (blocks until its previous tenant has been released)"""
        p = nat.lib().dsp_shm_ring_acquire(self.h, int(seq), self.TIMEOUT)
        if not p:
            raise RuntimeError(nat.lib().dsp_last_error().decode())
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(self.slot_bytes,))

    def publish(self, seq, length, first_row, n_rows):
        nat.check(int(nat.lib().dsp_shm_ring_publish(self.h, int(seq), int(length), int(first_row), int(n_rows))))

    def finish(self, n_blocks, status=0, message=None):
        nat.lib().dsp_shm_ring_finish(self.h, int(n_blocks), int(status), message.encode()[:250] if message else None)

    # -- consumer
    def take(self, seq):
        """(copy of block seq, global index of its first row, its row count), or None when the stream ended before it"""
        L = nat.lib()
        data, ln, fr, nr = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        rc = int(L.dsp_shm_ring_wait(self.h, int(seq), self.TIMEOUT, ctypes.byref(data), ctypes.byref(ln), ctypes.byref(fr),
                                     ctypes.byref(nr)))
        if rc == 1:
            return None
        nat.check(rc)
        view = np.ctypeslib.as_array(ctypes.cast(data, ctypes.POINTER(ctypes.c_uint8)), shape=(int(ln.value),))
        out = view.copy()     # the rows' sampleinfo strings are referenced until the block is written: own the bytes
        L.dsp_shm_ring_release(self.h, int(seq))
        return out, int(fr.value), int(nr.value)

    def abort(self):
        if self.h:
            nat.lib().dsp_shm_ring_abort(self.h)

    def unlink(self):
        """creator, once every consumer has attached: drop the name (the memory lives as long as it is mapped)"""
        if self.h and self.owner:
            nat.lib().dsp_shm_ring_unlink(self.h)
            self.owner = False

    def close(self):
        if self.h:
            nat.lib().dsp_shm_ring_close(self.h, 1 if self.owner else 0)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_text_chunks(path, chunk_bytes=64 << 20, nthreads=8):
    """any .gz as a sequence of uint8 arrays of text, in order, as fast as the file allows: BGZF members inflated in
    parallel batches, a single gzip stream through the parallel inflater, small files through zlib"""
    bz = BgzfFile(path)
    if bz.ok:
        m = 0
        while m < bz.n_members:
            e = int(np.searchsorted(bz.text_off, bz.text_off[m] + chunk_bytes, side="left"))
            e = min(bz.n_members, max(m + 1, e))
            buf, n = bz.inflate(m, e, nthreads=nthreads)
            m = e
            if n:
                yield buf[:n]
        return
    st = open_gz_stream(path, nthreads)
    try:
        while True:
            buf = np.empty(chunk_bytes, np.uint8)
            n = st.readinto(buf)
            if n == 0:
                return
            yield buf[:n]
    finally:
        st.close()


class BackgroundFileWriter(object):
    """file-like, write-only, plain bytes: write() hands the buffer to a background thread that appends it to the file
    (the page-cache copy of a gigabyte of rows takes as long as formatting the next one), so the caller must not modify
    a buffer after writing it; errors surface at the next write() / close()."""

    def __init__(self, path, depth=2):
        import queue
        import threading
        self.f = open(path, "wb")
        self.q = queue.Queue(maxsize=depth)
        self.error = None
        self.worker = threading.Thread(target=self._run, daemon=True)
        self.worker.start()

    def _run(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            data, on_done = item
            if self.error is None:
                try:
                    for piece in (data if isinstance(data, (list, tuple)) else (data,)):
                        self.f.write(piece)
                except BaseException as e:
                    self.error = e
            if on_done is not None:
                on_done()

    def write(self, data, on_done=None):
        """data: a buffer or a list of buffers (written in order); on_done() runs on the writer's thread once they are in
        the file -- the caller's signal that their memory may be reused"""
        if self.error is not None:
            raise self.error
        self.q.put((data, on_done))
        return sum(len(x) for x in data) if isinstance(data, (list, tuple)) else len(data)

    def close(self):
        if self.f is None:
            return
        self.q.put(None)
        self.worker.join()
        self.f.close()
        self.f = None
        if self.error is not None:
            raise self.error

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def open_write(path, is_gzip, nthreads=8, level=4, background=False):
    """binary writer: plain file (background=True: written by its own thread) or BGZF"""
    if is_gzip:
        return BgzfWriter(path, level=level, nthreads=nthreads)
    return BackgroundFileWriter(path) if background else open(path, "wb")
