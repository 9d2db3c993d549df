"""Read records: what the reference pulls out of one resquiggled fast5 before any arithmetic
(extract_features.py:44-91 `_get_label_raw`, :151-176 `_get_alignment_info_from_fast5`, :255-270
`_get_scaling_of_a_read`), as plain arrays.  `from_fast5` reads them from a fast5 through the native reader
(csrc/dsp_fast5.cpp over the HDF5 C library); `synth_reads` is the seeded generator used by tests and benchmarks."""
from __future__ import annotations

import os

import numpy as np


class ReadRecord(object):
    """One mapped read.
    raw       int16 [n_samples]    Raw/Reads/Read_x/Signal (DAQ values)
    scaling, offset  float         channel range / digitisation, channel offset  (pA = scaling * (raw + offset))
    ev_start  int64 [n_bases]      event starts in `raw` (read_start_rel_to_raw already added, :81)
    ev_len    int64 [n_bases]      event lengths
    ev_base   uint8 [n_bases]      event bases (ASCII)
    readname, strand ('t'/'c'), alignstrand ('+'/'-'), chrom, chrom_start   alignment attributes"""
    __slots__ = ("readname", "strand", "alignstrand", "chrom", "chrom_start", "raw", "scaling", "offset",
                 "ev_start", "ev_len", "ev_base")

    def __init__(self, readname, strand, alignstrand, chrom, chrom_start, raw, scaling, offset, ev_start, ev_len,
                 ev_base):
        self.readname, self.strand, self.alignstrand, self.chrom = readname, strand, alignstrand, chrom
        self.chrom_start = int(chrom_start)
        self.raw = np.ascontiguousarray(raw, np.int16)
        self.scaling, self.offset = float(scaling), float(offset)
        self.ev_start = np.ascontiguousarray(ev_start, np.int64)
        self.ev_len = np.ascontiguousarray(ev_len, np.int64)
        if isinstance(ev_base, (bytes, str)):
            ev_base = np.frombuffer(ev_base.encode() if isinstance(ev_base, str) else ev_base, np.uint8)
        self.ev_base = np.ascontiguousarray(ev_base, np.uint8)
        assert self.ev_start.shape == self.ev_len.shape == self.ev_base.shape

    @property
    def seq(self):
        return self.ev_base.tobytes().decode()


def synth_reads(n_reads, seed=0, mean_bases=400, max_len=40, long_every=97, n_chroms=3, cg_boost=0.15):
    """Seeded synthetic reads: random ACGT sequence with extra CG dinucleotides, per-base dwell 1..max_len samples
    (geometric-like, one long stall every `long_every` bases), DAQ values ~ level(base context) + noise."""
    rng = np.random.default_rng(seed)
    reads = []
    for r in range(n_reads):
        nb = int(max(30, rng.normal(mean_bases, mean_bases * 0.25)))
        codes = rng.integers(0, 4, size=nb)
        for i in np.nonzero(rng.random(nb - 1) < cg_boost)[0]:
            codes[i], codes[i + 1] = 1, 2
        bases = np.frombuffer(b"ACGT", np.uint8)[codes]
        lens = np.minimum(1 + rng.geometric(0.12, size=nb), max_len).astype(np.int64)
        lens[long_every - 1::long_every] = rng.integers(130, 420, size=len(lens[long_every - 1::long_every]))
        lead = int(rng.integers(0, 200))
        starts = lead + np.concatenate([[0], np.cumsum(lens)[:-1]])
        n_samples = int(starts[-1] + lens[-1] + rng.integers(0, 150))
        level = 420 + 45 * codes + 12 * np.roll(codes, 1) - 9 * np.roll(codes, -1)
        per_sample = np.repeat(level, lens)
        raw = rng.normal(470, 60, size=n_samples)
        raw[lead:lead + per_sample.size] = per_sample + rng.normal(0, 9, size=per_sample.size)
        raw = np.clip(np.rint(raw), 0, 8191).astype(np.int16)
        reads.append(ReadRecord("read_%05d-%04x" % (r, int(rng.integers(0, 65536))), "t", "+-"[r % 2],
                                "chr%d" % (r % n_chroms + 1), int(rng.integers(0, 5_000_000)), raw,
                                1467.61 / 8192.0, float(rng.integers(-5, 30)), starts, lens, bases))
    return reads


class NoAlignment(RuntimeError):
    """The fast5 has no Analyses/<corrected group>/<subgroup>/Alignment group: the reference carries on with empty
    alignment fields and fails the read at extract_features.py:327 (counted as an error)."""


def fast5_available():
    """True if the native reader found an HDF5 library (libdsp_amd.so dlopens it; DSP_HDF5_LIB overrides the search)."""
    from . import _native as nat
    return bool(nat.lib().dsp_fast5_available())


def from_fast5(path, corrected_group="RawGenomeCorrected_000", basecall_subgroup="BaseCalled_template", only_chrom=None):
    """One tombo-resquiggled single-read fast5 -> ReadRecord, through the native reader (csrc/dsp_fast5.cpp: the HDF5 C
    library, no h5py): the HDF5 paths and the failure texts are those of extract_features.py:36-37, :44-91, :94-176,
    :255-270.  only_chrom: the region filter's chromosome -- a read that maps elsewhere (or has no alignment, or cannot
    be opened) returns None before anything else is read, as at :308-309.  Errors raise RuntimeError (the caller counts
    the file as failed, :373-375)."""
    import ctypes
    from . import _native as nat
    L = nat.lib()
    rec = nat.Fast5Read()
    rc = L.dsp_fast5_load(os.fsencode(path), corrected_group.encode(), basecall_subgroup.encode(),
                          None if only_chrom is None else only_chrom.encode(), ctypes.byref(rec))
    if rc == 1:  # DSP_FAST5_SKIPPED
        return None
    if rc != 0:
        raise RuntimeError(nat.last_error())
    try:
        if not rec.has_alignment:
            raise NoAlignment("no Alignment group in %s" % path)
        n, m = int(rec.n_raw), int(rec.n_events)
        raw = np.ctypeslib.as_array(rec.raw, shape=(max(n, 1),))[:n].copy()
        ev_start = np.ctypeslib.as_array(rec.ev_start, shape=(max(m, 1),))[:m].copy()
        ev_len = np.ctypeslib.as_array(rec.ev_len, shape=(max(m, 1),))[:m].copy()
        ev_base = np.ctypeslib.as_array(rec.ev_base, shape=(max(m, 1),))[:m].copy()
        # scaling = range / digitisation in float64, as the reference divides the two attributes (:262)
        return ReadRecord(rec.read_id.decode("utf-8", "replace"), "t" if basecall_subgroup.endswith("template") else "c",
                          rec.mapped_strand.decode("utf-8", "replace"), rec.mapped_chrom.decode("utf-8", "replace"),
                          int(rec.mapped_start), raw, rec.range / rec.digitisation, rec.offset, ev_start, ev_len, ev_base)
    finally:
        L.dsp_fast5_free(ctypes.byref(rec))


# ---- read-record container (.reads.npz): the HDF5-free interchange of this build ------------------------------------
def save_reads(path, reads, compress=True):
    """Write reads as one npz: arrays back to back + CSR offsets + string arrays.  Any tool that can open fast5
    files can produce this (the fields are ReadRecord's); `load_reads` restores the list."""
    off = lambda xs: np.concatenate([[0], np.cumsum(xs)]).astype(np.int64)
    cat = lambda xs, dt: (np.concatenate(xs).astype(dt) if len(xs) else np.zeros(0, dt))
    (np.savez_compressed if compress else np.savez)(
        path if path.endswith(".npz") else path + ".npz",
        raw=cat([r.raw for r in reads], np.int16), raw_off=off([len(r.raw) for r in reads]),
        ev_start=cat([r.ev_start for r in reads], np.int64), ev_len=cat([r.ev_len for r in reads], np.int64),
        ev_base=cat([r.ev_base for r in reads], np.uint8), ev_off=off([len(r.ev_base) for r in reads]),
        scaling=np.array([r.scaling for r in reads], np.float64), offset=np.array([r.offset for r in reads], np.float64),
        chrom_start=np.array([r.chrom_start for r in reads], np.int64),
        readname=np.array([r.readname for r in reads]), chrom=np.array([r.chrom for r in reads]),
        strand=np.array([r.strand for r in reads]), alignstrand=np.array([r.alignstrand for r in reads]))


def load_reads(path):
    with np.load(path, allow_pickle=False) as z:
        d = {k: z[k] for k in z.files}  # NpzFile re-reads a member on every access: pull each one once
    ro, eo = d["raw_off"], d["ev_off"]
    raw, es, el, eb = d["raw"], d["ev_start"], d["ev_len"], d["ev_base"]
    names, strands, astrands, chroms = (d[k].tolist() for k in ("readname", "strand", "alignstrand", "chrom"))
    cstart, scaling, offset = d["chrom_start"].tolist(), d["scaling"].tolist(), d["offset"].tolist()
    return [ReadRecord(names[i], strands[i], astrands[i], chroms[i], cstart[i], raw[ro[i]:ro[i + 1]], scaling[i],
                       offset[i], es[eo[i]:eo[i + 1]], el[eo[i]:eo[i + 1]], eb[eo[i]:eo[i + 1]])
            for i in range(len(ro) - 1)]


def list_read_files(input_dir, recursive=True):
    """*.fast5 and *.reads.npz under a directory, sorted (the reference's get_fast5s walks the tree for
    *.fast5, utils/process_utils.py:148-161)."""
    import os
    found = []
    if recursive:
        for root, _dirs, files in os.walk(os.path.abspath(input_dir)):
            found += [os.path.join(root, f) for f in files if f.endswith(".fast5") or f.endswith(".reads.npz")]
    else:
        found = [os.path.join(os.path.abspath(input_dir), f) for f in os.listdir(input_dir)
                 if f.endswith(".fast5") or f.endswith(".reads.npz")]
    return sorted(found)


def load_read_file(path, corrected_group="RawGenomeCorrected_000", basecall_subgroup="BaseCalled_template", only_chrom=None):
    """-> list of ReadRecord.  A fast5 that cannot be parsed raises; the caller counts it as failed like the
    reference (extract_features.py:373-375).  only_chrom: see from_fast5."""
    if path.endswith(".reads.npz"):
        return load_reads(path)
    r = from_fast5(path, corrected_group, basecall_subgroup, only_chrom)
    return [] if r is None else [r]


def _reader_init():
    from . import _native
    _native.NO_TORCH = True  # reader processes never touch the GPU


def _load_chunk(task):
    paths, cg, sg, only_chrom = task
    out = []
    for p in paths:
        try:
            out.append(load_read_file(p, cg, sg, only_chrom))
        except Exception:
            out.append(None)
    return out


class ReadBatches(object):
    """Iterate (reads, uids) batches over a list of read files, delivered in FILE ORDER with bounded look-ahead; a file
    that cannot be parsed is counted in .failed and skipped, like the reference (extract_features.py:373-375).
    Decoding runs on `workers` threads, or -- procs > 1 and at least `procs_min_files` files -- on `procs` reader
    PROCESSES (spawned, no GPU, no torch): libhdf5 decodes one file at a time per process (≈ 200 files/s of 100 k samples),
    so fast5 directories scale with processes the way the reference's --nproc does (extract_features.py:589-651).
    uid of a read = ((first_file_index + file index) << 20) + index in the file: the key of the subsampler, independent
    of batching and of how files are dealt to ranks."""

    def __init__(self, files, batch_reads, corrected_group="RawGenomeCorrected_000",
                 basecall_subgroup="BaseCalled_template", first_file_index=0, workers=4, lookahead=8, only_chrom=None,
                 procs=0, procs_min_files=256, chunk_files=8):
        self.files, self.batch_reads = list(files), max(1, int(batch_reads))
        self.cg, self.sg, self.first = corrected_group, basecall_subgroup, int(first_file_index)
        self.workers, self.lookahead = max(1, int(workers)), max(1, int(lookahead))
        self.failed = 0
        self.only_chrom = only_chrom  # chromosome of the region of interest: reads elsewhere are dropped unread
        self.procs = int(procs) if (int(procs) > 1 and len(self.files) >= int(procs_min_files)) else 0
        self.chunk_files = max(1, int(chunk_files))

    def _load(self, path):
        try:
            return load_read_file(path, self.cg, self.sg, self.only_chrom)
        except Exception:
            return None

    def _decoded(self):
        """-> (file index, list of ReadRecord or None) in file order"""
        from collections import deque
        if self.procs:
            import multiprocessing as mp
            from concurrent.futures import ProcessPoolExecutor
            chunks = [(i, self.files[i:i + self.chunk_files]) for i in range(0, len(self.files), self.chunk_files)]
            with ProcessPoolExecutor(self.procs, mp_context=mp.get_context("spawn"), initializer=_reader_init) as pool:
                pending, it = deque(), iter(chunks)

                def refill():
                    while len(pending) < 3 * self.procs:
                        nxt = next(it, None)
                        if nxt is None:
                            return
                        pending.append((nxt[0], pool.submit(_load_chunk, (nxt[1], self.cg, self.sg, self.only_chrom))))
                refill()
                while pending:
                    i0, fut = pending.popleft()
                    got = fut.result()
                    refill()
                    for k, g in enumerate(got):
                        yield i0 + k, g
            return
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(self.workers) as pool:
            pending = deque()
            it = iter(enumerate(self.files))

            def refill():
                while len(pending) < self.lookahead:
                    nxt = next(it, None)
                    if nxt is None:
                        return
                    pending.append((nxt[0], pool.submit(self._load, nxt[1])))
            refill()
            while pending:
                fi, fut = pending.popleft()
                got = fut.result()
                refill()
                yield fi, got

    def __iter__(self):
        cur, uids = [], []
        for fi, got in self._decoded():
            if got is None:
                self.failed += 1
                continue
            cur += got
            uids += [((self.first + fi) << 20) + i for i in range(len(got))]
            if len(cur) >= self.batch_reads:
                yield cur, uids
                cur, uids = [], []
        if cur:
            yield cur, uids
