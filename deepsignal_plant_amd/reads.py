"""Read records: what the reference pulls out of one resquiggled fast5 before any arithmetic
(extract_features.py:44-91 `_get_label_raw`, :151-176 `_get_alignment_info_from_fast5`, :255-270
`_get_scaling_of_a_read`), as plain arrays.  The HDF5 side itself is outside this build (h5py is not in the
image); `from_fast5` is the gated loader for hosts that have it, `synth_reads` the seeded generator used by
tests and benchmarks."""
from __future__ import annotations

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


def from_fast5(path, corrected_group="RawGenomeCorrected_000", basecall_subgroup="BaseCalled_template"):
    """Gated single-read fast5 loader for hosts with h5py (not in this image, untested here): the HDF5 paths are
    those of extract_features.py:36-37, :57-89, :151-176, :255-266."""
    try:
        import h5py
    except ImportError:
        raise RuntimeError("reading fast5 files needs h5py, which this image does not have; feed ReadRecord "
                           "arrays (deepsignal_plant_amd.reads) instead")
    with h5py.File(path, "r") as h5:
        raw = list(h5["/Raw/Reads"].values())[0]
        readname = raw.attrs["read_id"]
        readname = readname.decode() if isinstance(readname, bytes) else str(readname)
        signal = raw["Signal"][()]
        ch = dict(h5["UniqueGlobalKey/channel_id"].attrs.items())
        ev = h5["Analyses/%s/%s/Events" % (corrected_group, basecall_subgroup)]
        rel = dict(ev.attrs.items())["read_start_rel_to_raw"]
        al = h5["Analyses/%s/%s/Alignment" % (corrected_group, basecall_subgroup)].attrs

        def s(x):
            return x.decode() if isinstance(x, bytes) else str(x)
        return ReadRecord(readname, "t" if basecall_subgroup.endswith("template") else "c", s(al["mapped_strand"]),
                          s(al["mapped_chrom"]), int(al["mapped_start"]), signal, ch["range"] / ch["digitisation"],
                          ch["offset"], ev["start"].astype(np.int64) + int(rel), ev["length"].astype(np.int64),
                          np.frombuffer(b"".join(ev["base"]), np.uint8))


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
    """*.fast5 (needs h5py) and *.reads.npz under a directory, sorted (the reference's get_fast5s walks the tree for
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


def load_read_file(path, corrected_group="RawGenomeCorrected_000", basecall_subgroup="BaseCalled_template"):
    """-> list of ReadRecord.  A fast5 that cannot be parsed raises; the caller counts it as failed like the
    reference (extract_features.py:373-375)."""
    if path.endswith(".reads.npz"):
        return load_reads(path)
    return [from_fast5(path, corrected_group, basecall_subgroup)]


class ReadBatches(object):
    """Iterate (reads, uids) batches over a list of read files: files are decoded by a small thread pool (HDF5 / zip
    decoding releases the GIL) with bounded look-ahead and delivered in FILE ORDER; a file that cannot be parsed is
    counted in .failed and skipped, like the reference (extract_features.py:373-375).  uid of a read =
    ((first_file_index + file index) << 20) + index in the file: the key of the subsampler, independent of batching
    and of how files are dealt to ranks."""

    def __init__(self, files, batch_reads, corrected_group="RawGenomeCorrected_000",
                 basecall_subgroup="BaseCalled_template", first_file_index=0, workers=4, lookahead=8):
        self.files, self.batch_reads = list(files), max(1, int(batch_reads))
        self.cg, self.sg, self.first = corrected_group, basecall_subgroup, int(first_file_index)
        self.workers, self.lookahead = max(1, int(workers)), max(1, int(lookahead))
        self.failed = 0

    def _load(self, path):
        try:
            return load_read_file(path, self.cg, self.sg)
        except Exception:
            return None

    def __iter__(self):
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        cur, uids = [], []
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
