"""Feature extraction from resquiggled reads on MI355X.

Host-side mirror of deepsignal_plant/extract_features.py for everything after the HDF5 reads: given read
records (reads.py: raw DAQ samples, channel scaling, resquiggle events, alignment attributes) it produces the
features of every motif site -- as device tensors in the layout dsp_forward consumes (the reference's fast5
route of call_mods, call_modifications.py:285-325), or rounded the way the feature TSV carries them
(_features_to_str, :381-395) for writing `extract` output.  The arithmetic runs in csrc/dsp_extract.hip
(float64, numpy's evaluation order); the motif scan / coordinates / sampleinfo strings in csrc/dsp_sites.cpp.

Difference from the reference, by necessity: bases longer than signal_len are subsampled with a counter-based
sampler keyed by (seed, read uid, base index) instead of the unseeded process-global random.sample
(:247-249), so results are reproducible and independent of batching."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _native as nat
from . import textio
from .utils.process_utils import get_motif_seqs, parse_region_str

KEY_SEP = "||"  # extract_features.py:40


class ReadBatchC(ctypes.Structure):
    _fields_ = [("n_reads", ctypes.c_int64), ("n_samples", ctypes.c_int64), ("n_events", ctypes.c_int64),
                ("raw", ctypes.c_void_p), ("raw_off", ctypes.c_void_p), ("scaling", ctypes.c_void_p),
                ("offset", ctypes.c_void_p), ("ev_start", ctypes.c_void_p), ("ev_len", ctypes.c_void_p),
                ("ev_base", ctypes.c_void_p), ("ev_off", ctypes.c_void_p)]


class _Staged(object):
    """host half of one batch: pinned, concatenated read arrays + the sites (FeatureExtractor.stage)"""
    __slots__ = ("set", "host", "rows", "n", "R", "n_samples", "E")


class ExtractedBatch(object):
    """rows: textio.ParsedRows whose text/row_off/... address the sites' sampleinfo strings (host; its feature
    arrays are filled only by to_host()); kmer/means/stds/lens/signals: device tensors [n, L(, S)]; site_keys: int64 [n]
    on the device, read uid << 24 | base index of the site in its read -- a name for the site that depends on neither the
    batching nor the rank count (the in-kernel initial-state generator of the forward is keyed by it)."""
    __slots__ = ("rows", "n", "kmer", "means", "stds", "lens", "signals", "shift", "scale", "site_keys")

    def to_host(self, pinned=None):
        """fill rows.kmer / means / stds / lens / signals.  pinned: a dict that keeps page-locked host buffers between
        calls (grown as needed) -- the arrays are then views of those buffers and valid until the dict's next use: the
        copy runs at the link's rate (57 GB/s measured) instead of a pageable .cpu()'s 5-9 GB/s, which for the float64
        rows of `extract` (2.5 GB per 1.35 M rows) was the longest stage left"""
        r = self.rows
        names = ("kmer", "means", "stds", "lens", "signals")
        if pinned is None:
            for k in names:
                setattr(r, k, getattr(self, k).cpu().numpy())
            return r
        import torch
        views = {}
        for k in names:
            t = getattr(self, k)
            n = t.numel()
            buf = pinned.get(k)
            if buf is None or buf.numel() < n or buf.dtype != t.dtype:
                buf = torch.empty(max(n + n // 4, 1024), dtype=t.dtype, pin_memory=True)
                pinned[k] = buf
            views[k] = buf[:n].view(t.shape)
            views[k].copy_(t, non_blocking=True)
        torch.cuda.current_stream(self.signals.device).synchronize()
        for k in names:
            setattr(r, k, views[k].numpy())
        return r


def _read_position_file(position_file):
    """chrom \\t pos \\t strand per line -> set of "chrom||pos||strand" (extract_features.py:520-529)"""
    positions = set()
    with open(position_file, "r") as rf:
        for line in rf:
            words = line.strip().split("\t")
            if len(words) < 3:
                raise ValueError("--position file in wrong format. If you didn't use Tab as delimiter, Please do.")
            positions.add(KEY_SEP.join(words[:3]))
    return positions


class FeatureExtractor(object):
    def __init__(self, motifs="CG", mod_loc=0, seq_len=13, signal_len=16, normalize_method="mad", chrom2len=None,
                 positions=None, region=None, methy_label=1, is_dna=True, device=0, seed=0, round_stats=False, nthreads=8):
        if seq_len % 2 == 0:
            raise ValueError("kmer_len must be odd")  # extract_features.py:296-297
        if normalize_method not in ("mad", "zscore"):
            raise ValueError("")  # :185
        import torch
        self.torch = torch
        self.dev = torch.device("cuda", device) if not isinstance(device, torch.device) else device
        self.motif_seqs = get_motif_seqs(motifs, is_dna) if isinstance(motifs, str) else list(motifs)
        mlens = set(len(m) for m in self.motif_seqs)
        self.motif_len = len(self.motif_seqs[0])
        # the reference takes the length of an arbitrary set element (:107): mixed-length motif sets are undefined there
        if len(mlens) != 1:
            raise ValueError("--motifs must all have the same length")
        self.mod_loc, self.L, self.S = int(mod_loc), int(seq_len), int(signal_len)
        self.method = nat.NORM_MAD if normalize_method == "mad" else nat.NORM_ZSCORE
        self.chrom2len, self.positions = chrom2len, positions
        self.regioninfo = parse_region_str(region) if (region is None or isinstance(region, str)) else tuple(region)
        self.methy_label, self.seed, self.round_stats = int(methy_label), int(seed), bool(round_stats)
        self._motif_blob = "".join(self.motif_seqs).encode()
        self.nthreads = max(1, int(nthreads))
        import threading
        self.n_stage_sets = 4      # sets of pinned staging buffers (grown on demand), rotated between stage() calls
        self._free_sets, self._set_lock = None, threading.Lock()
        nat.lib()

    # -- host: which reads / region bounds (extract_features.py:311-314, :337-341)
    def _select_reads(self, reads):
        rg_chrom, rg_start, rg_end = self.regioninfo
        keep, lo, hi = [], [], []
        for rd in reads:
            if rg_chrom is not None and rg_chrom != rd.chrom:
                continue
            nbases = int(rd.ev_base.shape[0])
            a = rd.chrom_start if rg_start is None else rg_start
            b = rd.chrom_start + nbases if rg_end is None else rg_end
            if a >= rd.chrom_start + nbases or b <= rd.chrom_start:
                continue
            keep.append(rd)
            lo.append(a)
            hi.append(b)
        return keep, np.array(lo, np.int64), np.array(hi, np.int64)

    def _sites(self, reads, ev_base, ev_off, rg_lo, rg_hi):
        L = nat.lib()
        n = len(reads)
        strs = lambda xs: (ctypes.c_char_p * n)(*[x.encode() for x in xs])
        chrom, names = strs([r.chrom for r in reads]), strs([r.readname for r in reads])
        rstrand = "".join(r.strand[:1] or "t" for r in reads).encode()
        astrand = "".join(r.alignstrand[:1] or "+" for r in reads).encode()
        cstart = np.array([r.chrom_start for r in reads], np.int64)
        clen = np.array([(self.chrom2len.get(r.chrom, -1) if self.chrom2len is not None else -1) for r in reads], np.int64)
        use_rg = self.regioninfo[0] is not None
        p = textio._ptr
        args = [n, p(ev_base), p(ev_off), chrom, names, rstrand, astrand, p(cstart), p(clen),
                p(rg_lo) if use_rg else None, p(rg_hi) if use_rg else None, self._motif_blob, len(self.motif_seqs),
                self.motif_len, self.mod_loc, self.L]
        need = ctypes.c_size_t()
        cnt = nat.check(int(L.dsp_extract_sites(*args, 0, None, None, None, 0, ctypes.byref(need), None, None, None, None, self.nthreads)))
        site_read, site_loc = np.empty(cnt, np.int32), np.empty(cnt, np.int32)
        info = np.empty(max(need.value, 1), np.uint8)
        row_off, info_len = np.empty(cnt, np.uint64), np.empty(cnt, np.uint32)
        read_off, read_len = np.empty(cnt, np.uint32), np.empty(cnt, np.uint32)
        got = nat.check(int(L.dsp_extract_sites(*args, cnt, p(site_read), p(site_loc), p(info), info.nbytes, None,
                                                p(row_off), p(info_len), p(read_off), p(read_len), self.nthreads)))
        assert got == cnt
        if self.positions is not None and cnt:  # :356-357 (hash-set lookup on the host)
            keep = np.zeros(cnt, bool)
            for i in range(cnt):
                w = bytes(info[int(row_off[i]):int(row_off[i]) + int(info_len[i])]).decode().split("\t")
                keep[i] = KEY_SEP.join(w[:3]) in self.positions
            site_read, site_loc, row_off = site_read[keep], site_loc[keep], row_off[keep]
            info_len, read_off, read_len = info_len[keep], read_off[keep], read_len[keep]
        return site_read, site_loc, info, row_off, info_len, read_off, read_len

    # -- host half: everything that does not need the GPU.  Thread-safe: several threads may stage batches at once,
    #    each into its own set of pinned buffers (K sets rotate; a set is reused once the uploads of the batch that
    #    last used it have completed)
    def _take_set(self):
        with self._set_lock:
            if self._free_sets is None:
                import queue
                self._free_sets = queue.Queue()
                for _ in range(self.n_stage_sets):
                    self._free_sets.put({})
        st = self._free_sets.get()
        if st.get("_event") is not None:
            st["_event"].synchronize()  # the uploads of the batch that used this set have left its buffers
            st["_event"] = None
        return st

    def stage(self, reads, first_read_uid=0, read_uids=None):
        """reads: sequence of reads.ReadRecord -> _Staged: region-selected reads concatenated into pinned staging
        buffers, the motif sites and their sampleinfo strings (csrc/dsp_sites.cpp).  No GPU call."""
        torch = self.torch
        torch.cuda.set_device(self.dev)  # staging threads start on device 0: pinned allocations belong with this rank's GPU
        uid_of = {id(r): (read_uids[i] if read_uids is not None else first_read_uid + i) for i, r in enumerate(reads)}
        reads, rg_lo, rg_hi = self._select_reads(reads)
        R = len(reads)
        sg = _Staged()
        rows = textio.ParsedRows()
        rows.seq_len, rows.signal_len = self.L, self.S
        i64 = lambda xs: np.concatenate([[0], np.cumsum(xs)]).astype(np.int64)
        raw_off = i64([r.raw.shape[0] for r in reads])
        ev_off = i64([r.ev_base.shape[0] for r in reads])
        st = self._take_set()

        def pin(name, parts, dt):
            """concatenate host arrays into a pinned staging buffer of this set"""
            if isinstance(parts, np.ndarray):
                parts = [parts]
            total = int(sum(p.shape[0] for p in parts))
            buf = st.get(name)
            if buf is None or buf.shape[0] < total or buf.dtype != dt:
                buf = torch.empty(max(total + total // 2, 1024), dtype=dt, pin_memory=True)
                st[name] = buf
            view = buf[:total]
            if total:
                np.concatenate(parts, out=view.numpy(), casting="same_kind")
            return view
        h = {}
        h["ev_base"] = pin("ev_base", [r.ev_base for r in reads], torch.uint8)
        ev_base = h["ev_base"].numpy()
        site_read, site_loc, info, row_off, info_len, read_off, read_len = (
            self._sites(reads, ev_base, ev_off, rg_lo, rg_hi) if R else
            (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(1, np.uint8), np.zeros(0, np.uint64),
             np.zeros(0, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint32)))
        n = int(site_read.shape[0])
        rows.text, rows.n = info, n
        rows.row_off, rows.info_len, rows.read_off, rows.read_len = row_off, info_len, read_off, read_len
        rows.labels = np.full(n, self.methy_label, np.int32)
        h["raw"] = pin("raw", [r.raw for r in reads], torch.int16)
        h["raw_off"], h["ev_off"] = pin("raw_off", raw_off, torch.int64), pin("ev_off", ev_off, torch.int64)
        h["scaling"] = pin("scaling", np.array([r.scaling for r in reads], np.float64), torch.float64)
        h["offset"] = pin("offset", np.array([r.offset for r in reads], np.float64), torch.float64)
        h["ev_start"] = pin("ev_start", [r.ev_start for r in reads], torch.int64)
        h["ev_len"] = pin("ev_len", [r.ev_len for r in reads], torch.int64)
        h["site_read"], h["site_loc"] = pin("site_read", site_read, torch.int32), pin("site_loc", site_loc, torch.int32)
        h["uid"] = pin("uid", np.array([uid_of[id(r)] for r in reads], np.uint64).view(np.int64), torch.int64)
        sg.set, sg.host, sg.rows, sg.n, sg.R = st, h, rows, n, R
        sg.n_samples, sg.E = int(raw_off[-1]), int(ev_off[-1])
        return sg

    # -- device half: uploads + the three kernels, asynchronous on `stream`
    def launch(self, sg, stream=None, f64=False):
        """_Staged -> ExtractedBatch (device tensors in dsp_forward's layout).  f64=True: means / stds / signals as float64
        (what the feature TSV prints) instead of the float32 the model eats."""
        torch = self.torch
        dev = self.dev
        out = ExtractedBatch()
        out.rows, out.n = sg.rows, sg.n
        n, R, E = sg.n, sg.R, sg.E
        st = stream if stream is not None else torch.cuda.current_stream(dev)
        with torch.cuda.device(dev), torch.cuda.stream(st):
            d = {k: v.to(dev, non_blocking=True) for k, v in sg.host.items()}
            batch = ReadBatchC(R, sg.n_samples, E, d["raw"].data_ptr(), d["raw_off"].data_ptr(), d["scaling"].data_ptr(),
                               d["offset"].data_ptr(), d["ev_start"].data_ptr(), d["ev_len"].data_ptr(), d["ev_base"].data_ptr(),
                               d["ev_off"].data_ptr())
            dbl = dict(dtype=torch.float64, device=dev)
            shift, scale = torch.empty(R, **dbl), torch.empty(R, **dbl)
            base_mean, base_std = torch.empty(E, **dbl), torch.empty(E, **dbl)
            base_len = torch.empty(E, dtype=torch.int32, device=dev)
            base_lo = torch.empty(E, dtype=torch.int64, device=dev)
            blk_off = torch.empty(R + 1, dtype=torch.int64, device=dev)
            out.kmer = torch.empty((n, self.L), dtype=torch.uint8, device=dev)
            fdt = torch.float64 if f64 else torch.float32
            out.means = torch.empty((n, self.L), dtype=fdt, device=dev)
            out.stds = torch.empty((n, self.L), dtype=fdt, device=dev)
            out.lens = torch.empty((n, self.L), dtype=torch.int32, device=dev)
            out.signals = torch.empty((n, self.L, self.S), dtype=fdt, device=dev)
            out.shift, out.scale = shift, scale
            out.site_keys = (d["uid"][d["site_read"].long()] << 24) | d["site_loc"].long() if n else \
                torch.empty(0, dtype=torch.int64, device=dev)
            if R:
                L = nat.lib()
                sp = ctypes.c_void_p(st.cuda_stream)
                ptr = lambda t: ctypes.c_void_p(t.data_ptr())
                nat.check(L.dsp_extract_normalize(sp, ctypes.byref(batch), self.method, ptr(shift), ptr(scale)))
                nat.check(L.dsp_extract_base_stats(sp, ctypes.byref(batch), ptr(shift), ptr(scale), ptr(blk_off), ptr(base_mean),
                                                   ptr(base_std), ptr(base_len), ptr(base_lo)))
                if n:
                    gather = L.dsp_extract_gather_f64 if f64 else L.dsp_extract_gather
                    nat.check(gather(sp, ctypes.byref(batch), ptr(shift), ptr(scale), ptr(base_mean),
                                     ptr(base_std), ptr(base_len), ptr(base_lo), n, ptr(d["site_read"]),
                                     ptr(d["site_loc"]), self.L, self.S, int(self.round_stats),
                                     ctypes.c_uint64(self.seed & ((1 << 64) - 1)), ptr(d["uid"]),
                                     ptr(out.kmer), ptr(out.means), ptr(out.stds), ptr(out.lens), ptr(out.signals)))
                # the inputs must outlive the asynchronous kernels
                for t in list(d.values()) + [base_mean, base_std, base_len, base_lo, blk_off]:
                    t.record_stream(st)
            ev = torch.cuda.Event()
            ev.record(st)
            sg.set["_event"] = ev
        self._free_sets.put(sg.set)  # reusable once `ev` has passed (checked by the next taker)
        sg.set = sg.host = None
        return out

    def discard(self, sg):
        """give a staged batch's buffers back without launching it"""
        if sg is not None and sg.set is not None:
            self._free_sets.put(sg.set)
            sg.set = sg.host = None

    def extract(self, reads, first_read_uid=0, stream=None, read_uids=None, f64=False):
        """reads: sequence of reads.ReadRecord -> ExtractedBatch (sites in read order, then position order).
        read_uids (default first_read_uid + index): 64-bit keys of the subsampler, one per read."""
        return self.launch(self.stage(reads, first_read_uid, read_uids), stream=stream, f64=f64)

    def extract_stream(self, batches, workers=4, stream=None, f64=False):
        """batches: iterable of (reads, read_uids or None) -> ExtractedBatch per batch, in order.  The host half of up to
        `n_stage_sets` batches is staged ahead by `workers` threads while the device half of earlier batches runs."""
        from concurrent.futures import ThreadPoolExecutor
        import collections
        pending = collections.deque()
        with ThreadPoolExecutor(max(1, workers)) as pool:
            for reads, uids in batches:
                pending.append(pool.submit(self.stage, reads, 0, uids))
                # never hold more staged batches than there are buffer sets: the next stage() would wait for a launch
                while len(pending) >= max(1, self.n_stage_sets):
                    yield self.launch(pending.popleft().result(), stream=stream, f64=f64)
            while pending:
                yield self.launch(pending.popleft().result(), stream=stream, f64=f64)


# ---- `deepsignal_plant extract` (extract_features.py:589-651, :654-767) ---------------------------------------------
def extract_features(args):
    """Reads in (a directory of *.fast5 / *.reads.npz), feature rows out: the reference's `extract`.
    Output: the feature TSV (plain / --gzip / --w_is_dir batches, byte-compatible with _features_to_str), or the
    binary container when --write_path ends with .dspf (what call_mods reads fastest)."""
    import os
    import queue
    import sys
    import threading
    import time
    import torch
    from . import featfile
    from . import reads as dsp_reads
    from .utils.process_utils import get_contig2len, str2bool
    print("[main] extract_features starts..")
    start = time.time()
    if not os.path.isdir(args.fast5_dir):
        raise ValueError("--fast5_dir is not a directory!")  # :598-599
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible: this build has no CPU path")
    files = dsp_reads.list_read_files(os.path.abspath(args.fast5_dir), str2bool(args.recursively))
    print("%d read files in total.." % len(files))
    chrom2len = get_contig2len(args.reference_path) if args.reference_path else None
    positions = _read_position_file(args.positions) if args.positions else None
    nthreads = min(max(1, args.nproc), os.cpu_count() or 1)
    to_dspf = args.write_path.endswith(".dspf")
    fx = FeatureExtractor(motifs=args.motifs, mod_loc=args.mod_loc, seq_len=args.seq_len, signal_len=args.signal_len,
                          normalize_method=args.normalize_method, chrom2len=chrom2len, positions=positions,
                          region=args.region, methy_label=args.methy_label, is_dna=str2bool(args.is_dna), device=0,
                          seed=getattr(args, "seed", 0), round_stats=True, nthreads=nthreads)
    batches = dsp_reads.ReadBatches(files, max(1, int(args.f5_batch_size)) * 8, args.corrected_group,
                                    args.basecall_subgroup, workers=min(8, nthreads), only_chrom=fx.regioninfo[0],
                                    procs=int(os.environ.get("DSP_READER_PROCS", "0")))  # decoding scales on the loader threads; reader processes are opt-in
    rq = queue.Queue(maxsize=3)

    def load():
        try:
            for item in batches:
                rq.put(fx.stage(item[0], read_uids=item[1]))  # host half here, GPU half in the main thread
        finally:
            rq.put(None)
    threading.Thread(target=load, daemon=True).start()

    is_dir, is_gzip = str2bool(args.w_is_dir), args.gzip
    n_rows = 0
    writer = None
    # three stages behind the loader thread: this thread runs the GPU half and brings the rows to the host, `sink`
    # formats them (nthreads threads inside the library) and the writer's own thread appends the text to the file
    # (plain: gzio.BackgroundFileWriter; --gzip: BgzfWriter deflates there as well)
    sq = queue.Queue(maxsize=2)
    sink_error = []
    host_sets = queue.Queue()   # page-locked landing buffers of the rows, rotating between this thread and the sink
    for _ in range(3):
        host_sets.put({})
    stage_s = {"wait for reads": 0.0, "gpu + to host": 0.0, "hand over": 0.0, "format": 0.0, "write (wait)": 0.0}  # DSP_TIMING=1
    if to_dspf:
        writer = featfile.FeatureFileWriter(args.write_path, args.seq_len, args.signal_len)
    else:
        from . import gzio
        op = lambda p: gzio.open_write(p, is_gzip, nthreads=nthreads, background=True)  # --gzip: BGZF members (seekable by call_mods ranks)
        if is_dir:  # :474-510
            if os.path.isfile(args.write_path):
                raise FileExistsError("{} already exists as a file, please use another write_dir".format(args.write_path))
            os.makedirs(args.write_path, exist_ok=True)
            first_path = os.path.join(args.write_path, "0.tsv" + (".gz" if is_gzip else ""))
        else:  # :451-471
            first_path = args.write_path + (".gz" if is_gzip and not args.write_path.endswith(".gz") else "")

        bufs = queue.Queue()
        for _ in range(3):
            bufs.put(None)

        def sink():
            wf = None
            file_count, batch_count = 0, 0
            try:
                wf = op(first_path)
                while True:
                    item = sq.get()
                    if item is None:
                        break
                    rows, hs = item
                    if is_dir and batch_count >= args.w_batch_num:
                        wf.close()
                        file_count += 1
                        batch_count = 0
                        wf = op(os.path.join(args.write_path, "%d.tsv%s" % (file_count, ".gz" if is_gzip else "")))
                    t0 = time.time()
                    # the formatting threads' parts go to the writer as they are, out of three rotating buffers: no
                    # compaction, no fresh pages after the first batches (plain: appended; --gzip: deflated part by part)
                    buf = bufs.get()
                    parts, buf = textio.format_feature_rows_parts(rows, rows.means, rows.stds, rows.signals,
                                                                  nthreads=nthreads, out=buf)
                    t1 = time.time()
                    host_sets.put(hs)   # the rows are text now
                    wf.write(parts, on_done=lambda b=buf: bufs.put(b))
                    stage_s["format"] += t1 - t0
                    stage_s["write (wait)"] += time.time() - t1
                    batch_count += 1
            except BaseException as e:
                sink_error.append(e)
                # a dead sink must not strand the producer: give it landing buffers and text buffers for as long as it
                # keeps handing over rows (it stops at its next look at sink_error), and empty the queue
                for _ in range(4):
                    host_sets.put({})
                    bufs.put(None)
                while sq.get() is not None:
                    host_sets.put({})
            finally:
                if wf is not None:
                    try:
                        wf.close()
                    except BaseException as e:
                        if not sink_error:
                            sink_error.append(e)
        sink_thread = threading.Thread(target=sink, daemon=True)
        sink_thread.start()
    try:
        while True:
            t0 = time.time()
            item = rq.get()
            t1 = time.time()
            stage_s["wait for reads"] += t1 - t0
            if item is None:
                break
            if sink_error:
                break
            out = fx.launch(item, f64=not to_dspf)
            if out.n == 0:
                continue
            hs = host_sets.get()
            rows = out.to_host(hs)
            t2 = time.time()
            stage_s["gpu + to host"] += t2 - t1
            if to_dspf:
                writer.add(rows)
                host_sets.put(hs)
            else:
                sq.put((rows, hs))
            stage_s["hand over"] += time.time() - t2
            n_rows += out.n
    finally:
        if writer is not None:
            writer.close()
        if not to_dspf:
            sq.put(None)
            sink_thread.join()
    if sink_error:
        raise sink_error[0]
    if os.environ.get("DSP_TIMING"):
        print("[extract] seconds per stage: " + ", ".join("%s %.2f" % kv for kv in stage_s.items()), file=sys.stderr)
    print("%d of %d read files failed.." % (batches.failed, len(files)))
    print("[main] extract_features costs %.1f seconds.. (%d feature rows)" % (time.time() - start, n_rows))
    return n_rows


def add_extract_args(p):
    """The reference's `extract` flags (extract_features.py:654-745) + --seed of the subsampler."""
    g = p.add_argument_group("INPUT")
    g.add_argument("--fast5_dir", "-i", type=str, required=True, help="directory of read files (*.fast5 / *.reads.npz)")
    g.add_argument("--recursively", "-r", type=str, default="yes")
    g.add_argument("--corrected_group", type=str, default="RawGenomeCorrected_000")
    g.add_argument("--basecall_subgroup", type=str, default="BaseCalled_template")
    g.add_argument("--is_dna", type=str, default="yes")
    g.add_argument("--reference_path", type=str, default=None, help="reference .fa (only its contig lengths are used)")
    g = p.add_argument_group("EXTRACTION")
    g.add_argument("--normalize_method", type=str, choices=["mad", "zscore"], default="mad")
    g.add_argument("--methy_label", type=int, choices=[1, 0], default=1)
    g.add_argument("--seq_len", type=int, default=13)
    g.add_argument("--signal_len", type=int, default=16)
    g.add_argument("--motifs", type=str, default="CG")
    g.add_argument("--mod_loc", type=int, default=0)
    g.add_argument("--region", type=str, default=None)
    g.add_argument("--positions", type=str, default=None)
    g.add_argument("--seed", type=int, default=0, help="seed of the deterministic subsampler of bases longer than --signal_len")
    g = p.add_argument_group("OUTPUT")
    g.add_argument("--write_path", "-o", type=str, required=True, help="feature file to write (.dspf = binary container)")
    g.add_argument("--w_is_dir", type=str, default="no")
    g.add_argument("--w_batch_num", type=int, default=200)
    g.add_argument("--gzip", action="store_true", default=False)
    p.add_argument("--nproc", "-p", type=int, default=10, help="host threads")
    p.add_argument("--f5_batch_size", type=int, default=30)
    return p
