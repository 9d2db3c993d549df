#! /usr/bin/env python
"""calculate modification frequency at genome level -- mirror of deepsignal_plant/call_mods_freq.py.

Same entry point ``call_mods_frequency_to_file(args)`` and flags (--input_path (append), --file_uid,
--result_file, --contigs, --nproc, --bed, --sort, --prob_cf, --gzip), same output bytes.  The per-line Python
of calculate_mods_frequency / write_sitekey2stats (call_mods_freq.py:29-122, utils/txt_formater.py:8-46) is
replaced by the native aggregator in libdsp_amd.so (csrc/dsp_freq.cpp); `call_mods --freq_file` feeds the
same aggregator straight from the GPU results without writing / re-reading the per-read file."""
from __future__ import absolute_import

import argparse
import ctypes
import gzip
import os
import sys
import time

import numpy as np

from . import _native as nat


class SiteFrequency(object):
    """Streaming per-site aggregator (sites keyed by chromosome + pos, txt_formater.py:12)."""

    def __init__(self, prob_cf=0.5, nthreads=None):
        self._h = ctypes.c_void_p(nat.lib().dsp_freq_create(float(prob_cf)))
        if not self._h:
            raise MemoryError("dsp_freq_create failed")
        if nthreads is None:
            import os
            nthreads = min(16, os.cpu_count() or 1)
        nat.lib().dsp_freq_set_threads(self._h, int(nthreads))

    def __del__(self):
        try:
            if self._h:
                nat.lib().dsp_freq_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def add_calls_text(self, data, contig=None):
        arr = np.frombuffer(memoryview(data), dtype=np.uint8)
        c = contig.encode() if contig is not None else None
        return nat.check(int(nat.lib().dsp_freq_add_calls_text(self._h, ctypes.c_void_p(arr.ctypes.data), arr.nbytes, c)))

    def add_calls_file(self, path, contig=None, chunk_bytes=64 << 20):
        """a per-read call file, plain or .gz (call_mods_freq.py:46-49 opens the latter with gzip.open; here BGZF members
        are inflated in parallel and a single gzip stream goes through the parallel inflater, gzio.read_text_chunks)"""
        def chunks():
            if path.endswith(".gz"):
                from . import gzio
                for arr in gzio.read_text_chunks(path, chunk_bytes, nthreads=min(16, os.cpu_count() or 1)):
                    yield arr.tobytes()
            else:
                with open(path, "rb") as f:
                    while True:
                        chunk = f.read(chunk_bytes)
                        if not chunk:
                            return
                        yield chunk
        carry = b""
        for chunk in chunks():
            data = carry + chunk
            nl = data.rfind(b"\n")
            if nl < 0:
                carry = data
                continue
            carry = data[nl + 1:]
            self.add_calls_text(data[:nl + 1], contig)
        if carry.strip():
            self.add_calls_text(carry, contig)

    def add_block(self, rows, probs, labels, start=0, stop=None):
        """parsed call_mods block + GPU results -> aggregator (the fused path)"""
        stop = rows.n if stop is None else stop
        n = stop - start
        if n <= 0:
            return 0
        probs = np.ascontiguousarray(probs, np.float32)
        labels = np.ascontiguousarray(labels, np.uint8)
        text = rows.text if isinstance(rows.text, np.ndarray) else np.frombuffer(memoryview(rows.text), np.uint8)
        p = ctypes.c_void_p
        return nat.check(int(nat.lib().dsp_freq_add_block(
            self._h, p(text.ctypes.data), p(rows.row_off[start:stop].ctypes.data), p(rows.info_len[start:stop].ctypes.data),
            p(probs.ctypes.data), probs.shape[1], p(labels.ctypes.data), p(rows.kmer[start:stop].ctypes.data), rows.seq_len, n)))

    def counts(self):
        c, u, s = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        nat.lib().dsp_freq_counts(self._h, ctypes.byref(c), ctypes.byref(u), ctypes.byref(s))
        return c.value, u.value, s.value

    def format(self, is_sort=False, is_bed=False):
        L = nat.lib()
        need = nat.check(int(L.dsp_freq_format(self._h, int(is_sort), int(is_bed), None, 0)))
        buf = np.empty(max(need, 1), np.uint8)
        nat.check(int(L.dsp_freq_format(self._h, int(is_sort), int(is_bed), ctypes.c_void_p(buf.ctypes.data), need)))
        return buf[:need].tobytes()

    def write(self, result_file, is_sort, is_bed, is_gzip):
        """write_sitekey2stats, call_mods_freq.py:77-122"""
        if is_gzip and not result_file.endswith(".gz"):
            result_file += ".gz"
        data = self.format(is_sort, is_bed)
        from . import gzio
        with gzio.open_write(result_file, is_gzip) as wf:
            wf.write(data)
        return result_file


class DeviceSiteFrequency(object):
    """`call_freq` fed from HBM and sharded over ranks (include/dsp_amd.h, "call_freq on the device").

    add_block() keeps every per-read call of this rank on its GPU as a 32-byte record (site key, packed printed
    probabilities + label + first-record metadata, pos_in_strand, global row index).  finish() then
      1. agrees on global chromosome ids (an all_gather of the name lists),
      2. deals the records to ranks by a hash of the site key -- ONE all_to_all over RCCL/xGMI, the only exchange
         step of the whole call_mods path; chunks arrive in rank order = file order,
      3. sorts them by site key with a stable sort and reduces every site on the device with the reference's own
         sequence of double additions (csrc/dsp_freq_dev.hip),
      4. gathers the per-site results (72 B per site) to rank 0, which prints them with the host formatter
         (insertion order = global row of the site's first used record, or sorted).
    Bytes are identical to `call_freq` run on the merged per-read file (tests/test_gpu_cli.py, tests/test_gpu_freq.py)."""

    UNUSED = 0x7fffffffffffffff

    def __init__(self, prob_cf, device, nthreads=None):
        import torch
        self.torch = torch
        self.dev = torch.device(device)
        self.prob_cf = float(prob_cf)
        self.host = SiteFrequency(prob_cf, nthreads)   # chromosome dictionary + final table / formatter
        self.keys, self.packed, self.pis, self.rows = [], [], [], []
        self.count = 0
        # Records stay resident (32 B per call) and finish() needs about five times that in temporaries (filter, exchange,
        # two-key sort): past the budget the aggregator stops collecting, says so, and the caller computes --freq_file from
        # the merged per-read file with the host table instead (same bytes) -- rather than running out of HBM after all the
        # forward work is done.  DSP_FREQ_DEV_MAX_BYTES overrides the budget (default: a tenth of the GPU's memory).
        env = os.environ.get("DSP_FREQ_DEV_MAX_BYTES")
        self.budget = int(env) if env else int(torch.cuda.get_device_properties(self.dev).total_memory // 10)
        self.overflow = False
        # finish() is the first user of a handful of torch operators (cat, masked indexing, the stable sort, gathers): the
        # HIP runtime loads their code objects on first use, 0.25 s measured -- after the last forward, when nothing can
        # hide it.  A daemon thread runs them once on 2 M elements (a millisecond of GPU time) while the forwards keep the GPU busy.
        import threading
        self._warm = threading.Thread(target=self._warm_up, daemon=True)
        self._warm.start()

    def _warm_up(self):
        torch = self.torch
        try:
            with torch.cuda.stream(torch.cuda.Stream(self.dev)):
                a = torch.arange(1 << 21, dtype=torch.int64, device=self.dev)   # large enough for the large-input sort kernels
                b = torch.cat([a, a])
                b = b[b != 3]
                k, perm = torch.sort(b, stable=True)   # (finish() sorts its records natively; rank 0's final ordering of the sites uses this)
                o = torch.sort(k[perm], stable=True)[1]
                c = torch.zeros(1, dtype=torch.int64, device=self.dev)
                d = torch.empty(4, dtype=torch.float64, device=self.dev).view(torch.int64)
                _ = (k[o] + c).cpu(), d.numel(), (a << 40) | (a & 7), ((a * -7046029254386353131) >> 24) % 3
                torch.cuda.current_stream(self.dev).synchronize()
        except Exception:   # a warm-up must never be the reason a run fails
            pass

    def add_block(self, rows, probs_dev, labels_dev, first_row, start=0, stop=None, stream=None):
        """rows: the parsed block (host); probs_dev [n, C] float32 and labels_dev [n] uint8: the forward's outputs,
        still on the GPU; first_row: global index of rows[start]."""
        torch = self.torch
        stop = rows.n if stop is None else stop
        n = stop - start
        if n <= 0:
            return 0
        if self.overflow:
            self.count += n
            return n
        if (self.count + n) * 32 > self.budget:
            sys.stderr.write("[call_freq] %d calls would exceed the device budget of %d bytes for resident records: "
                             "--freq_file will be computed from the per-read file by the host table\n" % (self.count + n, self.budget))
            self.overflow = True
            self.keys, self.packed, self.pis, self.rows = [], [], [], []
            self.count += n
            return n
        key = np.empty(n, np.int64)
        pis = np.empty(n, np.int64)
        meta = np.empty(n, np.uint32)
        text = rows.text if isinstance(rows.text, np.ndarray) else np.frombuffer(memoryview(rows.text), np.uint8)
        p = ctypes.c_void_p
        L = nat.lib()
        nat.check(int(L.dsp_freq_block_keys(self.host._h, p(text.ctypes.data), p(rows.row_off[start:stop].ctypes.data),
                                            p(rows.info_len[start:stop].ctypes.data), p(rows.kmer[start:stop].ctypes.data),
                                            rows.seq_len, n, p(key.ctypes.data), p(pis.ctypes.data), p(meta.ctypes.data))))
        s = stream if stream is not None else torch.cuda.current_stream(self.dev)
        # (this may run in a writer thread, whose current device is 0 whatever the rank: select ours for the launches)
        with torch.cuda.device(self.dev), torch.cuda.stream(s):
            if not torch.is_tensor(probs_dev):   # host copies (the reads branch feeds from its writer thread): 9 B per row up
                probs_dev = torch.from_numpy(np.ascontiguousarray(probs_dev[start:stop], np.float32)).to(self.dev)
                labels_dev = torch.from_numpy(np.ascontiguousarray(labels_dev[start:stop], np.uint8)).to(self.dev)
            key_d = torch.from_numpy(key).to(self.dev, non_blocking=True)
            meta_d = torch.from_numpy(meta.view(np.int32)).to(self.dev, non_blocking=True)
            pis_d = torch.from_numpy(pis).to(self.dev, non_blocking=True)
            key_o = torch.empty(n, dtype=torch.int64, device=self.dev)
            packed = torch.empty(n, dtype=torch.int64, device=self.dev)
            probs_dev = probs_dev.contiguous()
            nat.check(int(L.dsp_freq_dev_encode(p(s.cuda_stream), n, p(probs_dev.data_ptr()), int(probs_dev.shape[1]),
                                                p(labels_dev.data_ptr()), p(key_d.data_ptr()), p(meta_d.data_ptr()),
                                                self.prob_cf, p(key_o.data_ptr()), p(packed.data_ptr()))))
            row = torch.arange(first_row, first_row + n, dtype=torch.int64, device=self.dev)
        for t in (key_d, meta_d, probs_dev, labels_dev):
            t.record_stream(s)
        self.keys.append(key_o); self.packed.append(packed); self.pis.append(pis_d); self.rows.append(row)
        self.count += n
        self._keep = (key, meta, pis)  # pageable sources of the async copies stay alive until the next block
        return n

    # ---- collectives (RCCL on GPUs; gloo when ranks share a GPU on the dev box) -------------------------------
    def _chrom_names(self):
        L = nat.lib()
        out = []
        for i in range(L.dsp_freq_chrom_count(self.host._h)):
            k = int(L.dsp_freq_chrom_name(self.host._h, i, None, 0))
            buf = ctypes.create_string_buffer(max(k, 1))
            L.dsp_freq_chrom_name(self.host._h, i, buf, k)
            out.append(buf.raw[:k])
        return out

    def _exchange(self, cols, dest, world, dist=None):
        """deal the records (equally long int64 columns) to ranks by `dest`: dist.exchange_records -- ONE ragged
        all_to_all_single on either backend; what arrives is ordered by source rank, then source order"""
        from . import dist as dsp_dist
        return dsp_dist.exchange_records(cols, dest, world, self.dev)

    def can_finish(self, world=1):
        """collective: True when every rank kept all its records and has room for finish()'s temporaries"""
        torch = self.torch
        self._warm.join()
        ok = not self.overflow
        if ok and self.dev.type == "cuda":
            free, _total = torch.cuda.mem_get_info(self.dev)
            ok = self.count * 32 * 5 <= free
        from . import dist as dsp_dist
        if dsp_dist.collective(world):
            ok = bool(dsp_dist.all_reduce_int(1 if ok else 0, world, "min", self.dev))
        return ok

    def finish(self, rank=0, world=1):
        """-> the SiteFrequency holding every site on rank 0 (None on the other ranks)"""
        torch = self.torch
        L = nat.lib()
        p = ctypes.c_void_p
        dev = self.dev
        self._warm.join()
        marks = [("start", time.time())] if os.environ.get("DSP_TIMING") else None

        def mark(label):
            if marks is not None:
                torch.cuda.synchronize(dev)
                marks.append((label, time.time()))
        cat = lambda xs, dt: torch.cat(xs) if xs else torch.empty(0, dtype=dt, device=dev)
        key, packed = cat(self.keys, torch.int64), cat(self.packed, torch.int64)
        pis, row = cat(self.pis, torch.int64), cat(self.rows, torch.int64)
        self.keys, self.packed, self.pis, self.rows = [], [], [], []
        live = key != self.UNUSED
        key, packed, pis, row = key[live], packed[live], pis[live], row[live]
        names = self._chrom_names()
        total = self.count
        mark("records gathered and filtered")
        from . import dist as dsp_dist
        multi = dsp_dist.collective(world)   # several ranks, or a forced one-rank RCCL group (DSP_FORCE_DIST=1)
        if multi:
            lists = dsp_dist.all_gather_json(list(names), world, dev)   # (chromosome names: json through dist.comm_device)
            glob, ids = [], {}
            for lst in lists:
                for nm in lst:
                    if nm not in ids:
                        ids[nm] = len(glob)
                        glob.append(nm)
            if len(glob) >= (1 << 22):
                raise ValueError("more than 2^22 chromosomes: use the host aggregator (--freq_on host)")
            remap = torch.tensor([ids[nm] for nm in names] or [0], dtype=torch.int64, device=dev)
            key = (remap[key >> 40] << 40) | (key & ((1 << 40) - 1))
            mixed = (key * -7046029254386353131) >> 24            # 0x9E3779B97F4A7C15 as int64; wraps like uint64
            dest = (mixed & 0xffffff) % world
            key, packed, pis, row = self._exchange([key, packed, pis, row], dest, world)
            total = dsp_dist.all_reduce_int(total, world, "sum", dev)
            names = glob
        n = int(key.numel())
        torch.cuda.set_device(dev)
        s = torch.cuda.current_stream(dev)
        def sort_records(by, a, b, c):
            """the four columns stably sorted by `by` (csrc/dsp_freq_dev.hip: rocPRIM radix sort of (key, index) + gather)"""
            outs = [torch.empty_like(by) for _ in range(4)]
            need = ctypes.c_size_t(0)
            args = [p(s.cuda_stream), n] + [p(t.data_ptr()) for t in (by, a, b, c)] + [p(t.data_ptr()) for t in outs]
            nat.check(int(L.dsp_freq_dev_sort_records(*args, None, ctypes.byref(need))))
            tmp = torch.empty(max(need.value, 1), dtype=torch.uint8, device=dev)
            nat.check(int(L.dsp_freq_dev_sort_records(*args, p(tmp.data_ptr()), ctypes.byref(need))))
            tmp.record_stream(s)
            return outs
        if multi:
            # what arrived is ordered by source rank; ranks may own interleaved blocks of the input (a foreign .gz is dealt
            # block i -> rank i % world), so restore the global input order first: a site's sums are then taken in file
            # order -- the order `call_freq` sees on the merged per-read file -- whatever the sharding
            row, key, packed, pis = sort_records(row, key, packed, pis)
        key, packed, pis, row = sort_records(key, packed, pis, row)   # file order inside a site survives
        mark("exchanged and sorted")
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        nat.check(int(L.dsp_freq_dev_count_sites(p(s.cuda_stream), n, p(key.data_ptr()), p(cnt.data_ptr()))))
        ns = int(cnt.item())
        out_i = [torch.empty(ns, dtype=torch.int64, device=dev) for _ in range(6)]  # key, first_row, packed, pis, met, cov
        out_d = [torch.empty(ns, dtype=torch.float64, device=dev) for _ in range(2)]
        nat.check(int(L.dsp_freq_dev_reduce(p(s.cuda_stream), n, p(key.data_ptr()), p(packed.data_ptr()), p(pis.data_ptr()),
                                            p(row.data_ptr()), p(cnt.data_ptr()), ns, p(out_i[0].data_ptr()),
                                            p(out_i[1].data_ptr()), p(out_i[2].data_ptr()), p(out_i[3].data_ptr()),
                                            p(out_d[0].data_ptr()), p(out_d[1].data_ptr()), p(out_i[4].data_ptr()),
                                            p(out_i[5].data_ptr()))))
        cols = out_i + [o.view(torch.int64) for o in out_d]   # doubles travel as their bit patterns
        mark("sites reduced")
        if multi:
            gathered = dsp_dist.gather_columns(cols, world, dev)
            if rank != 0:
                return None
            cols = gathered
        order = torch.sort(cols[1], stable=True)[1]   # by first row: deterministic whatever order the slots were taken in
        host = [np.ascontiguousarray(c[order].cpu().numpy()) for c in cols]
        mark("sites on the host")
        table = SiteFrequency(self.prob_cf)
        for nm in names:
            L.dsp_freq_intern_chrom(table._h, nm, len(nm))
        m = len(host[0])
        nat.check(int(L.dsp_freq_add_sites(table._h, m, p(host[0].ctypes.data), p(host[1].ctypes.data), p(host[2].ctypes.data),
                                           p(host[3].ctypes.data), p(host[6].view(np.float64).ctypes.data),
                                           p(host[7].view(np.float64).ctypes.data), p(host[4].ctypes.data),
                                           p(host[5].ctypes.data))))
        L.dsp_freq_add_counts(table._h, total)
        if marks is not None and rank == 0:
            mark("host table filled")
            print("[call_freq on the device] seconds: " + ", ".join("%s %.3f" % (b[0], b[1] - a[1]) for a, b in zip(marks, marks[1:])),
                  file=sys.stderr)
        return table


def _contig_names(spec):
    """--contigs: genome fasta, a file of names, or a comma list (call_mods_freq.py:253-263).  A file is a fasta when it is
    named .fa / .fasta / .fna or when ANY of its lines starts with '>' (_is_file_a_genome_fasta, :142-149, only skips the
    lines that do not); fasta names keep the file's order (and duplicates), the first word of the header; a names file is
    sorted(set(lines)) -- comment and blank lines included, as contigs no call has."""
    if os.path.isfile(spec):
        lines = open(spec, "r").read().splitlines()
        if spec.endswith((".fa", ".fasta", ".fna")) or any(l.startswith(">") for l in lines):
            return [l.strip()[1:].split(' ')[0] for l in lines if l.startswith(">")]
        return sorted(set(lines))
    return sorted(set(spec.strip().split(",")))


def call_mods_frequency_to_file(args):
    print("[main]call_freq starts..")
    start = time.time()
    mods_files = []
    for ipath in args.input_path:
        input_path = os.path.abspath(ipath)
        if os.path.isdir(input_path):
            for ifile in os.listdir(input_path):
                if args.file_uid is None or ifile.find(args.file_uid) != -1:
                    mods_files.append('/'.join([input_path, ifile]))
        elif os.path.isfile(input_path):
            mods_files.append(input_path)
        else:
            raise ValueError("--input_path is not a file or a directory!")
    print("get {} input file(s)..".format(len(mods_files)))

    contigs = _contig_names(args.contigs) if args.contigs is not None else None
    if contigs is None:
        print("read the input files..")
        agg = SiteFrequency(args.prob_cf)
        for f in mods_files:
            agg.add_calls_file(f)
        count, used, _ = agg.counts()
        print("{:.2f}% ({} of {}) calls used..".format(used / float(max(count, 1)) * 100, used, count))
        print("write the result..")
        agg.write(args.result_file, args.sort, args.bed, args.gzip)
    else:
        # the reference processes one contig per subprocess and concatenates the per-contig results in the
        # order of their (contig-named) temporary files, call_mods_freq.py:264-311
        print("start processing {} contigs..".format(len(contigs)))
        chunks = []
        for contig in sorted(set(contigs), key=lambda c: c + "."):
            agg = SiteFrequency(args.prob_cf)
            for f in mods_files:
                agg.add_calls_file(f, contig)
            count, used, nsites = agg.counts()
            if count == 0:
                print("contig-{} -- the input file is empty..".format(contig))
                continue
            print("{:.2f}% ({} of {}) calls used for {}..".format(used / float(count) * 100, used, count, contig))
            chunks.append(agg.format(args.sort, args.bed))
        result_file = args.result_file
        if args.gzip and not result_file.endswith(".gz"):
            result_file += ".gz"
        with (gzip.open(result_file, "wb") if args.gzip else open(result_file, "wb")) as wf:
            for c in chunks:
                wf.write(c)
    print("[main]call_freq costs %.1f seconds.." % (time.time() - start))


def add_call_freq_args(p):
    g = p.add_argument_group("INPUT")
    g.add_argument('--input_path', '-i', action="append", type=str, required=True,
                   help="a per-read call file written by call_mods, or a directory of them; may be given several times")
    g.add_argument('--file_uid', type=str, default=None,
                   help="substring that input files in an input directory must contain")
    g = p.add_argument_group("OUTPUT")
    g.add_argument('--result_file', '-o', type=str, required=True, help="the file path to save the result")
    g.add_argument('--bed', action='store_true', default=False, help="save the result in bedMethyl format")
    g.add_argument('--sort', action='store_true', default=False, help="sort items in the result")
    g.add_argument("--gzip", action="store_true", default=False, help="gzip the output")
    g = p.add_argument_group("CALCULATE")
    g.add_argument('--prob_cf', type=float, default=0.5,
                   help="use a call only if abs(prob1-prob0) >= prob_cf; 0 uses all calls. range [0, 1], default 0.5")
    g = p.add_argument_group("PARALLEL")
    g.add_argument('--contigs', type=str, default=None,
                   help="genome fasta, a file of contig names, or a comma-separated list: process and write contig by contig")
    g.add_argument('--nproc', type=int, default=1, help="accepted for compatibility (the native aggregator needs no subprocesses)")
    return p


def main():
    args = add_call_freq_args(argparse.ArgumentParser(description='calculate frequency of interested sites at genome level')).parse_args()
    call_mods_frequency_to_file(args)


if __name__ == '__main__':
    sys.exit(main())
