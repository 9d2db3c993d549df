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
        op = gzip.open if path.endswith(".gz") else open  # call_mods_freq.py:46-49
        carry = b""
        with op(path, "rb") as f:
            while True:
                chunk = f.read(chunk_bytes)
                if not chunk:
                    break
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
        with (gzip.open(result_file, "wb") if is_gzip else open(result_file, "wb")) as wf:
            wf.write(data)
        return result_file


def _contig_names(spec):
    """--contigs: genome fasta, a file of names, or a comma list (call_mods_freq.py:253-263)"""
    if os.path.isfile(spec):
        lines = open(spec, "r").read().splitlines()
        is_fa = spec.endswith((".fa", ".fasta", ".fna"))
        if not is_fa:
            for l in lines:
                if l.startswith("#"):
                    continue
                if l.startswith(">"):
                    is_fa = True
                break
        if is_fa:
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
