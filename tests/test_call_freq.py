"""CPU: native call_freq (csrc/dsp_freq.cpp) vs the outputs of the reference's call_mods_frequency_to_file
captured in tests/golden/f5_* (make_golden_text.py:make_f5).  Byte-exact."""
import argparse
import gzip
import os

import numpy as np
import pytest

from deepsignal_plant_amd import call_mods_freq as cf
from deepsignal_plant_amd import textio
from tests.helpers import GOLDEN

CALLS = os.path.join(GOLDEN, "f5_calls.tsv")


def _args(out, **kw):
    d = dict(input_path=[CALLS], result_file=out, file_uid=None, contigs=None, nproc=1, gzip=False, bed=False,
             sort=False, prob_cf=0.5)
    d.update(kw)
    return argparse.Namespace(**d)


@pytest.mark.parametrize("tag,kw", [("tsv", {}), ("tsv_sorted", dict(sort=True)), ("bed_sorted", dict(bed=True, sort=True)),
                                    ("tsv_cf0", dict(prob_cf=0.0)), ("bed_cf02", dict(bed=True, prob_cf=0.2))])
def test_call_freq_matches_reference_output(tmp_path, tag, kw):
    out = str(tmp_path / "freq.txt")
    cf.call_mods_frequency_to_file(_args(out, **kw))
    assert open(out, "rb").read() == open(os.path.join(GOLDEN, "f5_freq_%s.txt" % tag), "rb").read()


def test_call_freq_gzip_dir_input_and_contigs(tmp_path):
    d = tmp_path / "in"
    d.mkdir()
    lines = open(CALLS).read().splitlines(True)
    (d / "a.calls.tsv").write_text("".join(lines[:3000]))
    with gzip.open(str(d / "b.calls.tsv.gz"), "wt") as f:
        f.write("".join(lines[3000:]))
    (d / "ignored.txt").write_text("not a call file\n")
    out = str(tmp_path / "freq.txt")
    # os.listdir order is arbitrary in the reference too; feed the two parts in file order explicitly
    cf.call_mods_frequency_to_file(_args(out, input_path=[str(d / "a.calls.tsv"), str(d / "b.calls.tsv.gz")], gzip=True))
    assert gzip.open(out + ".gz", "rb").read() == open(os.path.join(GOLDEN, "f5_freq_tsv.txt"), "rb").read()
    cf.call_mods_frequency_to_file(_args(out, input_path=[str(d)], file_uid=".calls.", sort=True))
    assert open(out, "rb").read() == open(os.path.join(GOLDEN, "f5_freq_tsv_sorted.txt"), "rb").read()
    # --contigs: only the listed contigs, contig by contig
    cf.call_mods_frequency_to_file(_args(out, contigs="chr1,chrT", sort=True))
    want = [l for l in open(os.path.join(GOLDEN, "f5_freq_tsv_sorted.txt")) if l.split("\t")[0] in ("chr1", "chrT")]
    assert open(out).read() == "".join(want)
    with pytest.raises(ValueError):
        cf.call_mods_frequency_to_file(_args(out, input_path=["/nonexistent"]))


CONTIG_RUNS = [("comma_tsv", dict(contigs="chr1,chrT,chr10,chrNone,chr1-x", nproc=1)),
               ("names_bed_sorted", dict(contigs=os.path.join(GOLDEN, "f5c_contig_names.txt"), nproc=2, bed=True, sort=True)),
               ("fasta_tsv_sorted", dict(contigs=os.path.join(GOLDEN, "f5c_genome.fa"), nproc=2, sort=True)),
               ("fasta_content_bed_cf02", dict(contigs=os.path.join(GOLDEN, "f5c_genome_noext.txt"), nproc=1, bed=True, prob_cf=0.2)),
               ("comma_gzip_cf0", dict(contigs="chr2,Chr1,chr1_2,chr1.1", nproc=2, prob_cf=0.0, gzip=True))]


@pytest.mark.parametrize("tag,kw", CONTIG_RUNS)
def test_call_freq_contigs_match_reference_output(tmp_path, tag, kw):
    """F5c (tests/golden/make_golden_text.py:make_f5_contigs -- the reference's call_mods_frequency_to_file run with
    --contigs): the contig list from a comma string / a names file / a genome fasta (by suffix and by content), two input
    files (one .gz), --nproc 1 and 2, per-contig results concatenated in the reference's sorted-file-name order (contig
    names that are prefixes of each other), --sort / --bed / --gzip.  Byte-exact."""
    out = str(tmp_path / "freq.txt")
    cf.call_mods_frequency_to_file(_args(out, input_path=[CALLS, os.path.join(GOLDEN, "f5c_calls_b.tsv.gz")], **kw))
    got = gzip.open(out + ".gz", "rb").read() if kw.get("gzip") else open(out, "rb").read()
    assert got == open(os.path.join(GOLDEN, "f5c_freq_%s.txt" % tag), "rb").read()
    assert os.listdir(str(tmp_path)) == [os.path.basename(out) + (".gz" if kw.get("gzip") else "")]   # no temporary files left


def test_contig_list_sources_follow_the_reference(tmp_path):
    """call_mods_freq.py:253-263 + :129-149: fasta by suffix or by ANY '>' line (not only the first non-comment line),
    header's first word, file order kept; names file and comma string: sorted(set(...))"""
    assert cf._contig_names("b,a,b, c") == [" c", "a", "b"]
    assert cf._contig_names(os.path.join(GOLDEN, "f5c_genome.fa")) == ["chr2", "chr1_2", "chr10", "chrM", "Chr1", "chr1", "absent_contig"]
    assert cf._contig_names(os.path.join(GOLDEN, "f5c_genome_noext.txt")) == ["chrT", "chr1-x", "scaffold_10"]
    assert cf._contig_names(os.path.join(GOLDEN, "f5c_contig_names.txt")) == sorted({"# contigs of interest", "chr10", "chrT", "chr1", "chr1-x",
                                                                                    "chrNone", "scaffold_10", "chr1_2"})
    late = tmp_path / "names_then_header.txt"
    late.write_text("chr1\nchr2\n>chr3 x\n")          # a '>' line anywhere makes it a fasta in the reference
    assert cf._contig_names(str(late)) == ["chr3"]
    fa = tmp_path / "empty.fasta"
    fa.write_text("no header here\n")
    assert cf._contig_names(str(fa)) == []


def test_fused_block_path_equals_text_round_trip():
    """call_mods results fed straight into the aggregator == writing the per-read file and re-reading it"""
    f3 = np.load(os.path.join(GOLDEN, "f3_format.npz"))
    probs, kmers = f3["probs"], f3["kmers"]
    n = probs.shape[0]
    rng = np.random.default_rng(5)
    info = ["chr%d\t%d\t%s\t%d\tread%d\tt" % (rng.integers(1, 4), 100 + 3 * rng.integers(0, 40), "+-"[i % 2], i, i) for i in range(n)]
    text = "\n".join(info).encode()
    offs, lens, pos = np.zeros(n, np.uint64), np.zeros(n, np.uint32), 0
    for i, s in enumerate(info):
        offs[i], lens[i] = pos, len(s)
        pos += len(s) + 1
    r = textio.ParsedRows()
    r.text, r.n, r.kmer, r.row_off, r.info_len, r.seq_len, r.signal_len = np.frombuffer(text, np.uint8), n, kmers, offs, lens, 13, 16
    labels = probs.argmax(1).astype(np.uint8)
    a = cf.SiteFrequency(0.5)
    a.add_calls_text(textio.format_calls(r, probs, labels))
    b = cf.SiteFrequency(0.5)
    b.add_block(r, probs[:5000], labels[:5000], 0, 5000)
    b.add_block(r, probs[5000:], labels[5000:], 5000, n)
    assert a.counts() == b.counts() and a.counts()[0] == n
    for sort in (False, True):
        for bed in (False, True):
            assert a.format(sort, bed) == b.format(sort, bed)


def test_call_freq_rejects_malformed_lines():
    a = cf.SiteFrequency(0.5)
    with pytest.raises(ValueError):
        a.add_calls_text(b"chr1\t10\t+\t10\tr\tt\t0.9\n")
    with pytest.raises(ValueError):
        a.add_calls_text(b"chr1\tx\t+\t10\tr\tt\t0.9\t0.1\t0\tAACGT\n")


def test_thread_count_does_not_change_the_result_and_numbers_parse_like_python():
    import ctypes
    import random
    from deepsignal_plant_amd import _native as nat
    from deepsignal_plant_amd.call_mods_freq import SiteFrequency
    text = open(os.path.join(GOLDEN, "f5_calls.tsv"), "rb").read() * 3
    outs = []
    for nt in (1, 2, 7, 16):
        agg = SiteFrequency(0.1, nthreads=nt)
        agg.add_calls_text(text)
        outs.append((agg.format(False, False), agg.format(True, True), agg.counts()))
    assert all(o == outs[0] for o in outs[1:]) and outs[0][2][0] == text.count(b"\n")
    # a malformed line: everything before it is applied, then the error (like a sequential pass)
    lines = text.splitlines(keepends=True)
    bad = b"".join(lines[:1000]) + b"chr1\t12\t+\t12\tr\tt\t0.4\tx\t1\tACGTA\n" + b"".join(lines[1000:])
    agg = SiteFrequency(0.1, nthreads=5)
    with pytest.raises(ValueError, match="line 1000"):
        agg.add_calls_text(bad)
    assert agg.counts()[0] == 1000
    # decimal -> double, bit for bit like float()
    L = nat.lib()
    L.dsp_parse_double_.restype = ctypes.c_int
    L.dsp_parse_double_.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double)]
    rng = random.Random(4)
    cases = ["0", "0.0", "-0.0", "1", "1.0", "1e-05", "9.9e-05", "0.000001", "0.123457", "1.", ".5", "+3.25", "1E5", "1e+22", "1e23",
             "123456789012345", "1234567890123456", "0.1234567890123456789", "1e-22", "1e-23", "5e-324", "1.7976931348623157e308",
             " 0.25 ", "0.500000", "00012.50", "1e0005"]
    for _ in range(20000):
        d = rng.randint(1, 19)
        m = str(rng.randint(0, 10 ** d - 1))
        k = rng.randint(0, len(m))
        t = m[:k] + "." + m[k:] if rng.random() < 0.8 else m
        if rng.random() < 0.3:
            t += "e%+d" % rng.randint(-30, 30)
        cases.append(t)
    out = ctypes.c_double()
    for t in cases:
        b = t.encode()
        assert L.dsp_parse_double_(b, len(b), ctypes.byref(out)) == 1, t
        assert out.value == float(t) and (out.value != 0 or str(out.value) == str(float(t))), (t, out.value, float(t))
    for t in ("", "abc", "1e", "1.2.3", "--1", "1e+", "0x10"):
        b = t.encode()
        assert L.dsp_parse_double_(b, len(b), ctypes.byref(out)) == 0, t


def test_printed_probability_of_a_rounded_float32_is_k_over_1e6_exactly():
    """The device aggregator (csrc/dsp_freq_dev.hip) never formats or parses text: it keeps k = rint(x * 1e6) and adds
    float(k) / 1e6.  That is only the reference's value if, for EVERY k in [0, 10^6], (a) str(np.float32(k / 1e6)) --
    what call_mods prints (call_modifications.py:177-187) -- is the decimal k * 1e-6, and (b) Python's float() of it
    (txt_formater.py:14-15) is the correctly rounded quotient k / 1e6.  Checked exhaustively."""
    k = np.arange(0, 1000001, dtype=np.float32)
    z = (k / np.float32(1e6)).astype(np.float32)               # np.round(x, 6) of a float32 ends with this division
    want = np.arange(0, 1000001, dtype=np.float64) / 1e6
    got = np.array([float(str(v)) for v in z])                 # str(np.float32) -> float(): the reference's round trip
    assert np.array_equal(got, want)
    # and the native formatter / parser pair used by the host aggregator agrees on a sample
    import ctypes
    from deepsignal_plant_amd import _native as nat
    L = nat.lib()
    L.dsp_format_prob_f32_.restype = ctypes.c_int
    L.dsp_format_prob_f32_.argtypes = [ctypes.c_float, ctypes.c_char_p]
    buf = ctypes.create_string_buffer(64)
    for i in list(range(0, 1000001, 997)) + [1, 9, 10, 99, 100, 999999, 1000000]:
        n = L.dsp_format_prob_f32_(float(z[i]), buf)
        assert float(buf.raw[:n].decode()) == want[i]


def _rows_from_calls(lines):
    """ParsedRows-like block (sampleinfo text + k-mer codes of a 5-mer window) and the float32 probabilities that print
    as the call lines' prob_0 / prob_1"""
    code = {b: i for i, b in enumerate("ACGTNWSMKRYBVDHZ")}
    info, probs, labels, kmer = [], [], [], []
    for l in lines:
        w = l.rstrip("\n").split("\t")
        info.append("\t".join(w[:6]))
        p0 = np.float32(w[6])
        probs.append((p0, np.float32(1.0) - p0))
        labels.append(int(w[8]))
        kmer.append([code[c] for c in w[9]])
    n = len(info)
    text = "\n".join(info).encode()
    offs, lens, pos = np.zeros(n, np.uint64), np.zeros(n, np.uint32), 0
    for i, s in enumerate(info):
        offs[i], lens[i] = pos, len(s)
        pos += len(s) + 1
    r = textio.ParsedRows()
    r.text, r.n, r.row_off, r.info_len, r.seq_len, r.signal_len = np.frombuffer(text, np.uint8), n, offs, lens, 5, 16
    r.kmer = np.array(kmer, np.uint8)
    return r, np.array(probs, np.float32), np.array(labels, np.uint8)


@pytest.mark.parametrize("tag,kw", [("tsv", {}), ("bed_sorted", dict(bed=True, sort=True)), ("tsv_cf0", dict(prob_cf=0.0)),
                                    ("bed_cf02", dict(bed=True, prob_cf=0.2))])
def test_host_side_of_the_device_aggregator_on_the_reference_outputs(tag, kw):
    """dsp_freq_block_keys + the record encoding + dsp_freq_add_sites, with the two device kernels (encode, sequential
    per-site reduce after a stable sort) restated in numpy: byte-identical to the reference's call_freq outputs (F5).
    The kernels themselves are checked against the same files in tests/test_gpu_freq.py."""
    import ctypes
    from deepsignal_plant_amd import _native as nat
    L = nat.lib()
    p = ctypes.c_void_p
    lines = open(CALLS).read().splitlines()
    r, probs, labels = _rows_from_calls(lines)
    cfv = kw.get("prob_cf", 0.5)
    agg = cf.SiteFrequency(cfv, nthreads=3)
    n = r.n
    key, pis, meta = np.empty(n, np.int64), np.empty(n, np.int64), np.empty(n, np.uint32)
    assert L.dsp_freq_block_keys(agg._h, p(r.text.ctypes.data), p(r.row_off.ctypes.data), p(r.info_len.ctypes.data),
                                 p(r.kmer.ctypes.data), 5, n, p(key.ctypes.data), p(pis.ctypes.data), p(meta.ctypes.data)) == n
    # encode (dsp_freq_dev.hip: freq_encode_kernel)
    a, b = probs[:, 0], probs[:, 1]
    q = (a / (a + b)).astype(np.float32)
    k0 = np.rint(q * np.float32(1e6)).astype(np.float32)
    z0 = (k0 / np.float32(1e6)).astype(np.float32)
    k1 = np.rint((np.float32(1.0) - z0) * np.float32(1e6)).astype(np.float32)
    k0, k1 = k0.astype(np.int64), k1.astype(np.int64)
    used = np.abs(k0 / 1e6 - k1 / 1e6) >= cfv
    packed = k0 | (k1 << 20) | ((labels == 1).astype(np.int64) << 40) | (meta.astype(np.int64) << 41)
    row = np.arange(n, dtype=np.int64)
    key, packed, pis, row = key[used], packed[used], pis[used], row[used]
    # stable sort + sequential per-site reduce (freq_reduce_kernel)
    order = np.argsort(key, kind="stable")
    key, packed, pis, row = key[order], packed[order], pis[order], row[order]
    heads = np.nonzero(np.r_[True, key[1:] != key[:-1]])[0] if len(key) else np.zeros(0, np.int64)
    ends = np.r_[heads[1:], len(key)]
    s0, s1, met, cov = [], [], [], []
    for h, e in zip(heads, ends):
        x0 = x1 = 0.0
        for j in range(h, e):
            x0 += float(packed[j] & 0xfffff) / 1e6
            x1 += float((packed[j] >> 20) & 0xfffff) / 1e6
        s0.append(x0); s1.append(x1)
        met.append(int(((packed[h:e] >> 40) & 1).sum())); cov.append(int(e - h))
    arr = lambda x, dt: np.ascontiguousarray(np.array(x, dt))
    sk, fr, pk, sp = arr(key[heads], np.int64), arr(row[heads], np.int64), arr(packed[heads], np.int64), arr(pis[heads], np.int64)
    s0, s1, met, cov = arr(s0, np.float64), arr(s1, np.float64), arr(met, np.int64), arr(cov, np.int64)
    by_first = np.argsort(fr, kind="stable")
    cols = [np.ascontiguousarray(c[by_first]) for c in (sk, fr, pk, sp, s0, s1, met, cov)]
    table = cf.SiteFrequency(cfv)
    for i in range(L.dsp_freq_chrom_count(agg._h)):
        k = int(L.dsp_freq_chrom_name(agg._h, i, None, 0))
        buf = ctypes.create_string_buffer(max(k, 1))
        L.dsp_freq_chrom_name(agg._h, i, buf, k)
        assert L.dsp_freq_intern_chrom(table._h, buf.raw[:k], k) == i
    assert L.dsp_freq_add_sites(table._h, len(cols[0]), *[p(c.ctypes.data) for c in cols]) == len(cols[0])
    L.dsp_freq_add_counts(table._h, n)
    assert table.format(kw.get("sort", False), kw.get("bed", False)) == open(os.path.join(GOLDEN, "f5_freq_%s.txt" % tag), "rb").read()
    assert table.counts()[0] == n and table.counts()[1] == int(used.sum())
    # rows the compact encoding cannot hold are refused loudly (the caller then uses the host aggregator)
    bad, _, _ = _rows_from_calls([lines[0].replace("\t+\t", "\t*\t").replace("\t-\t", "\t*\t")])
    with pytest.raises(ValueError, match="host aggregator"):
        nat.check(int(L.dsp_freq_block_keys(agg._h, p(bad.text.ctypes.data), p(bad.row_off.ctypes.data), p(bad.info_len.ctypes.data),
                                            p(bad.kmer.ctypes.data), 5, 1, p(key.ctypes.data), p(pis.ctypes.data), p(meta.ctypes.data))))


def test_call_freq_reads_gz_inputs_through_the_fast_gzip_layer(tmp_path, monkeypatch):
    """per-read call files as .gz (call_mods_freq.py:46-49 opens them with gzip.open): a BGZF file (what call_mods --gzip
    writes: members inflated in parallel), a plain single-member gzip through zlib, and the same through the parallel
    inflater (forced for this small file): the same bytes as from the text file"""
    from deepsignal_plant_amd import gzio
    want = open(os.path.join(GOLDEN, "f5_freq_tsv.txt"), "rb").read()
    calls = open(CALLS, "rb").read()
    bg = str(tmp_path / "calls.bgzf.tsv.gz")
    with gzio.BgzfWriter(bg, nthreads=3, chunk=1 << 16) as w:
        w.write(calls)
    sg = str(tmp_path / "calls.single.tsv.gz")
    open(sg, "wb").write(gzip.compress(calls, 6))
    for path, force_parallel in ((bg, False), (sg, False), (sg, True)):
        if force_parallel:
            monkeypatch.setattr(gzio, "PGZ_MIN_BYTES", 0)
        out = str(tmp_path / "freq.tsv")
        cf.call_mods_frequency_to_file(_args(out, input_path=[path]))
        assert open(out, "rb").read() == want, (path, force_parallel)
    assert [bytes(a) for a in gzio.read_text_chunks(sg, 50_000, nthreads=3)] != []
    assert b"".join(bytes(a) for a in gzio.read_text_chunks(bg, 50_000, nthreads=3)) == calls
