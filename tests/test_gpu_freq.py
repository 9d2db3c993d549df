"""GPU: `call_freq` on the device (csrc/dsp_freq_dev.hip + call_mods_freq.DeviceSiteFrequency) against the outputs of the
reference's call_mods_frequency_to_file (tests/golden/f5_*, captured by make_golden_text.py).  Byte-exact."""
import os

import numpy as np
import pytest

from tests.helpers import GOLDEN
from tests.test_call_freq import CALLS, _rows_from_calls

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,kw", [("tsv", {}), ("tsv_sorted", dict(sort=True)), ("bed_sorted", dict(bed=True, sort=True)),
                                    ("tsv_cf0", dict(prob_cf=0.0)), ("bed_cf02", dict(bed=True, prob_cf=0.2))])
@pytest.mark.parametrize("blocks", [1, 7])
def test_device_call_freq_matches_reference_output(tag, kw, blocks):
    import torch
    from deepsignal_plant_amd import call_mods_freq as cf
    lines = open(CALLS).read().splitlines()
    r, probs, labels = _rows_from_calls(lines)
    agg = cf.DeviceSiteFrequency(kw.get("prob_cf", 0.5), "cuda:0", nthreads=4)
    pd, ld = torch.from_numpy(probs).cuda(), torch.from_numpy(labels).cuda()
    cuts = np.linspace(0, r.n, blocks + 1).astype(int)
    for a, b in zip(cuts[:-1], cuts[1:]):
        agg.add_block(r, pd[a:b], ld[a:b], int(a), int(a), int(b))
    table = agg.finish()
    assert table.format(kw.get("sort", False), kw.get("bed", False)) == open(os.path.join(GOLDEN, "f5_freq_%s.txt" % tag), "rb").read()
    count, used, sites = table.counts()
    assert count == r.n and 0 < used <= count and sites > 0


def test_device_call_freq_reproduces_the_references_contig_runs():
    """F5c (the reference's call_freq --contigs runs, make_golden_text.py:make_f5_contigs) through the DEVICE reduction: the
    device table over both call files, its lines grouped contig by contig in the reference's concatenation order, is the
    reference's --contigs output -- sorted or in order of first appearance (a contig's sites appear in the whole stream in
    the order they appear in the contig's own stream), tsv and bedMethyl, three prob_cf values."""
    import gzip
    import torch
    from deepsignal_plant_amd import call_mods_freq as cf
    from tests.test_call_freq import CONTIG_RUNS
    lines = open(CALLS).read().splitlines() + gzip.open(os.path.join(GOLDEN, "f5c_calls_b.tsv.gz"), "rt").read().splitlines()
    r, probs, labels = _rows_from_calls(lines)
    pd, ld = torch.from_numpy(probs).cuda(), torch.from_numpy(labels).cuda()
    tables = {}
    for tag, kw in CONTIG_RUNS:
        pc = kw.get("prob_cf", 0.5)
        if pc not in tables:
            agg = cf.DeviceSiteFrequency(pc, "cuda:0", nthreads=4)
            for a in range(0, r.n, 1000):
                agg.add_block(r, pd[a:a + 1000], ld[a:a + 1000], a, a, min(r.n, a + 1000))
            tables[pc] = agg.finish()
        out = tables[pc].format(kw.get("sort", False), kw.get("bed", False)).decode().splitlines(True)
        contigs = sorted(set(cf._contig_names(kw["contigs"])), key=lambda c: c + ".")
        got = "".join(l for c in contigs for l in out if l.split("\t")[0] == c)
        assert got.encode() == open(os.path.join(GOLDEN, "f5c_freq_%s.txt" % tag), "rb").read(), tag


def test_device_call_freq_equals_host_aggregator_on_a_large_random_set():
    """2 M calls over 150 k sites with ties of the %.3f output by construction (probabilities on a coarse grid): the device
    reduction (sequential double sums per site after a stable sort) and the host table print the same bytes"""
    import torch
    from deepsignal_plant_amd import call_mods_freq as cf
    from deepsignal_plant_amd import textio
    rng = np.random.default_rng(11)
    n = 2_000_000
    pos = rng.integers(0, 50_000, n) * 3
    chrom = rng.integers(1, 4, n)
    strand = np.where(pos % 2 == 0, "+", "-")
    info = ["chr%d\t%d\t%s\t%d\tr%d\tt" % (c, p, s, p + 7, i // 50) for i, (c, p, s) in enumerate(zip(chrom, pos, strand))]
    text = "\n".join(info).encode()
    lens = np.array([len(s) for s in info], np.uint32)
    offs = np.r_[0, np.cumsum(lens.astype(np.uint64) + 1)[:-1]].astype(np.uint64)
    r = textio.ParsedRows()
    r.text, r.n, r.row_off, r.info_len, r.seq_len, r.signal_len = np.frombuffer(text, np.uint8), n, offs, lens, 13, 16
    r.kmer = rng.integers(0, 4, (n, 13)).astype(np.uint8)
    p0 = (rng.integers(0, 2001, n) * 0.0005).astype(np.float32)
    probs = np.stack((p0, np.float32(1) - p0), 1).astype(np.float32)
    labels = (probs[:, 1] > probs[:, 0]).astype(np.uint8)
    host = cf.SiteFrequency(0.3)
    host.add_block(r, probs, labels)
    dev = cf.DeviceSiteFrequency(0.3, "cuda:0")
    pd, ld = torch.from_numpy(probs).cuda(), torch.from_numpy(labels).cuda()
    for a in range(0, n, 262144):
        b = min(n, a + 262144)
        dev.add_block(r, pd[a:b], ld[a:b], a, a, b)
    table = dev.finish()
    assert table.counts() == host.counts()
    for sort in (False, True):
        for bed in (False, True):
            assert table.format(sort, bed) == host.format(sort, bed)


@pytest.mark.parametrize("n", [0, 1, 257, 100_003, 3_000_000])
def test_native_record_sort_is_numpys_stable_sort(n):
    """dsp_freq_dev_sort_records (rocPRIM radix sort of (key, index) + one gather): keys with many ties, the unused-record
    sentinel INT64_MAX and values across all 64 bits -- the four columns come out exactly as numpy's stable argsort
    orders them"""
    import ctypes
    import torch
    from deepsignal_plant_amd import _native as nat
    rng = np.random.default_rng(n + 1)
    key = rng.integers(0, max(2, n // 7), size=n).astype(np.int64) << rng.integers(0, 50, size=n).astype(np.int64)
    key[rng.random(n) < 0.1] = np.iinfo(np.int64).max
    cols = [rng.integers(-2 ** 62, 2 ** 62, size=n).astype(np.int64) for _ in range(3)]
    order = np.argsort(key, kind="stable")
    dev = [torch.from_numpy(x).cuda(0) for x in [key] + cols]
    outs = [torch.empty_like(dev[0]) for _ in range(4)]
    L, p = nat.lib(), ctypes.c_void_p
    s = torch.cuda.current_stream()
    need = ctypes.c_size_t(0)
    args = [p(s.cuda_stream), n] + [p(t.data_ptr()) for t in dev] + [p(t.data_ptr()) for t in outs]
    nat.check(int(L.dsp_freq_dev_sort_records(*args, None, ctypes.byref(need))))
    tmp = torch.empty(max(need.value, 1), dtype=torch.uint8, device="cuda:0")
    short = ctypes.c_size_t(max(need.value - 1, 0))
    if n:
        assert int(L.dsp_freq_dev_sort_records(*args, p(tmp.data_ptr()), ctypes.byref(short))) < 0   # scratch too small: refused
    nat.check(int(L.dsp_freq_dev_sort_records(*args, p(tmp.data_ptr()), ctypes.byref(need))))
    torch.cuda.synchronize()
    assert np.array_equal(outs[0].cpu().numpy(), key[order])
    for o, c in zip(outs[1:], cols):
        assert np.array_equal(o.cpu().numpy(), c[order])
