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
