"""CPU: the oracle's restatement of the reference's feature extraction (oracle/extract_np.py) against fixture F6,
captured from the reference's own _extract_features / _features_to_str (tests/golden/make_golden_extract.py).
Bit-exact, including the random.sample subsampling when replayed under the same random.seed."""
import ast
import os
import random

import numpy as np
import pytest

from deepsignal_plant_amd import reads as dsp_reads
from oracle import extract_np as ox
from tests.helpers import GOLDEN, ROOT

F6 = np.load(os.path.join(GOLDEN, "f6_extract.npz"))
CASES = ast.literal_eval(str(F6["cases"]))


def reads_checksum(rs):
    return int(sum(int(r.raw.astype(np.int64).sum()) + int(r.ev_start.sum()) + int(r.ev_len.sum()) +
                   int(r.ev_base.astype(np.int64).sum()) + r.chrom_start for r in rs))


def case_inputs(c):
    rs = dsp_reads.synth_reads(c["n"], seed=c["seed"], mean_bases=c["mean_bases"])
    assert reads_checksum(rs) == int(F6[c["name"] + "/reads_checksum"]), "read generator drifted"
    g = lambda k: F6["%s/%s" % (c["name"], k)]
    region = tuple(None if x == "None" else (x if i == 0 else int(x)) for i, x in enumerate(g("region").tolist()))
    positions = set(g("positions").tolist()) or None
    chrom2len = {"chr1": 30_000_000, "chr2": 20_000_000, "chr3": 10_000_000} if c["c2l"] else None
    return rs, g("motif_seqs").tolist(), chrom2len, region, positions, g


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_extraction(c):
    rs, motif_seqs, chrom2len, region, positions, g = case_inputs(c)
    random.seed(1000 + c["seed"])
    feats = ox.extract_features(rs, c["method"], motif_seqs, c["mod_loc"], chrom2len, c["k"], c["s"], 1, positions,
                                region, sampler="python")
    assert len(feats) == int(g("n_sites"))
    assert ["\t".join([f[0], str(f[1]), f[2], str(f[3]), f[4], f[5]]) for f in feats] == g("info").tolist()
    assert [f[6] for f in feats] == g("kmer").tolist()
    assert np.array_equal(np.array([f[7] for f in feats]), g("means"))
    assert np.array_equal(np.array([f[8] for f in feats]), g("stds"))
    assert np.array_equal(np.array([f[9] for f in feats]), g("lens"))
    assert np.array_equal(np.array([f[10] for f in feats]), g("signals"))
    assert [ox.features_to_str(f) for f in feats] == g("rows").tolist()


def test_hash_sampler_properties():
    seen = set()
    for n in (17, 18, 40, 129, 417):
        for b in range(50):
            idx = ox.hash_sample_sorted(n, 16, 5, 3, b)
            assert len(idx) == 16 and idx == sorted(set(idx)) and 0 <= idx[0] and idx[-1] < n
            seen.add(tuple(idx))
    assert len(seen) > 200  # different bases draw different samples
    assert ox.hash_sample_sorted(40, 16, 5, 3, 7) == ox.hash_sample_sorted(40, 16, 5, 3, 7)
    assert ox.hash_sample_sorted(40, 16, 5, 3, 7) != ox.hash_sample_sorted(40, 16, 6, 3, 7)
    # every element is (nearly) equally likely
    cnt = np.zeros(32)
    for b in range(4000):
        cnt[ox.hash_sample_sorted(32, 16, 1, 9, b)] += 1
    assert abs(cnt / 4000 - 0.5).max() < 0.04


def test_rows_feed_the_tsv_parser_identically():
    """features -> _features_to_str rows -> the call_mods parser == features_to_arrays(round_stats=True)"""
    from deepsignal_plant_amd import textio
    c = CASES[0]
    rs, motif_seqs, chrom2len, region, positions, g = case_inputs(c)
    feats = ox.extract_features(rs, c["method"], motif_seqs, c["mod_loc"], chrom2len, c["k"], c["s"], 1, positions,
                                region, sampler="hash", seed=3)
    text = ("\n".join(ox.features_to_str(f) for f in feats) + "\n").encode()
    rows = textio.parse_rows(text, c["k"], c["s"])
    arr = ox.features_to_arrays(feats, c["k"], c["s"], round_stats=True)
    for k in ("kmer", "means", "stds", "lens", "signals", "labels"):
        assert np.array_equal(getattr(rows, k), arr[k]), k
    assert [rows.sampleinfo(i) for i in range(rows.n)] == arr["sampleinfo"]


def test_mad_scale_constant_is_the_call_statsmodels_makes():
    """statsmodels is not in the image; its robust.mad divides by c = Gaussian.ppf(3/4.) with Gaussian = scipy.stats.norm
    (statsmodels/robust/scale.py).  scipy IS here: the constant restated in the oracle, in the fixture generator and in
    csrc/dsp_extract.hip (kMadC) is that very call's result, bit for bit."""
    import re
    from scipy.stats import norm
    from oracle import extract_np as ox
    c = float(norm.ppf(3 / 4.))
    assert c == ox.MAD_C == 0.6744897501960817
    hip = open(os.path.join(ROOT, "deepsignal_plant_amd", "csrc", "dsp_extract.hip")).read()
    assert float(re.search(r"kMadC = ([0-9.]+);", hip).group(1)) == c
    gen = open(os.path.join(ROOT, "tests", "golden", "make_golden_extract.py")).read()
    assert float(re.search(r"def _mad\(a, c=([0-9.]+)\)", gen).group(1)) == c
