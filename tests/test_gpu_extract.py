"""GPU: the extraction kernels (csrc/dsp_extract.hip) against the oracle's numpy restatement of the reference
(oracle/extract_np.py, itself pinned by fixture F6) and against F6 directly.  Bit-exact: every float64 operation
is evaluated in numpy's order, so means / stds / signals / lens / k-mers must agree to the last bit; bases longer
than signal_len are compared under the product's own sampler (the reference's is unseeded) and, against F6, through
the properties that do not depend on which samples were drawn."""
import numpy as np
import pytest

from deepsignal_plant_amd import extract_features as ef
from deepsignal_plant_amd import reads as R
from oracle import extract_np as ox
from tests.test_extract_oracle import CASES, case_inputs

pytestmark = pytest.mark.gpu
KEYS = ("kmer", "means", "stds", "lens", "signals", "labels")


def _oracle_arrays(rs, c, motif_seqs, chrom2len, region, positions, round_stats, seed, first_uid=0):
    feats = ox.extract_features(rs, c["method"], motif_seqs, c["mod_loc"], chrom2len, c["k"], c["s"], 1, positions,
                                region, sampler="hash", seed=seed, first_read_uid=first_uid)
    return ox.features_to_arrays(feats, c["k"], c["s"], round_stats=round_stats), feats


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint8) if a.dtype.kind != "f" else a.view(np.uint32),
                                                 b.view(np.uint8) if b.dtype.kind != "f" else b.view(np.uint32))


@pytest.mark.parametrize("round_stats", [False, True])
@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_kernels_match_oracle_bit_for_bit(c, round_stats):
    rs, motif_seqs, chrom2len, region, positions, g = case_inputs(c)
    fx = ef.FeatureExtractor(motifs=c["motifs"], mod_loc=c["mod_loc"], seq_len=c["k"], signal_len=c["s"],
                             normalize_method=c["method"], chrom2len=chrom2len, positions=positions, region=region,
                             seed=77, round_stats=round_stats)
    got = fx.extract(rs, first_read_uid=5).to_host()
    want, _ = _oracle_arrays(rs, c, motif_seqs, chrom2len, region, positions, round_stats, 77, 5)
    assert got.n == len(want["sampleinfo"]) == int(g("n_sites"))
    assert [got.sampleinfo(i) for i in range(got.n)] == want["sampleinfo"]
    for k in KEYS:
        assert _same(getattr(got, k), want[k]), k


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_kernels_match_reference_fixture(c):
    """straight against what the reference produced (F6): exact wherever it did not subsample"""
    rs, motif_seqs, chrom2len, region, positions, g = case_inputs(c)
    fx = ef.FeatureExtractor(motifs=c["motifs"], mod_loc=c["mod_loc"], seq_len=c["k"], signal_len=c["s"],
                             normalize_method=c["method"], chrom2len=chrom2len, positions=positions, region=region)
    got = fx.extract(rs).to_host()
    assert [got.sampleinfo(i) for i in range(got.n)] == g("info").tolist()
    assert np.array_equal(got.lens, g("lens"))
    assert _same(got.means, g("means").astype(np.float32)) and _same(got.stds, g("stds").astype(np.float32))
    ref_sig = g("signals").astype(np.float32)
    short = g("lens") <= c["s"]
    assert _same(got.signals[short], ref_sig[short])
    # subsampled bases: 16 samples of that base, in time order -> every value occurs in the base's own samples
    long_idx = np.argwhere(~short)
    assert len(long_idx) > 0 or c["name"] == "none"
    read_of = {r.readname: r for r in rs}
    for i, j in long_idx[:60]:
        rd = read_of[got.sampleinfo(int(i)).split("\t")[4]]
        norm = ox.normalize_signals(ox.rescale_signals(rd.raw, rd.scaling, rd.offset), c["method"])
        info = got.sampleinfo(int(i)).split("\t")
        pos = int(info[1])
        loc = (rd.chrom_start + len(rd.seq) - 1 - pos) if rd.alignstrand == "-" else pos - rd.chrom_start
        b = loc - (c["k"] - 1) // 2 + int(j)
        base = norm[int(rd.ev_start[b]):int(rd.ev_start[b] + rd.ev_len[b])].astype(np.float32)
        vals = got.signals[i, j]
        # an order-preserving embedding of vals into base
        p = 0
        for v in vals:
            while p < len(base) and base[p] != v:
                p += 1
            assert p < len(base)
            p += 1


def test_batching_invariance_empty_batches_and_edge_reads():
    rs = R.synth_reads(7, seed=21, mean_bases=200)
    fx = ef.FeatureExtractor(motifs="CG", seed=3)
    whole = fx.extract(rs).to_host()
    parts = [fx.extract(rs[:3], first_read_uid=0).to_host(), fx.extract(rs[3:], first_read_uid=3).to_host()]
    for k in KEYS:
        assert _same(getattr(whole, k), np.concatenate([getattr(p, k) for p in parts])), k
    assert fx.extract([]).n == 0
    # a read without any motif site, a read whose last event runs past the raw signal, a constant read (scale 0)
    a = R.synth_reads(1, seed=5, mean_bases=60)[0]
    a.ev_base[:] = ord("A")
    b = R.synth_reads(1, seed=6, mean_bases=80)[0]
    b.raw = b.raw[:int(b.ev_start[-3])]
    c = R.synth_reads(1, seed=7, mean_bases=80)[0]
    c.raw[:] = 500
    got = fx.extract([a, b, c]).to_host()
    feats = ox.extract_features([a, b, c], "mad", ["CG"], 0, None, 13, 16, 1, sampler="hash", seed=3)
    with np.errstate(invalid="ignore"):
        want = ox.features_to_arrays(feats, 13, 16, round_stats=False)
    assert got.n == len(feats) > 0
    for k in KEYS:
        w, gk = want[k], getattr(got, k)
        if w.dtype.kind == "f":  # empty bases give nan means in the reference too
            assert np.array_equal(np.isnan(w), np.isnan(gk)) and _same(np.nan_to_num(gk), np.nan_to_num(w)), k
        else:
            assert _same(gk, w), k


def test_long_read_statistics_match_numpy():
    """read-level shift/scale on reads far longer than numpy's 8192-element reduction buffer, both methods"""
    rs = R.synth_reads(3, seed=31, mean_bases=9000)
    assert max(len(r.raw) for r in rs) > 60000
    for method in ("mad", "zscore"):
        fx = ef.FeatureExtractor(normalize_method=method)
        out = fx.extract(rs)
        shift, scale = out.shift.cpu().numpy(), out.scale.cpu().numpy()
        for i, r in enumerate(rs):
            x = ox.rescale_signals(r.raw, r.scaling, r.offset)
            if method == "mad":
                want = (np.median(x), float(ox.mad(x)))
            else:
                want = (np.mean(x), float(np.std(x)))
            assert (shift[i], scale[i]) == want, (method, i, shift[i], scale[i], want)


def test_extracted_features_drive_call_mods_like_the_tsv_route(tmp_path):
    """reads -> device features (round_stats=True) -> forward  ==  reads -> oracle rows -> TSV parser -> forward"""
    import torch
    from deepsignal_plant_amd import textio
    from deepsignal_plant_amd.models import ModelBiLSTM
    from deepsignal_plant_amd import synth
    rs = R.synth_reads(4, seed=41, mean_bases=300)
    fx = ef.FeatureExtractor(seed=9, round_stats=True)
    out = fx.extract(rs)
    feats = ox.extract_features(rs, "mad", ["CG"], 0, None, 13, 16, 1, sampler="hash", seed=9)
    text = ("\n".join(ox.features_to_str(f) for f in feats) + "\n").encode()
    rows = textio.parse_rows(text, 13, 16)
    model = ModelBiLSTM(init_state="zeros")
    model.load_state_dict(synth.random_state_dict(model, seed=4))
    model.cuda(0)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(a).to(dev)
    _, p_tsv = model.forward(t(rows.kmer), t(rows.means), t(rows.stds), t(rows.lens), t(rows.signals))
    _, p_dev = model.forward(out.kmer, out.means, out.stds, out.lens, out.signals)
    assert torch.equal(p_tsv, p_dev) and p_dev.shape == (len(feats), 2)


def test_mad_fast_path_and_its_fallbacks_match_numpy():
    """the windowed-histogram MAD (two passes) and the generic radix select it falls back to, on reads built to hit
    each: ordinary DAQ ranges (odd / even lengths, ties at the median), codes spread over the whole int16 range, a
    bimodal read whose medians lie 40,000 codes apart (outside any window), negative scaling, a constant read"""
    rng = np.random.default_rng(5)
    base = R.synth_reads(1, seed=9, mean_bases=60)[0]

    def read_with(raw, scaling=0.18, offset=7.0):
        r = R.ReadRecord("r", "t", "+", "chr1", 0, raw, scaling, offset, base.ev_start[:5], base.ev_len[:5], base.ev_base[:5])
        return r
    raws = [
        rng.integers(300, 700, size=9999),                      # odd length
        rng.integers(300, 700, size=10000),                     # even length
        np.repeat(np.array([500, 501]), 5000),                   # medians straddle two codes, massive ties
        np.full(4096, 512),                                      # constant: scale 0
        rng.integers(-32768, 32768, size=20001),                 # whole int16 range: MAD level beyond the window
        np.concatenate([rng.integers(-32768, -20000, size=6000), rng.integers(20000, 32768, size=6000)]),  # bimodal
        np.concatenate([rng.integers(400, 600, size=50000), rng.integers(-30000, 30000, size=300)]),       # outliers outside the window
        rng.integers(0, 8192, size=131077),
        np.array([5]), np.array([5, 9]), np.array([9, 5, 7]),
    ]
    reads = [read_with(x.astype(np.int16)) for x in raws]
    reads.append(read_with(raws[1].astype(np.int16), scaling=-0.18))   # decreasing pA: generic path
    reads.append(read_with(raws[1].astype(np.int16), scaling=1e-30))   # degenerate spacing: guarded -> generic path
    out = ef.FeatureExtractor(normalize_method="mad").extract(reads)
    shift, scale = out.shift.cpu().numpy(), out.scale.cpu().numpy()
    for i, r in enumerate(reads):
        x = ox.rescale_signals(r.raw, r.scaling, r.offset)
        want = (np.median(x), float(ox.mad(x)))
        assert (shift[i], scale[i]) == want, (i, len(r.raw), shift[i], scale[i], want)
