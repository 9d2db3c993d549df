"""CPU: the host half of the extraction stage (motif sites, strand coordinates, region / positions filters,
sampleinfo strings: csrc/dsp_sites.cpp + extract_features.FeatureExtractor) against the site lists of fixture F6,
captured from the reference's _extract_features."""
import numpy as np
import pytest

from deepsignal_plant_amd import extract_features as ef
from deepsignal_plant_amd.utils import process_utils as pu
from tests.test_extract_oracle import CASES, case_inputs


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_site_lists_match_reference(c):
    rs, motif_seqs, chrom2len, region, positions, g = case_inputs(c)
    fx = ef.FeatureExtractor(motifs=c["motifs"], mod_loc=c["mod_loc"], seq_len=c["k"], signal_len=c["s"],
                             normalize_method=c["method"], chrom2len=chrom2len, positions=positions, region=region)
    assert fx.motif_seqs == motif_seqs
    keep, lo, hi = fx._select_reads(rs)
    ev_off = np.concatenate([[0], np.cumsum([r.ev_base.shape[0] for r in keep])]).astype(np.int64)
    ev_base = np.concatenate([r.ev_base for r in keep])
    site_read, site_loc, info, row_off, info_len, read_off, read_len = fx._sites(keep, ev_base, ev_off, lo, hi)
    got = [bytes(info[int(o):int(o) + int(n)]).decode() for o, n in zip(row_off, info_len)]
    assert got == g("info").tolist()
    nb = (c["k"] - 1) // 2
    kmers = [keep[r].seq[l - nb:l + nb + 1] for r, l in zip(site_read, site_loc)]
    assert kmers == g("kmer").tolist()
    names = [bytes(info[int(o) + int(a):int(o) + int(a) + int(n)]).decode() for o, a, n in zip(row_off, read_off, read_len)]
    assert names == [s.split("\t")[4] for s in got]


def test_motif_expansion_region_parsing_and_contig_lengths(tmp_path):
    assert pu.get_motif_seqs("CG") == ["CG"]
    assert pu.get_motif_seqs("chg, CHH") == ["CAG", "CCG", "CTG", "CAA", "CAC", "CAT", "CCA", "CCC", "CCT", "CTA", "CTC", "CTT"]
    assert pu.get_motif_seqs("N") == list("ACGT")
    with pytest.raises(KeyError):
        pu.get_motif_seqs("CX")
    assert pu.parse_region_str(None) == (None, None, None)
    assert pu.parse_region_str("chr1") == ("chr1", None, None)
    assert pu.parse_region_str("chr1:100") == ("chr1", 100, None)
    assert pu.parse_region_str("chr1:100-250") == ("chr1", 100, 250)
    with pytest.raises(ValueError):
        pu.parse_region_str("chr1:a-b")
    fa = tmp_path / "ref.fa"
    fa.write_text(">chr1 something\nACGT\nAC\n>chr2\n\nGGGTT\n")
    assert pu.get_contig2len(str(fa)) == {"chr1": 6, "chr2": 5}
    with pytest.raises(ValueError):
        ef.FeatureExtractor(seq_len=12)
    with pytest.raises(ValueError):
        ef.FeatureExtractor(motifs="CG,CHG")


def test_unknown_base_in_a_window_is_rejected():
    from deepsignal_plant_amd import reads as R
    rd = R.synth_reads(1, seed=3, mean_bases=120)[0]
    fx = ef.FeatureExtractor()
    loc = rd.seq.index("CG", 20)
    rd.ev_base[loc + 2] = ord("X")
    ev_off = np.array([0, rd.ev_base.shape[0]], np.int64)
    with pytest.raises(ValueError, match="not in the alphabet"):
        fx._sites([rd], rd.ev_base, ev_off, np.zeros(1, np.int64), np.zeros(1, np.int64))


def test_feature_row_formatter_reproduces_the_reference_rows():
    """dsp_format_feature_rows on the float64 features of F6 == the reference's _features_to_str, byte for byte"""
    import ctypes
    from deepsignal_plant_amd import _native as nat, textio
    from tests.test_extract_oracle import F6
    for c in CASES:
        g = lambda k: F6["%s/%s" % (c["name"], k)]
        n = int(g("n_sites"))
        rows = textio.ParsedRows()
        info = "".join(g("info").tolist()).encode()
        lens_info = np.array([len(s.encode()) for s in g("info").tolist()], np.uint32)
        rows.text, rows.n, rows.seq_len, rows.signal_len = np.frombuffer(info, np.uint8), n, c["k"], c["s"]
        rows.info_len = lens_info
        rows.row_off = np.concatenate([[0], np.cumsum(lens_info)[:-1]]).astype(np.uint64)
        code = {ch: i for i, ch in enumerate("ACGTNWSMKRYBVDHZ")}
        rows.kmer = np.array([[code[ch] for ch in k] for k in g("kmer").tolist()], np.uint8)
        rows.lens, rows.labels = g("lens").astype(np.int32), g("labels").astype(np.int32)
        # the row prints means/stds rounded to 6 decimals, signals as they are
        got = textio.format_feature_rows(rows, np.around(g("means"), 6), np.around(g("stds"), 6), g("signals"), nthreads=3)
        assert got.decode().splitlines() == g("rows").tolist()
        view = textio.format_feature_rows(rows, np.around(g("means"), 6), np.around(g("stds"), 6), g("signals"), nthreads=2,
                                          as_view=True)   # the zero-copy form `extract` writes
        assert isinstance(view, memoryview) and bytes(view) == got
        # the form `extract` writes: every thread's rows left where it formatted them, in a buffer the caller keeps
        buf = None
        for nt in (1, 3, 64):
            parts, buf = textio.format_feature_rows_parts(rows, np.around(g("means"), 6), np.around(g("stds"), 6), g("signals"),
                                                          nthreads=nt, out=buf)
            assert 1 <= len(parts) <= min(nt, n) and b"".join(bytes(x) for x in parts) == got
            assert buf.nbytes >= textio.feature_rows_capacity(rows)
    # the capacity is a true bound: the longest float64 and int32 there are, in every position
    rows = textio.ParsedRows()
    rows.text, rows.n, rows.seq_len, rows.signal_len = np.frombuffer(b"c\t1\t+\t2\tr\tt", np.uint8), 3, 5, 4
    rows.info_len, rows.row_off = np.full(3, 11, np.uint32), np.zeros(3, np.uint64)
    rows.kmer = np.zeros((3, 5), np.uint8)
    rows.lens, rows.labels = np.full((3, 5), -2147483648, np.int32), np.full(3, -2147483648, np.int32)
    worst = -2.2250738585072014e-308
    m, sg = np.full((3, 5), worst), np.full((3, 5, 4), worst)
    parts, buf = textio.format_feature_rows_parts(rows, m, m, sg, nthreads=3)
    text = b"".join(bytes(x) for x in parts)
    assert text == textio.format_feature_rows(rows, m, m, sg, nthreads=2)
    assert len(text) <= textio.feature_rows_capacity(rows) and text.count(b"-2.2250738585072014e-308") == 3 * (5 + 5 + 20)
    # float64 -> str corner cases against numpy itself
    L = nat.lib()
    L.dsp_format_f64_.restype = ctypes.c_int
    L.dsp_format_f64_.argtypes = [ctypes.c_double, ctypes.c_char_p]
    buf = ctypes.create_string_buffer(64)
    rng = np.random.default_rng(0)
    vals = [0.0, -0.0, 1.0, -1.5, 1e-4, 9.999e-5, 1e-5, 1.5e-7, 123456.789, 1e15, 1e16, 1.2345e17, 5e-324, 1.7976931348623157e308,
            0.1, 0.30000000000000004, 2.5e-05, float("nan"), float("inf"), -float("inf"), 1e22, 123456789012345680.0]
    vals += list(np.around(rng.normal(size=3000) * rng.choice([1e-3, 1, 30], size=3000), 6))
    vals += list(rng.normal(size=2000) * 10.0 ** rng.integers(-12, 20, size=2000))
    # the 6-decimal fast path (x == k / 1e6, 1e-4 <= |x| < 1e9) and its borders
    vals += list(np.around(rng.uniform(-1e9, 1e9, size=20000) * 10.0 ** -rng.integers(0, 10, size=20000), 6))
    vals += list(rng.integers(0, 10 ** 15, size=20000) / 1e6) + list(-(rng.integers(0, 10 ** 9, size=5000) / 1e6))
    vals += [1e-4, 0.0001, 0.000101, 0.000099, 999999999.999999, 1e9, 1000000000.000001, 1073741823.999999, 1073741824.000001,
             0.1 + 0.2, 0.5, 0.25, 123.456789, 123.4567891, 1.0000005, 2.0000015, 4503599627.370496, 0.000123, 33554432.000001]
    for v in vals:
        k = L.dsp_format_f64_(float(v), buf)
        assert buf.raw[:k].decode() == str(np.float64(v)), (v, buf.raw[:k], str(np.float64(v)))


def test_read_container_roundtrip_and_ordered_batches(tmp_path):
    from deepsignal_plant_amd import reads as R
    rs = R.synth_reads(10, seed=2, mean_bases=100)
    d = str(tmp_path)
    R.save_reads(d + "/a.reads.npz", rs[:3])
    R.save_reads(d + "/b.reads", rs[3:4], compress=False)  # ".npz" appended
    R.save_reads(d + "/d.reads.npz", rs[4:])
    open(d + "/c.fast5", "wb").write(b"junk")  # unreadable (and h5py is absent): counted, skipped
    files = R.list_read_files(d)
    assert [f.rsplit("/", 1)[1] for f in files] == ["a.reads.npz", "b.reads.npz", "c.fast5", "d.reads.npz"]
    back = R.load_reads(files[0])
    for a, b in zip(back, rs[:3]):
        assert (a.readname, a.strand, a.alignstrand, a.chrom, a.chrom_start, a.scaling, a.offset) == \
               (b.readname, b.strand, b.alignstrand, b.chrom, b.chrom_start, b.scaling, b.offset)
        assert np.array_equal(a.raw, b.raw) and np.array_equal(a.ev_start, b.ev_start)
        assert np.array_equal(a.ev_len, b.ev_len) and np.array_equal(a.ev_base, b.ev_base)
    bt = R.ReadBatches(files, 4, first_file_index=5, workers=3, lookahead=2)
    out = list(bt)
    assert bt.failed == 1 and [len(x[0]) for x in out] == [4, 6]
    assert [r.readname for x in out for r in x[0]] == [r.readname for r in rs]
    assert [u for x in out for u in x[1]] == [(5 << 20) + i for i in range(3)] + [6 << 20] + [(8 << 20) + i for i in range(6)]
    with pytest.raises(RuntimeError, match="Error opening file"):
        R.from_fast5(d + "/c.fast5")


def test_feature_row_formatter_parts_equal_the_contiguous_form_on_random_rows():
    """differential: the in-place parts formatter (what `extract` writes) against the contiguous one (pinned to the
    reference's rows above), on values of every magnitude and sign, ragged sampleinfo lengths, several shapes and thread
    counts; and the text parses back to the values (round trip through this build's parser)"""
    from deepsignal_plant_amd import textio
    rng = np.random.default_rng(12)
    for L, S, n in ((13, 16, 701), (5, 4, 64), (21, 1, 33), (9, 40, 17)):
        infos = ["chr%d\t%d\t%s\t%d\tread_%d\tt" % (rng.integers(1, 30), rng.integers(0, 10 ** int(rng.integers(1, 10))), "+-"[int(rng.integers(0, 2))],
                                                   rng.integers(0, 10 ** 9), rng.integers(0, 10 ** 6)) for _ in range(n)]
        rows = textio.ParsedRows()
        rows.text = np.frombuffer("".join(infos).encode(), np.uint8)
        rows.n, rows.seq_len, rows.signal_len = n, L, S
        rows.info_len = np.array([len(x) for x in infos], np.uint32)
        rows.row_off = np.concatenate([[0], np.cumsum(rows.info_len)[:-1]]).astype(np.uint64)
        rows.kmer = rng.integers(0, 16, size=(n, L)).astype(np.uint8)
        rows.lens = rng.integers(-5, 100000, size=(n, L)).astype(np.int32)
        rows.labels = rng.integers(0, 2, size=n).astype(np.int32)
        mag = 10.0 ** rng.integers(-9, 12, size=(n, L))
        means = np.around(rng.standard_normal((n, L)) * mag, 6)
        stds = np.abs(rng.standard_normal((n, L))) * 10.0 ** rng.integers(-320, 300, size=(n, L)).astype(np.float64)
        signals = np.where(rng.random((n, L, S)) < 0.3, 0.0, np.around(rng.standard_normal((n, L, S)) * 3, 6))
        want = textio.format_feature_rows(rows, means, stds, signals, nthreads=2)
        buf = None
        for nt in (1, 2, 7, 16):
            parts, buf = textio.format_feature_rows_parts(rows, means, stds, signals, nthreads=nt, out=buf)
            assert b"".join(bytes(x) for x in parts) == want, (L, S, nt)
        back = textio.parse_rows(want, L, S)
        assert back.n == n and np.array_equal(back.kmer, rows.kmer) and np.array_equal(back.lens, rows.lens)
        assert np.array_equal(back.signals, signals.astype(np.float32))
