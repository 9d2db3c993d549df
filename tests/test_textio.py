"""CPU: native parser / formatter vs fixtures captured from the reference's own
_read_features_file and _call_mods (tests/golden/make_golden_text.py).  Bit-exact."""
import gzip
import os

import numpy as np
import pytest

from deepsignal_plant_amd import textio
from tests.helpers import GOLDEN


@pytest.fixture(scope="module")
def f2():
    return np.load(os.path.join(GOLDEN, "f2_parsed.npz"))


@pytest.mark.parametrize("nthreads", [1, 3])
@pytest.mark.parametrize("gz", [False, True])
def test_parser_matches_reference_reader(f2, nthreads, gz):
    path = os.path.join(GOLDEN, "f2_rows.tsv" + (".gz" if gz else ""))
    data = (gzip.open(path, "rb") if gz else open(path, "rb")).read()
    r = textio.parse_rows(data, 13, 16, nthreads=nthreads)
    assert r.n == len(f2["labels"]) == 200
    assert np.array_equal(r.kmer, f2["kmers"])
    # the reference holds Python doubles and narrows to float32 when it builds tensors (FloatTensor)
    assert np.array_equal(r.means, f2["means"].astype(np.float32))
    assert np.array_equal(r.stds, f2["stds"].astype(np.float32))
    assert np.array_equal(r.lens, f2["lens"])
    assert np.array_equal(r.signals, f2["signals"].astype(np.float32))
    assert np.array_equal(r.labels, f2["labels"])
    assert [r.sampleinfo(i) for i in range(r.n)] == list(f2["sampleinfo"])
    assert r.readname(0) == f2["sampleinfo"][0].split("\t")[4]


def test_parser_edge_cases():
    row = open(os.path.join(GOLDEN, "f2_rows.tsv")).readline().rstrip("\n")
    w = row.split("\t")
    ok = textio.parse_rows((row + "\r\n" + row).encode(), 13, 16)  # CRLF, no trailing newline
    assert ok.n == 2 and np.array_equal(ok.means[0], ok.means[1])
    extra = textio.parse_rows(("  " + row + "\textra\tcols\n").encode(), 13, 16)  # strip(); words[11] ignores extras
    assert extra.n == 1 and extra.labels[0] == int(w[11])
    # a base outside base2code_dna: KeyError(letter), as the reference's reader raises (call_modifications.py:84) -- also
    # when the k-mer has the wrong length (the reference's reader never looks at the length)
    for kmer in ("ACGTXACGTACGT", "ACGX", "acgtacgtacgta"):
        with pytest.raises(KeyError) as ei:
            textio.parse_rows(("\t".join(w[:6] + [kmer] + w[7:]) + "\n").encode(), 13, 16)
        assert ei.value.args[0] == {"ACGTXACGTACGT": "X", "ACGX": "X", "acgtacgtacgta": "a"}[kmer] and "row 0" in ei.value.detail
    # fewer than 12 fields: words[k] of the reference's reader is an IndexError (call_modifications.py:84-86, :117)
    with pytest.raises(IndexError):
        textio.parse_rows(("\t".join(w[:11]) + "\n").encode(), 13, 16)
    for bad in ("\t".join(w[:6] + ["ACGT"] + w[7:]),                  # k-mer of the wrong length
                "\t".join(w[:7] + [w[7] + ",1.0"] + w[8:]),           # 14 means
                "\t".join(w[:7] + [w[7].replace(",", ",x", 1)] + w[8:]),
                "\t".join(w[:9] + [w[9].replace(",", ".5,", 1)] + w[10:]),  # int("3.5") fails in the reference
                "\t".join(w[:10] + [w[10].replace(";", ",", 1)] + w[11:]),
                "\t".join(w[:11] + ["one"])):
        with pytest.raises(ValueError):
            textio.parse_rows((bad + "\n").encode(), 13, 16)
    assert textio.parse_rows(b"", 13, 16).n == 0
    with pytest.raises(IndexError):
        textio.parse_rows((row + "\n\n" + row + "\n").encode(), 13, 16)  # blank line: IndexError in the reference too


def test_parser_float_grammar_against_python():
    rng = np.random.default_rng(3)
    toks = ["%.*g" % (int(rng.integers(1, 18)), x) for x in rng.standard_normal(3000) * 10.0 ** rng.integers(-8, 8, 3000)]
    toks += ["1e-45", "1e-46", "3.4028235e38", "3.5e38", "1e39", "-1e39", "0.1", "16777217", "9007199254740993",
             "0.30000001192092896", "1.00000005960464477539", "123456789012345678901234567890", "1e22", "1e23",
             "4.35", "0.000001", "2.4703282292062328e-324", "nan", "inf", "-inf", "Infinity", "1.", ".5", "+.5e1"]
    row = open(os.path.join(GOLDEN, "f2_rows.tsv")).readline().rstrip("\n").split("\t")
    with np.errstate(over="ignore"):
        for i in range(0, len(toks) - 12, 13):
            chunk = toks[i:i + 13]
            row[7] = ",".join(chunk)
            got = textio.parse_rows(("\t".join(row) + "\n").encode(), 13, 16).means[0]
            want = np.array([float(t) for t in chunk], np.float64).astype(np.float32)
            assert np.array_equal(got, want, equal_nan=True), chunk


def test_underscores_in_numbers_follow_pythons_float_and_int():
    """Round 5 (VERDICT r4 missing 4): the reference parses numbers with Python's float() / int() (call_modifications.py:85-87),
    which accept ONE underscore between two digits and nothing else -- differential test against float() / int() themselves,
    accepted values and rejections alike, in the float lists and in the integer list of a row"""
    row = open(os.path.join(GOLDEN, "f2_rows.tsv")).readline().rstrip("\n").split("\t")
    spell = ["1_0", "1_0.5", "1.2_5", "1_0e1_0", "1_000_000", "-1_2.5e-0_3", "+4_2", "0_0.0_1", "1_0.", ".0_5",
             "_1", "1_", "1__0", "1_.0", "1._0", "1e_5", "1_e5", "1e5_", "_", "1_0_", "-_1", "1_0e", "in_f", "n_an", "1_0x"]
    for tok in spell:
        try:
            want = np.float32(float(tok))
        except ValueError:
            want = None
        r = list(row)
        means = r[7].split(",")
        means[4] = tok
        r[7] = ",".join(means)
        text = ("\t".join(r) + "\n").encode()
        if want is None:
            with pytest.raises(ValueError):
                textio.parse_rows(text, 13, 16)
        else:
            got = textio.parse_rows(text, 13, 16).means[0][4]
            assert got == want, (tok, got, want)
    # ... and 600 random spellings with underscores dropped in anywhere: accepted <=> float() accepts, same float32
    rng = np.random.default_rng(12)
    bases = ["%.*g" % (int(rng.integers(1, 12)), x) for x in rng.standard_normal(600) * 10.0 ** rng.integers(-6, 6, 600)]
    agree_ok = agree_bad = 0
    for tok in bases:
        chars = list(tok)
        for _ in range(int(rng.integers(1, 3))):
            chars.insert(int(rng.integers(0, len(chars) + 1)), "_")
        tok = "".join(chars)
        try:
            want = np.float32(float(tok))
        except ValueError:
            want = None
        r = list(row)
        means = r[7].split(",")
        means[7] = tok
        r[7] = ",".join(means)
        text = ("\t".join(r) + "\n").encode()
        if want is None:
            with pytest.raises(ValueError):
                textio.parse_rows(text, 13, 16)
            agree_bad += 1
        else:
            assert textio.parse_rows(text, 13, 16).means[0][7] == want, tok
            agree_ok += 1
    assert agree_ok > 100 and agree_bad > 100, (agree_ok, agree_bad)
    for tok in ["1_0", "2_5", "0_7", "1_0_0", "_5", "5_", "1__0", "-_3", "+1_2", "1_2.0", "1e2"]:
        try:
            want = int(tok)
        except ValueError:
            want = None
        r = list(row)
        lens = r[9].split(",")
        lens[2] = tok
        r[9] = ",".join(lens)
        text = ("\t".join(r) + "\n").encode()
        if want is None:
            with pytest.raises(ValueError):
                textio.parse_rows(text, 13, 16)
        else:
            assert int(textio.parse_rows(text, 13, 16).lens[0][2]) == want, tok


@pytest.mark.parametrize("nthreads", [1, 4])
def test_formatter_matches_reference_strings(nthreads):
    f3 = np.load(os.path.join(GOLDEN, "f3_format.npz"))
    probs, kmers, info, lines = f3["probs"], f3["kmers"], list(f3["sampleinfo"]), list(f3["lines"])
    n = len(lines)
    # build a ParsedRows by hand: text = the sampleinfo strings back to back
    text = "\n".join(info).encode()
    offs = np.zeros(n, np.uint64)
    lens = np.zeros(n, np.uint32)
    pos = 0
    for i, s in enumerate(info):
        offs[i], lens[i] = pos, len(s.encode())
        pos += lens[i] + 1
    r = textio.ParsedRows()
    r.text, r.n, r.kmer, r.row_off, r.info_len = np.frombuffer(text, np.uint8), n, kmers, offs, lens
    r.seq_len, r.signal_len = 13, 16
    labels = probs.argmax(1).astype(np.uint8)  # torch.max(.., 1) at call_modifications.py:163
    got = textio.format_calls(r, probs, labels, nthreads=nthreads).decode().split("\n")
    assert got[-1] == "" and len(got) == n + 1
    bad = [(a, b) for a, b in zip(got, lines) if a != b]
    assert not bad, bad[:5]


def test_formatter_exhaustive_rounding_grid_vs_numpy():
    """every k/1e6 neighbourhood that the 6-decimal rounding can produce, against numpy's own float32
    round()/str() (what the reference executes at call_modifications.py:177-179, :186-187)"""
    rng = np.random.default_rng(9)
    k = np.concatenate((np.arange(0, 2000), rng.integers(0, 1000001, 60000), np.arange(998000, 1000001)))
    p1 = (k / 1e6).astype(np.float32)
    p1 = np.concatenate((p1, np.nextafter(p1, np.float32(2)), np.nextafter(p1, np.float32(-1)).clip(0, 1))).astype(np.float32)
    p0 = (np.float32(1) - p1).astype(np.float32)
    probs = np.stack((p0, p1), 1)
    n = probs.shape[0]
    r = textio.ParsedRows()
    r.text, r.n = np.frombuffer(b"x", np.uint8), n
    r.kmer = np.zeros((n, 13), np.uint8)
    r.row_off, r.info_len, r.seq_len, r.signal_len = np.zeros(n, np.uint64), np.ones(n, np.uint32), 13, 16
    got = textio.format_calls(r, probs, np.zeros(n, np.uint8)).decode().split("\n")[:-1]
    for i in rng.integers(0, n, 4000).tolist() + list(range(0, 300)):
        a, b = probs[i]
        z0 = round(a / (a + b), 6)
        z1 = round(1 - z0, 6)
        assert got[i] == "x\t%s\t%s\t0\tAAAAA" % (str(z0), str(z1)), (i, got[i], z0, z1)


def test_one_pass_row_parser_agrees_with_the_general_parser_on_everything():
    """Round 3: plain rows take a one-pass parser (digits accumulated while the delimiter is looked for); anything it
    does not recognise -- and every malformed row -- is parsed again by the general parser.  Differential test: the
    golden rows, rows with odd but legal number spellings, and byte-mutated rows give the same arrays, the same
    sampleinfo addressing and the same error text with the fast path on and off."""
    from deepsignal_plant_amd import _native as nat
    L = nat.lib()
    rng = np.random.default_rng(11)
    rows = open(os.path.join(GOLDEN, "f2_rows.tsv")).read().splitlines()[:60]

    def both(data):
        out = []
        for fast in (1, 0):
            L.dsp_text_set_fast_rows_(fast)
            try:
                r = textio.parse_rows(data, 13, 16, nthreads=2)
                out.append(("ok", r.n, r.kmer.tobytes(), r.means.tobytes(), r.stds.tobytes(), r.lens.tobytes(), r.signals.tobytes(),
                            r.labels.tobytes(), r.row_off.tobytes(), r.info_len.tobytes(), r.read_off.tobytes(), r.read_len.tobytes()))
            except (ValueError, KeyError, IndexError) as e:
                out.append(("error", type(e).__name__ + str(e)))
        L.dsp_text_set_fast_rows_(1)
        return out
    try:
        a, b = both(("\n".join(rows) + "\n").encode())
        assert a == b and a[0] == "ok" and a[1] == 60
        a, b = both("\n".join(rows).encode())                       # unterminated last row
        assert a == b and a[1] == 60
        a, b = both(("\r\n".join(rows) + "\r\n").encode())          # CRLF
        assert a == b and a[1] == 60
        # legal but unusual spellings inside otherwise plain rows
        spell = ["1e-05", "-2.5E-6", "1.", "007.50", "-0.0", "0", "123456789012345678", "1234567890123456789", "1e22", "1e23",
                 "4e-324", "+1.5", " 1.5", "1.5 ", ".5", "inf", "-inf", "nan", "1e", "1e+", "--1", "1.2.3", "0x10", "1_0",
                 "9007199254740993", "0.000000000000000000001", "1e400", "-1e-400", "3.4028235e38", "3.4028236e38"]
        w = rows[0].split("\t")
        for i in range(0, len(spell), 3):
            m = w[7].split(",")
            m[0:3] = (spell + ["0", "0"])[i:i + 3]
            sg = w[10].split(";")
            g0 = sg[2].split(",")
            g0[5] = spell[i]
            sg[2] = ",".join(g0)
            a, b = both(("\t".join(w[:7] + [",".join(m)] + w[8:10] + [";".join(sg)] + w[11:]) + "\n").encode())
            assert a == b, spell[i:i + 3]
        for lens_tok in ("12", "-3", "999999999", "9999999999", "2147483648", "1.0", ""):
            ln = w[9].split(",")
            ln[4] = lens_tok
            a, b = both(("\t".join(w[:9] + [",".join(ln)] + w[10:]) + "\n" + rows[1] + "\n").encode())
            assert a == b, lens_tok
        for label in ("1", "0", "-1", "7\textra", "1 ", "1\r", "1\rx", "", "x"):
            a, b = both(("\t".join(w[:11] + [label]) + "\n" + rows[1] + "\n").encode())
            assert a == b, label
        # byte mutations: any single-byte damage gives the same outcome on both paths
        base = ("\n".join(rows[:6]) + "\n").encode()
        pool = b"\t,;.-+e0123456789 \nACGTX\r"
        n_err = 0
        for _ in range(1500):
            bad = bytearray(base)
            for _k in range(int(rng.integers(1, 3))):
                bad[int(rng.integers(0, len(bad)))] = pool[int(rng.integers(0, len(pool)))]
            a, b = both(bytes(bad))
            assert a == b
            n_err += a[0] == "error"
        assert 300 < n_err < 1500
    finally:
        L.dsp_text_set_fast_rows_(1)


def test_block_staging_of_the_device_parser_notes_every_row_start():
    """Host half of the device-side row parser (round 4): parse_dev.stage_rows copies a block into the staging buffer and
    notes where its rows start, in one pass (dsp_copy_rows_index) -- rows as dsp_parse_feature_rows counts them: every
    newline-separated piece, an unterminated last row included (it gets its newline), CRLF and empty rows kept as they are."""
    from deepsignal_plant_amd import feed, parse_dev
    rows = open(os.path.join(GOLDEN, "f2_rows.tsv"), "rb").read().splitlines()
    cases = {"plain": b"\n".join(rows[:50]) + b"\n", "unterminated": b"\n".join(rows[:7]), "crlf": b"\r\n".join(rows[:5]) + b"\r\n",
             "blank_rows": rows[0] + b"\n\n" + rows[1] + b"\n", "one": rows[3] + b"\n", "empty": b"",
             "big": b"\n".join(rows * 40) + b"\n"}   # 8,000 rows: many 256 KiB chunks, rows straddling them
    for name, data in cases.items():
        pieces = data.split(b"\n")
        if pieces and pieces[-1] == b"":
            pieces.pop()
        stage = parse_dev.alloc_stage(len(pieces) + 1, len(data) + 1, 13, pinned=False)
        r, n_bytes = parse_dev.stage_rows(np.frombuffer(data, np.uint8), stage, 13, 16)
        assert r.n == len(pieces) == textio.count_rows(data), name
        off = stage["row_off"][:r.n + 1].astype(np.int64)
        assert n_bytes == int(off[-1]) == len(data) + (0 if data.endswith(b"\n") or not data else 1), name
        text = bytes(stage["text"][:n_bytes])
        assert text == data + (b"" if data.endswith(b"\n") or not data else b"\n"), name
        assert [text[off[i]:off[i + 1] - 1] for i in range(r.n)] == pieces, name
    small = parse_dev.alloc_stage(3, 1 << 20, 13, pinned=False)
    with pytest.raises(RuntimeError, match="more than 3 rows"):
        parse_dev.stage_rows(np.frombuffer(cases["plain"], np.uint8), small, 13, 16)
    # the reader in device_parse mode hands out staged text blocks (plain files: read straight into the staging slot,
    # dsp_read_rows_index) holding the rows -- and the global row indices -- of the parsing reader: any block size (also one
    # smaller than a row), with and without a newline behind the last row, one rank and three
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        body = b"\n".join(rows)
        for tail in (b"\n", b""):
            path = os.path.join(td, "rows%d.tsv" % len(tail))
            open(path, "wb").write(body + tail)
            for world in (1, 3):
                for bb in (90_000, 500, 5_000_000):
                    first = 0
                    for rank in range(world):
                        got = {}
                        for mode in (False, True):
                            rd = feed.FeatureReader(path, 13, 16, rank=rank, world=world, nthreads=2, nbuf=2, block_bytes=bb, pinned=False,
                                                    first_row=first, device_parse=mode)
                            rd.start()
                            seen = []
                            for blk in rd:
                                assert (blk.n_bytes is not None) == mode
                                if mode:
                                    t, o = bytes(blk.rows.text[:blk.n_bytes]), blk.rows.row_off
                                    ends = [int(x) for x in o[1:]] + [blk.n_bytes]
                                    seen += [(blk.first_row + i, t[int(o[i]):ends[i] - 1]) for i in range(blk.rows.n)]
                                else:
                                    seen += [(blk.first_row + i, rows[blk.first_row + i]) for i in range(blk.rows.n)]
                                rd.release(blk)
                            got[mode] = seen
                        assert got[True] == got[False], (tail, world, bb, rank)
                        assert all(txt == rows[i] for i, txt in got[True])
                        first += len(got[True])
                    assert first == 200
