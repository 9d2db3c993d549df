"""GPU: the feature-row parser on the device (csrc/dsp_parse_dev.hip, round 4) against the host parser
(csrc/dsp_text.cpp = the row grammar of deepsignal_plant/call_modifications.py:76-86, pinned by F2 in tests/test_textio.py).

The device parser takes the plain rows the reference's writer emits and FLAGS everything else; flagged blocks go through the
host parser.  So the property to hold is: every row it accepts has exactly the host parser's values (bit for bit: decimal
-> correctly rounded float64 -> float32), every row the host parser rejects is flagged -- never accepted."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def _device_parse(data, L=13, S=16):
    """-> (status[n], dict of arrays as the device wrote them, host-staged ParsedRows)"""
    import torch
    from deepsignal_plant_amd import parse_dev
    n_rows = data.count(b"\n") + (0 if data.endswith(b"\n") or not data else 1)
    stage = parse_dev.alloc_stage(n_rows + 1, len(data) + 1, L)
    rows, n_bytes = parse_dev.stage_rows(np.frombuffer(data, np.uint8), stage, L, S)
    dp = parse_dev.DeviceRowParser(torch.device("cuda", 0), L, S)
    b, ev = dp.submit(rows, n_bytes, stage, torch.cuda.current_stream())
    ev.synchronize()
    out = {k: b[k][:rows.n].cpu().numpy() for k in ("kmer", "means", "stds", "lens", "signals", "labels", "info_len", "read_off", "read_len", "status")}
    assert (int(stage["_torch"]["n_flagged"][0]) != 0) == bool(out["status"].any())   # (flag events: zero = every row was plain)
    return out, rows, stage


def _same(dev, host, i, j):
    """row i of the device arrays == row j of the host parser's, bit for bit"""
    for k in ("kmer", "means", "stds", "lens", "signals", "labels", "info_len", "read_off", "read_len"):
        a = np.atleast_1d(np.asarray(dev[k][i]))
        b = np.atleast_1d(np.asarray(getattr(host, k)[j])).astype(a.dtype)
        if not np.array_equal(a.view(np.uint8), b.view(np.uint8)):
            return k
    return None


def test_device_parser_reproduces_the_host_parser_on_the_golden_rows():
    from deepsignal_plant_amd import textio
    data = open(os.path.join(GOLDEN, "f2_rows.tsv"), "rb").read()
    host = textio.parse_rows(data, 13, 16)
    dev, rows, stage = _device_parse(data)
    # (row 3 of F2 is the one with the odd spellings -- "+4.25", ".5", "7.", "5e-324" --: not a plain row, flagged)
    assert host.n == rows.n == 200 and list(np.flatnonzero(dev["status"])) == [3]
    ok = dev["status"] == 0
    for i in np.flatnonzero(ok):
        assert _same(dev, host, i, i) is None, i
    # ... and the arrays F2 holds from the reference's own reader
    f2 = np.load(os.path.join(GOLDEN, "f2_parsed.npz"), allow_pickle=True)
    assert np.array_equal(dev["means"][ok], f2["means"].astype(np.float32)[ok]) and np.array_equal(dev["signals"][ok], f2["signals"].astype(np.float32)[ok])
    assert np.array_equal(dev["kmer"][ok], f2["kmers"][ok]) and np.array_equal(dev["lens"][ok], f2["lens"][ok])
    # the writer's view of the block (sampleinfo strings through the staged text + the arrays copied back)
    assert [rows.sampleinfo(i) for i in np.flatnonzero(ok)] == [f2["sampleinfo"][i] for i in np.flatnonzero(ok)]
    # CRLF rows, an unterminated last row, extra columns: plain enough
    lines = data.splitlines()
    lines = lines[10:19]
    # CRLF rows and an unterminated last row are plain enough for the device; extra columns behind the label (the reference
    # ignores them: words[11]) give a row another token count: flagged, the host parser takes the block -- never a wrong value
    for variant, on_device in ((b"\r\n".join(lines) + b"\r\n", True), (b"\n".join(lines), True),
                               (b"\n".join(l + b"\textra\tcols" for l in lines) + b"\n", False)):
        dev, rows, _ = _device_parse(variant)
        host = textio.parse_rows(variant, 13, 16)
        assert rows.n == host.n == 9
        if on_device:
            assert dev["status"].sum() == 0
        assert all(dev["status"][i] == 1 or _same(dev, host, i, i) is None for i in range(9))


def test_device_parser_on_the_extreme_rows_and_every_float_spelling():
    """the extreme legal rows of F1 (values of 1e-3 .. 1e6, all-padding signal rectangles, the whole alphabet) printed with 9
    significant digits, and the float grammar corpus of tests/test_textio.py: accepted tokens carry the host parser's bits,
    the others flag their row"""
    from deepsignal_plant_amd import textio
    from oracle import forward_np as onp
    from tests.helpers import rows_to_tsv
    import tempfile
    cfg = onp.OracleConfig()
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "x.tsv")
        ins = onp.make_extreme_inputs(cfg, 200, 114)
        rows_to_tsv(p, *ins)
        data = open(p, "rb").read()
    host = textio.parse_rows(data, 13, 16)
    dev, rows, _ = _device_parse(data)
    assert dev["status"].sum() == 0 and all(_same(dev, host, i, i) is None for i in range(200))
    assert np.array_equal(dev["means"], ins[1]) and np.array_equal(dev["signals"], ins[4])
    rng = np.random.default_rng(3)
    toks = ["%.*g" % (int(rng.integers(1, 18)), x) for x in rng.standard_normal(6000) * 10.0 ** rng.integers(-8, 8, 6000)]
    toks += ["1e-45", "1e-46", "3.4028235e38", "3.5e38", "1e39", "-1e39", "0.1", "16777217", "9007199254740993", "9007199254740991",
             "0.30000001192092896", "1.00000005960464477539", "123456789012345678901234567890", "1e22", "1e23", "1e-22", "1e-23",
             "4.35", "0.000001", "2.4703282292062328e-324", "nan", "inf", "-inf", "Infinity", "1.", ".5", "+.5e1", "-0", "-0.0",
             "0e99", "0.0e-99", "1E5", "1e+05", "1e05", "12e0003", "1e00004", "000001.5", "1.17549435e-38", "1.4e-45", "7e-46",
             "123456789012345678", "1234567890123456789", "0.123456789012345678", "99999999999999999e-17", "1 ", " 1", "1e", "e5", "--1", ""]
    row = open(os.path.join(GOLDEN, "f2_rows.tsv")).readline().rstrip("\n").split("\t")
    lines, chunks = [], []
    for i in range(0, len(toks) - 12, 13):
        chunk = toks[i:i + 13]
        chunks.append(chunk)
        lines.append("\t".join(row[:7] + [",".join(chunk)] + row[8:]))
    data = ("\n".join(lines) + "\n").encode()
    dev, rows, _ = _device_parse(data)
    n_ok = 0
    for i, chunk in enumerate(chunks):
        try:
            h = textio.parse_rows((lines[i] + "\n").encode(), 13, 16)
        except ValueError:
            assert dev["status"][i] == 1, chunk      # what the host parser rejects is never accepted
            continue
        if dev["status"][i] == 0:
            assert _same(dev, h, i, 0) is None, chunk
            n_ok += 1
    assert n_ok > 100       # (13 tokens per row, every third of up to 17 digits: most rows hold one beyond 2^53 and are flagged)


def test_device_parser_never_accepts_what_the_host_parser_rejects_and_never_differs():
    """byte-mutated rows (the corpus of tests/test_textio.py's differential test, 3,000 mutations): a row is either flagged
    or equal to the host parser's row; rows of a block the host parser rejects as a whole are checked one by one"""
    from deepsignal_plant_amd import textio
    rng = np.random.default_rng(11)
    rows = open(os.path.join(GOLDEN, "f2_rows.tsv")).read().splitlines()[:40]
    pool = b"\t,;.-+eE0123456789 \nACGTNX\r:_"
    blocks = []
    for _ in range(3000):
        k = int(rng.integers(1, 4))
        base = ("\n".join(rows[int(i)] for i in rng.integers(0, len(rows), k)) + "\n").encode()
        bad = bytearray(base)
        for _k in range(int(rng.integers(1, 3))):
            bad[int(rng.integers(0, len(bad)))] = pool[int(rng.integers(0, len(pool)))]
        if not bad.endswith(b"\n"):
            bad += b"\n"
        blocks.append(bytes(bad))
    data = b"".join(blocks)
    dev, staged, _ = _device_parse(data)
    i = 0
    n_flag = n_same = 0
    for blk in blocks:
        pieces = blk.split(b"\n")[:-1]
        for piece in pieces:
            try:
                h = textio.parse_rows(piece + b"\n", 13, 16) if piece else None
            except (ValueError, KeyError, IndexError):
                h = None
            if h is None or h.n != 1:
                assert dev["status"][i] == 1, piece[:80]
                n_flag += 1
            elif dev["status"][i] == 0:
                assert _same(dev, h, i, 0) is None, piece[:80]
                n_same += 1
            else:
                n_flag += 1
            i += 1
    assert i == staged.n and n_same > 2000 and n_flag > 300


def test_device_parser_other_row_shapes_and_the_thread_per_row_kernels():
    """rows of other k-mer lengths / signal windows: short ones through the token-parallel kernel, rows too long for its LDS
    (seq_len 21 x signal_len 24: 5 kB) through the thread-per-row pair -- which DSP_PARSE_KERNEL=rows also selects for the
    default shape (a child process: the switch is read once)"""
    from deepsignal_plant_amd import textio, tsv
    for L, S, n in ((5, 8, 300), (9, 12, 200), (21, 24, 150)):
        data = ("\n".join(tsv.synth_rows(n, seq_len=L, signal_len=S, seed=L)) + "\n").encode()
        host = textio.parse_rows(data, L, S)
        dev, rows, _ = _device_parse(data, L, S)
        assert rows.n == host.n == n and dev["status"].sum() == 0, (L, S)
        assert all(_same(dev, host, i, i) is None for i in range(n)), (L, S)
    code = ("import sys; sys.path.insert(0, %r); import tests.test_gpu_parse as t; "
            "t.test_device_parser_reproduces_the_host_parser_on_the_golden_rows(); "
            "t.test_device_parser_never_accepts_what_the_host_parser_rejects_and_never_differs(); print('rows kernels ok')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, DSP_PARSE_KERNEL="rows"))
    assert r.returncode == 0 and "rows kernels ok" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def _cli(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    cmd = [sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant", "call_mods"] + args
    return subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)


def test_cli_gives_the_same_bytes_whichever_side_parses(tmp_path):
    """call_mods --parse_on device (the default) and --parse_on host: per-read calls and --freq_file byte-identical, for plain
    text in many small blocks, BGZF and a foreign .gz, one rank and two; a block holding an odd-but-legal row (leading
    blank, '+' sign, 20-digit mantissa) takes the host parser for that block and still gives the same bytes; a malformed row
    and an unknown base end the run with the host parser's message"""
    import gzip
    from deepsignal_plant_amd import gzio
    from tests.test_gpu_cli import _ckpt, _folded_rows, _two_ranks
    ck = _ckpt(tmp_path)
    data = _folded_rows(n_rep=6)
    lines = data.split(b"\n")
    w = lines[700].split(b"\t")
    w[7] = b"+" + w[7]                                   # '+0.123...' : legal for float(), not plain
    w[8] = w[8].replace(b",", b",0000000000000000000", 1)  # a 20+ digit mantissa
    lines[700] = b" " + b"\t".join(w)                     # leading blank (line.strip())
    w = lines[333].split(b"\t")
    m = w[7].split(b",")
    m[2] = b"-1_0.2_5"                                    # Python's float(): one underscore between two digits (round 5)
    w[7] = b",".join(m)
    lines[333] = b"\t".join(w)
    data = b"\n".join(lines)
    plain = str(tmp_path / "rows.tsv")
    open(plain, "wb").write(data)
    bg = str(tmp_path / "rows_bgzf.tsv.gz")
    with gzio.open_write(bg, True, nthreads=2) as wf:
        wf.write(data)
    fz = str(tmp_path / "rows_foreign.tsv.gz")
    open(fz, "wb").write(gzip.compress(data, 1))
    blk = {"DSP_BLOCK_BYTES": "120000"}
    # round 5: eleven command lines as jobs of two launches (tests/cli_jobs.py) instead of eleven processes
    from tests.helpers import run_cli_jobs
    T = lambda name: str(tmp_path / name)
    cm = lambda inp, out, extra=(): {"argv": ["call_mods", "-i", inp, "-m", ck, "-o", T(out), "--seed", "3"] + list(extra), "env": blk}
    fq = lambda out: ["--freq_file", T(out + ".freq"), "--prob_cf", "0.02"]
    jobs, names = [], []
    for k, inp in enumerate((plain, bg, fz)):
        for mode in ("host", "device"):
            names.append("o_%d_%s.tsv" % (k, mode))
            jobs.append(cm(inp, names[-1], fq(names[-1]) + ["--parse_on", mode]))
    # errors: the host parser's, whichever side parses
    bad = lines[:]
    wb = bad[300].split(b"\t")
    wb[7] = wb[7].replace(b",", b";", 1)
    bad[300] = b"\t".join(wb)
    open(T("bad_number.tsv"), "wb").write(b"\n".join(bad))
    wb = lines[10].split(b"\t")
    wb[6] = wb[6][:5] + b"X" + wb[6][6:]
    bad = lines[:]
    bad[10] = b"\t".join(wb)
    open(T("bad_base.tsv"), "wb").write(b"\n".join(bad))
    for name in ("bad_number.tsv", "bad_base.tsv"):
        for mode in ("host", "device"):
            jobs.append(cm(T(name), "x.tsv", ["--parse_on", mode]))
    res = run_cli_jobs(tmp_path, jobs, world=1)
    assert len(res) == len(jobs), (res.proc.stdout[-2000:], res.proc.stderr[-3000:])
    ref = None
    for name, r in zip(names, res):
        assert r["rc"] == 0, (name, r["stderr"][-3000:])
        got = (open(T(name), "rb").read(), open(T(name + ".freq"), "rb").read())
        ref = ref or got
        assert got == ref, name
    assert ref[0].count(b"\n") == 1200
    for r, mode in zip(res[6:8], ("host", "device")):
        assert r["rc"] != 0 and "malformed feature row" in r["stderr"] and "signal_means" in r["stderr"], mode
    for r, mode in zip(res[8:10], ("host", "device")):
        assert r["rc"] != 0 and "KeyError: 'X'" in r["stderr"], mode
    res2 = run_cli_jobs(tmp_path, [cm(plain, "two.tsv", fq("two.tsv"))], world=2, tag="two")
    assert len(res2) == 1 and res2[0]["rc"] == 0, (res2.proc.stderr[-3000:], [r["stderr"][-2000:] for r in res2])
    assert (open(T("two.tsv"), "rb").read(), open(T("two.tsv.freq"), "rb").read()) == ref


@pytest.mark.parametrize("mode", ["device", "host"])
def test_a_writer_that_lags_behind_the_gpu_does_not_lose_a_block(tmp_path, mode):
    """The results of a block wait for the writer in one of a few page-locked slots.  A slot may only be refilled once the
    writer has FORMATTED its previous contents -- not merely once the copy into it has finished: with a writer slower than the
    GPU (a single deflate thread, a slow disk; here DSP_WRITER_DELAY_MS) the last blocks of a run used to land in slots the
    writer was still reading (device parser: five reader slots against four result slots).  20 small blocks, the writer 40 ms
    behind on each: the calls must be the bytes of the undelayed run."""
    from tests.test_gpu_cli import _ckpt, _folded_rows
    ck = _ckpt(tmp_path)
    plain = str(tmp_path / "rows.tsv")
    open(plain, "wb").write(_folded_rows(n_rep=6))
    blk = {"DSP_BLOCK_BYTES": "120000"}
    ref, out = str(tmp_path / "ref.tsv"), str(tmp_path / "slow.tsv")
    r = _cli(["-i", plain, "-m", ck, "-o", ref, "--seed", "3", "--parse_on", mode], env=blk)
    assert r.returncode == 0, r.stderr[-3000:]
    r = _cli(["-i", plain, "-m", ck, "-o", out, "--seed", "3", "--parse_on", mode], env=dict(blk, DSP_WRITER_DELAY_MS="40"))
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = open(ref, "rb").read(), open(out, "rb").read()
    assert a.count(b"\n") == 1200 and a == b
    if mode == "device":
        # ... and with two ranks dealing the blocks of a foreign .gz between them, calls deflated on the way out
        import gzip
        from tests.test_gpu_cli import _two_ranks
        fz = str(tmp_path / "rows_foreign.tsv.gz")
        open(fz, "wb").write(gzip.compress(open(plain, "rb").read(), 1))
        r = _two_ranks(["-i", fz, "-m", ck, "-o", out, "--seed", "3", "--gzip"], env=dict(blk, DSP_WRITER_DELAY_MS="40"))
        assert r.returncode == 0, r.stderr[-3000:]
        assert gzip.open(out + ".gz", "rb").read() == a
