"""GPU: `deepsignal_plant extract` and `call_mods` on a directory of REAL fast5 files (fixture F7: HDF5 files written with
h5py; expected rows = the reference's own _extract_features + _features_to_str run on them with h5py and statsmodels'
robust.mad, tests/golden/make_golden_fast5.py).  Every column the reference computes deterministically must be
byte-identical; the signal groups of bases longer than signal_len (which the reference subsamples with the unseeded global
`random`, extract_features.py:247-249) must be an order-preserving selection of that base's own normalised samples."""
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from deepsignal_plant_amd import reads as R
from oracle import extract_np as ox
from tests.helpers import ROOT

pytestmark = pytest.mark.gpu
F7 = os.path.join(ROOT, "tests", "golden", "fast5")
EXPECT = json.load(open(os.path.join(F7, "expect.json")))


def _need_hdf5():
    """fail if the image's HDF5 library is there but was not loaded; skip on a host that has none"""
    if not R.fast5_available():
        from deepsignal_plant_amd import _native as nat
        if os.path.exists("/opt/conda/lib/libhdf5.so"):
            pytest.fail("the image's libhdf5 was not loaded: " + nat.last_error())
        pytest.skip("no HDF5 library on this host: " + nat.last_error())


def _ref_fasta(tmp_path):
    fa = tmp_path / "ref.fa"
    with open(fa, "w") as f:
        for c, n in EXPECT["chrom_len"].items():
            f.write(">%s synthetic\n" % c)
            for o in range(0, n, 60):
                f.write("N" * min(60, n - o) + "\n")
    return str(fa)


@pytest.mark.parametrize("name", sorted(EXPECT["cases"]))
def test_extract_cli_on_fast5_files_writes_the_reference_rows(name, tmp_path):
    _need_hdf5()
    c = EXPECT["cases"][name]
    out = str(tmp_path / "feats.tsv")
    cmd = [sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant", "extract", "-i", os.path.join(F7, "reads"), "-o", out,
           "--normalize_method", c["method"], "--motifs", c["motifs"], "--mod_loc", str(c["mod_loc"]), "--seq_len", str(c["k"]),
           "--signal_len", str(c["s"]), "--methy_label", str(c["label"]), "--f5_batch_size", "2"]
    if c["c2l"]:
        cmd += ["--reference_path", _ref_fasta(tmp_path)]
    if c["region"]:
        cmd += ["--region", c["region"]]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "%d of %d read files failed" % (c["errors"], len(EXPECT["files"])) in r.stdout
    got = open(out).read().splitlines()
    want = gzip.open(os.path.join(F7, "expect_%s.tsv.gz" % name), "rt").read().splitlines()
    assert len(got) == len(want) == c["rows"]
    reads = {}
    for rel, info in EXPECT["files"].items():
        if info["ok"] and info["chrom"]:
            rd = R.from_fast5(os.path.join(F7, "reads", rel))
            reads[rd.readname] = rd
    n_long = n_short = 0
    half = (c["k"] - 1) // 2
    for g, w in zip(got, want):
        fg, fw = g.split("\t"), w.split("\t")
        assert fg[:10] == fw[:10] and fg[11] == fw[11], (fg[:7], fw[:7])
        lens = [int(x) for x in fw[9].split(",")]
        gg, gw = fg[10].split(";"), fw[10].split(";")
        rd = norm = None
        for j, n in enumerate(lens):
            if n <= c["s"]:
                assert gg[j] == gw[j]
                n_short += 1
                continue
            if rd is None:  # the base's own samples, normalised as the reference does (oracle restatement, pinned by F6)
                rd = reads[fw[4]]
                norm = ox.normalize_signals(ox.rescale_signals(rd.raw, rd.scaling, rd.offset), c["method"])
                pos = int(fw[1])
                loc = (rd.chrom_start + len(rd.seq) - 1 - pos) if rd.alignstrand == "-" else pos - rd.chrom_start
            b = loc - half + j
            base = [str(x) for x in np.around(norm[int(rd.ev_start[b]):int(rd.ev_start[b] + rd.ev_len[b])], decimals=6)]
            for vals in (gg[j].split(","), gw[j].split(",")):  # this build's selection and the reference's own
                assert len(vals) == c["s"]
                p = 0
                for v in vals:
                    while p < len(base) and base[p] != v:
                        p += 1
                    assert p < len(base), (fw[:7], j)
                    p += 1
            n_long += 1
    assert n_short > 1000 and n_long > 10


def test_call_mods_on_fast5_files_equals_call_mods_on_the_same_reads_as_records(tmp_path):
    """call_mods -i <dir of fast5> (the reference's fast5 branch, call_modifications.py:559-583) == the same reads handed
    over as read records (the path the extraction tests pin against the oracle)"""
    import torch
    from oracle import forward_np as onp
    _need_hdf5()
    w = onp.make_weights(onp.OracleConfig(), 23, 2.0)
    ck = str(tmp_path / "m.ckpt")
    torch.save({k: torch.from_numpy(v) for k, v in w.items()}, ck)
    files = R.list_read_files(os.path.join(F7, "reads"))
    d = tmp_path / "records"
    d.mkdir()
    for i, p in enumerate(files):  # one record file per fast5, same order => same subsampler keys
        try:
            R.save_reads(str(d / ("%04d.reads.npz" % i)), [R.from_fast5(p)])
        except RuntimeError:
            (d / ("%04d.fast5" % i)).write_bytes(b"broken")
    outs = []
    # default --init_state randn: initial states keyed by (read uid, base index in the read), so the two routes and the
    # two reader batch sizes must agree to the byte
    for src, bs in ((os.path.join(F7, "reads"), "1"), (str(d), "30")):
        out = str(tmp_path / ("calls_%d.tsv" % len(outs)))
        cmd = [sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant", "call_mods", "-i", src, "-m", ck, "-o", out,
               "--seed", "9", "--f5_batch_size", bs]
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        assert "4 of 15 read files failed" in r.stdout
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and outs[0].count(b"\n") == EXPECT["cases"]["mad_cg"]["rows"]


def _key_rows(lines, s):
    """rows -> sorted tuples of everything the reference computes deterministically (over-long bases' groups dropped)"""
    out = []
    for ln in lines:
        f = ln.split("\t")
        lens = [int(x) for x in f[9].split(",")]
        groups = f[10].split(";")
        out.append(tuple(f[:10]) + tuple(g if n <= s else "*" for g, n in zip(groups, lens)) + (f[11],))
    return sorted(out)


@pytest.mark.parametrize("name", sorted(EXPECT["cli"]))
def test_extract_cli_equals_the_reference_command_line(name, tmp_path):
    """The reference's real `deepsignal_plant extract` (multi-process, run under the image's python3.9 by
    make_golden_fast5.py) and this build's, same flags, same fast5 directory: plain file, --w_is_dir + --gzip batches,
    --positions + --methy_label 0 + zscore.  Row ORDER is not compared (the reference's depends on process timing)."""
    _need_hdf5()
    c = EXPECT["cli"][name]
    flags = [(_ref_fasta(tmp_path) if x == "<ref.fa>" else os.path.join(F7, "positions.tsv") if x == "<positions.tsv>" else x)
             for x in c["flags"]]
    is_dir = "--w_is_dir" in flags
    out = str(tmp_path / ("out.d" if is_dir else "out.tsv"))
    cmd = [sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant", "extract", "-i", os.path.join(F7, "reads"), "-o", out] + flags
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    if is_dir:
        names = sorted(os.listdir(out))
        assert names[0] == "0.tsv.gz" and all(n.endswith(".tsv.gz") for n in names)  # extract_features.py:474-510
        got = [ln for n in names for ln in gzip.open(os.path.join(out, n), "rt").read().splitlines()]
    else:
        got = open(out).read().splitlines()
    want = gzip.open(os.path.join(F7, "expect_%s.tsv.gz" % name), "rt").read().splitlines()
    assert len(got) == len(want) == c["rows"]
    assert _key_rows(got, 16) == _key_rows(want, 16)
