"""GPU: the `deepsignal_plant call_mods` CLI end to end (feature TSV in -> per-read-call TSV out) against
the output of the reference's own TSV branch captured with pinned (zero) initial states
(tests/golden/f4_expected.tsv, made by tests/golden/make_golden_text.py)."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def _ckpt(tmp_path, seed=23, scale=2.0):
    import torch
    from oracle import forward_np as onp
    cfg = onp.OracleConfig()
    w = onp.make_weights(cfg, seed, scale)
    p = os.path.join(str(tmp_path), "model.ckpt")
    torch.save({k: torch.from_numpy(v) for k, v in w.items()}, p)  # a bare state_dict, like train.py:161-164
    return p


def _keyed(lines):
    out = {}
    for l in lines:
        w = l.rstrip("\n").split("\t")
        out[(w[0], w[1], w[2], w[4])] = (float(w[6]), float(w[7]), int(w[8]), w[9], "\t".join(w[:6]))
    return out


def _run_cli(args, env=None):
    cmd = [sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant", "call_mods"] + args
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("gz_in,gz_out", [(False, False), (True, True)])
def test_cli_matches_reference_tsv_branch(tmp_path, gz_in, gz_out):
    ck = _ckpt(tmp_path)
    inp = os.path.join(GOLDEN, "f2_rows.tsv" + (".gz" if gz_in else ""))
    out = os.path.join(str(tmp_path), "calls.tsv")
    args = ["-i", inp, "-m", ck, "-o", out, "--init_state", "zeros", "-p", "4"] + (["--gzip"] if gz_out else [])
    r = _run_cli(args)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "[main] call_mods costs" in r.stdout
    if gz_out:
        out += ".gz"
        got_lines = gzip.open(out, "rt").read().splitlines()
    else:
        got_lines = open(out).read().splitlines()
    want_lines = open(os.path.join(GOLDEN, "f4_expected.tsv")).read().splitlines()
    assert len(got_lines) == len(want_lines) == 200
    got, want = _keyed(got_lines), _keyed(want_lines)
    assert set(got) == set(want)
    for k, (p0, p1, lab, kmer5, info) in want.items():
        g = got[k]
        assert g[3] == kmer5 and g[4] == info
        # the printed values are rounded to 6 decimals; fp32 summation-order differences (<= ~2e-7) can move
        # the last printed digit, so compare numerically: 1e-4 is the contract, 2e-6 what is observed
        assert abs(g[0] - p0) <= 2e-6 and abs(g[1] - p1) <= 2e-6
        if abs(p1 - 0.5) >= 1e-4:
            assert g[2] == lab
    # rows come out in input order
    assert [l.split("\t")[1] for l in got_lines] == [l.split("\t")[1] for l in open(os.path.join(GOLDEN, "f2_rows.tsv")).read().splitlines()]


def test_cli_randn_mode_is_batching_invariant_and_matches_oracle(tmp_path):
    """default --init_state randn: results keyed by global row index -> identical for different block sizes,
    and equal to the oracle's same generator"""
    from deepsignal_plant_amd import textio
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    ck = _ckpt(tmp_path)
    inp = os.path.join(GOLDEN, "f2_rows.tsv")
    outs = []
    for i, blk in enumerate(("100000", "")):
        out = os.path.join(str(tmp_path), "c%d.tsv" % i)
        r = _run_cli(["-i", inp, "-m", ck, "-o", out, "--seed", "77"], env={"DSP_BLOCK_BYTES": blk} if blk else None)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(open(out).read())
    assert outs[0] == outs[1]
    cfg = onp.OracleConfig()
    w = onp.make_weights(cfg, 23, 2.0)
    rows = textio.parse_rows(open(inp, "rb").read(), 13, 16)
    lo, po = oc.forward(cfg, w, rows.kmer.astype(np.float32), rows.means, rows.stds, rows.lens.astype(np.float32),
                        rows.signals, init_mode="philox", seed=77)
    got = np.array([[float(x) for x in l.split("\t")[6:8]] for l in outs[0].splitlines()])
    assert np.abs(got[:, 1] - po[:, 1] / (po[:, 0] + po[:, 1])).max() <= 2e-6


def test_cli_errors(tmp_path):
    ck = _ckpt(tmp_path)
    r = _run_cli(["-i", "/nonexistent.tsv", "-m", ck, "-o", os.path.join(str(tmp_path), "o.tsv")])
    assert r.returncode != 0 and "--input_path does not exist!" in r.stderr
    r = _run_cli(["-i", os.path.join(GOLDEN, "f2_rows.tsv"), "-m", "/nonexistent.ckpt", "-o", os.path.join(str(tmp_path), "o.tsv")])
    assert r.returncode != 0 and "--model_path is not set right!" in r.stderr
    bad = tmp_path / "bad.tsv"
    bad.write_text("chr1\t1\t+\t1\tr\tt\tACGTXACGTACGT\t0\t0\t0\t0\t0\n")
    r = _run_cli(["-i", str(bad), "-m", ck, "-o", os.path.join(str(tmp_path), "o.tsv")])
    assert r.returncode != 0 and "malformed feature row" in r.stderr


def test_cli_two_ranks_give_the_same_calls_as_one(tmp_path):
    """range-shard invariance (SURVEY.md 4 'Multi-GPU'): 2 ranks (sharing this box's GPU, control plane over
    gloo) write byte-identical output to 1 rank, in the default randn (Philox) mode"""
    import socket
    ck = _ckpt(tmp_path)
    inp = os.path.join(GOLDEN, "f2_rows.tsv")
    one = os.path.join(str(tmp_path), "one.tsv")
    r = _run_cli(["-i", inp, "-m", ck, "-o", one, "--seed", "5"])
    assert r.returncode == 0, r.stderr[-2000:]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    two = os.path.join(str(tmp_path), "two.tsv")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), "-m", "deepsignal_plant_amd.deepsignal_plant", "call_mods",
           "-i", inp, "-m", ck, "-o", two, "--seed", "5"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(one).read() == open(two).read()
    assert not [f for f in os.listdir(str(tmp_path)) if ".part" in f]


def test_cli_fused_freq_file_equals_call_freq_on_the_result(tmp_path):
    """call_mods --freq_file (aggregated straight from GPU results) == call_freq run on the per-read result
    file, for the tsv and the sorted bedMethyl flavours"""
    ck = _ckpt(tmp_path)
    inp = os.path.join(GOLDEN, "f2_rows.tsv")
    # many reads per site: reuse the 200 rows with positions folded onto a few sites
    rows = open(inp).read().splitlines()
    folded = []
    for i, l in enumerate(rows * 6):
        w = l.split("\t")
        w[1] = str(1000 + 7 * (i % 23))
        w[4] = "r%d" % i
        folded.append("\t".join(w))
    inp2 = os.path.join(str(tmp_path), "folded.tsv")
    open(inp2, "w").write("\n".join(folded) + "\n")
    out = os.path.join(str(tmp_path), "calls.tsv")
    for flags, cf_flags in (([], []), (["--freq_bed", "--freq_sort"], ["--bed", "--sort"])):
        fq = os.path.join(str(tmp_path), "fused.freq")
        r = _run_cli(["-i", inp2, "-m", ck, "-o", out, "--freq_file", fq, "--prob_cf", "0.02"] + flags)
        assert r.returncode == 0, r.stderr[-2000:]
        fq2 = os.path.join(str(tmp_path), "two_step.freq")
        cmd = [sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant", "call_freq", "-i", out, "-o", fq2,
               "--prob_cf", "0.02"] + cf_flags
        r2 = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert r2.returncode == 0, r2.stderr[-2000:]
        assert open(fq, "rb").read() == open(fq2, "rb").read() and os.path.getsize(fq) > 0


def test_cli_binary_feature_container_gives_the_same_calls_as_the_tsv(tmp_path):
    """call_mods on a .dspf (pack_features of the TSV) writes byte-identical per-read calls,
    on one rank and on two (block-range sharding, no collective for the row offsets)"""
    import socket
    from deepsignal_plant_amd import featfile
    ck = _ckpt(tmp_path)
    inp = os.path.join(GOLDEN, "f2_rows.tsv")
    ref_out = os.path.join(str(tmp_path), "tsv.tsv")
    ref_fq = os.path.join(str(tmp_path), "tsv.freq")
    r = _run_cli(["-i", inp, "-m", ck, "-o", ref_out, "--seed", "9", "--freq_file", ref_fq, "--prob_cf", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    packed = os.path.join(str(tmp_path), "f2.dspf")
    assert featfile.pack_features(inp, packed, block_rows=48) == 200
    out = os.path.join(str(tmp_path), "bin.tsv")
    fq = os.path.join(str(tmp_path), "bin.freq")
    r = _run_cli(["-i", packed, "-m", ck, "-o", out, "--seed", "9", "--freq_file", fq, "--prob_cf", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(out, "rb").read() == open(ref_out, "rb").read()
    assert open(fq, "rb").read() == open(ref_fq, "rb").read() and os.path.getsize(fq) > 0
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    two = os.path.join(str(tmp_path), "bin2.tsv")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), "-m", "deepsignal_plant_amd.deepsignal_plant", "call_mods",
           "-i", packed, "-m", ck, "-o", two, "--seed", "9"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert open(two, "rb").read() == open(ref_out, "rb").read()
    bad = os.path.join(str(tmp_path), "bad.dspf")
    open(bad, "wb").write(open(packed, "rb").read()[:5000])
    r = _run_cli(["-i", bad, "-m", ck, "-o", out])
    assert r.returncode != 0 and "dsp_feat_open" in r.stderr and "truncated" in r.stderr
