"""GPU: DSP_SLOT_CANARY=1 (deepsignal_plant_amd/canary.py) -- the systematic check for the slot-reuse class of bug in the host
pipeline of `call_mods`.

Every page-locked input slot and result slot is poisoned (0xFF) the moment its owner releases it, verified intact when the
next owner takes it, and every consumer verifies that what it reads holds no poison.  ONE parametrised run drives the shapes
of the CLI / rank / parser tests -- every input format, both parsers, --freq_file, --gzip, 1 and 2 ranks -- under the canary
with jittered block sizes and writer delays, on small inputs, all in two launches (tests/cli_jobs.py), and the output bytes
must be those of the plain run.  A second test re-enacts round 4's result-slot bug and shows that the canary catches it with
NO slow writer."""
import gzip
import os

import pytest

from tests.helpers import ROOT, run_cli_jobs
from tests.test_gpu_cli import _ckpt, _folded_rows

pytestmark = pytest.mark.gpu


def _inputs(tmp_path, data):
    from deepsignal_plant_amd import gzio
    plain = str(tmp_path / "rows.tsv")
    open(plain, "wb").write(data)
    bgzf = str(tmp_path / "rows_bgzf.tsv.gz")
    with gzio.open_write(bgzf, True, nthreads=2) as wf:
        wf.write(data)
    foreign = str(tmp_path / "rows_foreign.tsv.gz")
    open(foreign, "wb").write(gzip.compress(data, 1))
    return {"plain": plain, "bgzf": bgzf, "foreign_gz": foreign, "dspf": str(tmp_path / "rows.dspf")}


def test_the_pipeline_under_the_slot_canary_writes_the_same_bytes(tmp_path):
    ck = _ckpt(tmp_path)
    data = _folded_rows(n_rep=6)            # 1,200 rows, 23 sites
    paths = _inputs(tmp_path, data)
    common = ["-m", ck, "--seed", "31", "--prob_cf", "0.02"]

    def job(tag, fmt, env, extra=()):
        out = str(tmp_path / ("%s.tsv" % tag))
        return {"argv": ["call_mods", "-i", paths[fmt], "-o", out, "--freq_file", out + ".freq"] + common + list(extra), "env": env,
                "out": out, "tag": tag, "gz": "--gzip" in extra}
    jobs = [{"argv": ["pack_features", "-i", paths["plain"], "-o", paths["dspf"]], "env": {}, "tag": "pack"},
            job("ref", "plain", {})]       # the plain run: no canary, default block size
    # the canary runs: block sizes from a handful of rows to the whole file, a writer slower than the GPU or not
    jitter = [(20000, 0), (61000, 3), (150000, 0), (33000, 11), (400000, 1), (97000, 0), (10_000_000, 0), (25000, 25)]
    i = 0
    for fmt in ("plain", "bgzf", "foreign_gz", "dspf"):
        for parse in ("device", "host"):
            if fmt == "dspf" and parse == "host":
                continue                   # (a container holds parsed rows: one reader path)
            bb, delay = jitter[i % len(jitter)]
            env = {"DSP_SLOT_CANARY": "1", "DSP_BLOCK_BYTES": str(bb), "DSP_WRITER_DELAY_MS": str(delay)}
            jobs.append(job("c%d_%s_%s" % (i, fmt, parse), fmt, env, ["--parse_on", parse] + (["--gzip"] if i % 3 == 2 else [])))
            i += 1
    res = run_cli_jobs(tmp_path, jobs, world=1)
    assert len(res) == len(jobs), (res.proc.stdout[-2000:], res.proc.stderr[-4000:])
    for j, r in zip(jobs, res):
        assert r["rc"] == 0, (j["tag"], r["stderr"][-3000:])
    ref_calls, ref_freq = open(jobs[1]["out"], "rb").read(), open(jobs[1]["out"] + ".freq", "rb").read()
    assert ref_calls.count(b"\n") == 1200 and len(ref_freq) > 0
    for j in jobs[2:]:
        rd = (lambda p: gzip.open(p + ".gz", "rb").read()) if j["gz"] else (lambda p: open(p, "rb").read())
        assert rd(j["out"]) == ref_calls, j["tag"]
        assert rd(j["out"] + ".freq") == ref_freq, j["tag"]
    # two ranks (sharing the GPU), one launch: the range split, the BGZF member split, the shared-memory ring of a foreign .gz
    jobs2 = []
    for i, (fmt, parse) in enumerate([("plain", "device"), ("bgzf", "host"), ("foreign_gz", "device"), ("plain", "host")]):
        bb, delay = jitter[(i + 3) % len(jitter)]
        env = {"DSP_SLOT_CANARY": "1", "DSP_BLOCK_BYTES": str(bb), "DSP_WRITER_DELAY_MS": str(delay)}
        jobs2.append(job("r2_%d_%s_%s" % (i, fmt, parse), fmt, env, ["--parse_on", parse] + (["--gzip"] if i == 1 else [])))
    res2 = run_cli_jobs(tmp_path, jobs2, world=2, tag="jobs2")
    assert len(res2) == len(jobs2) and all(r["rc"] == 0 for r in res2), (res2.proc.stderr[-4000:], [r["stderr"][-1500:] for r in res2])
    for j in jobs2:
        rd = (lambda p: gzip.open(p + ".gz", "rb").read()) if j["gz"] else (lambda p: open(p, "rb").read())
        assert rd(j["out"]) == ref_calls, j["tag"]
        assert rd(j["out"] + ".freq") == ref_freq, j["tag"]
    try:
        d = os.path.join(ROOT, "gpurun_out", "r5")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "slot_canary_runs.txt"), "w") as f:
            for j, r in list(zip(jobs, res)) + list(zip(jobs2, res2)):
                f.write("%-28s %6.2f s  env %s\n" % (j["tag"], r["seconds"], j["env"]))
    except OSError:
        pass


def test_the_canary_catches_a_result_ring_that_goes_round_by_block_number(tmp_path):
    """Round 4's bug: result slots chosen by block number, guarded only by the event of the copy INTO the slot -- the writer
    may still be formatting the slot's previous block.  It needed a writer slower than the GPU to show (and only showed as
    wrong numbers in the last blocks).  DSP_TEST_RESULT_RING_BY_BLOCK=1 re-enacts it (a ring of 2 slots by block number:
    the main loop is ahead of the GPU by more than that at once); under the canary the run ENDS at the first slot taken
    before it was handed back -- no DSP_WRITER_DELAY_MS, no luck needed.  The fixed hand-back ring passes the same run."""
    ck = _ckpt(tmp_path)
    inp = str(tmp_path / "rows.tsv")
    open(inp, "wb").write(_folded_rows(n_rep=3))
    argv = ["call_mods", "-i", inp, "-m", ck, "-o", str(tmp_path / "o.tsv"), "--seed", "5"]
    small = {"DSP_BLOCK_BYTES": "20000", "DSP_SLOT_CANARY": "1"}
    res = run_cli_jobs(tmp_path, [{"argv": argv, "env": dict(small)},
                                  {"argv": argv, "env": dict(small, DSP_TEST_RESULT_RING_BY_BLOCK="2")}])
    assert len(res) == 2
    assert res[0]["rc"] == 0, res[0]["stderr"][-3000:]
    assert res[1]["rc"] != 0 and "SlotCanaryError" in res[1]["stderr"] and "main loop takes result slot" in res[1]["stderr"], res[1]["stderr"][-3000:]
