"""GPU: BASELINE.json configs[3] and configs[4] at EIGHT ranks under the driver's own `pytest -m gpu`.

No 8-GPU node is available to this build, so the eight ranks share the one MI355X (control plane over gloo, exactly as
bench.py / call_mods fall back when fewer GPUs than ranks are visible); everything else is the 8-rank code path:

  configs[3]  bench.py --gpus 8 --gather: the range split of the global site index space over eight ranks
              (call_modifications.py:613-621 is the reference's replica-DP), batch 65,536, and the optional
              final gather of every step's per-site probabilities to rank 0 (dist.gather_probs, point to point);
  configs[4]  torchrun x8 `call_mods --freq_file` in the DEFAULT randn mode on plain text, on a BGZF .gz and on a foreign
              single-member .gz (what the reference's `extract --gzip` writes): per-read file and frequency file
              byte-identical to the one-rank run -- and the same with fewer rows than ranks.
"""
import gzip
import json
import os
import socket
import subprocess
import sys

import pytest

from tests.helpers import ROOT, run_cli_jobs
from tests.test_gpu_cli import _ckpt, _folded_rows, _keep, _run_cli

pytestmark = pytest.mark.gpu

WORLD = 8


def _ranks(args, world=WORLD, timeout=1200, env=None):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), "-m", "deepsignal_plant_amd.deepsignal_plant", "call_mods"] + args
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    e.update(env or {})
    e["DSP_TIMING"] = "1"      # the run's milestones on rank 0's stdout
    import time
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=e)
    dt = time.time() - t0
    if dt > 40:                # eight ranks on a few thousand rows take 4-5 s (cold imports: 10-15): keep what a slow run says
        print("[slow 8-rank run] %.1f s: %s" % (dt, " ".join(args)))
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out", "r5"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "r5", "slow_8rank_runs.txt"), "a") as f:
                f.write("%.1f s: %s\n%s\n%s\n\n" % (dt, " ".join(args), r.stdout[-3000:], r.stderr[-3000:]))
        except OSError:
            pass
    return r


def test_config4_bench_at_eight_ranks_tiles_the_site_space_and_gathers_every_call():
    """`python bench.py --gpus 8 --steps 2 --gather` started plainly: eight ranks of batch 65,536 (configs[3]'s batch; its
    100 M sites are 191 such steps per rank -- the step count is the only thing scaled down), rank_site_ranges tile
    [0, 8 * 2 * 65,536) without gap or overlap in rank order, and the gather brings exactly those rows to rank 0"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(WORLD), "--steps", "2", "--warmup", "1",
                        "--gather", "--no_cpu_baseline"], cwd=ROOT, capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    B = 65536
    assert d["n_gpus"] == WORLD and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["batch"] == B
    ranges = d["config"]["rank_site_ranges"]
    assert len(ranges) == WORLD and ranges[0][0] == 0 and ranges[-1][1] == WORLD * 2 * B
    assert all(ranges[i][1] == ranges[i + 1][0] for i in range(WORLD - 1))          # no gap, no overlap, rank order
    assert all(b - a == 2 * B for a, b in ranges)
    assert d["config"]["sites"] == WORLD * 2 * B and "configs[3]" in d["config"]["workload"]
    assert abs(d["value"] - WORLD * 2 * B / (d["ms_per_step"] * 2e-3)) / d["value"] < 0.01
    g = d["gather"]
    assert g["sites"] == WORLD * 2 * B and g["bytes"] == 8 * g["sites"] and g["to_rank"] == 0
    assert "cpu_baseline" not in d and d["vs_baseline"] is None
    # round 5: the line proves which devices ran it -- every rank's PCI name, uuid, host, CPUs and its
    # OWN ms_per_step; here the eight ranks share ONE GPU and the line says so (backend gloo, distinct_gpus 1)
    c = d["config"]
    assert len(c["ranks"]) == WORLD and [x["rank"] for x in c["ranks"]] == list(range(WORLD))
    for x in c["ranks"]:
        assert set(x) >= {"host", "local_rank", "hip_device", "pci_bdf", "uuid", "numa_node", "cpus", "pinned", "ms_per_step", "name"}
        assert len(x["pci_bdf"].split(":")) == 3 and len(x["uuid"]) == 32 and x["ms_per_step"] > 0
        assert x["ms_per_step"] <= d["ms_per_step"] * 1.0001          # the line's time is the MAX over ranks
    assert c["distinct_gpus"] == 1 and c["backend"].startswith("gloo") and c["rccl_version"] is None
    assert max(x["ms_per_step"] for x in c["ranks"]) >= 0.9 * d["ms_per_step"]
    assert len({x["cpus"] for x in c["ranks"]}) == WORLD or not all(x["pinned"] for x in c["ranks"])   # pinned ranks: disjoint CPU slices
    _keep("bench_8ranks_shared_gpu.json", json.dumps(d) + "\n")


def test_a_scale_line_is_refused_when_ranks_do_not_sit_on_distinct_gpus():
    """what RCCL runs check unconditionally, forced here on the 1-GPU box (DSP_REQUIRE_DISTINCT_GPUS=1): two ranks on one
    device must NOT print a 2-GPU line -- non-zero exit, the reason on stderr.  And --n1_ms puts the driver's own
    efficiency arithmetic on the line of a run that is allowed to print."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "4096",
           "--no_cpu_baseline", "--n1_ms", "3.5"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(env, DSP_REQUIRE_DISTINCT_GPUS="1"))
    assert r.returncode != 0 and "2 ranks but only 1 distinct GPUs" in r.stderr, (r.returncode, r.stderr[-2000:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n1_ms_per_step"] == 3.5 and abs(d["scaling_efficiency_vs_n1"] - 3.5 / d["ms_per_step"]) < 1e-3
    assert d["config"]["distinct_gpus"] == 1


def _inputs(tmp_path, data, tag):
    """the same rows as plain text, as BGZF (this build's own writer) and as one foreign gzip member"""
    from deepsignal_plant_amd import gzio
    plain = str(tmp_path / ("%s.tsv" % tag))
    open(plain, "wb").write(data)
    bgzf = str(tmp_path / ("%s_bgzf.tsv.gz" % tag))
    with gzio.open_write(bgzf, True, nthreads=2) as wf:
        wf.write(data)
    foreign = str(tmp_path / ("%s_foreign.tsv.gz" % tag))
    open(foreign, "wb").write(gzip.compress(data, 1))
    assert gzio.BgzfFile(bgzf).ok and not gzio.BgzfFile(foreign).ok
    return {"plain": plain, "bgzf": bgzf, "foreign_gz": foreign}


COMMON = ["--prob_cf", "0.02", "--seed", "31"]
BLK = {"DSP_BLOCK_BYTES": "150000"}     # many reader blocks per rank (and ~33 ring blocks for the foreign .gz)


def _job(inp, ck, out, env=None, extra=()):
    return {"argv": ["call_mods", "-i", inp, "-m", ck, "-o", out, "--freq_file", out + ".freq"] + COMMON + list(extra),
            "env": dict(env or {}, DSP_TIMING="1"), "out": out}


@pytest.fixture(scope="module")
def eight_rank_runs(tmp_path_factory):
    """Round 5: the eight-rank cases share their launches.  ONE launch of eight ranks runs all six
    configs[4] command lines (three input forms x plain / --gzip output) and ONE more the six fewer-rows-than-ranks ones
    (tests/cli_jobs.py: the ranks, their process group and the HIP runtime are set up once per launch instead of once per
    command line: 10-15 s of cold imports each); the one-rank references run as jobs of one process.  Every command line is
    still `deepsignal_plant call_mods ...` through the CLI's own main()."""
    import time
    tmp_path = tmp_path_factory.mktemp("ranks8")
    ck = _ckpt(tmp_path)
    data = _folded_rows(n_rep=12)
    paths = _inputs(tmp_path, data, "rows")
    few = {}
    for n_rows in (5, 1):
        few[n_rows] = _inputs(tmp_path, b"".join(_folded_rows(n_rep=1).splitlines(True)[:n_rows]), "few%d" % n_rows)
    ref_jobs = [_job(paths["plain"], ck, str(tmp_path / "rows_one.tsv"), BLK)]
    ref_jobs += [_job(few[n]["plain"], ck, str(tmp_path / ("few%d_one.tsv" % n))) for n in (5, 1)]
    t0 = time.time()
    refs = run_cli_jobs(tmp_path, ref_jobs, world=1, tag="refs")
    assert len(refs) == 3 and all(r["rc"] == 0 for r in refs), (refs.proc.stderr[-3000:], [r["stderr"][-2000:] for r in refs])
    big = {}
    for fmt in ("plain", "bgzf", "foreign_gz"):
        big[(fmt, False)] = _job(paths[fmt], ck, str(tmp_path / ("eight_%s.tsv" % fmt)), BLK)
        big[(fmt, True)] = _job(paths[fmt], ck, str(tmp_path / ("eight_%s_gz.tsv" % fmt)), BLK, ["--gzip"])
    res_big = run_cli_jobs(tmp_path, list(big.values()), world=WORLD, tag="big", timeout=1200)
    small = {}
    for n in (5, 1):
        for fmt in ("plain", "bgzf", "foreign_gz"):
            small[(n, fmt)] = _job(few[n][fmt], ck, str(tmp_path / ("eight_few%d_%s.tsv" % (n, fmt))))
    res_small = run_cli_jobs(tmp_path, list(small.values()), world=WORLD, tag="small", timeout=1200)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out", "r5"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "r5", "ranks8_launches.txt"), "w") as f:
            f.write("3 launches (1 rank x 3 jobs, 8 ranks x 6 jobs, 8 ranks x 6 jobs): %.1f s in all\n" % (time.time() - t0))
            for name, jobs, res in (("refs", ref_jobs, refs), ("big", list(big.values()), res_big), ("small", list(small.values()), res_small)):
                for j, r in zip(jobs, res):
                    f.write("%-6s %6.2f s  %s\n" % (name, r["seconds"], " ".join(j["argv"][1:])))
    except OSError:
        pass
    return dict(tmp=tmp_path, refs=dict(zip(("rows", 5, 1), ref_jobs)), big=big, res_big=dict(zip(big, res_big)), small=small,
                res_small=dict(zip(small, res_small)), procs=(res_big.proc, res_small.proc))


def _read(job, gz=False):
    rd = (lambda p: gzip.open(p + ".gz", "rb").read()) if gz else (lambda p: open(p, "rb").read())
    return rd(job["out"]), rd(job["out"] + ".freq")


@pytest.mark.parametrize("fmt", ["plain", "bgzf", "foreign_gz"])
def test_config5_eight_ranks_write_the_bytes_of_one(eight_rank_runs, fmt):
    """configs[4]'s pipeline (feature TSV -> per-read calls -> call_freq aggregation) with eight ranks, default randn mode:
    2,400 rows / 23 sites.  Plain text is split by byte range, BGZF by member range (row counts from the member
    headers), the foreign .gz is inflated once into the shared-memory ring and dealt block by block; --freq_file is reduced
    on the device (records exchanged by site: one all_to_all over the eight ranks -- since round 5 the SAME ragged
    all_to_all_single that runs on RCCL, dist.exchange_records).  Bytes == the one-rank run on the plain file."""
    R = eight_rank_runs
    ref_calls, ref_freq = _read(R["refs"]["rows"])
    assert ref_calls.count(b"\n") == 2400 and len(ref_freq) > 0
    for gz in (False, True):   # and with the calls gzip-compressed on the way out (whole BGZF members per rank, one end-of-file member)
        r = R["res_big"].get((fmt, gz))
        assert r is not None and r["rc"] == 0, (fmt, gz, R["procs"][0].stderr[-3000:], r and r["stderr"][-3000:])
        assert "2400 sites on 8 GPU(s)" in r["stdout"]
        assert _read(R["big"][(fmt, gz)], gz) == (ref_calls, ref_freq), (fmt, gz)
    left = os.listdir(str(R["tmp"]))
    assert not [f for f in left if ".part" in f or f.endswith(".blocks")]
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("dsp_gz_")]
    _keep("cli_8ranks_%s.txt" % fmt, "8 ranks sharing one GPU, %s input, 2400 rows, --freq_file (device), randn mode: calls and "
          "frequencies byte-identical to one rank, plain and --gzip output\n" % fmt)


@pytest.mark.parametrize("n_rows", [5, 1])
def test_config5_fewer_rows_than_ranks(eight_rank_runs, n_rows):
    """five rows (and one) for eight ranks, in all three input forms: ranks without rows take part in every collective
    (row counts, the exchange of call_freq records, the merge) and the result is the one-rank file"""
    R = eight_rank_runs
    ref_calls, ref_freq = _read(R["refs"][n_rows])
    assert ref_calls.count(b"\n") == n_rows
    for fmt in ("plain", "bgzf", "foreign_gz"):
        r = R["res_small"].get((n_rows, fmt))
        assert r is not None and r["rc"] == 0, (fmt, R["procs"][1].stderr[-3000:], r and r["stderr"][-3000:])
        assert _read(R["small"][(n_rows, fmt)]) == (ref_calls, ref_freq), fmt
    assert not [f for f in os.listdir(str(R["tmp"])) if ".part" in f or f.endswith(".blocks")]
