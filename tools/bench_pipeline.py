#!/usr/bin/env python3
"""End-to-end `call_mods` throughput (feature TSV -> per-read-call TSV) on this box: BASELINE.json configs[0]
shape (100k synthetic k=13 rows) and a longer file for the steady state.  Prints one JSON line per run."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeline(stderr):
    """the run's milestones (DSP_TIMING=1: seconds since call_mods started)"""
    lines = [l.split(": ", 1)[1] for l in stderr.splitlines() if l.startswith("[call_mods] seconds at")]
    return lines[0] if lines else None


def main():
    import torch
    from deepsignal_plant_amd import synth
    from deepsignal_plant_amd.models import ModelBiLSTM
    work = os.environ.get("DSP_WORK", "/tmp/dsp_pipe")
    os.makedirs(work, exist_ok=True)
    ck = os.path.join(work, "model.ckpt")
    torch.save(synth.random_state_dict(ModelBiLSTM(), seed=1234), ck)
    sizes = [int(x) for x in (sys.argv[1:] or ["100000", "1000000"])]
    for n in sizes:
        tsv = os.path.join(work, "feat_%d.tsv" % n)
        t0 = time.time()
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_tsv.py"), tsv, str(n)])
        gen = time.time() - t0
        for nproc in [int(x) for x in os.environ.get("DSP_BENCH_THREADS", "2,4,8,16").split(",")]:  # host threads of the one rank
            out = os.path.join(work, "calls_%d.tsv" % n)
            t0 = time.time()
            r = subprocess.run([sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant", "call_mods", "-i", tsv, "-m", ck,
                                "-o", out, "-p", str(nproc)], cwd=ROOT, capture_output=True, text=True,
                               env=dict(os.environ, DSP_TIMING="1"))
            wall = time.time() - t0
            assert r.returncode == 0, r.stderr[-3000:]
            inner = [l for l in r.stdout.splitlines() if "call_mods costs" in l][0]
            secs = float(inner.split("costs")[1].split("seconds")[0])
            rows = sum(1 for _ in open(out))
            assert rows == n
            print(json.dumps({"rows": n, "tsv_mb": round(os.path.getsize(tsv) / 1e6, 1), "parse_threads": nproc,
                              "parse_on": os.environ.get("DSP_PARSE_ON", "device"),
                              "call_mods_s": secs, "process_wall_s": round(wall, 2), "sites_per_s": round(n / secs, 1),
                              "text_mb_per_s": round(os.path.getsize(tsv) / 1e6 / secs, 1), "gen_s": round(gen, 1),
                              "timeline": timeline(r.stderr)}), flush=True)
            os.remove(out)
        if os.environ.get("DSP_BENCH_NO_DSPF"):
            continue
        # the same rows as a binary feature container (pack_features): no parsing on the call_mods side
        packed = os.path.join(work, "feat_%d.dspf" % n)
        t0 = time.time()
        subprocess.check_call([sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant", "pack_features", "-i", tsv,
                               "-o", packed, "-p", "16"], cwd=ROOT, stdout=subprocess.DEVNULL)
        pack_s = time.time() - t0
        os.remove(tsv)
        for rep in range(2):
            out = os.path.join(work, "calls_%d.tsv" % n)
            t0 = time.time()
            r = subprocess.run([sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant", "call_mods", "-i", packed,
                                "-m", ck, "-o", out, "-p", "16"], cwd=ROOT, capture_output=True, text=True,
                               env=dict(os.environ, DSP_TIMING="1"))
            wall = time.time() - t0
            assert r.returncode == 0, r.stderr[-3000:]
            inner = [l for l in r.stdout.splitlines() if "call_mods costs" in l][0]
            secs = float(inner.split("costs")[1].split("seconds")[0])
            assert sum(1 for _ in open(out)) == n
            print(json.dumps({"rows": n, "input": "dspf", "dspf_mb": round(os.path.getsize(packed) / 1e6, 1),
                              "pack_s": round(pack_s, 2), "call_mods_s": secs, "process_wall_s": round(wall, 2),
                              "sites_per_s": round(n / secs, 1), "timeline": timeline(r.stderr)}), flush=True)
            os.remove(out)
        os.remove(packed)


if __name__ == "__main__":
    main()
