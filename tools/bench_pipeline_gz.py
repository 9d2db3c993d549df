#!/usr/bin/env python3
"""End-to-end `call_mods` throughput on gzipped feature files next to the plain-text rate (same rows): BGZF (what this
build's `extract --gzip` / tools write; inflated on all parser threads) and a foreign single-member .gz (`gzip -1`; one
inflate thread by nature).  Also `call_mods --gzip` output (BGZF writer).  One JSON line per run.
usage: bench_pipeline_gz.py [rows]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(inp, ck, out, extra=()):
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant", "call_mods", "-i", inp, "-m", ck,
                        "-o", out, "-p", "16"] + list(extra), cwd=ROOT, capture_output=True, text=True)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    inner = [l for l in r.stdout.splitlines() if "call_mods costs" in l][0]
    return float(inner.split("costs")[1].split("seconds")[0]), wall


def main():
    import torch
    from deepsignal_plant_amd import gzio, synth
    from deepsignal_plant_amd.models import ModelBiLSTM
    work = os.environ.get("DSP_WORK", "/tmp/dsp_pipe")
    os.makedirs(work, exist_ok=True)
    ck = os.path.join(work, "model.ckpt")
    torch.save(synth.random_state_dict(ModelBiLSTM(), seed=1234), ck)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
    tsv = os.path.join(work, "feat_%d.tsv" % n)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_tsv.py"), tsv, str(n)])
    size = os.path.getsize(tsv)
    out = os.path.join(work, "calls.tsv")
    secs, wall = run(tsv, ck, out)
    plain = n / secs
    print(json.dumps({"rows": n, "input": "plain text", "mb": round(size / 1e6, 1), "call_mods_s": secs, "process_wall_s": round(wall, 2),
                      "sites_per_s": round(plain, 1)}), flush=True)
    ref_calls = open(out, "rb").read()
    # BGZF
    bg = tsv + ".bgzf.gz"
    t0 = time.time()
    with gzio.BgzfWriter(bg, level=4, nthreads=16) as w, open(tsv, "rb") as f:
        while True:
            chunk = f.read(64 << 20)
            if not chunk:
                break
            w.write(chunk)
    comp_s = time.time() - t0
    secs, wall = run(bg, ck, out)
    assert open(out, "rb").read() == ref_calls
    print(json.dumps({"rows": n, "input": "BGZF .gz (this build's --gzip)", "mb": round(os.path.getsize(bg) / 1e6, 1),
                      "compress_s_16_threads": round(comp_s, 1), "call_mods_s": secs, "process_wall_s": round(wall, 2),
                      "sites_per_s": round(n / secs, 1), "vs_plain": round(n / secs / plain, 3)}), flush=True)
    # --gzip output next to it
    secs, wall = run(bg, ck, out, ["--gzip"])
    import gzip
    assert gzip.open(out + ".gz", "rb").read() == ref_calls
    print(json.dumps({"rows": n, "input": "BGZF .gz", "output": "--gzip (BGZF)", "call_mods_s": secs,
                      "sites_per_s": round(n / secs, 1), "vs_plain": round(n / secs / plain, 3)}), flush=True)
    os.remove(out + ".gz")
    os.remove(bg)
    # a foreign single-member .gz
    sg = tsv + ".single.gz"
    t0 = time.time()
    with open(sg, "wb") as f:
        subprocess.check_call(["gzip", "-1", "-c", tsv], stdout=f)
    comp_s = time.time() - t0
    secs, wall = run(sg, ck, out)
    assert open(out, "rb").read() == ref_calls
    print(json.dumps({"rows": n, "input": "single-member .gz (gzip -1)", "mb": round(os.path.getsize(sg) / 1e6, 1),
                      "compress_s": round(comp_s, 1), "call_mods_s": secs, "process_wall_s": round(wall, 2),
                      "sites_per_s": round(n / secs, 1), "vs_plain": round(n / secs / plain, 3)}), flush=True)
    for p in (sg, tsv, out):
        os.remove(p)


if __name__ == "__main__":
    main()
