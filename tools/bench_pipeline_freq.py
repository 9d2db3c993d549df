#!/usr/bin/env python3
"""BASELINE.json configs[4] scaled to one GPU and to `rows` rows (default 4 M; the config says ~1 B on 8 GPUs -- every line this prints
carries its row count): feature TSV -> call_mods -> per-read calls, with the per-site frequency
(`--freq_file`, what the reference gets by piping the calls into call_mods_freq) from the device-side reduction and from
the host table, next to the plain call_mods run and to `call_freq` on the written file.  One JSON line per run.
usage: bench_pipeline_freq.py [rows] [coverage]   (coverage: calls per genome site; default 1 = every call its own site, the
worst case for the frequency printer)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(cmd):
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant"] + cmd, cwd=ROOT, capture_output=True, text=True)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    inner = [l for l in r.stdout.splitlines() if "costs" in l][0]
    return float(inner.split("costs")[1].split("seconds")[0]), wall


def main():
    import torch
    from deepsignal_plant_amd import synth
    from deepsignal_plant_amd.models import ModelBiLSTM
    work = os.environ.get("DSP_WORK", "/tmp/dsp_pipe")
    os.makedirs(work, exist_ok=True)
    ck = os.path.join(work, "model.ckpt")
    torch.save(synth.random_state_dict(ModelBiLSTM(), seed=1234), ck)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
    cov = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    tsv = os.path.join(work, "feat_%d_cov%d.tsv" % (n, cov))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_tsv.py"), tsv, str(n)] +
                          (["--sites", str(n // cov)] if cov > 1 else []))
    out = os.path.join(work, "calls.tsv")
    base = ["call_mods", "-i", tsv, "-m", ck, "-o", out, "-p", "16"]
    secs, wall = run(base)
    print(json.dumps({"rows": n, "coverage": cov, "run": "call_mods", "call_mods_s": secs, "process_wall_s": round(wall, 2), "sites_per_s": round(n / secs, 1)}), flush=True)
    outs = {}
    for where in ("device", "host"):
        fq = os.path.join(work, "freq_%s.tsv" % where)
        secs, wall = run(base + ["--freq_file", fq, "--freq_on", where, "--prob_cf", "0"])
        outs[where] = open(fq, "rb").read()
        print(json.dumps({"rows": n, "run": "call_mods --freq_file --freq_on %s" % where, "call_mods_s": secs,
                          "process_wall_s": round(wall, 2), "sites_per_s": round(n / secs, 1), "freq_sites": outs[where].count(b"\n")}), flush=True)
    fq = os.path.join(work, "freq_two_step.tsv")
    secs, wall = run(["call_freq", "-i", out, "-o", fq, "--prob_cf", "0"])
    two = open(fq, "rb").read()
    print(json.dumps({"rows": n, "run": "call_freq on the per-read file (host table, 16 parser threads)", "call_freq_s": secs,
                      "process_wall_s": round(wall, 2), "identical_to_device": two == outs["device"], "identical_to_host": two == outs["host"]}), flush=True)
    for p in (tsv, out):
        os.remove(p)


if __name__ == "__main__":
    main()
