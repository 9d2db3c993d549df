#!/usr/bin/env python3
"""BASELINE.json configs[4] in miniature on the 1-GPU box (N rows, default 2 M, printed on the line; the config says ~1 B on 8 GPUs): a reference-style single-member .gz of N feature rows -> call_mods
on TWO ranks sharing the GPU (one inflater for the node through the shared-memory ring, parallel inflate, blocks dealt
round-robin, device-side call_freq with the all_to_all over gloo, parts merged back into input order) against the same
command on one rank: byte-identical per-read calls and frequency file, wall times.  One JSON line.
usage: bench_two_ranks_gz.py [rows] [ranks=2] [plain]   (plain: the feature TSV itself, byte ranges per rank, every rank
copying its part of the calls into the result at once)"""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from deepsignal_plant_amd import synth
    from deepsignal_plant_amd.models import ModelBiLSTM
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
    nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    plain = len(sys.argv) > 3 and sys.argv[3] == "plain"
    work = os.environ.get("DSP_WORK", "/tmp/dsp_pipe")
    os.makedirs(work, exist_ok=True)
    ck = os.path.join(work, "model.ckpt")
    torch.save(synth.random_state_dict(ModelBiLSTM(), seed=1234), ck)
    tsv = os.path.join(work, "feat_%d.tsv" % n)
    if not os.path.exists(tsv):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_tsv.py"), tsv, str(n)])
    gz = tsv + ".single.gz"
    if plain:
        gz = tsv
    elif not os.path.exists(gz):
        with open(gz, "wb") as f:
            subprocess.check_call(["gzip", "-1", "-c", tsv], stdout=f)
    res = {"rows": n, "input": "plain text" if plain else "single-member .gz (gzip -1)", "input_mb": round(os.path.getsize(gz) / 1e6, 1)}
    outs = {}
    for ranks in (1, nranks):
        out, fq = os.path.join(work, "calls_%d.tsv" % ranks), os.path.join(work, "freq_%d.tsv" % ranks)
        args = ["call_mods", "-i", gz, "-m", ck, "-o", out, "-p", "16", "--freq_file", fq, "--prob_cf", "0"]
        if ranks == 1:
            cmd = [sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant"] + args
        else:
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
            s.close()
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
                   "--master-port", str(port), "-m", "deepsignal_plant_amd.deepsignal_plant"] + args
        t0 = time.time()
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, env=dict(os.environ, DSP_TIMING="1"))
        wall = time.time() - t0
        assert r.returncode == 0, r.stderr[-3000:]
        inner = [l for l in r.stdout.splitlines() if "call_mods costs" in l][0]
        secs = float(inner.split("costs")[1].split("seconds")[0])
        outs[ranks] = (open(out, "rb").read(), open(fq, "rb").read())
        tl = [l.split(": ", 1)[1] for l in r.stderr.splitlines() if "[call_mods] seconds at" in l]
        res["ranks_%d" % ranks] = {"call_mods_s": secs, "process_wall_s": round(wall, 2), "sites_per_s": round(n / secs, 1),
                                   "timeline_rank0": tl[0] if tl else None}
        os.remove(out)
        os.remove(fq)
    res["per_read_calls_identical"] = outs[1][0] == outs[nranks][0]
    res["freq_file_identical"] = outs[1][1] == outs[nranks][1]
    res["calls"] = outs[1][0].count(b"\n")
    print(json.dumps(res))
    assert res["per_read_calls_identical"] and res["freq_file_identical"]


if __name__ == "__main__":
    main()
