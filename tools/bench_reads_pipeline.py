#!/usr/bin/env python3
"""End-to-end `call_mods -i <directory of reads>` on this box: synthetic reads (about 12 samples per base) ->
GPU feature extraction -> forward -> per-read calls.  Prints one JSON line."""
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from deepsignal_plant_amd import reads as R
    from deepsignal_plant_amd import synth
    from deepsignal_plant_amd.models import ModelBiLSTM
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    mean_bases = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    cg_boost = float(sys.argv[3]) if len(sys.argv) > 3 else 0.15
    work = os.environ.get("DSP_WORK", "/tmp/dsp_reads")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(os.path.join(work, "reads"))
    ck = os.path.join(work, "model.ckpt")
    torch.save(synth.random_state_dict(ModelBiLSTM(), seed=1234), ck)
    t0 = time.time()
    samples = bases = 0
    per_file = 64
    for i in range(0, n_reads, per_file):
        rs = R.synth_reads(min(per_file, n_reads - i), seed=100 + i, mean_bases=mean_bases, cg_boost=cg_boost)
        samples += sum(len(r.raw) for r in rs)
        bases += sum(len(r.ev_len) for r in rs)
        R.save_reads(os.path.join(work, "reads", "batch_%05d.reads.npz" % i), rs, compress=False)
    gen = time.time() - t0
    in_bytes = sum(os.path.getsize(os.path.join(work, "reads", f)) for f in os.listdir(os.path.join(work, "reads")))
    for rep in range(2):
        out = os.path.join(work, "calls.tsv")
        t0 = time.time()
        r = subprocess.run([sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant", "call_mods", "-i",
                            os.path.join(work, "reads"), "-m", ck, "-o", out, "-p", "16", "--f5_batch_size", "32"],
                           cwd=ROOT, capture_output=True, text=True)
        wall = time.time() - t0
        assert r.returncode == 0, r.stderr[-3000:]
        inner = [l for l in r.stdout.splitlines() if "call_mods costs" in l][0]
        secs = float(inner.split("costs")[1].split("seconds")[0])
        sites = sum(1 for _ in open(out))
        print(json.dumps({"pipeline": "reads -> extract (GPU) -> forward -> calls", "reads": n_reads, "samples": samples,
                          "bases": bases, "sites": sites, "input_mb": round(in_bytes / 1e6, 1), "call_mods_s": secs,
                          "process_wall_s": round(wall, 2), "sites_per_s": round(sites / secs, 1),
                          "msamples_per_s": round(samples / secs / 1e6, 1), "gen_s": round(gen, 1)}), flush=True)
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
