#!/usr/bin/env python3
"""End-to-end `call_mods -i <directory of fast5 files>` on this box: synthetic single-read fast5 files written with h5py
(tools/gen_fast5.py under /opt/conda/bin/python3.9) -> native HDF5 reads (loader threads) -> GPU feature extraction ->
forward -> per-read calls.  Prints one JSON line per repetition, for --nproc 1 and 16 (loader threads; set
DSP_READER_PROCS=N for reader processes instead)."""
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
    from deepsignal_plant_amd import synth
    from deepsignal_plant_amd.models import ModelBiLSTM
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    mean_bases = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    work = os.environ.get("DSP_WORK", "/tmp/dsp_fast5")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(work)
    ck = os.path.join(work, "model.ckpt")
    torch.save(synth.random_state_dict(ModelBiLSTM(), seed=1234), ck)
    t0 = time.time()
    r = subprocess.run(["/opt/conda/bin/python3.9", os.path.join(ROOT, "tools", "gen_fast5.py"), os.path.join(work, "reads"),
                        str(n_reads), str(mean_bases)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    samples, bases = (int(x) for x in r.stdout.split()[-2:])
    gen = time.time() - t0
    in_bytes = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(os.path.join(work, "reads")) for f in fs)
    for nproc in (1, 16, 16):
        out = os.path.join(work, "calls.tsv")
        t0 = time.time()
        r = subprocess.run([sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant", "call_mods", "-i",
                            os.path.join(work, "reads"), "-m", ck, "-o", out, "-p", str(nproc), "--f5_batch_size", "32"],
                           cwd=ROOT, capture_output=True, text=True)
        wall = time.time() - t0
        assert r.returncode == 0, r.stderr[-3000:]
        inner = [l for l in r.stdout.splitlines() if "call_mods costs" in l][0]
        secs = float(inner.split("costs")[1].split("seconds")[0])
        sites = sum(1 for _ in open(out))
        print(json.dumps({"pipeline": "fast5 -> native HDF5 reads -> extract (GPU) -> forward -> calls", "reads": n_reads,
                          "nproc": nproc, "samples": samples, "bases": bases, "sites": sites,
                          "input_mb": round(in_bytes / 1e6, 1), "call_mods_s": secs, "process_wall_s": round(wall, 2),
                          "sites_per_s": round(sites / secs, 1), "files_per_s": round(n_reads / secs, 1),
                          "msamples_per_s": round(samples / secs / 1e6, 1), "gen_s": round(gen, 1)}), flush=True)
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
