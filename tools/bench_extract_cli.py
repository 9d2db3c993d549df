#!/usr/bin/env python3
"""The `extract` sub-command end to end on this box: synthetic reads (.reads.npz) -> GPU feature extraction -> feature
TSV (plain / --gzip = BGZF) or the binary container (.dspf).  One JSON line per output kind.
usage: bench_extract_cli.py [reads] [mean bases]"""
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from deepsignal_plant_amd import reads as R
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    mean_bases = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    work = os.environ.get("DSP_WORK", "/tmp/dsp_extract_cli")
    shutil.rmtree(work, ignore_errors=True)
    os.makedirs(os.path.join(work, "reads"))
    samples = 0
    for i in range(0, n_reads, 64):
        rs = R.synth_reads(min(64, n_reads - i), seed=100 + i, mean_bases=mean_bases)
        samples += sum(len(r.raw) for r in rs)
        R.save_reads(os.path.join(work, "reads", "batch_%05d.reads.npz" % i), rs, compress=False)
    for kind, out, extra in (("tsv", "feats.tsv", []), ("tsv --gzip (BGZF)", "feats_gz.tsv", ["--gzip"]), ("dspf", "feats.dspf", [])):
        path = os.path.join(work, out)
        t0 = time.time()
        r = subprocess.run([sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant", "extract", "-i", os.path.join(work, "reads"),
                            "-o", path, "-p", "16", "--f5_batch_size", "32"] + extra, cwd=ROOT, capture_output=True, text=True,
                           env=dict(os.environ, DSP_TIMING="1"))
        wall = time.time() - t0
        assert r.returncode == 0, r.stderr[-3000:]
        inner = [l for l in r.stdout.splitlines() if "extract_features costs" in l][0]
        secs = float(inner.split("costs")[1].split("seconds")[0])
        rows = int(inner.split("(")[1].split()[0])
        real = path + (".gz" if extra and not path.endswith(".gz") else "")
        stages = [l.split(": ", 1)[1] for l in r.stderr.splitlines() if l.startswith("[extract] seconds per stage")]
        print(json.dumps({"extract_to": kind, "reads": n_reads, "samples": samples, "rows": rows, "out_mb": round(os.path.getsize(real) / 1e6, 1),
                          "extract_s": secs, "process_wall_s": round(wall, 2), "rows_per_s": round(rows / secs, 1),
                          "stage_seconds": stages[0] if stages else None}), flush=True)
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
