#!/usr/bin/env python3
"""Write a synthetic feature TSV (extract format) in parallel: tools/make_tsv.py OUT N [--procs P]."""
import argparse
import os
import sys
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _part(job):
    path, start, n, seed, n_sites = job
    from deepsignal_plant_amd import tsv
    with open(path, "w") as f:
        # rows are numbered globally so that read ids / positions continue across parts
        for i, row in enumerate(tsv.synth_rows(n, seed=seed, first_index=start, n_sites=n_sites)):
            f.write(row)
            f.write("\n")
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("n", type=int)
    ap.add_argument("--procs", type=int, default=min(64, os.cpu_count() or 1))
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--sites", type=int, default=0, help="the rows cycle over this many genome sites (coverage = N / sites); "
                    "default: every row its own site")
    a = ap.parse_args()
    per = 5000  # a multiple of sites_per_read (50): reads never straddle parts
    jobs = [(a.out + ".part%06d" % k, s, min(per, a.n - s), a.seed * 1000003 + k, a.sites or None) for k, s in enumerate(range(0, a.n, per))]
    with Pool(a.procs) as pool:
        parts = pool.map(_part, jobs)
    with open(a.out, "wb") as out:
        for p in parts:
            with open(p, "rb") as f:
                out.write(f.read())
            os.remove(p)
    print("wrote %s: %d rows, %.1f MB" % (a.out, a.n, os.path.getsize(a.out) / 1e6))


if __name__ == "__main__":
    main()
