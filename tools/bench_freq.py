#!/usr/bin/env python3
"""call_freq aggregator throughput (host only): per-read call lines -> per-site table, by parser thread count."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from deepsignal_plant_amd.call_mods_freq import SiteFrequency

rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
pos = rng.integers(0, 2000000, size=n)
p0 = np.round(rng.random(n), 6)
lines = ["chr%d\t%d\t+\t%d\tread_%07d\tt\t%s\t%s\t%d\tAACGT" % (i % 5 + 1, pos[i], pos[i], i // 40, p0[i], round(1 - p0[i], 6), p0[i] < 0.5)
         for i in range(n)]
text = ("\n".join(lines) + "\n").encode()
for nt in (1, 4, 16):
    agg = SiteFrequency(0.2, nthreads=nt)
    t = time.time(); agg.add_calls_text(text); dt = time.time() - t
    print("%2d parser threads: %.2f M lines/s (%d lines, %d sites, %.0f MB)" % (nt, n / dt / 1e6, n, agg.counts()[2], len(text) / 1e6), flush=True)
