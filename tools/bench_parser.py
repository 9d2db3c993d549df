#!/usr/bin/env python3
"""Row-parser throughput by thread count, one-pass fast path on and off (csrc/dsp_text.cpp): MB/s of feature text into
SoA arrays, no GPU.  usage: bench_parser.py [rows]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from deepsignal_plant_amd import _native as nat
    from deepsignal_plant_amd import textio
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    work = os.environ.get("DSP_WORK", "/tmp/dsp_pipe")
    os.makedirs(work, exist_ok=True)
    tsv = os.path.join(work, "feat_%d.tsv" % n)
    if not os.path.exists(tsv):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_tsv.py"), tsv, str(n)])
    data = np.fromfile(tsv, dtype=np.uint8)
    out = textio.alloc_rows(n + 16, 13, 16, pinned=False)
    res = {}
    for fast in (1, 0):
        nat.lib().dsp_text_set_fast_rows_(fast)
        for nt in (1, 2, 4, 8, 16):
            best = 1e9
            for _ in range(3):
                t0 = time.time()
                rows = textio.parse_rows(data, 13, 16, nthreads=nt, out=out)
                best = min(best, time.time() - t0)
            assert rows.n == n
            res["%s_%d_threads_mb_per_s" % ("one_pass" if fast else "general", nt)] = round(len(data) / best / 1e6, 1)
    nat.lib().dsp_text_set_fast_rows_(1)
    print(json.dumps({"what": "row parser, text -> SoA arrays", "rows": n, "text_mb": round(len(data) / 1e6, 1), **res}))


if __name__ == "__main__":
    main()
