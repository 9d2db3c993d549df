#!/usr/bin/env python3
"""Inflate rate of one gzip stream of feature text: zlib (sequential reader) vs the parallel inflater by thread count
(csrc/dsp_pgz.cpp), no GPU.  usage: bench_inflate.py [rows] [gzip level]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from deepsignal_plant_amd import gzio
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
    level = sys.argv[2] if len(sys.argv) > 2 else "6"
    work = os.environ.get("DSP_WORK", "/tmp/dsp_pipe")
    os.makedirs(work, exist_ok=True)
    tsv = os.path.join(work, "feat_%d.tsv" % n)
    if not os.path.exists(tsv):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_tsv.py"), tsv, str(n)])
    gz = tsv + ".l%s.gz" % level
    if not os.path.exists(gz):
        with open(gz, "wb") as f:
            subprocess.check_call(["gzip", "-%s" % level, "-c", tsv], stdout=f)
    size = os.path.getsize(tsv)
    buf = np.empty(64 << 20, np.uint8)

    def run(st):
        t0 = time.time()
        tot = 0
        while True:
            k = st.readinto(buf)
            if k == 0:
                break
            tot += k
        dt = time.time() - t0
        st.close()
        assert tot == size
        return round(size / dt / 1e6, 1)
    res = {"what": "inflate of one gzip stream, MB/s of text", "rows": n, "text_mb": round(size / 1e6, 1),
           "compressed_mb": round(os.path.getsize(gz) / 1e6, 1), "gzip_level": level, "zlib_sequential": run(gzio.GzStream(gz))}
    for nt in (1, 2, 4, 8, 16):
        st = gzio.PgzStream(gz, nt)
        res["parallel_%d_threads" % nt] = run(st)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
