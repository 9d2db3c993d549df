#!/usr/bin/env python3
"""round 4: a larger soak of the device row parser against the host parser than the -m gpu suite holds.
(1) N synthetic rows with random float spellings (fixed / scientific notation, 1..17 significant digits, signs, leading
    zeros, exponents up to +-25): every row the device accepts equals the host row bit for bit; no row is flagged unless the
    host-side plain grammar excludes it.
(2) M byte-mutated blocks: a row is flagged or equal; what the host rejects is never accepted.
usage: r4_parse_soak.py [N=200000] [M=30000]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

from deepsignal_plant_amd import textio
from tests.test_gpu_parse import _device_parse, _same

L, S = 13, 16


def spell(rng, x, maxd, exotic):
    """one of many decimal spellings of a random value: at most maxd significant digits (a row of 15-digit tokens is longer
    than the LDS kernel's rows and takes the thread-per-row kernels), exponents within +-22 unless `exotic`"""
    kind = int(rng.integers(0, 6))
    digs = int(rng.integers(1, maxd + 1))
    if exotic:
        return ["%.17e" % x, "%.3e" % (x * 1e30), "+%.3f" % abs(x), "%.3f " % x, "1e400", "nan", "%.25f" % x][int(rng.integers(0, 7))]
    if kind == 0:
        return "%.*f" % (min(digs, 9), x)
    if kind == 1:
        return "%.*e" % (max(digs - 1, 0), x)
    if kind == 2:
        return "%.*g" % (digs, x)
    if kind == 3:
        return "%.*E" % (max(digs - 1, 0), x * 10.0 ** float(rng.integers(-18, 19)))
    if kind == 4:
        return "%d" % int(x * 1000)
    return ("-" if x < 0 else "") + "0" * int(rng.integers(0, 3)) + ("%.*f" % (min(digs, 7), abs(x)))


def rows_random(rng, n):
    bases = "ACGTN"
    out = []
    for i in range(n):
        maxd = int(rng.integers(1, 16))
        ex = int(rng.integers(0, 234 * 8)) if rng.random() < 0.1 else -1      # one row in ten holds ONE token outside the plain grammar
        cnt = [0]

        def tok(v):
            cnt[0] += 1
            return spell(rng, float(v), maxd, cnt[0] - 1 == ex)
        kmer = "".join(bases[int(b)] for b in rng.integers(0, 5, L))
        f = lambda: ",".join(tok(v) for v in rng.normal(0, 1.5, L))
        means, stds = f(), f()
        lens = ",".join(str(int(v)) for v in rng.integers(1, 400, L))
        sig = ";".join(",".join(tok(v) for v in rng.normal(0, 1.5, S)) for _ in range(L))
        out.append("\t".join(["chr%d" % (i % 7), str(i * 3), "+-"[i & 1], str(i), "read_%d" % (i // 50), "t", kmer, means, stds, lens, sig,
                              str(i & 1)]))
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
    rng = np.random.default_rng(2024)
    t0 = time.time()
    n_same = n_flag = n_long = 0
    for c0 in range(0, n, 20000):
        lines = rows_random(rng, min(20000, n - c0))
        data = ("\n".join(lines) + "\n").encode()
        dev, staged, _ = _device_parse(data)
        assert staged.n == len(lines)
        try:
            host = textio.parse_rows(data, L, S)
        except (ValueError, KeyError, IndexError):
            host = None                                   # (an exotic token the host parser rejects: row by row then)
        for i in range(staged.n):
            if dev["status"][i]:
                n_flag += 1
                continue
            if host is not None:
                k = _same(dev, host, i, i)
            else:
                k = _same(dev, textio.parse_rows((lines[i] + "\n").encode(), L, S), i, 0)   # (raises if the device took a bad row)
            assert k is None, (k, lines[i][:200])
            n_same += 1
            n_long += len(lines[i]) + 1 > 2560
    print("random spellings: %d rows, %d accepted and bit-identical to the host parser (%d of them longer than the LDS kernel's rows), "
          "%d flagged (left to the host parser), %.0f s" % (n, n_same, n_long, n_flag, time.time() - t0))
    base = rows_random(rng, 64)
    pool = b"\t,;.-+eE0123456789 \nACGTNX\r:_"
    t0 = time.time()
    n_same = n_flag = n_rej = 0
    for c0 in range(0, m, 3000):
        blocks = []
        for _ in range(min(3000, m - c0)):
            k = int(rng.integers(1, 4))
            bad = bytearray(("\n".join(base[int(i)] for i in rng.integers(0, len(base), k)) + "\n").encode())
            for _k in range(int(rng.integers(1, 4))):
                bad[int(rng.integers(0, len(bad)))] = pool[int(rng.integers(0, len(pool)))]
            if not bad.endswith(b"\n"):
                bad += b"\n"
            blocks.append(bytes(bad))
        dev, staged, _ = _device_parse(b"".join(blocks))
        i = 0
        for blk in blocks:
            for piece in blk.split(b"\n")[:-1]:
                try:
                    h = textio.parse_rows(piece + b"\n", L, S) if piece else None
                except (ValueError, KeyError, IndexError):
                    h = None
                if h is None or h.n != 1:
                    assert dev["status"][i] == 1, piece[:120]
                    n_rej += 1
                elif dev["status"][i] == 0:
                    assert _same(dev, h, i, 0) is None, piece[:120]
                    n_same += 1
                else:
                    n_flag += 1
                i += 1
        assert i == staged.n
    print("mutated blocks: %d blocks; rows accepted and bit-identical %d, flagged though the host parser takes them %d, rejected by the host "
          "parser and flagged %d; never accepted what the host rejects; %.0f s" % (m, n_same, n_flag, n_rej, time.time() - t0))


if __name__ == "__main__":
    main()
