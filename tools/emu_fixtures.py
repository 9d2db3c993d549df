#!/usr/bin/env python3
"""Every F1 fixture (the reference's own inputs -> outputs, tests/golden/) through the KERNELS' SOURCE on the host -- the SIMT
interpreter of the test-suite (tests/native/emu, tests/test_kernel_emu.py) -- in the default form and, where the call is small
enough for it, with the opt-in "x ahead" form (DSP_LSTM_XAHEAD=1).  Test infrastructure, CPU only; round 6 had no GPU.
    python tools/emu_fixtures.py [workers] > profiles/r6/kernel_emu_all_fixtures.txt"""
import multiprocessing
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(name):
    import numpy as np
    from tests import test_kernel_emu as T
    from tests.helpers import load_f1
    L = T._build(os.path.join(T._cache_dir("emu"), "libdsp_amd_emu.so"))
    f = load_f1(name)
    out = []
    ref = None
    for xa in ("0", "1"):
        t0 = time.time()
        with T.env(DSP_LSTM_XAHEAD=xa):
            m = T.Model(L, f["cfg"], f["w"])
            probs, logits, labels = m.forward(f["inputs"], states=f["states"])
            m.close()
        dp = float(np.abs(probs - f["probs"]).max())
        if ref is None:
            ref = probs
        out.append("%-28s n %4d  x ahead %s  max|dprob| vs the reference %.3e  max|dlogit| %.3e  same bytes as the default form: %s  (%.0f s)" % (
            name, f["n"], "on " if xa == "1" else "off", dp, float(np.abs(logits - f["logits"]).max()), bool(np.array_equal(probs, ref)), time.time() - t0))
    return "\n".join(out)


def main():
    from tests import test_kernel_emu as T
    from tests.helpers import f1_names
    T._build(os.path.join(T._cache_dir("emu"), "libdsp_amd_emu.so"))     # (once, before the workers)
    names = list(f1_names())
    with multiprocessing.Pool(int(sys.argv[1]) if len(sys.argv) > 1 else 4) as pool:
        for text in pool.imap(one, names):
            print(text, flush=True)


if __name__ == "__main__":
    main()
