#!/usr/bin/env python3
"""Every F1 fixture (the reference's own inputs -> outputs, tests/golden/) through the KERNELS' SOURCE on the host -- the SIMT
interpreter of the test-suite (tests/native/emu, tests/test_kernel_emu.py) -- in the default form and, where the call is small
enough for it, with the opt-in "x ahead" form (DSP_LSTM_XAHEAD=1).  Test infrastructure, CPU only; round 6 had no GPU.
    python tools/emu_fixtures.py [workers] > profiles/r6/kernel_emu_all_fixtures.txt
    python tools/emu_fixtures.py [workers] precisions > profiles/r6/kernel_emu_all_fixtures_split_precision.txt
(the second form: the opt-in split-precision modes -- bf16x9, bf16x6, fp16x3 -- on the same fixtures, with fp32's bounds)"""
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


def one_precision(arg):
    import numpy as np
    from deepsignal_plant_amd import _native as nat
    from tests import test_kernel_emu as T
    from tests.helpers import f1_tolerances, load_f1
    name, precision = arg
    L = T._build(os.path.join(T._cache_dir("emu"), "libdsp_amd_emu.so"))
    f = load_f1(name)
    t0 = time.time()
    with T.env():
        m = T.Model(L, f["cfg"], f["w"])
        rc = L.dsp_model_set_precision(m.h, nat.PRECISION[precision])
        if rc != 0:      # fp16 pieces refused for this checkpoint (operands not provably inside the fp16 range): the documented refusal
            msg = L.dsp_last_error().decode()
            m.close()
            return "%-22s %-7s refused: %s" % (name, precision, msg[:110])
        probs, logits, labels = m.forward(f["inputs"], states=f["states"])
        m.close()
    dp, tol = float(np.abs(probs - f["probs"]).max()), f1_tolerances(name)[1]
    return "%-22s %-7s n %4d  max|dprob| vs the reference %.3e  (the GPU suite's bound for fp32: %.1e)  %s  (%.0f s)" % (
        name, precision, f["n"], dp, tol, "inside" if dp <= tol else "OUTSIDE", time.time() - t0)


def main():
    from tests import test_kernel_emu as T
    from tests.helpers import f1_names
    T._build(os.path.join(T._cache_dir("emu"), "libdsp_amd_emu.so"))     # (once, before the workers)
    names = list(f1_names())
    with multiprocessing.Pool(int(sys.argv[1]) if len(sys.argv) > 1 else 4) as pool:
        if len(sys.argv) > 2 and sys.argv[2] == "precisions":
            for text in pool.imap(one_precision, [(n, p) for n in names for p in ("bf16x9", "bf16x6", "fp16x3")]):
                print(text, flush=True)
            return
        for text in pool.imap(one, names):
            print(text, flush=True)


if __name__ == "__main__":
    main()
