#!/usr/bin/env python3
"""A full batch through the KERNELS' SOURCE on the host (tests/native/emu, the test-suite's SIMT interpreter), one dsp_forward,
against the C oracle over ALL sites -- the script behind profiles/r6/kernel_emu_default_arch_full_batches.txt (round 6 had no GPU).
Test infrastructure, CPU only; 65,536 sites of the default architecture take 2.5 hours and 19 GB.
    python tools/emu_full_batch.py default 65536                 # BASELINE configs[1]'s batch, Philox states
    python tools/emu_full_batch.py cfg3 65536                    # configs[2]: seq_bilstm, hidden 256, 2 combined layers
    python tools/emu_full_batch.py default 9001 explicit         # explicit N(0,1) states in the reference's layout (never cut)
    python tools/emu_full_batch.py default 9001 philox bf16x9    # the split-precision kernels' full-batch form"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    from tests import test_kernel_emu as T
    model, n = sys.argv[1], int(sys.argv[2])
    states = sys.argv[3] if len(sys.argv) > 3 else "philox"
    precision = sys.argv[4] if len(sys.argv) > 4 else None
    L = T._build(os.path.join(T._cache_dir("emu"), "libdsp_amd_emu.so"))
    cfg = onp.OracleConfig() if model == "default" else onp.OracleConfig(module="seq_bilstm", num_layers1=2)
    w, ins = onp.make_weights(cfg, 5, 2.0), onp.make_inputs(cfg, n, 100 + n)
    st = onp.make_init_states(cfg, n, 7) if states == "explicit" else None
    t0 = time.time()
    with T.env():
        m = T.Model(L, cfg, w, precision=precision)
        pp = m.forward(ins, states=st, philox=None if st is not None else (7, 5 * n))[0]
        m.close()
    t1 = time.time()
    want = oc.forward(cfg, w, *ins, states=st)[1] if st is not None else oc.forward(cfg, w, *ins, init_mode="philox", seed=7, site_offset=5 * n)[1]
    d = np.abs(pp - want).max(axis=1)
    print("%s, %s states%s: %6d sites in one dsp_forward through the interpreted kernels: max|dprob| vs the C oracle over ALL sites %.2e "
          "(99.9th percentile %.2e); labels equal %d / %d  (interpreter %.0f s, oracle %.0f s)" % (
              "the default architecture (both_bilstm, hidden 256, 3 combined layers, T 13)" if model == "default" else "configs[2] (seq_bilstm, hidden 256, 2 combined layers, T 13)",
              states, ", " + precision if precision else "", n, float(d.max()), float(np.quantile(d, 0.999)),
              int((pp.argmax(1) == want.argmax(1)).sum()), n, t1 - t0, time.time() - t1), flush=True)


if __name__ == "__main__":
    main()
