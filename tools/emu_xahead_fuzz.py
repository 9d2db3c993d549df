#!/usr/bin/env python3
"""Random model shapes / batch sizes / switch sets through the interpreted kernels (tests/native/emu), with and without the
opt-in "x ahead" form: the probabilities must be the same bytes, and within 1e-6 of the C oracle.  Test infrastructure, CPU only.
    python tools/emu_xahead_fuzz.py [cases] [workers] > profiles/r6/xahead_fuzz.txt"""
import multiprocessing
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SWITCHES = [{}, {"DSP_LSTM_CLUSTER": "2"}, {"DSP_LSTM_CLUSTER": "4"}, {"DSP_LSTM_HANDOFF": "0"}, {"DSP_CLUSTER_TIMEOUT": "0"}, {"DSP_LSTM_XAHEAD_RING": "8"},
            {"DSP_RSRC_EXTENTS": "tight"}, {"DSP_TWO_STREAMS": "0"}, {"EMU_CUS": "128"}, {"DSP_LSTM_XAHEAD_TILES": "3"}]


def one(k):
    import numpy as np
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    from tests import test_kernel_emu as T
    rng = np.random.default_rng(1000 + k)
    L = T._build(os.path.join(T._cache_dir("emu"), "libdsp_amd_emu.so"))
    hidden = int(rng.choice([int(rng.integers(97, 129)), int(rng.integers(193, 257)), 256, 128]))
    module = str(rng.choice(["both_bilstm", "both_bilstm", "seq_bilstm", "signal_bilstm"]))
    if module == "both_bilstm" and hidden < 194:
        hidden = int(rng.integers(97, 129))           # (both: the combined stack is `hidden` wide, the front ends half of it)
    kw = dict(seq_len=int(rng.integers(1, 4)), signal_len=int(rng.choice([8, 16])), hidden_size=hidden, num_layers1=int(rng.integers(1, 4)),
              num_layers2=int(rng.integers(1, 3)), module=module, is_base=bool(rng.integers(0, 2)), is_signallen=bool(rng.integers(0, 2)))
    n = int(rng.choice([1, int(rng.integers(2, 65)), int(rng.integers(65, 257)), int(rng.integers(257, 400))]))
    sw = dict(SWITCHES[int(rng.integers(0, len(SWITCHES)))])
    seed = int(rng.integers(0, 4))
    if seed:
        sw["DSP_EMU_SEED"] = str(seed)
    mode = ("explicit", "philox", "zeros")[int(rng.integers(0, 3))]
    cfg = onp.OracleConfig(**kw)
    w, ins = onp.make_weights(cfg, 50 + k, 2.0), onp.make_inputs(cfg, n, 60 + k)
    st = onp.make_init_states(cfg, n, 70 + k) if mode == "explicit" else None
    t0 = time.time()
    out = []
    for xa in ("0", "1"):
        with T.env(DSP_LSTM_XAHEAD=xa, **sw):
            m = T.Model(L, cfg, w)
            out.append(m.forward(ins, states=st, philox=(9, 77) if mode == "philox" else None)[0])
            m.close()
    want = oc.forward(cfg, w, *ins, states=st)[1] if mode == "explicit" else oc.forward(cfg, w, *ins, init_mode=mode, **(dict(seed=9, site_offset=77) if mode == "philox" else {}))[1]
    same, d = bool(np.array_equal(out[0], out[1])), float(np.abs(out[1] - want).max())
    return "%s case %3d  %-13s hidden %3d layers %d/%d T %d  n %3d  %-8s %-48s same bytes %s  vs the C oracle %.2e  (%.0f s)" % (
        "ok  " if same and d <= 2e-6 else "FAIL", k, module, hidden, kw["num_layers1"], kw["num_layers2"], kw["seq_len"], n, mode, sw, same, d, time.time() - t0)


def main():
    from tests import test_kernel_emu as T
    T._build(os.path.join(T._cache_dir("emu"), "libdsp_amd_emu.so"))
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    bad = 0
    with multiprocessing.Pool(int(sys.argv[2]) if len(sys.argv) > 2 else 4) as pool:
        for text in pool.imap(one, range(cases)):
            print(text, flush=True)
            bad += text.startswith("FAIL")
    print("%d cases, %d failed" % (cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
