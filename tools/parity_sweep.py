#!/usr/bin/env python3
"""Randomised parity sweep on this box: N random model shapes (every flag of the reference constructor, hidden sizes up to
1,280, 1-3 layers, odd k-mer lengths, any signal window) x random batch sizes, HIP forward vs the oracle's C port with explicit
N(0,1) initial states and with in-kernel Philox states.  Prints one line per case and the maxima; exits non-zero if any case
exceeds the tolerance the parity tests assert.  usage: parity_sweep.py [N=200] [seed=0]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def rand_cfg(rng):
    from oracle import forward_np as onp
    module = ["both_bilstm", "seq_bilstm", "signal_bilstm"][int(rng.integers(0, 3))]
    hidden = int(rng.choice([20, 32, 50, 64, 96, 128, 160, 200, 256, 258, 320, 384, 450, 512, 514, 640, 700, 1024, 1280]))
    if module == "both_bilstm" and hidden % 2:
        hidden += 1
    return onp.OracleConfig(seq_len=int(rng.choice([5, 9, 13, 17, 21])), signal_len=int(rng.choice([4, 8, 12, 16, 24, 32, 40])),
                            num_layers1=int(rng.integers(1, 4)), num_layers2=int(rng.integers(1, 3)),
                            num_classes=int(rng.choice([2, 2, 3, 5])), hidden_size=hidden,
                            vocab_size=int(rng.choice([5, 16])), embedding_size=int(rng.choice([2, 4, 6])),
                            is_base=bool(rng.integers(0, 2)), is_signallen=bool(rng.integers(0, 2)), module=module)


def main():
    import torch
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    from tests.test_gpu_parity import TOL_TIGHT, build_model, to_dev
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    precision = "fp32"
    for i, a in enumerate(sys.argv):
        if a == "--precision":
            precision = sys.argv[i + 1]
            argv = [x for x in argv if x != precision]
    n_cases = int(argv[0]) if len(argv) > 0 else 200
    seed = int(argv[1]) if len(argv) > 1 else 0
    refused = 0
    worst = worst_p = 0.0
    t0 = time.time()
    bad = 0
    for case in range(n_cases):
        rng = np.random.default_rng(77_000 + 1000 * seed + case)
        cfg = rand_cfg(rng)
        n = int(rng.choice([1, 5, 31, 32, 33, 64, 65, 100, 129, 257, 400]))
        if cfg.hidden_size > 512:
            n = min(n, 65)   # the CPU restatement of a hidden-1,280 model runs at a few sites per second and core
        scale = float(rng.choice([1.0, 2.0, 3.0, 4.0]))
        w = onp.make_weights(cfg, 10_000 + case, scale)
        ins = onp.make_inputs(cfg, n, 20_000 + case, wide_alphabet=cfg.vocab_size == 16)
        if cfg.vocab_size < 16:
            ins = (np.minimum(ins[0], cfg.vocab_size - 1),) + ins[1:]
        st = onp.make_init_states(cfg, n, 30_000 + case)
        m = build_model(cfg, w)
        try:
            m.set_precision(precision)   # --precision bf16x9 | bf16x6 | fp16x3: the same sweep, the same tolerance
        except ValueError as e:          # fp16 pieces refused for this checkpoint (documented: operands outside the fp16 range)
            assert precision == "fp16x3" and "fp16 range" in str(e), e
            refused += 1
            continue
        _, probs = m.forward(*to_dev(ins), init_states={k: torch.from_numpy(v).cuda(0) for k, v in st.items()})
        _, po = oc.forward(cfg, w, *ins, states=st, init_mode="explicit")
        d = float(np.abs(probs.cpu().numpy() - po).max())
        m.init_state, m.seed = "randn", 40_000 + case
        _, pp = m.forward(*to_dev(ins))
        _, pq = oc.forward(cfg, w, *ins, init_mode="philox", seed=40_000 + case)
        d2 = float(np.abs(pp.cpu().numpy() - pq).max())
        worst, worst_p = max(worst, d), max(worst_p, d2)
        flag = "" if max(d, d2) <= TOL_TIGHT else "  <-- above %.0e" % TOL_TIGHT
        bad += bool(flag)
        c = cfg.as_dict()
        print("%3d %-14s hid %3d l1 %d l2 %d k %2d s %2d base %d len %d vocab %2d n %3d x%.0f: max|dprob| %.2e explicit, %.2e philox%s" % (
            case, c["module"], c["hidden_size"], c["num_layers1"], c["num_layers2"], c["seq_len"], c["signal_len"], c["is_base"],
            c["is_signallen"], c["vocab_size"], n, scale, d, d2, flag), flush=True)
        del m
    print("%d cases (precision %s, %d refused) in %.0f s: max|dprob| %.2e (explicit states), %.2e (Philox states); tolerance %.0e; %d above" % (
        n_cases, precision, refused, time.time() - t0, worst, worst_p, TOL_TIGHT, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
