#!/usr/bin/env python3
"""Soak: the same forward (Philox initial states, full batch) N times on this box; every output must be bit-identical to the
first (the h exchange goes through global memory + a workgroup barrier without an explicit vmcnt(0): this is the test that
would see a lost or late store).  usage: soak_determinism.py [N=300] [config: 1|3]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from deepsignal_plant_amd import synth
    from deepsignal_plant_amd.models import ModelBiLSTM
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    cfg3 = len(sys.argv) > 2 and sys.argv[2] == "3"
    m = ModelBiLSTM(13, 16, 2 if cfg3 else 3, 1, 2, 0, 256, 16, 4, True, True, module="seq_bilstm" if cfg3 else "both_bilstm",
                    device=0, init_state="randn", seed=11)
    m.load_state_dict(synth.random_state_dict(m, seed=3))
    m.cuda(0).eval()
    bad = 0
    for b, B in enumerate((65536, 65536 - 37, 300_000)):
        ins = synth.feature_batch(B, device="cuda:0", seed=20 + b)
        ref = [t.clone() for t in m(*ins)]
        for i in range(n if B < 100_000 else max(3, n // 10)):
            out = m(*ins)
            if not all(torch.equal(a, r) for a, r in zip(out, ref)):
                bad += 1
                d = (out[1] - ref[1]).abs()
                print("batch %d run %d differs: %d sites, max %.3e" % (B, i, int((d.amax(1) > 0).sum()), float(d.max())))
        torch.cuda.synchronize()
        print("batch %d: %d repeats, %s" % (B, n if B < 100_000 else max(3, n // 10), "all bit-identical" if not bad else "%d differ" % bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
