#!/usr/bin/env python3
"""round 4: dsp_lstmc_kernel (a site tile's unit tiles spread over a cluster of CUs) -- bit-identity against the
unclustered path and the latency of a forward by batch size.  usage: r4_cluster_check.py [explicit|philox|zeros]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
from oracle import forward_np as onp


def build(cfg, w, env):
    for k in ("DSP_LSTM_CLUSTER", "DSP_LSTM_TILING", "DSP_HEAD_ST4", "DSP_TWO_STREAMS", "DSP_LSTM_LOCAL8"):
        os.environ.pop(k, None)
    os.environ.update(env)
    m = ModelBiLSTM(cfg.seq_len, cfg.signal_len, cfg.num_layers1, cfg.num_layers2, cfg.num_classes, 0, cfg.hidden_size,
                    cfg.vocab_size, cfg.embedding_size, cfg.is_base, cfg.is_signallen, module=cfg.module, device=0,
                    init_state="randn", seed=17)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return m.cuda(0).eval()


def main():
    cfg = onp.OracleConfig()
    w = onp.make_weights(cfg, 91, 2.0)
    sizes = (1, 33, 512, 513, 1024, 1025, 2048, 2049, 4096, 4097, 8192)
    ins = {n: synth.feature_batch(n, device="cuda:0", seed=400 + n) for n in sizes}
    modes = [("off", {"DSP_LSTM_CLUSTER": "0", "DSP_HEAD_ST4": "1", "DSP_TWO_STREAMS": "0"}), ("auto", {}), ("1stream", {"DSP_TWO_STREAMS": "0"}), ("local8", {"DSP_LSTM_LOCAL8": "1"}), ("G4", {"DSP_LSTM_CLUSTER": "4"}),
]
    res, ms = {}, {}
    for name, env in modes:
        m = build(cfg, w, env)
        for n in sizes:
            m.site_offset = 10 * n
            out = m.forward(*ins[n])[1]
            torch.cuda.synchronize()
            res[name, n] = out.clone()
            for _ in range(5):
                m.forward(*ins[n])
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(40):
                m.forward(*ins[n])
            b.record()
            torch.cuda.synchronize()
            ms[name, n] = a.elapsed_time(b) / 40
        del m
    print("%-6s" % "sites" + "".join("%10s" % name for name, _ in modes) + "   (ms per forward; * = bits differ from 'off')")
    ok = True
    for n in sizes:
        row = "%-6d" % n
        for name, _ in modes:
            same = torch.equal(res[name, n], res["off", n])
            ok &= same
            row += "%9.3f%s" % (ms[name, n], " " if same else "*")
        print(row)
    from oracle import c_oracle as oc
    n = 513
    sample = [t.cpu().numpy() for t in ins[n]]
    _, po = oc.forward(cfg, w, *sample, init_mode="philox", seed=17, site_offset=10 * n)
    print("auto vs oracle at %d sites: max |dprob| = %.2e" % (n, np.abs(res["auto", n].cpu().numpy() - po).max()))
    print("bit-identical everywhere:", ok)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
