#!/usr/bin/env python3
"""round 5: latency of a forward at small batches by switch setting (same process, same box, alternating), bit-identity
against the round-3 path, and the per-launch HIP-event times of the 'auto' setting.
usage: r5_small_ab.py [sizes=512,1024,2048,4096] [reps=200]   (modes: edit MODES / pass DSP_R5_MODES=name:K=V;K=V,...)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
from oracle import forward_np as onp

SWITCHES = ("DSP_LSTM_CLUSTER", "DSP_LSTM_TILING", "DSP_HEAD_ST4", "DSP_TWO_STREAMS", "DSP_LSTM_LOCAL8", "DSP_LSTM_FRONT_CLUSTER",
            "DSP_FC_FUSED", "DSP_CLUSTER_TIMEOUT", "DSP_LSTM_HANDOFF", "DSP_FC_SMALL", "DSP_FORWARD_SPLIT")
MODES = [("round3", {"DSP_LSTM_CLUSTER": "0", "DSP_LSTM_LOCAL8": "0", "DSP_TWO_STREAMS": "0", "DSP_HEAD_ST4": "1", "DSP_FC_FUSED": "0", "DSP_FC_SMALL": "0"}),
         ("round4", {"DSP_LSTM_FRONT_CLUSTER": "0", "DSP_FC_SMALL": "0"}),
         ("fc_small", {"DSP_LSTM_FRONT_CLUSTER": "0"}),
         ("auto", {}),
         ("frontG1", {"DSP_LSTM_FRONT_CLUSTER": "1"}),
         ("frontG2", {"DSP_LSTM_FRONT_CLUSTER": "2"})]
if os.environ.get("DSP_R5_MODES"):
    MODES = [MODES[0]]
    for item in os.environ["DSP_R5_MODES"].split(","):
        name, _, kv = item.partition(":")
        MODES.append((name, dict(x.split("=") for x in kv.split(";") if x)))


def build(cfg, w, env):
    for k in SWITCHES:
        os.environ.pop(k, None)
    os.environ.update(env)
    m = ModelBiLSTM(cfg.seq_len, cfg.signal_len, cfg.num_layers1, cfg.num_layers2, cfg.num_classes, 0, cfg.hidden_size,
                    cfg.vocab_size, cfg.embedding_size, cfg.is_base, cfg.is_signallen, module=cfg.module, device=0,
                    init_state="randn", seed=17)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return m.cuda(0).eval()


def timed(m, ins, reps):
    for _ in range(10):
        m.forward(*ins)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        m.forward(*ins)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    sizes = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "512,1024,2048,4096").split(","))
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    cfg = onp.OracleConfig()
    w = onp.make_weights(cfg, 91, 2.0)
    ins = {n: synth.feature_batch(n, device="cuda:0", seed=400 + n) for n in sizes}
    models = [(name, build(cfg, w, env)) for name, env in MODES]
    res, ms = {}, {name: {n: [] for n in sizes} for name, _ in MODES}
    for name, m in models:
        for n in sizes:
            m.site_offset = 10 * n
            res[name, n] = m.forward(*ins[n])[1].clone()
    torch.cuda.synchronize()
    for rnd in range(3):               # alternating rounds: box drift hits every mode alike
        for name, m in models:
            for n in sizes:
                m.site_offset = 10 * n
                ms[name][n].append(timed(m, ins[n], reps))
    flops = 118447104
    print("%-7s" % "sites" + "".join("%22s" % name for name, _ in MODES) + "   (ms per forward, min of 3 rounds [max]; fraction of 157.3 TFLOP/s; * = bits differ from round3)")
    ok = True
    for n in sizes:
        row = "%-7d" % n
        for name, _ in MODES:
            same = torch.equal(res[name, n], res["round3", n])
            ok &= same
            best = min(ms[name][n])
            row += "  %7.4f [%6.4f] %.3f%s" % (best, max(ms[name][n]), n * flops / (best * 1e-3) / 157.3e12, " " if same else "*")
        print(row)
    print("bit-identical everywhere:", ok)
    # per-launch times of the last mode and of 'auto'
    for name, m in models:
        if name not in ("auto", MODES[-1][0], "round4"):
            continue
        for n in sizes[:2]:
            m.site_offset = 10 * n
            m.profile(True)
            R = 20
            for _ in range(R):
                m.forward(*ins[n])
            torch.cuda.synchronize()
            pr = m.profile_read()
            m.profile(False)
            k = len(pr) // R
            print("per launch, %s, %d sites: " % (name, n) + ", ".join("%s %.1f" % (pr[i][0], 1e3 * sum(pr[i + r * k][1] for r in range(R)) / R) for i in range(k)) + " us")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
