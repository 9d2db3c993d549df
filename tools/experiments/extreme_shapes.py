"""Smallest and oddest model shapes the C ABI accepts, HIP vs the C oracle (Philox states)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from oracle import c_oracle as oc
from oracle import forward_np as onp
from tests.test_gpu_parity import build_model, to_dev
cases = [dict(hidden_size=2, seq_len=1, signal_len=1, num_classes=2), dict(hidden_size=2, seq_len=3, signal_len=2, num_classes=64),
         dict(hidden_size=6, seq_len=101, signal_len=3, num_layers1=1), dict(hidden_size=34, seq_len=5, signal_len=64, num_layers2=3),
         dict(hidden_size=3, module="seq_bilstm", seq_len=2, embedding_size=1, vocab_size=1, num_classes=1),
         dict(hidden_size=3, module="signal_bilstm", seq_len=2, signal_len=1, num_classes=3, num_layers1=15, num_layers2=1)]
worst = 0.0
for kw in cases:
    cfg = onp.OracleConfig(**kw)
    w = onp.make_weights(cfg, 7, 2.0)
    for n in (1, 67):
        ins = onp.make_inputs(cfg, n, 8)
        if cfg.vocab_size < 4:
            ins[0][:] = 0
        m = build_model(cfg, w, init_state="randn", seed=5)
        lo, po = m(*to_dev(ins))
        torch.cuda.synchronize()
        _, pr = oc.forward(cfg, w, *ins, init_mode="philox", seed=5)
        d = float(np.abs(po.cpu().numpy() - pr).max())
        worst = max(worst, d)
        print(kw, n, "%.2e" % d, flush=True)
print("worst %.2e" % worst)
assert worst <= 2e-5
