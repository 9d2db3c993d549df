#!/usr/bin/env python3
"""what the per-launch HIP events of dsp_profile cost a forward, by batch size (bench.py times its steps with them on)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
from oracle import forward_np as onp

cfg = onp.OracleConfig(); w = onp.make_weights(cfg, 91, 2.0)
m = ModelBiLSTM(cfg.seq_len, cfg.signal_len, cfg.num_layers1, cfg.num_layers2, cfg.num_classes, 0, cfg.hidden_size, cfg.vocab_size,
                cfg.embedding_size, cfg.is_base, cfg.is_signallen, module=cfg.module, device=0, init_state="randn", seed=17)
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); m = m.cuda(0).eval()
for n in (512, 2048, 4096, 65536):
    ins = synth.feature_batch(n, device="cuda:0", seed=n)
    R = 200 if n <= 4096 else 20
    out = []
    for rep in range(2):
        for prof in (False, True):
            m.profile(prof)
            for _ in range(5): m.forward(*ins)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(R): m.forward(*ins)
            b.record(); torch.cuda.synchronize()
            if prof: m.profile_read()
            out.append("%s %.4f" % ("events on" if prof else "events off", a.elapsed_time(b) / R))
    m.profile(False)
    print(n, "sites, ms per forward:", " | ".join(out))
