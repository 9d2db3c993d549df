#!/usr/bin/env python3
"""Per-launch timing of one forward (HIP events on the launch stream): name, ms, TFLOP/s of the LSTM launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
B = 65536
m = ModelBiLSTM(13, 16, 3, 1, 2, 0, 256, 16, 4, True, True, device=0, init_state="randn")
m.load_state_dict(synth.random_state_dict(m)); m.cuda(0)
ins = synth.feature_batch(B, device="cuda:0", seed=1)
for _ in range(2): m(*ins)
torch.cuda.synchronize(); m.profile(True)
R = 5
for _ in range(R): m(*ins)
torch.cuda.synchronize()
pr = m.profile_read()
n = len(pr) // R
# algorithmic MACs per site of every launch, in launch order
macs = [("pack", 0), ("lstm_seq", 2*13*4*128*(7+128)), ("fc_seq", 13*128*256), ("lstm_signal", 2*13*4*128*(16+128)),
        ("fc_signal", 13*128*256), ("lstm_comb", 2*13*4*256*(256+256)), ("lstm_comb", 2*13*4*256*(512+256)),
        ("lstm_comb", 2*13*4*256*(512+256)), ("head", 256*512 + 2*256)]
for i in range(n):
    ms = sum(pr[i + r*n][1] for r in range(R)) / R
    name, mac = macs[i] if i < len(macs) and macs[i][0] == pr[i][0] else (pr[i][0], 0)
    tf = (" %6.1f TFLOP/s" % (2*mac*B/ms/1e9)) if mac else ""
    print("%d %-12s %8.3f ms%s" % (i, pr[i][0], ms, tf))
