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
mac = {1: 2*13*4*128*(7+128), 2: 2*13*4*128*(16+128), 5: 2*13*4*256*(256+256), 6: 2*13*4*256*(512+256), 7: 2*13*4*256*(512+256)}
for i in range(n):
    ms = sum(pr[i + r*n][1] for r in range(R)) / R
    tf = (" %6.1f TFLOP/s" % (2*mac[i]*B/ms/1e9)) if i in mac else ""
    print("%d %-12s %8.3f ms%s" % (i, pr[i][0], ms, tf))
