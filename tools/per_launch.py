#!/usr/bin/env python3
"""Per-launch timing of one forward (HIP events on the launch stream): name, ms, TFLOP/s of every launch.
usage: per_launch.py [--model_type M] [--layernum1 N] [--hid_rnn H] [--batch B] [--precision P]"""
import argparse
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
ap = argparse.ArgumentParser()
ap.add_argument("--model_type", default="both_bilstm")
ap.add_argument("--layernum1", type=int, default=3)
ap.add_argument("--layernum2", type=int, default=1)
ap.add_argument("--hid_rnn", type=int, default=256)
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--precision", default="fp32")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
B, H, T = a.batch, a.hid_rnn, 13
m = ModelBiLSTM(T, 16, a.layernum1, a.layernum2, 2, 0, H, 16, 4, True, True, module=a.model_type, device=0, init_state="randn")
m.load_state_dict(synth.random_state_dict(m)); m.cuda(0); m.set_precision(a.precision)
ins = synth.feature_batch(B, device="cuda:0", seed=1)
for _ in range(2): m(*ins)
torch.cuda.synchronize(); m.profile(True)
R = a.reps
for _ in range(R): m(*ins)
torch.cuda.synchronize()
pr = m.profile_read()
n = len(pr) // R
# algorithmic MACs per site of every launch, in launch order
hs = {"both_bilstm": H // 2, "seq_bilstm": H, "signal_bilstm": 0}[a.model_type]
hg = {"both_bilstm": H - H // 2, "seq_bilstm": 0, "signal_bilstm": H}[a.model_type]
def lstm_macs(i, h, layers):
    return [2 * T * 4 * h * ((i if k == 0 else 2 * h) + h) for k in range(layers)]
macs = {"pack": [0], "lstm_seq": lstm_macs(7, hs, a.layernum2), "fc_seq": [T * hs * 2 * hs],
        "lstm_signal": lstm_macs(16, hg, a.layernum2), "fc_signal": [T * hg * 2 * hg],
        "fc_seq+fc_signal": [T * hs * 2 * hs + T * hg * 2 * hg],
        "lstm_comb": lstm_macs(H, H, a.layernum1), "head": [H * 2 * H + 2 * H], "front": [0]}
seen = {}
tot = 0.0
for i in range(n):
    ms = sum(pr[i + r * n][1] for r in range(R)) / R
    name = pr[i][0]
    k = seen.get(name, 0); seen[name] = k + 1
    mac = macs.get(name, [0])
    mac = mac[k] if k < len(mac) else 0
    tf = (" %6.1f TFLOP/s  %5.1f %% of 157.3" % (2 * mac * B / ms / 1e9, 2 * mac * B / ms / 1e9 / 1.573)) if mac else ""
    tot += ms
    print("%d %-16s %8.3f ms%s" % (i, name, ms, tf))
fl = m.flops_per_site()
print("sum %.3f ms  -> %.4f M sites/s, %.1f TFLOP/s = %.1f %% of the fp32 MFMA peak  [DSP_LSTM_SG=%s]" % (
    tot, B / tot / 1e3, fl * B / tot / 1e9, fl * B / tot / 1e9 / 1.573, os.environ.get("DSP_LSTM_SG", "")))
