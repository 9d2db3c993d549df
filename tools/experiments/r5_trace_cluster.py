#!/usr/bin/env python3
"""round 5: shader-clock stamps inside ONE clustered LSTM launch, prologue, first and last step included (the TSTAMPs of
lstmc_layer in a DSP_TRACE build: make -C deepsignal_plant_amd/csrc trace).
step rows: 0 step start, 1 before the poll, 2 after it, 3 k-loop done, 4 gates exchanged, 5 cell + h stores issued, 6 published;
row 13: 0 prologue start, 1 ring fill issued, 2 states stored, 3 h0 published; row 14: 0 layer done.
usage: DSP_AMD_LIB=deepsignal_plant_amd/libdsp_amd_trace.so DSP_TRACE_LAUNCH=3 DSP_LSTM_PERSIST=0 DSP_TWO_STREAMS=0 python3 tools/experiments/r5_trace_cluster.py [--batch 512]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from deepsignal_plant_amd import _native as nat
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=512); a = ap.parse_args()
m = ModelBiLSTM(13, 16, 3, 1, 2, 0, 256, 16, 4, True, True, module="both_bilstm", device=0, init_state="randn")
m.load_state_dict(synth.random_state_dict(m)); m.cuda(0)
ins = synth.feature_batch(a.batch, device="cuda:0", seed=1)
for _ in range(5): m(*ins)
torch.cuda.synchronize()
W = 8192
t = np.zeros((W, 16, 8), np.uint64); hw = np.zeros((W, 4), np.uint32)
rc = nat.lib().dsp_k_trace_read(t.ctypes.data_as(ctypes.c_void_p), hw.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
t = t.astype(np.int64)
used = np.nonzero(t[:, 1, 0])[0]
T = 13
print("launch %s, %d sites: traced workgroups: %d" % (os.environ.get("DSP_TRACE_LAUNCH"), a.batch, len(used)))
t0 = t[used, 13, 0].min()
med = lambda x: float(np.median(x))
print("prologue start after the first workgroup's: median %.0f, max %.0f cycles" % (med(t[used, 13, 0] - t0), (t[used, 13, 0] - t0).max()))
print("prologue: ring fill issued %.0f, states stored %.0f, h0 published %.0f, first step starts %.0f (median cycles since the workgroup's prologue start)" % (
    med(t[used, 13, 1] - t[used, 13, 0]), med(t[used, 13, 2] - t[used, 13, 0]), med(t[used, 13, 3] - t[used, 13, 0]), med(t[used, 0, 0] - t[used, 13, 0])))
names = ["x part", "poll", "h part", "exchange", "cell+stores", "publish"]
print("%-6s" % "step" + "".join("%12s" % n for n in names) + "%12s" % "whole")
for s in range(T):
    d = [t[used, s, k + 1] - t[used, s, k] for k in range(6)]
    end = t[used, s + 1, 0] if s + 1 < T else t[used, 14, 0]
    print("%-6d" % s + "".join("%12.0f" % med(x) for x in d) + "%12.0f" % med(end - t[used, s, 0]))
print("layer done after the first prologue start: median %.0f, max %.0f cycles; the workgroup's own span: median %.0f" % (
    med(t[used, 14, 0] - t0), (t[used, 14, 0] - t0).max(), med(t[used, 14, 0] - t[used, 13, 0])))
