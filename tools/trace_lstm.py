#!/usr/bin/env python3
"""Shader-clock trace of one LSTM launch (DSP_TRACE build: `make -C deepsignal_plant_amd/csrc trace`).
usage: DSP_AMD_LIB=deepsignal_plant_amd/libdsp_amd_trace.so DSP_TRACE_LAUNCH=<i> python3 tools/trace_lstm.py [--batch B]
Prints, per CU sample, the step timeline of the workgroups that ran on it (k-loop / cell / barrier wait, cycles)."""
import argparse
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from deepsignal_plant_amd import _native as nat
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--model_type", default="both_bilstm")
ap.add_argument("--layernum1", type=int, default=3)
ap.add_argument("--cus", type=int, default=2)
a = ap.parse_args()
m = ModelBiLSTM(13, 16, a.layernum1, 1, 2, 0, 256, 16, 4, True, True, module=a.model_type, device=0, init_state="randn")
m.load_state_dict(synth.random_state_dict(m)); m.cuda(0)
ins = synth.feature_batch(a.batch, device="cuda:0", seed=1)
for _ in range(3): m(*ins)
torch.cuda.synchronize()
W = 8192
t = np.zeros((W, 16, 8), np.uint64); hw = np.zeros((W, 4), np.uint32)
L = nat.lib()
rc = L.dsp_k_trace_read(t.ctypes.data_as(ctypes.c_void_p), hw.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
used = np.nonzero(t[:, 0, 1])[0]
print("traced workgroups: %d" % len(used))
hwid, xcc = hw[:, 0], hw[:, 1] & 0xf
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7; simd = (hwid >> 4) & 3; wid = hwid & 0xf
key = (xcc.astype(np.int64) << 16) | (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 4) | cu
t = t.astype(np.int64)
shown = 0
for k in np.unique(key[used]):
    wg = [i for i in used if key[i] == k]
    wg.sort(key=lambda i: t[i, 0, 0])
    t0 = t[wg[0], 0, 0]
    print("== CU key %06x: %d workgroups" % (k, len(wg)))
    for i in wg[:6]:
        steps = []
        for s in range(13):
            steps.append("%d:%d+%d+%d" % (s, t[i, s, 1] - t[i, s, 0], t[i, s, 2] - t[i, s, 1], t[i, s, 3] - t[i, s, 2]))
        print(" wg %5d dir %d simd %d wave %2d start %9d end %9d | step: barrier+kloop+cell | %s" % (
            i, i & 1, simd[i], wid[i], t[i, 0, 0] - t0, t[i, 12, 3] - t0, " ".join(steps)))
    shown += 1
    if shown >= a.cus:
        break
# between workgroups on a CU: end of the previous one -> entry of the next (dispatch) -> its first step (prologue)
entry = hw[:, 2].astype(np.int64) | (hw[:, 3].astype(np.int64) << 32)
last = int(np.max(np.nonzero(t[used[0], :, 3])[0]))  # index of the last stamped step
gaps = []
for k in np.unique(key[used]):
    wg = sorted([int(i) for i in used if key[i] == k], key=lambda i: t[i, 0, 0])
    for a_, b_ in zip(wg[:-1], wg[1:]):
        if entry[b_] >= t[a_, last, 3] > 0:  # b entered after a had finished: they followed each other on the same slot
            gaps.append(int(entry[b_] - t[a_, last, 3]))
pros = [int(t[i, 0, 0] - entry[i]) for i in used if entry[i] > 0]
print("per workgroup: entry -> first step (prologue: initial states, h0 stores, ring fill) %d cycles; last stamp of the previous "
      "workgroup on the slot -> entry (dispatch) %s cycles (median of %d)" % (
          np.median(pros) if pros else -1, ("%d" % np.median(gaps)) if gaps else "n/a", len(gaps)))
# aggregate over all traced workgroups: median per-phase cycles of the middle steps
mid = t[used][:, 2:12, :]
bar = np.median(mid[:, :, 1] - mid[:, :, 0]); kl = np.median(mid[:, :, 2] - mid[:, :, 1]); ce = np.median(mid[:, :, 3] - mid[:, :, 2])
step = np.median(mid[:, 1:, 0] - mid[:, :-1, 0])
print("median cycles per step %d = barrier wait %d + k-loop %d + cell %d (+ stamp overhead)" % (step, bar, kl, ce))
if mid[:, :, 4].any():
    print("k-loop split: first four k-groups %d, middle %d, last four %d" % (
        np.median(mid[:, :, 4] - mid[:, :, 1]), np.median(mid[:, :, 5] - mid[:, :, 4]), np.median(mid[:, :, 2] - mid[:, :, 5])))
if mid[:, :, 6].any():
    print("cell split: first quarter %d, second+third %d, last %d" % (
        np.median(mid[:, :, 6] - mid[:, :, 2]), np.median(mid[:, :, 7] - mid[:, :, 6]), np.median(mid[:, :, 3] - mid[:, :, 7])))
