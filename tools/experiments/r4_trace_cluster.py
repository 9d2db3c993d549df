#!/usr/bin/env python3
"""round 4: shader-clock stamps inside one clustered LSTM launch (an instrumented, uncommitted build of dsp_lstmc_kernel:
TSTAMP 0 step start, 1 before the poll, 2 after it, 3 k-loop done, 4 gates exchanged, 5 cell + h stores issued, 6 published).
usage: DSP_AMD_LIB=variants/libdsp_T.so DSP_TRACE_LAUNCH=3 DSP_TWO_STREAMS=0 python3 tools/experiments/r4_trace_cluster.py [--batch 512]"""
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
print("traced workgroups: %d" % len(used))
names = ["x part", "poll", "h part", "exchange", "cell+stores", "publish", "to next step"]
rows = []
for i in used:
    for s in range(1, 12):
        d = [t[i, s, k + 1] - t[i, s, k] for k in range(6)] + [t[i, s + 1, 0] - t[i, s, 6]]
        rows.append(d + [t[i, s + 1, 0] - t[i, s, 0]])
rows = np.array(rows)
print("cycles per step, steps 1..11 of %d workgroups: median / mean / p90" % len(used))
for k, nme in enumerate(names + ["whole step"]):
    print("  %-13s %8.0f %8.0f %8.0f" % (nme, np.median(rows[:, k]), rows[:, k].mean(), np.percentile(rows[:, k], 90)))
for i in used[:4]:
    print("wg %d:" % i, " | ".join("/".join(str(int(t[i, s, k + 1] - t[i, s, k])) for k in range(6)) for s in range(1, 5)))
