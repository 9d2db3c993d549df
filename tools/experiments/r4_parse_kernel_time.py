#!/usr/bin/env python3
"""round 4: the device-side row parser alone -- ms per block of 32,768 rows (HIP events, 30 launches on one staged block)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from deepsignal_plant_amd import parse_dev, tsv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
rows = list(tsv.synth_rows(2048, seed=1))
data = ("\n".join(rows[i % 2048] for i in range(n)) + "\n").encode()
stage = parse_dev.alloc_stage(n + 1, len(data) + 1, 13)
t0 = time.time()
r, nb = parse_dev.stage_rows(np.frombuffer(data, np.uint8), stage, 13, 16)
t_stage = time.time() - t0
dp = parse_dev.DeviceRowParser(torch.device("cuda", 0), 13, 16)
s = torch.cuda.current_stream()
for _ in range(3):
    b, ev = dp.submit(r, nb, stage, s)
ev.synchronize()
assert int(stage["_torch"]["n_flagged"][0]) == 0
import ctypes
from deepsignal_plant_amd import _native as nat
p = lambda x: ctypes.c_void_p(x.data_ptr())
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(30):
    nat.check(int(nat.lib().dsp_parse_rows_device(ctypes.c_void_p(s.cuda_stream), p(b["text"]), p(b["row_off"]), n, 13, 16, p(b["kmer"]),
                                                  p(b["means"]), p(b["stds"]), p(b["lens"]), p(b["signals"]), p(b["labels"]), p(b["info_len"]),
                                                  p(b["read_off"]), p(b["read_len"]), p(b["status"]), p(b["n_flagged"]), p(b["seg"]), int(nb))))
e.record()
torch.cuda.synchronize()
ms = a.elapsed_time(e) / 30
print("parse kernel: %.3f ms per block of %d rows (%.1f MB of text: %.1f GB/s); host staging %.1f ms (%.2f GB/s)" % (
    ms, n, nb / 1e6, nb / ms / 1e6, t_stage * 1e3, nb / t_stage / 1e9))
