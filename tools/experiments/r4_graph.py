#!/usr/bin/env python3
"""round 4: dsp_forward captured into a HIP graph (torch.cuda.CUDAGraph) -- does it capture (side stream fork / join, the
clean-up launches), does the replay give the eager bits, and what does a replay cost at small batches?"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM

m = ModelBiLSTM(13, 16, 3, 1, 2, 0, 256, 16, 4, True, True, module="both_bilstm", device=0, init_state="randn", seed=3)
m.load_state_dict(synth.random_state_dict(m, seed=1234))
m.cuda(0).eval()
for n in (512, 2048, 4096, 65536):
    m.reserve(n)
    static = [t.clone() for t in synth.feature_batch(n, device="cuda:0", seed=1)]
    other = synth.feature_batch(n, device="cuda:0", seed=2)
    m.site_offset = 77
    eager1 = m.forward(*static)[1].clone()
    eager2 = m.forward(*other)[1].clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        m.forward(*static)          # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = m.forward(*static)[1]
    g.replay()
    torch.cuda.synchronize()
    same1 = torch.equal(out, eager1)
    for a, b in zip(static, other):
        a.copy_(b)
    g.replay()
    torch.cuda.synchronize()
    same2 = torch.equal(out, eager2)
    reps = 200 if n <= 4096 else 20
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        m.forward(*static)
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / reps * 1e3
    print("n = %6d: captured; replay == eager: %s / %s (new inputs); %.3f ms per replay, %.3f ms per eager forward (wall)" % (n, same1, same2, tg, te))
