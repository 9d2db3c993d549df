import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deepsignal_plant_amd import extract_features as ef, reads as R
for tag, kw in (("long bases (default)", {}), ("no base > 16", dict(max_len=16, long_every=10**9))):
    rs = R.synth_reads(512, seed=1, mean_bases=8000, **kw)
    fx = ef.FeatureExtractor(seed=1)
    fx.extract(rs); torch.cuda.synchronize()
    fx.torch.cuda.synchronize()
    m = __import__("deepsignal_plant_amd._native", fromlist=["x"])
    t0 = time.perf_counter()
    for _ in range(5): out = fx.extract(rs)
    torch.cuda.synchronize()
    print(tag, "sites", out.n, "samples", sum(len(r.raw) for r in rs), "wall ms", (time.perf_counter()-t0)/5*1e3)
