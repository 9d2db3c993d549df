#!/usr/bin/env python3
"""Device-side call_freq throughput: N synthetic calls already in HBM (keys / pos_in_strand / meta as the host hands them
over, probabilities and labels as the forward leaves them) -> encode -> stable sort by site -> sequential per-site reduce.
Prints the rate of each stage and the algorithmic HBM traffic (profiles/LAB_NOTEBOOK_r1_r3.md section 6c: 25 B/record into encode, 16 out; the
reduce reads 32 B/record and writes 72 B/site).  usage: bench_freq_dev.py [N records] [sites]"""
import ctypes
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deepsignal_plant_amd import _native as nat

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64_000_000
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
key_in = (torch.randint(0, 5, (n,), generator=g, device=dev) << 40) | torch.randint(0, sites // 5, (n,), generator=g, device=dev)
meta = torch.randint(0, 1 << 22, (n,), generator=g, device=dev, dtype=torch.int32)
pis = torch.randint(0, 1 << 30, (n,), generator=g, device=dev)
p0 = torch.rand((n,), generator=g, device=dev)
probs = torch.stack((p0, 1 - p0), 1).contiguous()
labels = (p0 < 0.5).to(torch.uint8)
row = torch.arange(n, device=dev)
L = nat.lib()
p = ctypes.c_void_p
s = torch.cuda.current_stream(dev)


def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps, out


key = torch.empty(n, dtype=torch.int64, device=dev)
packed = torch.empty(n, dtype=torch.int64, device=dev)
t_enc, _ = timed(lambda: nat.check(L.dsp_freq_dev_encode(p(s.cuda_stream), n, p(probs.data_ptr()), 2, p(labels.data_ptr()),
                                                         p(key_in.data_ptr()), p(meta.data_ptr()), 0.2, p(key.data_ptr()), p(packed.data_ptr()))))
# the library's own stable sort of the four columns (rocPRIM radix sort of (key, index) + one gather); torch.sort + three
# indexed gathers, what finish() used before, is timed next to it
outs = [torch.empty_like(key) for _ in range(4)]
need = ctypes.c_size_t(0)
sargs = [p(s.cuda_stream), n] + [p(t.data_ptr()) for t in (key, packed, pis, row)] + [p(t.data_ptr()) for t in outs]
nat.check(int(L.dsp_freq_dev_sort_records(*sargs, None, ctypes.byref(need))))
tmp = torch.empty(max(need.value, 1), dtype=torch.uint8, device=dev)
t_sort, _ = timed(lambda: nat.check(int(L.dsp_freq_dev_sort_records(*sargs, p(tmp.data_ptr()), ctypes.byref(need)))))
ks, pk, ps, rw = outs
t_torch, (ks_t, perm) = timed(lambda: torch.sort(key, stable=True))
t_gather_t, (pk_t, ps_t, rw_t) = timed(lambda: (packed[perm], pis[perm], row[perm]))
assert torch.equal(ks, ks_t) and torch.equal(pk, pk_t) and torch.equal(ps, ps_t) and torch.equal(rw, rw_t)
t_gather = 0.0
cnt = torch.zeros(1, dtype=torch.int64, device=dev)
nat.check(L.dsp_freq_dev_count_sites(p(s.cuda_stream), n, p(ks.data_ptr()), p(cnt.data_ptr())))
ns = int(cnt.item())
oi = [torch.empty(ns, dtype=torch.int64, device=dev) for _ in range(6)]
od = [torch.empty(ns, dtype=torch.float64, device=dev) for _ in range(2)]
t_red, _ = timed(lambda: nat.check(L.dsp_freq_dev_reduce(p(s.cuda_stream), n, p(ks.data_ptr()), p(pk.data_ptr()), p(ps.data_ptr()), p(rw.data_ptr()),
                                                         p(cnt.data_ptr()), ns, p(oi[0].data_ptr()), p(oi[1].data_ptr()), p(oi[2].data_ptr()),
                                                         p(oi[3].data_ptr()), p(od[0].data_ptr()), p(od[1].data_ptr()), p(oi[4].data_ptr()), p(oi[5].data_ptr()))))
used = int((key != 0x7fffffffffffffff).sum())
tot = t_enc + t_sort + t_gather + t_red
print('{"records": %d, "used": %d, "sites": %d, "encode_ms": %.2f, "encode_GBps": %.0f, "stable_sort_with_gather_ms": %.2f, "torch_sort_plus_gathers_ms": %.2f, '
      '"reduce_ms": %.2f, "reduce_GBps": %.0f, "total_ms": %.2f, "records_per_s": %.3e}' % (
          n, used, ns, t_enc * 1e3, n * 41 / t_enc / 1e9, t_sort * 1e3, (t_torch + t_gather_t) * 1e3, t_red * 1e3,
          (n * 32 + ns * 72) / t_red / 1e9, tot * 1e3, n / tot))
