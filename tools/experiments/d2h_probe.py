"""D2H of a large tensor: pageable .cpu() vs a pinned destination, and what pinning costs (GPU box)."""
import time
import torch
n = 450 << 20
t = torch.empty(n, dtype=torch.uint8, device="cuda:0")
t.fill_(3); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.time(); h = t.cpu(); dt = time.time() - t0
    print("pageable .cpu(): %.3f s  %.1f GB/s" % (dt, n / dt / 1e9))
t0 = time.time(); p = torch.empty(n, dtype=torch.uint8, pin_memory=True); dt = time.time() - t0
print("pin_memory alloc of %d MB: %.3f s  %.1f GB/s" % (n >> 20, dt, n / dt / 1e9))
for rep in range(3):
    t0 = time.time(); p.copy_(t, non_blocking=True); torch.cuda.synchronize(); dt = time.time() - t0
    print("pinned copy: %.3f s  %.1f GB/s" % (dt, n / dt / 1e9))
t0 = time.time(); a = p.numpy().copy(); dt = time.time() - t0
print("host memcpy of the same bytes: %.3f s  %.1f GB/s" % (dt, n / dt / 1e9))
