"""One forward of a very large batch (default 1,048,576 + 37 sites, ~76 GB of workspace): contiguous windows against the
C oracle with Philox states, and bit-identity with the same sites run as 16 batches of 65,536 (+ the tail)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
from oracle import c_oracle as oc
from oracle import forward_np as onp
n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 20) + 37
cfg = onp.OracleConfig()
w = onp.make_weights(cfg, 61, 2.0)
m = ModelBiLSTM(13, 16, 3, 1, 2, 0, 256, 16, 4, True, True, device=0, init_state="randn", seed=11)
m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
m.cuda(0).eval()
ins = synth.feature_batch(n, device="cuda:0", seed=62)
m.site_offset = 5_000_000_000
big = m(*ins)[1].clone()
torch.cuda.synchronize()
print("forward of %d sites done; workspace %.1f GB" % (n, torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9))
worst = 0.0
for a in (0, n // 2 + 4001, n - 2304):
    b = min(n, a + 2304)
    sample = [t[a:b].cpu().numpy() for t in ins]
    _, po = oc.forward(cfg, w, *sample, init_mode="philox", seed=11, site_offset=m.site_offset + a)
    worst = max(worst, float(np.abs(big[a:b].cpu().numpy() - po).max()))
print("windows vs oracle: max|dprob| %.2e" % worst)
same = True
for a in range(0, n, 65536):
    b = min(n, a + 65536)
    m.site_offset = 5_000_000_000 + a
    same &= bool(torch.equal(m(*[t[a:b] for t in ins])[1], big[a:b]))
print("identical to the same sites in batches of 65,536:", same)
assert worst <= 2e-5 and same
