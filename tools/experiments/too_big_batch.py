"""A batch whose workspace cannot fit (5 M sites ~ 360 GB of the 309 GB there are): the forward must refuse with an error, and the handle must
still work afterwards."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
m = ModelBiLSTM(13, 16, 3, 1, 2, 0, 256, 16, 4, True, True, device=0, init_state="zeros")
m.load_state_dict(synth.random_state_dict(m, seed=3))
m.cuda(0).eval()
small = synth.feature_batch(1000, device="cuda:0", seed=1)
ref = m(*small)[1].clone()
try:
    m.reserve(int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000)
    print("reserve succeeded?!")
except RuntimeError as e:
    print("refused:", str(e)[:160])
assert torch.equal(m(*small)[1], ref)
print("handle still fine")
free, total = torch.cuda.mem_get_info()
print("after reserve: free %.1f GB of %.1f GB" % (free / 1e9, total / 1e9))
