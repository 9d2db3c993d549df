import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from deepsignal_plant_amd.models import ModelBiLSTM
from deepsignal_plant_amd import synth
hid = int(sys.argv[1])
m = ModelBiLSTM(13, 16, 2, 1, 2, 0, hid, 16, 4, True, True, device=0, init_state="zeros")
m.load_state_dict(synth.random_state_dict(m, seed=3)); m.cuda(0)
ins = synth.feature_batch(33, device="cuda:0", seed=1)
_, p0 = m(*ins); torch.cuda.synchronize()
print("ok fp32", hid)
