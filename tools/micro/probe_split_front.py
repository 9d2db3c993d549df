import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from deepsignal_plant_amd.models import ModelBiLSTM
from deepsignal_plant_amd import synth
hid, n = int(sys.argv[1]), int(sys.argv[2])
m = ModelBiLSTM(13, 16, 2, 1, 2, 0, hid, 16, 4, True, True, device=0, init_state=sys.argv[3])
m.load_state_dict(synth.random_state_dict(m, seed=3))
m.cuda(0)
ins = synth.feature_batch(n, device="cuda:0", seed=1)
m.set_precision("fp32"); _, p0 = m(*ins); torch.cuda.synchronize()
m.set_precision("bf16x6"); _, p1 = m(*ins); torch.cuda.synchronize()
print("hid", hid, "n", n, sys.argv[3], "max diff", float((p0 - p1).abs().max()))
