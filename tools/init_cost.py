import sys
sys.path.insert(0, "/root/repo")
import torch
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
B = 65536
for mode in ("randn", "zeros"):
    m = ModelBiLSTM(13, 16, 3, 1, 2, 0, 256, 16, 4, True, True, device=0, init_state=mode)
    m.load_state_dict(synth.random_state_dict(m)); m.cuda(0)
    ins = synth.feature_batch(B, device="cuda:0", seed=1)
    for _ in range(2): m(*ins)
    torch.cuda.synchronize(); m.profile(True)
    R = 10
    for _ in range(R): m(*ins)
    torch.cuda.synchronize()
    pr = m.profile_read(); n = len(pr) // R
    print(mode, " ".join("%s %.3f" % (pr[i][0], sum(pr[i + r * n][1] for r in range(R)) / R) for i in range(n)),
          "sum %.3f" % (sum(x[1] for x in pr) / R))
