import sys, os
sys.path.insert(0, "/root/repo")
os.environ["DSP_DEBUG_LSTM"]="1"
import torch
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
m = ModelBiLSTM(init_state="randn", seed=3)
m.load_state_dict(synth.random_state_dict(m, seed=5)); m.cuda(0)
sys.stderr.write("== default\n"); sys.stderr.flush()
m(*synth.feature_batch(512, device="cuda:0", seed=9)); torch.cuda.synchronize()
