import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
t=time.time(); torch.zeros(1, device="cuda:0"); torch.cuda.synchronize(); print("hip init %.3f" % (time.time()-t))
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
m = ModelBiLSTM(13, 16, 3, 1, 2, 0, 256, 16, 4, True, True, device=0)
sd = synth.random_state_dict(m, seed=1)
torch.save(sd, "/tmp/m.ckpt")
for rep in range(3):
    t0=time.time(); m = ModelBiLSTM(13, 16, 3, 1, 2, 0, 256, 16, 4, True, True, device=0)
    t1=time.time(); pd = torch.load("/tmp/m.ckpt", map_location="cpu")
    t2=time.time(); d = m.state_dict(); d.update(pd); m.load_state_dict(d)
    t3=time.time(); m.cuda(0)
    t4=time.time(); m.reserve(65536); torch.cuda.synchronize()
    t5=time.time()
    print("construct %.3f  torch.load %.3f  load_state_dict %.3f  cuda()=dsp_model_create %.3f  reserve %.3f" % (t1-t0,t2-t1,t3-t2,t4-t3,t5-t4))
