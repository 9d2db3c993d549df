#!/usr/bin/env python3
"""round 4 soak: the small-batch paths (clusters of 8 / 4 / 2 CUs, the one-workgroup forms, two streams) repeated N times per
size; every output must be bit-identical to the first AND to the round-3 path (a lost or late cross-CU hand-off would show)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM

n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 2000


def build(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    m = ModelBiLSTM(13, 16, 3, 1, 2, 0, 256, 16, 4, True, True, module="both_bilstm", device=0, init_state="randn", seed=11)
    m.load_state_dict(synth.random_state_dict(m, seed=3))
    m.cuda(0).eval()
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    return m


ref_m = build({"DSP_LSTM_CLUSTER": "0", "DSP_LSTM_LOCAL8": "0", "DSP_TWO_STREAMS": "0", "DSP_HEAD_ST4": "1", "DSP_FC_SMALL": "0"})
m = build({})
bad = 0
for n in (37, 512, 1000, 1024, 2048, 3000, 4096):
    ins = synth.feature_batch(n, device="cuda:0", seed=40 + n)
    ref_m.site_offset = m.site_offset = 5 * n
    ref = ref_m(*ins)[1].clone()
    outs = []
    for i in range(n_rep):
        outs.append(m(*ins)[1])
        if len(outs) == 100:
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
            outs = []
    torch.cuda.synchronize()
    bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
    print("%5d sites: %d forwards, %d differ from the round-3 path so far" % (n, n_rep, bad), flush=True)
print("soak small batches: %d of %d forwards differ" % (bad, 7 * n_rep))
sys.exit(1 if bad else 0)
