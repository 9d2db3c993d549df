# where the time of `call_mods --freq_file` goes at a realistic coverage (GPU box): milestones of the three variants
cd ${GRAFT_REPO_ROOT:-/root/repo}
W=/tmp/dsp_pipe; mkdir -p $W
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
torch.save(synth.random_state_dict(ModelBiLSTM(), seed=1234), "/tmp/dsp_pipe/model.ckpt")
PY
python tools/make_tsv.py $W/f.tsv 4000000 --sites 160000 > /dev/null
for v in "" "--freq_file $W/fq.tsv --freq_on device --prob_cf 0" "--freq_file $W/fq.tsv --freq_on host --prob_cf 0"; do
  echo "== $v"
  DSP_TIMING=1 python -m deepsignal_plant_amd.deepsignal_plant call_mods -i $W/f.tsv -m $W/model.ckpt -o $W/o.tsv -p 16 $v 2>&1 | grep "seconds\|costs"
done
