#!/bin/bash
# round 4: what the device-side row parser costs on the GPU (rocprofv3 kernel trace of one call_mods run on 400,000 rows)
export DSP_WORK=/tmp/dsp_pipe
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $DSP_WORK $REPO/gpurun_out/r4
cd $REPO
python tools/make_tsv.py $DSP_WORK/feat_400000.tsv 400000 > /dev/null
python - <<'PY'
import os, sys, torch
sys.path.insert(0, '.')
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
torch.save(synth.random_state_dict(ModelBiLSTM(), seed=1234), os.path.join(os.environ["DSP_WORK"], "model.ckpt"))
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_parse
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_parse -o run -- python3 $REPO/tools/experiments/r4_run_cli.py > $REPO/gpurun_out/r4/parse_prof.log 2>&1
python3 $REPO/tools/summarize_prof.py trace /tmp/prof_parse $REPO/gpurun_out/r4/parse_kernel_trace_summary.txt
head -12 $REPO/gpurun_out/r4/parse_kernel_trace_summary.txt
