set -x
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0,'.')
from tests.helpers import load_f1, rows_to_tsv, GOLDEN
f = load_f1("randn_capture")
torch.save({k: torch.from_numpy(v) for k, v in f["w"].items()}, "/tmp/model.ckpt")
rows_to_tsv("/tmp/rows.tsv", *f["inputs"])
z = np.load(os.path.join(GOLDEN, "f1_randn_capture.npz"))
np.savez("/tmp/few.npz", **{k: z[k][:, :2] for k in z.files if k.startswith("state_")})
PY
for st in "file:tests/golden/f1_randn_capture.npz" "zeros" "file:/tmp/few.npz" "rand"; do
  T0=$(date +%s.%N); timeout 300 python -X faulthandler -c "
import faulthandler, sys
faulthandler.dump_traceback_later(25, repeat=True)
sys.argv=['deepsignal_plant','call_mods','-i','/tmp/rows.tsv','-m','/tmp/model.ckpt','-o','/tmp/o.tsv','--init_state','$st']
import os
os.environ['DSP_TIMING']='1'
from deepsignal_plant_amd.deepsignal_plant import main
main()
" 2>&1 | tail -40; python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),1), \"s wall: --init_state\", sys.argv[2])" $T0 "$st"
done
