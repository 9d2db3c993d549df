set -x
cd $GRAFT_REPO_ROOT
W=/tmp/edge; rm -rf $W; mkdir -p $W/emptydir
python - <<'PY'
import torch, sys
sys.path.insert(0,'.')
from oracle import forward_np as onp
cfg=onp.OracleConfig(); w=onp.make_weights(cfg,23,2.0)
torch.save({k: torch.from_numpy(v) for k,v in w.items()}, "/tmp/edge/m.ckpt")
PY
CLI="python -m deepsignal_plant_amd.deepsignal_plant"
: > $W/empty.tsv
$CLI call_mods -i $W/empty.tsv -m $W/m.ckpt -o $W/o1.tsv --gzip 2>&1 | tail -2; ls -la $W/o1.tsv.gz; python -c "import gzip; print(repr(gzip.open('$W/o1.tsv.gz').read()))"
$CLI call_mods -i $W/emptydir -m $W/m.ckpt -o $W/o2.tsv 2>&1 | tail -3; ls -la $W/o2.tsv
$CLI extract -i $W/emptydir -o $W/f.tsv 2>&1 | tail -3; ls -la $W/f.tsv
$CLI call_freq -i $W/empty.tsv -o $W/freq.tsv 2>&1 | tail -3; ls -la $W/freq.tsv
$CLI call_mods -i $W/empty.tsv -m $W/m.ckpt -o $W/o3.tsv --freq_file $W/fq.tsv 2>&1 | tail -2; ls -la $W/fq.tsv
