import os, sys, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests.test_gpu_cli import _folded_rows
from oracle import forward_np as onp
td = tempfile.mkdtemp()
cfg = onp.OracleConfig()
w = onp.make_weights(cfg, 23, 2.0)
ck = os.path.join(td, "m.ckpt"); torch.save({k: torch.from_numpy(v) for k, v in w.items()}, ck)
data = _folded_rows(n_rep=6)
inp = os.path.join(td, "rows.tsv"); open(inp, "wb").write(data)
outs = {}
for mode in ("host", "device"):
    out = os.path.join(td, mode + ".tsv")
    env = dict(os.environ, DSP_BLOCK_BYTES=sys.argv[1] if len(sys.argv) > 1 else "120000")
    r = subprocess.run([sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant", "call_mods", "-i", inp, "-m", ck, "-o", out, "--seed", "3"] + (["--parse_on", mode] if not os.environ.get("DSP_PARSE_ON") else []),
                       cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    print("\n".join(l for l in r.stderr.splitlines() if "parse_dev" in l)[:3000])
    outs[mode] = open(out).read().splitlines()
a, b = outs["host"], outs["device"]
print(len(a), len(b))
bad = [i for i in range(min(len(a), len(b))) if a[i] != b[i]]
print("differing rows:", len(bad), bad[:40])
for i in bad[:5]:
    print(a[i]); print(b[i])
