#!/usr/bin/env python3
"""How far do the opt-in split modes move the CLI's output?  Calls the same synthetic feature TSV with --precision fp32 and
with each split mode (same seed, in-kernel Philox states) and counts the per-read calls whose printed 6-decimal
probabilities or labels differ."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from deepsignal_plant_amd import synth
    from deepsignal_plant_amd.models import ModelBiLSTM
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    work = os.environ.get("DSP_WORK", "/tmp/dsp_prec")
    os.makedirs(work, exist_ok=True)
    ck = os.path.join(work, "model.ckpt")
    torch.save(synth.random_state_dict(ModelBiLSTM(), seed=1234, scale=2.0), ck)
    tsv = os.path.join(work, "feat.tsv")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_tsv.py"), tsv, str(n)], stdout=subprocess.DEVNULL)
    outs = {}
    for prec in ("fp32", "bf16x9", "bf16x6", "fp16x3"):
        out = os.path.join(work, "calls_%s.tsv" % prec)
        r = subprocess.run([sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant", "call_mods", "-i", tsv, "-m", ck, "-o", out,
                            "-p", "16", "--precision", prec, "--seed", "7"], cwd=ROOT, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        secs = float([l for l in r.stdout.splitlines() if "call_mods costs" in l][0].split("costs")[1].split("seconds")[0])
        outs[prec] = (open(out).read().splitlines(), secs)
    ref = outs["fp32"][0]
    for prec in ("bf16x9", "bf16x6", "fp16x3"):
        got, secs = outs[prec]
        assert len(got) == len(ref) == n
        diff = lab = 0
        worst = 0.0
        for a, b in zip(ref, got):
            if a != b:
                diff += 1
                wa, wb = a.split("\t"), b.split("\t")
                lab += wa[8] != wb[8]
                worst = max(worst, abs(float(wa[6]) - float(wb[6])))
        print(json.dumps({"precision": prec, "rows": n, "rows_with_a_different_printed_digit": diff, "labels_changed": lab,
                          "largest_printed_difference": worst, "call_mods_s": secs, "fp32_call_mods_s": outs["fp32"][1]}), flush=True)


if __name__ == "__main__":
    main()
