#!/usr/bin/env python3
"""A/B of the "x ahead" form of the small-batch LSTM launches (round 6, DSP_LSTM_XAHEAD=1; dsp_kernels.hip dsp_xahead_kernel):
same bytes, and what a forward costs with and without it.  One JSON line:
    {"identical": bool, "differs": [...], "ms": {"<model>/<n>": [ms without, ms with, ms with rings 8 deep]}}
Written with no GPU at hand (the round's GPU access was closed): the first run of this script on an MI355X is the form's first
contact with hardware -- tests/test_gpu_zz_extents.py runs it and holds the bytes; the timings are printed, not asserted."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODELS = [
    ("default", dict()),
    ("cfg3_seq_only", dict(module="seq_bilstm", num_layers1=2)),
    ("one_combined_layer_s40", dict(signal_len=40, hidden_size=256, num_layers1=1)),
    ("hid128", dict(hidden_size=128, num_layers1=2, num_layers2=2)),
]
SIZES = (1, 16, 33, 64, 100, 128, 200, 256, 257, 300)
SWITCHES = [("", {}), ("/round4_handoff", {"DSP_LSTM_HANDOFF": "0"}), ("/G2", {"DSP_LSTM_CLUSTER": "2"})]


def main():
    import torch
    from oracle import forward_np as onp
    from tests.test_gpu_parity import build_model, to_dev
    reps = int(os.environ.get("XAHEAD_AB_REPS", "40"))
    digest, ms = {}, {}
    for label, kw in MODELS:
        cfg = onp.OracleConfig(**kw)
        w = onp.make_weights(cfg, 4242, 2.0)
        for tag, sw in SWITCHES:
            if tag and label != "default":
                continue
            for xa in ("0", "1", "ring8"):
                for k in ("DSP_LSTM_HANDOFF", "DSP_LSTM_CLUSTER", "DSP_LSTM_XAHEAD_RING"):
                    os.environ.pop(k, None)
                os.environ.update(sw)
                os.environ["DSP_LSTM_XAHEAD"] = "0" if xa == "0" else "1"      # (read when the handle is made)
                if xa == "ring8":
                    os.environ["DSP_LSTM_XAHEAD_RING"] = "8"
                m = build_model(cfg, w, init_state="randn", seed=5)
                for n in SIZES:
                    ins = to_dev(onp.make_inputs(cfg, n, 9000 + n))
                    m.site_offset = 3 * n
                    probs = m(*ins)[1]
                    torch.cuda.synchronize()
                    digest.setdefault("%s%s/%d" % (label, tag, n), {})[xa] = hashlib.sha256(probs.cpu().numpy().tobytes()).hexdigest()[:16]
                    if not tag:
                        for _ in range(5):
                            m(*ins)
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(reps):
                            m(*ins)
                        torch.cuda.synchronize()
                        ms.setdefault("%s/%d" % (label, n), []).append(round((time.perf_counter() - t0) * 1e3 / reps, 4))
                # explicit states on one ragged size
                n = 45
                ins = to_dev(onp.make_inputs(cfg, n, 77))
                st = {k: torch.from_numpy(v).cuda(0) for k, v in onp.make_init_states(cfg, n, 78).items()}
                probs = m.forward(*ins, init_states=st)[1]
                torch.cuda.synchronize()
                digest.setdefault("%s%s/%d/explicit" % (label, tag, n), {})[xa] = hashlib.sha256(probs.cpu().numpy().tobytes()).hexdigest()[:16]
                del m
    os.environ.pop("DSP_LSTM_XAHEAD", None)
    os.environ.pop("DSP_LSTM_XAHEAD_RING", None)
    differs = sorted(k for k, v in digest.items() if not (v.get("0") == v.get("1") == v.get("ring8")))
    print(json.dumps({"identical": not differs, "differs": differs, "cases": len(digest), "ms": ms}, sort_keys=True))


if __name__ == "__main__":
    main()
