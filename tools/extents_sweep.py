#!/usr/bin/env python3
"""Forwards over the shapes and batch sizes that exercise every kernel form, digested: one JSON line {case: sha256 of the
probabilities}.  Run it once per library / extents mode and compare the lines (tests/test_gpu_zz_extents.py does):
    DSP_AMD_LIB=deepsignal_plant_amd/libdsp_amd_bounds.so python tools/extents_sweep.py    # the bounds-recording build
    DSP_RSRC_EXTENTS=wide python tools/extents_sweep.py                                    # the 2 GiB windows of rounds 1-5
A DSP_EBOUNDS from the bounds build ends the sweep with its message (exit status 3)."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [
    # (label, OracleConfig kwargs, batch sizes, precisions)
    ("default", dict(), (1, 33, 300, 512, 1024, 1100, 2048, 3000, 4096, 4097, 9001, 16500), ("fp32", "bf16x9", "fp16x3")),
    ("cfg3_seq_only", dict(module="seq_bilstm", num_layers1=2), (31, 512, 1024, 4097, 9001), ("fp32", "bf16x6")),
    ("signal_only", dict(module="signal_bilstm", num_layers1=1), (33, 513, 4100), ("fp32",)),
    ("hid128", dict(hidden_size=128, num_layers1=2, num_layers2=2), (1, 512, 1024, 2048, 5000), ("fp32", "bf16x9")),
    ("hid200_padded", dict(hidden_size=200, num_layers1=2), (65, 1025, 4097), ("fp32",)),
    ("hid100_ut4_padded", dict(hidden_size=100), (64, 700), ("fp32",)),
    ("hid64_ut2", dict(hidden_size=64, num_layers2=2, is_base=False), (33, 2000), ("fp32",)),
    ("hid320_many_pass", dict(hidden_size=320, num_layers1=2), (40, 600, 4200), ("fp32",)),
    ("hid640_three_classes", dict(hidden_size=640, num_layers1=1, num_classes=3, is_signallen=False), (50, 300), ("fp32",)),
    ("k9_s24", dict(seq_len=9, signal_len=24, hidden_size=96), (100, 1500), ("fp32",)),
    ("s40_wide_window", dict(signal_len=40, hidden_size=256, num_layers1=1), (90, 600), ("fp32",)),
]


def main():
    import numpy as np
    import torch
    from oracle import forward_np as onp
    from tests.test_gpu_parity import build_model, to_dev
    only = set(sys.argv[1:])
    out = {}
    for label, kw, sizes, precisions in CASES:
        if only and label not in only:
            continue
        cfg = onp.OracleConfig(**kw)
        w = onp.make_weights(cfg, 4242, 2.0)
        for precision in precisions:
            m = build_model(cfg, w, init_state="randn", seed=5)
            try:
                m.set_precision(precision)
            except ValueError:
                continue   # fp16 pieces refused for this checkpoint (documented)
            for n in sizes:
                ins = onp.make_inputs(cfg, n, 9000 + n)
                m.site_offset = 3 * n
                try:
                    probs = m(*to_dev(ins))[1]
                    torch.cuda.synchronize()
                except RuntimeError as e:
                    print("%s %s n=%d: %s" % (label, precision, n, e), file=sys.stderr)
                    sys.exit(3)
                out["%s/%s/%d/philox" % (label, precision, n)] = hashlib.sha256(probs.cpu().numpy().tobytes()).hexdigest()[:16]
            # explicit N(0,1) states (the reference's layout) and zero states on one ragged size
            n = sizes[1] if len(sizes) > 1 else sizes[0]
            ins = onp.make_inputs(cfg, n, 77)
            st = {k: torch.from_numpy(v).cuda(0) for k, v in onp.make_init_states(cfg, n, 78).items()}
            try:
                probs = m.forward(*to_dev(ins), init_states=st)[1]
                mz = build_model(cfg, w, init_state="zeros")
                mz.set_precision(precision)
                pz = mz(*to_dev(ins))[1]
                torch.cuda.synchronize()
            except RuntimeError as e:
                print("%s %s n=%d explicit/zeros: %s" % (label, precision, n, e), file=sys.stderr)
                sys.exit(3)
            out["%s/%s/%d/explicit" % (label, precision, n)] = hashlib.sha256(probs.cpu().numpy().tobytes()).hexdigest()[:16]
            out["%s/%s/%d/zeros" % (label, precision, n)] = hashlib.sha256(pz.cpu().numpy().tobytes()).hexdigest()[:16]
    print(json.dumps(out, sort_keys=True))


if __name__ == "__main__":
    main()
