#!/usr/bin/env python3
"""F9 -- what the reference's DEFAULT run mode looks like as a distribution (build container only: imports /root/reference).

The reference draws the LSTM initial states h0, c0 ~ N(0,1) with torch.randn on EVERY forward
(deepsignal_plant/models.py:169-176, used at :196-198, :212-214, :226-228): the same row gives a different probability
each time it is called.  This build's default mode draws them in the kernel from Philox4x32-10 + Box-Muller; every other
parity test compares it with an oracle that implements the SAME generator, so a mis-keyed stream (h and c sharing a
counter, a variance of 0.9, correlated directions) would pass them all.  This fixture records the reference's own output
distribution so that the stand-in can be held against it:

  for each of three models -- seeded default-scale weights, the same x3 ("sharp"), and F8's checkpoint that the reference's
  `train` optimised (hid_rnn 256, the default architecture) -- and 64 fixed rows: ModelBiLSTM.forward under
  torch.manual_seed(0..255), the 64 rows repeated 4 times in the batch (rows are independent: 1,024 draws per row),
  p1 = softmax[:, 1] of every draw as float32  ->  p1[model][1024, 64]

plus the per-row summary the tests print (mean / std / 5-50-95 % quantiles, label-flip rate) and, from the same seeds, the
moments of the DRAWS themselves (mean, std, correlation between h and c, between the two directions, between layers).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_randn_dist.py      (about four minutes on 8 cores)
Output: tests/golden/f9_randn_dist.npz (data only).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import torch  # noqa: E402

from deepsignal_plant.models import ModelBiLSTM  # noqa: E402  (the reference)
from oracle import forward_np as onp  # noqa: E402

N_SITES, N_SEEDS, REPS = 64, 256, 4
ISEED = 909
MODELS = [("default", 31, 1.0), ("sharp_x3", 32, 3.0), ("f8_trained_h256", None, None)]


def build_ref(cfg, weights):
    m = ModelBiLSTM(cfg.seq_len, cfg.signal_len, cfg.num_layers1, cfg.num_layers2, cfg.num_classes, 0, cfg.hidden_size,
                    cfg.vocab_size, cfg.embedding_size, cfg.is_base, cfg.is_signallen, module=cfg.module, device=0)
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()})
    m.eval()
    return m


def inputs_for(name, cfg):
    """the 64 rows: synthetic rows of the bench's statistics; for the trained model the first 64 of the rows it was
    evaluated on in F8 (they carry the 3-mer structure it learnt)"""
    if name.startswith("f8"):
        from tests.helpers import load_f8
        f8 = load_f8(256)
        return [np.ascontiguousarray(a[:N_SITES]) for a in f8["inputs"]], f8["w"]
    return list(onp.make_inputs(cfg, N_SITES, ISEED)), None


def main():
    torch.set_num_threads(8)
    cfg = onp.OracleConfig()
    out = {"n_sites": N_SITES, "n_seeds": N_SEEDS, "reps": REPS, "iseed": ISEED,
           "models": np.array([m[0] for m in MODELS]), "wseeds": np.array([m[1] or -1 for m in MODELS]),
           "wscales": np.array([m[2] or 0.0 for m in MODELS])}
    for name, wseed, wscale in MODELS:
        ins, w = inputs_for(name, cfg)
        if w is None:
            w = onp.make_weights(cfg, wseed, wscale)
        model = build_ref(cfg, w)
        tin = [torch.from_numpy(np.tile(a, (REPS,) + (1,) * (a.ndim - 1))) for a in ins]
        p1 = np.empty((N_SEEDS * REPS, N_SITES), np.float32)
        with torch.no_grad():
            for s in range(N_SEEDS):
                torch.manual_seed(s)
                _lg, pr = model(*tin)
                p1[s * REPS:(s + 1) * REPS] = pr[:, 1].numpy().reshape(REPS, N_SITES)
        out["p1_" + name] = p1
        q = np.quantile(p1.astype(np.float64), [0.05, 0.5, 0.95], axis=0)
        maj = (np.median(p1, axis=0) > 0.5)
        flip = ((p1 > 0.5) != maj[None, :]).mean(axis=0)
        out["summary_" + name] = np.stack([p1.mean(0, dtype=np.float64), p1.std(0, dtype=np.float64, ddof=1), q[0], q[1], q[2], flip])
        print("%-16s  std of p1 over draws: median %.4f, max %.4f; label-flip rate: mean %.4f, max %.3f" % (
            name, np.median(p1.std(0)), p1.std(0).max(), flip.mean(), flip.max()))
    # the draws themselves (init_hidden, models.py:169-176): what torch.randn hands the three LSTMs under those seeds
    hs, cs = [], []
    for s in range(32):
        torch.manual_seed(s)
        h = torch.randn(6, 64, 256).numpy()   # (2 * layers, batch, hidden): one h, then one c, as init_hidden draws them
        c = torch.randn(6, 64, 256).numpy()
        hs.append(h)
        cs.append(c)
    h, c = np.stack(hs).astype(np.float64), np.stack(cs).astype(np.float64)   # [seed, layer*dir, site, unit]
    corr = lambda a, b: float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))
    out["draw_moments"] = np.array([h.mean(), h.std(), c.mean(), c.std(), corr(h, c), corr(h[:, 0], h[:, 1]),
                                    corr(h[:, 0], h[:, 2]), float(((np.abs(h) > 3).mean())), float((h ** 4).mean())])
    print("draws: mean %.4f std %.4f | c: %.4f %.4f | corr(h,c) %.4f corr(fwd,bwd) %.4f corr(l0,l1) %.4f | P(|x|>3) %.5f, E x^4 %.3f"
          % tuple(out["draw_moments"]))
    np.savez_compressed(os.path.join(HERE, "f9_randn_dist.npz"), **out)
    print("wrote", os.path.join(HERE, "f9_randn_dist.npz"))


if __name__ == "__main__":
    main()
