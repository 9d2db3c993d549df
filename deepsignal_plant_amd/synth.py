"""Synthetic checkpoints and feature batches of the shape BASELINE.json names (no network: there is no
trained checkpoint or dataset in the tree).  Statistics follow SURVEY.md 8(d): kmer codes uniform{A,C,G,T}
with the centre forced to C, MAD-normalised signal statistics, centred zero padding of short bases
(the row producer's format, deepsignal_plant/extract_features.py:232-251, :381-395)."""
from __future__ import annotations

import math
from collections import OrderedDict

import torch


def random_state_dict(model, seed=1234, scale=1.0):
    """Seeded random state_dict with PyTorch's default-init scales: U(-1/sqrt(H), 1/sqrt(H)) for LSTM
    tensors, U(-1/sqrt(fan_in), ..) for Linear, N(0,1) for the embedding."""
    g = torch.Generator().manual_seed(int(seed))
    sd = OrderedDict()
    for name, shape in model._spec:
        if name == "embed.weight":
            t = torch.randn(shape, generator=g)
        else:
            if name.startswith("lstm"):
                k = 1.0 / math.sqrt(shape[0] // 4)
            else:
                fan_in = shape[1] if len(shape) == 2 else dict(model._spec)[name.replace(".bias", ".weight")][1]
                k = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2.0 - 1.0) * k
        sd[name] = (t * scale).float()
    return sd


def feature_batch(n, seq_len=13, signal_len=16, device="cuda:0", seed=0, compact=False):
    """(kmer, means, stds, lens, signals) device tensors for n sites; all float32 like the reference feeds
    them (call_modifications.py:159-162) unless compact=True (uint8 codes, uint16 lens)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(int(seed))
    L, S = seq_len, signal_len
    kmer = torch.randint(0, 4, (n, L), generator=g, device=dev, dtype=torch.int32)
    kmer[:, L // 2] = 1
    means = torch.round(torch.randn((n, L), generator=g, device=dev) * 1e6) / 1e6
    stds = torch.round((torch.randn((n, L), generator=g, device=dev) * 0.1 + 0.25).abs() * 1e6) / 1e6
    lens = torch.randint(2, 40, (n, L), generator=g, device=dev, dtype=torch.int32)
    sig = torch.round(torch.randn((n, L, S), generator=g, device=dev) * 1e6) / 1e6
    ln = lens.clamp(max=S).unsqueeze(-1)
    left = (S - ln) // 2
    idx = torch.arange(S, device=dev).view(1, 1, S)
    signals = torch.where((idx >= left) & (idx < left + ln), sig, torch.zeros_like(sig))
    if compact:
        return kmer.to(torch.uint8), means, stds, lens.to(torch.uint16), signals
    return kmer.float(), means, stds, lens.float(), signals
