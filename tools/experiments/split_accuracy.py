#!/usr/bin/env python3
"""What would emulating the fp32 products of the combined BiLSTM stack with split-bf16 MFMAs cost in accuracy?
(CPU experiment with the oracle; nothing here ships.)  Operands are split into 3 bf16 pieces (round to nearest
even at every level); a product keeps the NPROD largest piece products (9 = all, 6 = drop ml, lm, ll, 3 = hh, hm,
mh); accumulation is float64 here (the MFMA accumulates in fp32, like the fp32 path, so that part cancels).  Prints
max |dprob| against the float64 forward for the default model with ordinary and "sharp" (x3) weights."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import forward_np as onp  # noqa: E402


def bf16(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = x.astype(np.float32)
    hi = bf16(x)
    r = x - hi
    mid = bf16(r)
    lo = bf16(r - mid)
    return [hi.astype(np.float64), mid.astype(np.float64), lo.astype(np.float64)]


ORDER = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1), (1, 2), (2, 1), (2, 2)]  # decreasing magnitude


def split2_f16(x, scale=1.0):
    """two fp16 pieces (11 + 11 mantissa bits) of scale * x"""
    x = (x.astype(np.float32) * np.float32(scale))
    hi = x.astype(np.float16).astype(np.float32)
    lo = (x - hi).astype(np.float16).astype(np.float32)
    return [hi.astype(np.float64), lo.astype(np.float64)]


def split_matmul_f16(a, bt, wscale):
    """3 products hh + hl + lh on fp16 pieces; the weights (bt) are pre-scaled by a power of two"""
    A, B = split2_f16(a), split2_f16(bt, wscale)
    return (A[1] @ B[0] + A[0] @ B[1] + A[0] @ B[0]) / wscale


def split_matmul(a, bt, nprod):
    """a [n,K] @ bt[K,m] with every scalar product replaced by its NPROD largest piece products"""
    A, B = split3(a), split3(bt)
    out = 0.0
    for p, q in ORDER[:nprod]:
        out = out + A[p] @ B[q]
    return out


def forward_split(cfg, w, ins, states, nprod):
    """float64 forward, except the gate pre-activations of lstm_comb"""
    orig = onp.lstm_bidir

    def lstm(x, w_, prefix, layers, hid, h0, c0, dtype):
        if prefix != "lstm_comb" or nprod == 0:
            return orig(x, w_, prefix, layers, hid, h0, c0, dtype)
        n, L, _ = x.shape
        inp = x
        for k in range(layers):
            out = np.zeros((n, L, 2 * hid), dtype)
            for d, suf in enumerate(("", "_reverse")):
                wih = w_["%s.weight_ih_l%d%s" % (prefix, k, suf)]
                whh = w_["%s.weight_hh_l%d%s" % (prefix, k, suf)]
                b = (w_["%s.bias_ih_l%d%s" % (prefix, k, suf)].astype(dtype) + w_["%s.bias_hh_l%d%s" % (prefix, k, suf)].astype(dtype))
                h = h0[2 * k + d].astype(dtype)
                c = c0[2 * k + d].astype(dtype)
                for t in (range(L) if d == 0 else range(L - 1, -1, -1)):
                    if nprod < 0:  # fp16x3 with weights pre-scaled by 2^-nprod
                        g = split_matmul_f16(inp[:, t, :], wih.T, 2.0 ** -nprod) + split_matmul_f16(h, whh.T, 2.0 ** -nprod) + b
                    else:
                        g = split_matmul(inp[:, t, :], wih.T, nprod) + split_matmul(h, whh.T, nprod) + b
                    i_, f_, g_, o_ = g[:, :hid], g[:, hid:2 * hid], g[:, 2 * hid:3 * hid], g[:, 3 * hid:]
                    c = onp._sigmoid(f_) * c + onp._sigmoid(i_) * np.tanh(g_)
                    h = onp._sigmoid(o_) * np.tanh(c)
                    out[:, t, d * hid:(d + 1) * hid] = h
            inp = out
        return inp
    onp.lstm_bidir = lstm
    try:
        return onp.forward(cfg, w, *ins, states, dtype=np.float64)
    finally:
        onp.lstm_bidir = orig


def main():
    cfg = onp.OracleConfig()
    n = 256
    for scale in (1.0, 3.0):
        w = onp.make_weights(cfg, 11, scale)
        ins = onp.make_inputs(cfg, n, 12)
        states = onp.make_init_states(cfg, n, 13)
        _, ref = forward_split(cfg, w, ins, states, 0)
        _, p32 = onp.forward(cfg, w, *ins, states, dtype=np.float32)
        print("weights x%.0f: fp32 forward vs float64        max|dprob| = %.2e" % (scale, np.abs(p32 - ref).max()))
        for nprod in (9, 6, 3):
            _, p = forward_split(cfg, w, ins, states, nprod)
            print("weights x%.0f: bf16x%d products (float64 accumulate) max|dprob| = %.2e" % (scale, nprod, np.abs(p - ref).max()))
        for sh in (-1, -5, -9):  # weight pre-scale 2^0 (encoded -1 -> 2^1 .. keep simple), 2^5, 2^9
            _, p = forward_split(cfg, w, ins, states, sh)
            print("weights x%.0f: fp16x3 products, weights pre-scaled 2^%d  max|dprob| = %.2e" % (scale, -sh, np.abs(p - ref).max()))


if __name__ == "__main__":
    main()
