"""CPU: the exactness statement of the split-precision modes.  bf16x9 / bf16x6 emulate an fp32 product by
products of bf16 PIECES: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), each rounded to nearest even -- three
8-bit significands (and their signs) for fp32's 24 bits.  The claim "hi + mid + lo == x exactly" is what makes bf16x9 ("all
nine piece products") exact up to the accumulation; it is checked here on the host packer's own arithmetic
(dsp_debug_split_bf16 = the statements of pack_lstm_dir_split), over 1e7 random fp32 bit patterns and every exponent, together
with the two places where it CANNOT hold and what happens there:
  * |x| < 2^-109: the low piece falls below bf16's smallest subnormal (2^-133).  The residue is < 2^-133 in absolute terms --
    against gate sums of order 1 and an fp32 accumulation that rounds at 2^-24 relative, nothing: harmless, not refused
    (a kernel that flushed bf16 subnormals entirely would lose at most 2^-126 per weight: the same verdict);
  * |x| > 3.3895e38 (the largest finite bf16): hi rounds to infinity and the residual is -inf: inf / NaN in the output --
    visible, never silently wrong.  (fp16x3, whose pieces have 5 exponent bits, IS refused outside its range:
    dsp_model_set_precision, tests/test_gpu_parity.py.)
The in-kernel split of the activations (dsp_kernels.hip split_bf16x3) is the same three statements on v_cvt's round-to-nearest-even."""
import ctypes

import numpy as np

from deepsignal_plant_amd import _native as nat


def split(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    hi, mid, lo = (np.empty(x.size, np.uint16) for _ in range(3))
    nat.lib().dsp_debug_split_bf16(x.ctypes.data, x.size, hi.ctypes.data, mid.ctypes.data, lo.ctypes.data)
    f = lambda h: (h.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    return f(hi), f(mid), f(lo)


def test_three_bf16_pieces_sum_to_the_fp32_value_exactly():
    rng = np.random.default_rng(8)
    bits = rng.integers(0, 1 << 32, 10_000_000, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    x = x[np.isfinite(x)]
    mag = np.abs(x.astype(np.float64))
    inside = (mag >= 2.0 ** -109) & (mag <= 3.3895313892515355e38)
    hi, mid, lo = split(x)
    with np.errstate(invalid="ignore", over="ignore"):
        s = hi + mid + lo                       # float64: the sum of three bf16 values of one fp32 is exact here
    assert inside.sum() > 8_000_000
    assert np.array_equal(s[inside], x[inside].astype(np.float64))
    # below 2^-109 the low piece may underflow: the residue stays under half of bf16's smallest subnormal step
    small = mag < 2.0 ** -109
    assert small.sum() > 100_000
    assert np.abs(s[small] - x[small].astype(np.float64)).max() <= 2.0 ** -134


def test_every_exponent_edge():
    """for every binade of fp32 (subnormals included): the smallest and largest significand, all-ones low bits, a lone lowest
    bit, the round-to-even ties of both cuts -- exact from 2^-109 up to the largest finite bf16"""
    pats = []
    for e in range(0, 255):
        for m in (0, 1, 0x7fffff, 0x7ffffe, 0x008000, 0x018000, 0x007fff, 0x008001, 0x000080, 0x000180, 0x00007f, 0x400000, 0x3fffff, 0x555555, 0x2aaaaa):
            for sgn in (0, 1):
                pats.append((sgn << 31) | (e << 23) | m)
    x = np.array(pats, dtype=np.uint32).view(np.float32)
    mag = np.abs(x.astype(np.float64))
    hi, mid, lo = split(x)
    with np.errstate(invalid="ignore", over="ignore"):
        s = hi + mid + lo
    inside = (mag >= 2.0 ** -109) & (mag <= 3.3895313892515355e38)
    bad = inside & (s != x.astype(np.float64))
    assert not bad.any(), x[bad][:10]
    # the edge of exactness is where it is said to be: somewhere below 2^-109 a value does lose its low bits ...
    lost = (mag < 2.0 ** -109) & (mag > 0) & (s != x.astype(np.float64))
    assert lost.any() and np.abs(s[lost] - x[lost].astype(np.float64)).max() <= 2.0 ** -134
    # ... and above the largest finite bf16 the high piece is infinite (visible)
    top = mag > 3.3895313892515355e38
    assert top.any() and np.isinf(hi[top]).all()
    # zero and the pieces of an exactly representable bf16 value
    z = split(np.array([0.0, -0.0, 1.0, -1.5, 3.3895313892515355e38], np.float32))
    assert np.array_equal(z[1], np.zeros(5)) and np.array_equal(z[2], np.zeros(5))
