"""CPU: how dsp_forward cuts a call into pieces (round 6: from the model's launch geometry and the device's CU count, not from
five constants measured on the default model at 256 CUs -- ADVICE r5 medium).  dsp_debug_plan is the
host-only twin of the decision dsp_forward makes: same functions, no device.  A cut never changes a bit of the result (the GPU
suite's test_small_batch_kernels_do_not_change_a_bit holds that); what is checked here is that the cost model tracks the
measurements on record and that the plan is the cheapest cover under it."""
import ctypes
import itertools

import pytest

from deepsignal_plant_amd import _native as nat


def cfg(**kw):
    d = dict(seq_len=13, signal_len=16, num_layers1=3, num_layers2=1, num_classes=2, hidden_size=256, vocab_size=16,
             embedding_size=4, is_base=1, is_signallen=1, module=0)
    d.update(kw)
    return nat.ModelCfg(*[d[k] for k, _ in nat.ModelCfg._fields_])


def plan(c, n, cus=256):
    p, t = (ctypes.c_int64 * 9)(), (ctypes.c_double * 9)()
    k = nat.check(int(nat.lib().dsp_debug_plan(ctypes.byref(c), cus, n, p, t)))
    return [int(p[i]) for i in range(k)], [float(t[i]) for i in range(k)]


def cost(c, n, cus=256):
    """estimated microseconds of n sites as ONE piece (n inside one class)"""
    t = float(nat.lib().dsp_debug_piece_cost(ctypes.byref(c), cus, n))
    assert t > 0
    return t


DEFAULT, HID128, CFG3 = cfg(), cfg(hidden_size=128), cfg(module=1, num_layers1=2)

# measured on MI355X, ms per forward: profiles/r5/batch_sweep.txt (default, events on), DESIGN.md 3b (hid 128, configs[2]'s model)
MEASURED = [(DEFAULT, 512, 0.578, 0.08), (DEFAULT, 1024, 0.987, 0.08), (DEFAULT, 2048, 1.785, 0.08), (DEFAULT, 4096, 3.393, 0.08),
            (DEFAULT, 8192, 6.789, 0.08), (DEFAULT, 65536, 52.929, 0.08),
            (HID128, 512, 0.474, 0.2), (HID128, 1024, 0.479, 0.2), (HID128, 2048, 0.80, 0.2),
            (CFG3, 512, 0.413, 0.2), (CFG3, 1024, 0.710, 0.2)]


@pytest.mark.parametrize("c,n,ms,tol", MEASURED)
def test_the_cost_model_tracks_the_measurements_on_record(c, n, ms, tol):
    est = cost(c, n) / 1000.0
    assert abs(est - ms) <= tol * ms, (n, est, ms)


def test_the_default_model_keeps_its_cuts_on_256_compute_units():
    """what round 5's table decided for the sizes its A/B measured (profiles/r5/forward_split_ab.txt)"""
    want = {512: [512], 1100: [1024, 76], 2500: [2048, 452], 3000: [2048, 952], 4096: [4096], 4097: [4096, 1], 5000: [4096, 904],
            7000: [4096, 2048, 856], 8192: [8192], 9000: [8192, 808], 10000: [8192, 1808], 65536: [65536], 66000: [65536, 464],
            16384 + 8192: [16384 + 8192], 8193: [8192, 1], 14000: [8192, 4096, 1712]}
    for n, pieces in want.items():
        assert plan(DEFAULT, n)[0] == pieces, (n, plan(DEFAULT, n)[0])


def test_sizes_follow_the_compute_unit_count():
    """304 CUs (MI300X): a round is 9,728 sites, the workgroup-local forms reach 4,608 -- a 9,000-site call is ONE round there,
    not 8,192 + 808 (ADVICE r5's example); 128 CUs: half of everything"""
    assert plan(DEFAULT, 9000, 304)[0] == [9000]
    assert plan(DEFAULT, 9728, 304)[0] == [9728]
    assert plan(DEFAULT, 9729, 304)[0] == [9728, 1]
    assert plan(DEFAULT, 4609, 304)[0] == [4608, 1]
    assert plan(DEFAULT, 2 * 9728 + 544, 304)[0] == [2 * 9728, 544]
    assert plan(DEFAULT, 4096, 128)[0] == [4096]
    assert plan(DEFAULT, 4097, 128)[0] == [4096, 1]
    assert plan(DEFAULT, 2049, 128)[0] == [2048, 1]


def class_caps(cus):
    return sorted({(cus // d) // 16 * 16 * 32 for d in (16, 8, 4, 2, 1) if (cus // d) // 16 * 16 >= 16})


@pytest.mark.parametrize("c,label", [(DEFAULT, "default"), (HID128, "hid128"), (CFG3, "configs2"), (cfg(hidden_size=200), "hid200"),
                                     (cfg(module=2, num_layers1=1), "signal_only"), (cfg(hidden_size=320, num_layers1=2), "hid320")])
@pytest.mark.parametrize("cus", [256, 304])
def test_the_plan_is_the_cheapest_cover_under_the_cost_model(c, label, cus):
    caps = class_caps(cus)
    rnd = caps[-1]
    ccost = {cap: cost(c, cap, cus) + 40.0 for cap in caps}
    unit = 512
    for n in [1, 300, 513, 1100, 1500, 2500, 3000, 3600, 4097, 5000, 5700, 6500, 7000, 7700, rnd - 1]:
        if n >= rnd:
            continue
        pieces, _ = plan(c, n, cus)
        assert sum(pieces) == n and all(p > 0 for p in pieces)
        need = -(-n // unit)
        best = None
        for k in range(1, 5):   # brute force: every multiset of up to four classes that covers the call
            for combo in itertools.combinations_with_replacement(caps, k):
                if sum(cc // unit for cc in combo) >= need:
                    t = sum(ccost[cc] for cc in combo)
                    best = t if best is None or t < best else best
        # the plan's own cost: each piece at its class
        def cls(p):
            return next(cap for cap in caps if p <= cap)
        got = sum(ccost[cls(p)] for p in pieces)
        assert got <= best + 1e-6, (label, cus, n, pieces, got, best)
