"""F9 (CPU half): the Philox stand-in of the default run mode against the reference's OWN draws.

The reference draws h0, c0 ~ N(0,1) with torch.randn on every forward (deepsignal_plant/models.py:169-176); this build's
default mode draws them in the kernel (Philox4x32-10 + Box-Muller).  Every other "Philox parity" test compares the HIP
path with an oracle that implements the same generator, so a mis-keyed generator (h and c sharing a counter, a variance
of 0.9, both directions on one stream) would pass them all.  Here the generator is held against the reference itself,
in two levels -- and each level is shown to REJECT the three mis-keyings (negative controls):

  level 1, the draws: what the oracle's generator hands the three LSTMs (c_oracle.philox_states -- bit for bit what the
    forward uses, asserted) against what torch.randn hands them under torch.manual_seed (the very call of init_hidden):
    moments, normality, two-sample KS, and independence across {h, c}, directions, layers, LSTMs, sites, units, seeds.
  level 2, the outputs: p1 over draws for 64 fixed rows and three models against tests/golden/f9_randn_dist.npz
    (make_golden_randn_dist.py: the reference's forward under torch.manual_seed(0..255), 1,024 draws per row):
    per-row location and spread, per-row and pooled two-sample KS.

Why both: given the states, this build's forward equals the reference's to 1e-6 (F1 / F8 fixtures), so equality of the
output distributions FOLLOWS from equality of the state distributions (level 1); level 2 checks the chain end to end,
but the outputs are by construction insensitive to some mis-keyings (both directions on one stream moves the spread of
p1 by < 1 %), which only level 1 can see.  tests/test_gpu_randn_dist.py runs level 2 for the HIP path with 4,096 draws
per row and ties the kernel's keying to the oracle's.
"""
import os

import numpy as np
import pytest
from scipy import stats

from oracle import c_oracle as oc
from oracle import forward_np as onp
from tests.helpers import GOLDEN

F9 = os.path.join(GOLDEN, "f9_randn_dist.npz")
CONTROLS = ("h_and_c_on_one_stream", "sigma_0p9", "both_directions_on_one_stream")


def perturb(states, control):
    """the three mis-keyings of VERDICT r3 item 2, applied to a dict of states in init_hidden's layout (numpy or torch)"""
    out = {k: (v.copy() if isinstance(v, np.ndarray) else v.clone()) for k, v in states.items()}
    if control == "h_and_c_on_one_stream":
        for k in list(out):
            if k.startswith("c_"):
                out[k] = out["h_" + k[2:]].copy() if isinstance(out[k], np.ndarray) else out["h_" + k[2:]].clone()
    elif control == "sigma_0p9":
        for k in out:
            out[k] = out[k] * 0.9
            if isinstance(out[k], np.ndarray):
                out[k] = out[k].astype(np.float32)
    elif control == "both_directions_on_one_stream":
        for k in out:
            out[k][1::2] = out[k][0::2]
    else:
        raise KeyError(control)
    return out


# ---- level 1: the draws ------------------------------------------------------------------------------------------------
def draws_violations(st, ref):
    """st, ref: dicts of float arrays in init_hidden's layout (2 * layers, n, H), n even.  Returns the list of checks that
    `st` fails as a set of i.i.d. N(0,1) draws distributed like `ref` (torch.randn's)."""
    bad = []
    allv = np.concatenate([v.ravel() for v in st.values()]).astype(np.float64)
    N = allv.size
    z = 5.0
    if abs(allv.mean()) > z / np.sqrt(N):
        bad.append("mean %.5f" % allv.mean())
    if abs(allv.std() - 1.0) > z / np.sqrt(2 * N):
        bad.append("std %.5f" % allv.std())
    if abs((allv ** 4).mean() - 3.0) > z * np.sqrt(96.0 / N):
        bad.append("fourth moment %.4f" % (allv ** 4).mean())
    sub = allv[:: max(1, N // 400000)]
    if stats.kstest(sub, "norm").pvalue < 1e-3:
        bad.append("KS against N(0,1)")
    refv = np.concatenate([v.ravel() for v in ref.values()]).astype(np.float64)
    if stats.ks_2samp(sub, refv[:: max(1, refv.size // 400000)]).pvalue < 1e-3:
        bad.append("two-sample KS against torch.randn")

    def corr(a, b, what):
        a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
        r = float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))
        if abs(r) > z / np.sqrt(a.size):
            bad.append("corr(%s) = %.4f" % (what, r))
    for name in ("seq", "sig", "comb"):
        if "h_" + name not in st:
            continue
        h, c = st["h_" + name], st["c_" + name]
        corr(h, c, "h, c of " + name)
        corr(h[0::2], h[1::2], "forward, backward h of " + name)
        corr(c[0::2], c[1::2], "forward, backward c of " + name)
        corr(h[:, 0::2], h[:, 1::2], "neighbouring sites of " + name)
        corr(h[:, :, 0::2], h[:, :, 1::2], "neighbouring units of " + name)
        corr(h[:, :, :-4], h[:, :, 4:], "units 4 apart (Philox blocks) of " + name)
        corr(h[:, :, 0::4], h[:, :, 1::4], "Box-Muller partners of " + name)
        corr(h[:, :, 0::4], c[:, :, 1::4], "h, c' of " + name)
        if h.shape[0] >= 4:
            corr(h[0:-2], h[2:], "layers l, l+1 of " + name)
    if "h_seq" in st and "h_sig" in st:
        corr(st["h_seq"], st["h_sig"], "seq, signal LSTM")
        corr(st["h_seq"][0], st["h_comb"][0][:, :st["h_seq"].shape[2]], "seq, combined LSTM")
    return bad


def torch_randn_states(cfg, n, seed):
    """what init_hidden draws for a batch of n under torch.manual_seed(seed): the reference's own call sequence
    (models.py:169-176 called at :196-198, :212-214, :226-228): h then c, for seq, signal, combined"""
    import torch
    torch.manual_seed(seed)
    return {k: torch.randn(*shape).numpy() for k, shape in onp.init_state_shapes(cfg, n)}


def test_philox_states_are_what_the_forward_draws():
    """c_oracle.philox_states is the generator of init_mode='philox', not a restatement of it"""
    cfg = onp.OracleConfig(hidden_size=64, num_layers1=2)
    w = onp.make_weights(cfg, 3, 2.0)
    ins = onp.make_inputs(cfg, 37, 4)
    a = oc.forward(cfg, w, *ins, init_mode="philox", seed=99, site_offset=12345)
    b = oc.forward(cfg, w, *ins, init_mode="explicit", states=oc.philox_states(cfg, 37, 99, 12345))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    keys = np.arange(37, dtype=np.uint64) * 7919 + (1 << 40)
    a = oc.forward(cfg, w, *ins, init_mode="philox", seed=5, site_keys=keys)
    b = oc.forward(cfg, w, *ins, init_mode="explicit", states=oc.philox_states(cfg, 37, 5, site_keys=keys))
    assert np.array_equal(a[1], b[1])


def test_level1_the_draws_are_iid_standard_normal_like_torch_randn():
    cfg = onp.OracleConfig()
    n = 512
    ref = torch_randn_states(cfg, n, 0)
    assert draws_violations(ref, torch_randn_states(cfg, n, 1)) == []          # the yardstick passes its own test
    for seed, off in ((0, 0), (2024, 1 << 33), (7, 65536 * 153)):
        st = oc.philox_states(cfg, n, seed, off)
        assert draws_violations(st, ref) == [], (seed, off)
    # consecutive seeds and consecutive batches are independent draws too
    a, b = oc.philox_states(cfg, n, 11, 0), oc.philox_states(cfg, n, 12, 0)
    c = oc.philox_states(cfg, n, 11, n)
    for x, y, what in ((a, b, "seeds"), (a, c, "batches")):
        for k in a:
            r = float((x[k].astype(np.float64) * y[k]).mean())
            assert abs(r) < 5 / np.sqrt(a[k].size), (what, k, r)


@pytest.mark.parametrize("control", CONTROLS)
def test_level1_rejects_the_mis_keyed_generators(control):
    cfg = onp.OracleConfig()
    ref = torch_randn_states(cfg, 512, 0)
    bad = draws_violations(perturb(oc.philox_states(cfg, 512, 0, 0), control), ref)
    print(control, "->", bad)
    assert bad, control
    assert draws_violations(perturb(ref, control), ref), "the control must fail on torch.randn's own draws as well"


# ---- level 2: the outputs ----------------------------------------------------------------------------------------------
def load_f9():
    d = np.load(F9)
    return d, [str(m) for m in d["models"]]


def f9_model(d, name):
    """(weights, the 64 rows) of one of F9's models"""
    cfg = onp.OracleConfig()
    i = [str(m) for m in d["models"]].index(name)
    if name.startswith("f8"):
        from tests.helpers import load_f8
        f8 = load_f8(256)
        return cfg, f8["w"], [np.ascontiguousarray(a[:int(d["n_sites"])]) for a in f8["inputs"]]
    return cfg, onp.make_weights(cfg, int(d["wseeds"][i]), float(d["wscales"][i])), list(onp.make_inputs(cfg, int(d["n_sites"]), int(d["iseed"])))


def output_violations(got, ref, z=5.0, p_site=1e-6, p_pool=1e-3):
    """got [draws, sites], ref [1024, sites]: p1 over draws of the same rows.  Location and spread by order statistics (the
    trained model's p1 has heavy tails: rare draws flip a confident call, and a standard deviation estimated from 256 or
    1,024 draws is then dominated by a handful of them -- the reference against ITSELF fails a +-15 % std band there).
    Returns (violations, numbers to print)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    ng, nr = got.shape[0], ref.shape[0]
    q = lambda a, p: np.quantile(a, p, axis=0)
    med_r, med_g = q(ref, 0.5), q(got, 0.5)
    iqr_r, iqr_g = q(ref, 0.84) - q(ref, 0.16), q(got, 0.84) - q(got, 0.16)
    # median: asymptotic sd = 1.2533 * sigma / sqrt(n) with sigma ~ iqr / 2 (normal core)
    se_med = 1.2533 * (iqr_r / 2) * np.sqrt(1.0 / ng + 1.0 / nr)
    z_loc = np.abs(med_g - med_r) / np.maximum(se_med, 1e-12)
    # spread: the 16-84 % range; relative sd of one estimate ~ 1.04 / sqrt(n) (normal core, two quantiles at +-1 sigma)
    ratio = iqr_g / np.maximum(iqr_r, 1e-12)
    se_ratio = 1.04 * np.sqrt(1.0 / ng + 1.0 / nr)
    bad = []
    if z_loc.max() > z:
        bad.append("median of row %d off by %.1f standard errors" % (int(z_loc.argmax()), z_loc.max()))
    if np.abs(ratio - 1).max() > z * se_ratio:
        bad.append("spread of row %d: ratio %.3f (allowed +-%.3f)" % (int(np.abs(ratio - 1).argmax()), ratio[np.abs(ratio - 1).argmax()], z * se_ratio))
    pooled = float(np.median(ratio))
    if abs(pooled - 1) > z * se_ratio / np.sqrt(ratio.size) * 1.2533:
        bad.append("median spread ratio over the rows %.4f" % pooled)
    # rows whose p1 is light-tailed (excess kurtosis < 1: the linear regime of untrained weights): there the classical
    # standard deviation is a sharp estimate, and its ratio pooled over the rows sees a few per cent of scale.  The rows are
    # chosen on the POOLED sample of both sides: choosing them on the reference sample alone picks, among heavy-tailed rows,
    # the ones whose 1,024 reference draws happened to hold no excursion -- and whose reference std is therefore biased low
    # (measured on the trained model: the correct generator came out at 1.068 on rows selected that way)
    kurt = stats.kurtosis(np.concatenate([got, ref], axis=0), axis=0)
    light = kurt < 1.0
    if light.sum() >= 8:
        lr = np.log(got[:, light].std(0, ddof=1) / ref[:, light].std(0, ddof=1))
        se = np.sqrt((1.0 / (2 * ng) + 1.0 / (2 * nr)) * (1 + 0.5 * max(0.0, float(kurt[light].mean()))) / light.sum())
        if abs(lr.mean()) > z * se:
            bad.append("pooled std ratio over the %d light-tailed rows: %.4f (allowed +-%.4f)" % (light.sum(), np.exp(lr.mean()), z * se))
    p_rows = np.array([stats.ks_2samp(got[:, j], ref[:, j]).pvalue for j in range(got.shape[1])])
    if p_rows.min() < p_site:
        bad.append("two-sample KS of row %d: p = %.1e" % (int(p_rows.argmin()), p_rows.min()))
    zg = ((got - med_r) / np.maximum(iqr_r, 1e-12)).ravel()
    zr = ((ref - med_r) / np.maximum(iqr_r, 1e-12)).ravel()
    ks = stats.ks_2samp(zg, zr)
    if ks.pvalue < p_pool:
        bad.append("pooled two-sample KS: D = %.4f, p = %.1e" % (ks.statistic, ks.pvalue))
    flip_r = ((ref > 0.5) != (med_r > 0.5)[None, :]).mean(0)
    flip_g = ((got > 0.5) != (med_r > 0.5)[None, :]).mean(0)
    se_flip = np.sqrt(np.maximum(flip_r * (1 - flip_r), 1e-4) * (1.0 / ng + 1.0 / nr))
    if (np.abs(flip_g - flip_r) / se_flip).max() > z + 1:
        bad.append("label-flip rate of row %d: %.3f vs %.3f" % (int((np.abs(flip_g - flip_r) / se_flip).argmax()),
                                                                flip_g[(np.abs(flip_g - flip_r) / se_flip).argmax()], flip_r[(np.abs(flip_g - flip_r) / se_flip).argmax()]))
    info = "median off <= %.2f se, spread ratio %.3f..%.3f (median %.4f), min row KS p %.1e, pooled KS D %.4f p %.2e, flip rate %.4f vs %.4f" % (
        z_loc.max(), ratio.min(), ratio.max(), pooled, p_rows.min(), ks.statistic, ks.pvalue, flip_g.mean(), flip_r.mean())
    return bad, info


def oracle_p1(cfg, w, ins, sites, draws, seed, control=None):
    """p1[draws, sites] of the first `sites` rows from the oracle: Philox mode, or -- for a control -- the same draws
    perturbed and fed back as explicit states"""
    rows = [np.tile(a[:sites], (draws,) + (1,) * (a.ndim - 1)) for a in ins]    # draw-major: row r = draw r // sites
    n = sites * draws
    if control is None:
        _lg, pr = oc.forward(cfg, w, *rows, init_mode="philox", seed=seed)
    else:
        _lg, pr = oc.forward(cfg, w, *rows, init_mode="explicit", states=perturb(oc.philox_states(cfg, n, seed), control))
    return pr[:, 1].reshape(draws, sites)


def test_f9_fixture_is_self_consistent():
    d, models = load_f9()
    assert models == ["default", "sharp_x3", "f8_trained_h256"] and int(d["n_sites"]) == 64
    for m in models:
        p1 = d["p1_" + m]
        assert p1.shape == (1024, 64) and p1.dtype == np.float32 and np.isfinite(p1).all() and 0 <= p1.min() and p1.max() <= 1
        s = d["summary_" + m]
        assert np.allclose(s[0], p1.mean(0, dtype=np.float64)) and np.allclose(s[3], np.median(p1, axis=0), atol=1e-6)
        # the reference against itself (odd against even seeds) passes the test its stand-in has to pass
        bad, info = output_violations(p1[0::2], p1[1::2])
        assert bad == [], (m, bad)
    dm = d["draw_moments"]
    assert abs(dm[0]) < 1e-3 and abs(dm[1] - 1) < 1e-3 and abs(dm[4]) < 5e-3 and abs(dm[5]) < 5e-3 and abs(dm[8] - 3) < 0.02


# the oracle's forwards of the level-2 tests (15-30 s each, in C with the GIL released) are background jobs (tests/bgjobs.py):
# started when collection ends, next to the tests in front of them
from tests import bgjobs  # noqa: E402


def _p1_job(model, sites, control=None):
    d, _ = load_f9()
    cfg, w, ins = f9_model(d, model)
    return oracle_p1(cfg, w, ins, sites, 256, seed=4242, control=control)


for _m in ("default", "sharp_x3", "f8_trained_h256"):
    bgjobs.job("randn_p1_" + _m)(lambda m=_m: _p1_job(m, 12))
bgjobs.job("randn_ctrl_h_and_c_on_one_stream")(lambda: _p1_job("f8_trained_h256", 16, "h_and_c_on_one_stream"))
bgjobs.job("randn_ctrl_sigma_0p9")(lambda: _p1_job("default", 32, "sigma_0p9"))


@pytest.mark.parametrize("model", ["default", "sharp_x3", "f8_trained_h256"])
@bgjobs.uses(lambda p: ["randn_p1_" + p["model"]])
def test_level2_oracle_philox_outputs_are_distributed_like_the_references(model):
    """12 rows x 256 Philox draws from the oracle against the reference's 1,024 torch.randn draws of the same rows
    (the GPU half does all 64 rows x 4,096 draws)"""
    d, _ = load_f9()
    got = bgjobs.result("randn_p1_" + model)
    bad, info = output_violations(got, d["p1_" + model][:, :12])
    print(model, info)
    assert bad == [], (model, bad, info)


@bgjobs.uses(lambda p: ["randn_ctrl_h_and_c_on_one_stream", "randn_ctrl_sigma_0p9"])
def test_level2_rejects_what_it_can_see_of_the_mis_keyed_generators():
    """What the OUTPUT level sees of the three mis-keyings with the few hundred draws the CPU suite can afford: 'h and c on
    one stream' on the trained model (the most sensitive to its states: pooled KS), 'sigma 0.9' on the default-scale
    weights (linear regime: the spread of p1 follows the spread of the states, pooled std ratio 0.94).  'Both directions
    on one stream' leaves the distribution of p1 where it was (spread within 1 %): level 1 rejects it (asserted above),
    here it is only shown not to be visible.  The GPU half repeats all three with 4,096 draws of all 64 rows."""
    d, _ = load_f9()
    seen = {}
    # ('both directions on one stream' is not run here: 256 draws of 16 rows showed a spread ratio of 1.017 and a pooled KS p
    # of 0.69 -- nothing; the GPU half prints what 4,096 draws of 64 rows show of it)
    for model, sites, controls in (("f8_trained_h256", 16, ("h_and_c_on_one_stream",)), ("default", 32, ("sigma_0p9",))):
        ref = d["p1_" + model][:, :sites]
        for control in controls:
            got = bgjobs.result("randn_ctrl_" + control)
            bad, info = output_violations(got, ref)
            print(model, control, "->", bad or "not visible in the outputs", "|", info)
            seen[control] = bool(bad)
    assert seen["h_and_c_on_one_stream"] and seen["sigma_0p9"]
