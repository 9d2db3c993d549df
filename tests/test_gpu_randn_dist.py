"""F9 (GPU half): the HIP path's default run mode -- in-kernel Philox N(0,1) initial states -- against the distribution of the
reference's own outputs under torch.randn (tests/golden/f9_randn_dist.npz, tests/golden/make_golden_randn_dist.py;
deepsignal_plant/models.py:169-176).  See tests/test_randn_dist.py for the two-level argument and the statistics.

Here, through the C ABI, for the three models of F9 and all 64 rows with 4,096 draws per row (the reference: 1,024):
  * Philox mode passes;
  * explicit states drawn by torch.randn ON THE DEVICE (the reference's generator call, `.cuda()` side) pass;
  * the same draws mis-keyed -- h and c on one stream, sigma 0.9, both directions on one stream -- are rejected wherever the
    outputs can show it (negative controls: the test has the power it claims);
  * the kernel's keying IS the oracle's: a Philox-mode forward equals the oracle run on c_oracle.philox_states (the draws
    level 1 of tests/test_randn_dist.py examines) and differs visibly from the oracle run on any of the mis-keyed sets.
"""
import numpy as np
import pytest

from oracle import c_oracle as oc
from oracle import forward_np as onp
from tests.test_randn_dist import CONTROLS, f9_model, load_f9, output_violations, perturb

pytestmark = pytest.mark.gpu

DRAWS, CHUNK = 4096, 1024   # draws per row; draws per forward (x 64 rows = 65,536 sites)


def _hip_model(cfg, w, init_state="randn", seed=0):
    import torch
    from deepsignal_plant_amd.models import ModelBiLSTM
    m = ModelBiLSTM(cfg.seq_len, cfg.signal_len, cfg.num_layers1, cfg.num_layers2, cfg.num_classes, 0, cfg.hidden_size,
                    cfg.vocab_size, cfg.embedding_size, cfg.is_base, cfg.is_signallen, module=cfg.module, device=0,
                    init_state=init_state, seed=seed)
    m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()})
    return m.cuda(0).eval()


def _p1_over_draws(model, cfg, ins, sites, draws, states_fn=None):
    """p1[draws, sites]: the rows repeated draw-major, forwards of CHUNK draws; Philox mode (site_offset advancing with the
    global row index) or explicit states from states_fn(n_rows, chunk index)"""
    import torch
    dev = torch.device("cuda", 0)
    out = np.empty((draws, sites), np.float32)
    per = min(CHUNK, draws)
    rows = [torch.from_numpy(np.tile(a[:sites], (per,) + (1,) * (a.ndim - 1))).to(dev) for a in ins]
    for c in range(0, draws, per):
        model.site_offset = (1 << 36) + c * sites
        st = states_fn(per * sites, c // per) if states_fn is not None else None
        _lg, pr = model.forward(*rows, init_states=st)
        out[c:c + per] = pr[:, 1].float().cpu().numpy().reshape(per, sites)
        del st
    return out


def _device_randn_states(cfg, control=None):
    """states_fn: init_hidden's draws for a batch of n, made by torch.randn on the device under a fixed seed"""
    import torch
    gen = torch.Generator(device="cuda:0")

    def fn(n, k):
        gen.manual_seed(1000 + k)
        st = {name: torch.randn(*shape, generator=gen, device="cuda:0") for name, shape in onp.init_state_shapes(cfg, n)}
        return perturb(st, control) if control else st
    return fn


@pytest.mark.parametrize("model_name", ["default", "sharp_x3", "f8_trained_h256"])
def test_hip_default_mode_is_distributed_like_the_reference(model_name):
    d, _ = load_f9()
    cfg, w, ins = f9_model(d, model_name)
    ref = d["p1_" + model_name]
    m = _hip_model(cfg, w, "randn", seed=20241002)
    got = _p1_over_draws(m, cfg, ins, 64, DRAWS)
    bad, info = output_violations(got, ref)
    print("\n[F9] %-16s Philox, %d draws x 64 rows: %s" % (model_name, DRAWS, info))
    assert bad == [], (model_name, bad)
    got = _p1_over_draws(m, cfg, ins, 64, DRAWS, _device_randn_states(cfg))
    bad, info = output_violations(got, ref)
    print("[F9] %-16s torch.randn on the device:  %s" % (model_name, info))
    assert bad == [], (model_name, bad)


def test_the_output_level_rejects_mis_keyed_generators_at_this_sample_size():
    """negative controls through the HIP path: torch.randn's draws, mis-keyed.  Every control must be rejected on at least
    one of the three models -- except 'both directions on one stream' where the outputs may not carry it (level 1 of
    tests/test_randn_dist.py rejects it on the draws themselves); what each model shows is printed."""
    d, models = load_f9()
    caught = {c: [] for c in CONTROLS}
    for model_name in models:
        cfg, w, ins = f9_model(d, model_name)
        m = _hip_model(cfg, w, "randn", seed=1)
        for control in CONTROLS:
            got = _p1_over_draws(m, cfg, ins, 64, DRAWS, _device_randn_states(cfg, control))
            bad, info = output_violations(got, d["p1_" + model_name])
            print("\n[F9 control] %-16s %-30s -> %s | %s" % (model_name, control, "; ".join(bad) or "not visible", info))
            if bad:
                caught[control].append(model_name)
    assert caught["h_and_c_on_one_stream"] and caught["sigma_0p9"], caught
    assert len(caught["sigma_0p9"]) >= 2 and len(caught["h_and_c_on_one_stream"]) >= 2, caught


def test_the_kernels_keying_is_the_oracles():
    """the Philox-mode forward of the HIP path == the oracle on the explicit draws level 1 examined; any mis-keying of
    them moves the probabilities far beyond the parity tolerance, so a kernel keyed differently could not pass"""
    import torch
    d, _ = load_f9()
    cfg, w, ins = f9_model(d, "f8_trained_h256")
    n = 64 * 6
    rows = [np.tile(a, (6,) + (1,) * (a.ndim - 1)) for a in ins]
    m = _hip_model(cfg, w, "randn", seed=77)
    m.site_offset = 5000
    _lg, pr = m.forward(*[torch.from_numpy(a).cuda(0) for a in rows])
    hip = pr.cpu().numpy()
    st = oc.philox_states(cfg, n, 77, 5000)
    same = oc.forward(cfg, w, *rows, init_mode="explicit", states=st)[1]
    assert np.abs(hip - same).max() <= 2e-6
    for control in CONTROLS:
        other = oc.forward(cfg, w, *rows, init_mode="explicit", states=perturb(st, control))[1]
        dist = np.abs(hip - other).max()
        print("[F9 keying] %-30s max |dprob| vs the kernel's Philox mode: %.3e" % (control, dist))
        assert dist > 1e-3, control
