"""GPU parity: the HIP path (through the C ABI) against the golden fixtures captured from the reference
and against the CPU oracle on seeded inputs.  Contract (BASELINE.json north_star): per-site probabilities
within 1e-4 (fp32) of the reference forward on identical feature rows with pinned initial states."""
import numpy as np
import pytest

from tests.helpers import f1_names, f1_tolerances, load_f1

pytestmark = pytest.mark.gpu

TOL_PROB = 1e-4   # the contract
TOL_TIGHT = 2e-5  # what fp32 MFMA + v_exp/v_rcp actually achieves; a regression guard


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def build_model(cfg, w, **kw):
    torch = _torch()
    from deepsignal_plant_amd.models import ModelBiLSTM
    m = ModelBiLSTM(cfg.seq_len, cfg.signal_len, cfg.num_layers1, cfg.num_layers2, cfg.num_classes, 0,
                    cfg.hidden_size, cfg.vocab_size, cfg.embedding_size, cfg.is_base, cfg.is_signallen,
                    module=cfg.module, device=0, **kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return m.cuda(0).eval()


def to_dev(arrs):
    torch = _torch()
    return [torch.from_numpy(np.ascontiguousarray(a)).cuda(0) for a in arrs]


PRECISIONS = ["fp32", "bf16x9", "bf16x6", "fp16x3"]   # include/dsp_amd.h DSP_PREC_*: fp32 is the product, the rest opt-in


def _set_precision_or_refusal(m, precision):
    """True when the mode is on; False when the library REFUSED it for this checkpoint (fp16 pieces for operands that are
    not provably inside the fp16 range: the documented refusal, dsp_model_set_precision) -- anything else raises"""
    try:
        m.set_precision(precision)
        return True
    except ValueError as e:
        assert precision == "fp16x3" and "fp16 range" in str(e), (precision, e)
        return False


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", f1_names())
def test_hip_matches_reference_fixture(name, precision):
    """All 17 F1 fixtures -- the saturating ladder x5 ... x8, the extreme rows, every model variant -- in EVERY precision
    mode with the SAME per-fixture bounds as fp32 (round 5, VERDICT r4 weak 5: the split-precision modes quoted beside
    `value` were pinned on 3 fixtures only).  Layers a split mode has no form for (hidden > 256, odd input widths) run the
    fp32 kernel; a checkpoint fp16 pieces are not safe for is refused (and that is asserted, not skipped over)."""
    torch = _torch()
    f = load_f1(name)
    m = build_model(f["cfg"], f["w"])
    if not _set_precision_or_refusal(m, precision):
        return
    st = {k: torch.from_numpy(v).cuda(0) for k, v in f["states"].items()}
    logits, probs, labels = m.forward(*to_dev(f["inputs"]), init_states=st, want_labels=True)
    torch.cuda.synchronize()
    dp = np.abs(probs.cpu().numpy() - f["probs"]).max()
    dl = np.abs(logits.cpu().numpy() - f["logits"]).max()
    print(name, precision, "HIP vs reference: max|dprob| %.2e max|dlogit| %.2e (n = %d, p1 in [%.4g, %.4g])" % (
        dp, dl, f["n"], f["probs"][:, -1].min(), f["probs"][:, -1].max()))
    assert dp <= TOL_PROB                   # the contract
    assert dp <= f1_tolerances(name)[1]     # regression guard (= the contract for the saturating-weight fixtures)
    # labels identical except where |p1 - 0.5| < 1e-4
    ref_lab = f["probs"].argmax(1)
    sure = np.abs(f["probs"][:, 1] - 0.5) >= 1e-4 if f["probs"].shape[1] == 2 else np.ones(len(ref_lab), bool)
    assert np.array_equal(labels.cpu().numpy()[sure], ref_lab[sure])
    # intermediates recorded from the reference through forward hooks (first 8 sites)
    if "lstm_comb" in f["inter"]:
        got = m.debug_activation(1, f["n"])[: f["inter"]["lstm_comb"].shape[0]]
        assert np.abs(got - f["inter"]["lstm_comb"]).max() <= 5e-5
    if f["cfg"].module == "both_bilstm" and "relu_seq" in f["inter"]:
        got = m.debug_activation(0, f["n"])[: f["inter"]["relu_seq"].shape[0]]
        ref = np.concatenate((f["inter"]["relu_seq"], f["inter"]["relu_signal"]), axis=2)
        assert np.abs(got - ref).max() <= 5e-5


@pytest.mark.parametrize("n", [1, 31, 32, 33, 64, 513, 2000])
def test_hip_matches_oracle_ragged_sizes_philox(n):
    """ragged batch sizes (tile tails), in-kernel Philox initial states vs the oracle's same generator"""
    torch = _torch()
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    cfg = onp.OracleConfig()
    w = onp.make_weights(cfg, 21, 2.0)
    ins = onp.make_inputs(cfg, n, 22)
    m = build_model(cfg, w, init_state="randn", seed=99)
    m.site_offset = 1000
    logits, probs = m(*to_dev(ins))
    torch.cuda.synchronize()
    lo, po = oc.forward(cfg, w, *ins, init_mode="philox", seed=99, site_offset=1000)
    assert np.abs(probs.cpu().numpy() - po).max() <= TOL_TIGHT


def test_compact_dtypes_and_zero_states_match_float_path():
    torch = _torch()
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    cfg = onp.OracleConfig()
    w = onp.make_weights(cfg, 31, 1.0)
    n = 300
    kmer, means, stds, lens, signals = onp.make_inputs(cfg, n, 32, wide_alphabet=True)
    m = build_model(cfg, w, init_state="zeros")
    a = m(*to_dev([kmer, means, stds, lens, signals]))[1]
    b = m(*to_dev([kmer.astype(np.uint8), means, stds, lens.astype(np.uint16), signals]))[1]
    c = m(*to_dev([kmer.astype(np.int32), means, stds, lens.astype(np.int32), signals]))[1]
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, c)
    lo, po = oc.forward(cfg, w, kmer, means, stds, lens, signals, init_mode="zeros")
    assert np.abs(a.cpu().numpy() - po).max() <= TOL_TIGHT


def test_determinism_and_batch_split_invariance():
    """same input twice -> bit-identical (catches LDS races); splitting a batch -> identical per-site
    results when the Philox site offset follows the split (what range-sharding relies on)"""
    torch = _torch()
    from oracle import forward_np as onp
    cfg = onp.OracleConfig()
    w = onp.make_weights(cfg, 41, 3.0)
    n = 1000
    ins = to_dev(onp.make_inputs(cfg, n, 42))
    m = build_model(cfg, w, init_state="randn", seed=5)
    p1 = m(*ins)[1].clone()
    p2 = m(*ins)[1].clone()
    assert torch.equal(p1, p2)
    cut = 389
    m.site_offset = 0
    pa = m(*[t[:cut] for t in ins])[1].clone()
    m.site_offset = cut
    pb = m(*[t[cut:] for t in ins])[1].clone()
    torch.cuda.synchronize()
    assert torch.equal(torch.cat((pa, pb)), p1)


def test_errors_are_loud():
    torch = _torch()
    from deepsignal_plant_amd.models import ModelBiLSTM
    from oracle import forward_np as onp
    with pytest.raises(ValueError):
        ModelBiLSTM(module="bogus")
    with pytest.raises(ValueError):
        ModelBiLSTM(hidden_size=2050).cuda(0)  # the one documented limit: hid_rnn <= 2048 (include/dsp_amd.h)
    cfg = onp.OracleConfig()
    m = build_model(cfg, onp.make_weights(cfg, 1))
    cpu_ins = [torch.from_numpy(a) for a in onp.make_inputs(cfg, 4, 2)]
    with pytest.raises(RuntimeError):
        m(*cpu_ins)
    sd = m.state_dict()
    sd["fc1.weight"] = sd["fc1.weight"][:, :10]
    with pytest.raises(RuntimeError):
        m.load_state_dict(sd)


def _rand_cfg(rng):
    from oracle import forward_np as onp
    module = ["both_bilstm", "seq_bilstm", "signal_bilstm"][int(rng.integers(0, 3))]
    hidden = int(rng.choice([20, 32, 50, 64, 96, 128, 160, 200, 256]))
    if module == "both_bilstm" and hidden % 2:
        hidden += 1
    return onp.OracleConfig(seq_len=int(rng.choice([5, 9, 13, 17, 21])), signal_len=int(rng.choice([8, 12, 16, 24])),
                            num_layers1=int(rng.integers(1, 4)), num_layers2=int(rng.integers(1, 3)),
                            num_classes=int(rng.choice([2, 2, 3, 5])), hidden_size=hidden,
                            vocab_size=int(rng.choice([5, 16])), embedding_size=int(rng.choice([2, 4, 6])),
                            is_base=bool(rng.integers(0, 2)), is_signallen=bool(rng.integers(0, 2)), module=module)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("case", range(14))
def test_random_model_shapes_match_oracle(case, precision):
    """the generic kernels (hidden-size padding, unit-tile pairing, k-group padding, feature maps) over random
    model shapes: every flag of the reference constructor is exercised, explicit N(0,1) initial states -- in every
    precision mode, same bound"""
    torch = _torch()
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    rng = np.random.default_rng(1000 + case)
    cfg = _rand_cfg(rng)
    n = int(rng.choice([1, 7, 33, 100, 257]))
    w = onp.make_weights(cfg, 2000 + case, float(rng.choice([1.0, 2.0, 3.0])))
    ins = onp.make_inputs(cfg, n, 3000 + case, wide_alphabet=cfg.vocab_size == 16)
    if cfg.vocab_size < 16:
        ins = (np.minimum(ins[0], cfg.vocab_size - 1),) + ins[1:]
    st = onp.make_init_states(cfg, n, 4000 + case)
    m = build_model(cfg, w)
    if not _set_precision_or_refusal(m, precision):
        return
    logits, probs = m.forward(*to_dev(ins), init_states={k: torch.from_numpy(v).cuda(0) for k, v in st.items()})
    torch.cuda.synchronize()
    lo, po = oc.forward(cfg, w, *ins, states=st, init_mode="explicit")
    d = np.abs(probs.cpu().numpy() - po).max()
    assert d <= TOL_TIGHT, (cfg.as_dict(), n, d, precision)
    assert np.abs(logits.cpu().numpy() - lo).max() <= 1e-4


@pytest.mark.parametrize("which", ["configs1_both_bilstm", "configs2_seq_only_hid256x2"])
def test_full_batch_properties(which):
    """BASELINE size (65,536 + a ragged tail), for the models of BASELINE.json configs[1] and configs[2]:
    size-independent properties -- determinism, permutation equivariance under pinned zero states, split invariance,
    finite normalised probabilities -- plus an oracle spot check on a strided sample of the same batch"""
    torch = _torch()
    from deepsignal_plant_amd import synth
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    cfg = onp.OracleConfig() if which.startswith("configs1") else onp.OracleConfig(module="seq_bilstm", num_layers1=2)
    w = onp.make_weights(cfg, 51, 2.0)
    n = 65536 + 1234
    ins = synth.feature_batch(n, device="cuda:0", seed=52)
    m = build_model(cfg, w, init_state="zeros")
    p = m(*ins)[1].clone()
    assert torch.isfinite(p).all() and float((p.sum(1) - 1).abs().max()) <= 1e-6
    assert torch.equal(p, m(*ins)[1])
    perm = torch.randperm(n, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(1))
    pp = m(*[t[perm] for t in ins])[1]
    assert torch.equal(pp, p[perm])
    cut = 40000
    assert torch.equal(torch.cat((m(*[t[:cut] for t in ins])[1], m(*[t[cut:] for t in ins])[1])), p)
    idx = torch.arange(0, n, 97, device="cuda:0")
    sample = [t[idx].cpu().numpy() for t in ins]
    lo, po = oc.forward(cfg, w, *sample, init_mode="zeros")
    assert np.abs(p[idx].cpu().numpy() - po).max() <= TOL_TIGHT


def test_empty_batch():
    torch = _torch()
    from oracle import forward_np as onp
    cfg = onp.OracleConfig()
    m = build_model(cfg, onp.make_weights(cfg, 1))
    z = [torch.zeros((0, 13), device="cuda:0")] * 4 + [torch.zeros((0, 13, 16), device="cuda:0")]
    logits, probs = m(*z)
    assert logits.shape == (0, 2) and probs.shape == (0, 2)


@pytest.mark.parametrize("precision", ["bf16x9", "bf16x6", "fp16x3"])
def test_split_bf16_product_emulation_stays_within_the_fp32_path(precision):
    """opt-in split-bf16 evaluation of the combined stack's products (include/dsp_amd.h, DSP_PREC_*): against the
    fixtures captured from the reference it has to meet the same bound as the fp32 path, and it must agree with
    the fp32 path itself far below the 1e-4 contract -- also with sharp (x3) weights and Philox states at full
    batch"""
    torch = _torch()
    from deepsignal_plant_amd.models import ModelBiLSTM
    for name in ("both_default", "both_sharp", "seq_cfg3"):
        f = load_f1(name)
        model = build_model(f["cfg"], f["w"])
        st = {k: torch.from_numpy(v).cuda(0) for k, v in f["states"].items()}
        ins = to_dev(f["inputs"])
        model.set_precision("fp32")  # whatever DSP_PRECISION says
        _, p32 = model.forward(*ins, init_states=st)
        model.set_precision(precision)
        _, ps = model.forward(*ins, init_states=st)
        err_ref = float(np.abs(ps.cpu().numpy() - f["probs"]).max())
        err_32 = float((ps - p32).abs().max())
        print(name, precision, "vs reference %.2e, vs fp32 path %.2e" % (err_ref, err_32))
        assert err_ref <= TOL_TIGHT and err_32 <= 5e-6, (name, precision, err_ref, err_32)
        model.set_precision("fp32")
        _, again = model.forward(*ins, init_states=st)
        assert torch.equal(again, p32)
    # full batch, in-kernel Philox states
    from deepsignal_plant_amd import synth
    model = ModelBiLSTM(init_state="randn", seed=3)
    model.load_state_dict(synth.random_state_dict(model, seed=5, scale=2.0))
    model.cuda(0)
    ins = synth.feature_batch(65536 + 77, device="cuda:0", seed=9)
    model.set_precision("fp32")
    _, p32 = model.forward(*ins)
    model.set_precision(precision)
    _, ps = model.forward(*ins)
    assert float((ps - p32).abs().max()) <= 5e-6
    with pytest.raises(ValueError):
        model.set_precision("fp8")
    # a checkpoint whose fc outputs could leave the fp16 range is refused fp16 pieces (bf16 pieces have fp32's range)
    sd = synth.random_state_dict(model, seed=5)
    sd["fc_seq.weight"] = sd["fc_seq.weight"] * 1e4
    big = ModelBiLSTM(init_state="zeros")
    big.load_state_dict(sd)
    big.cuda(0)
    with pytest.raises(ValueError, match="fp16 range"):
        big.set_precision("fp16x3")
    big.set_precision("bf16x6")
    _, pb = big.forward(*synth.feature_batch(64, device="cuda:0", seed=2))
    big.set_precision("fp32")
    _, pf = big.forward(*synth.feature_batch(64, device="cuda:0", seed=2))
    assert bool(torch.isfinite(pb).all()) and float((pb - pf).abs().max()) <= 1e-4


@pytest.mark.parametrize("hidden", [160, 192, 80])
def test_hidden_sizes_whose_padding_adds_row_tiles(hidden):
    """hid_rnn 160 / 192 / 80: the per-branch width (80 / 96 / 40) is padded to more 32-row tiles than it fills; the
    padding rows of fc_seq / fc_signal / fc1 must be packed as zeros (they once were not packed at all and the
    kernels read past the weight buffers) -- all precision modes against the oracle"""
    torch = _torch()
    from deepsignal_plant_amd.models import ModelBiLSTM
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    cfg = onp.OracleConfig(hidden_size=hidden, num_layers1=2)
    w = onp.make_weights(cfg, 31, 2.0)
    n = 300
    ins = onp.make_inputs(cfg, n, 32)
    m = build_model(cfg, w)
    m.init_state, m.seed = "randn", 17
    _, po = oc.forward(cfg, w, *ins, init_mode="philox", seed=17)
    for precision in ("fp32", "bf16x6", "fp16x3"):
        m.set_precision(precision)
        _, probs = m.forward(*to_dev(ins))
        assert np.abs(probs.cpu().numpy() - po).max() <= TOL_TIGHT, precision


@pytest.mark.parametrize("sg", ["1", "2"])
def test_site_groups_per_workgroup_do_not_change_a_bit(sg, monkeypatch):
    """DSP_LSTM_SG (site groups per LSTM workgroup: 4-wave or 8-wave workgroups on the front ends) only changes which
    workgroup owns which sites, never the order of a sum: bit-identical probabilities, and they match the fixture"""
    torch = _torch()
    fx = load_f1("both_default")
    st = {k: torch.from_numpy(v).cuda(0) for k, v in fx["states"].items()}
    monkeypatch.delenv("DSP_LSTM_SG", raising=False)
    _, p0 = build_model(fx["cfg"], fx["w"]).forward(*to_dev(fx["inputs"]), init_states=st)
    monkeypatch.setenv("DSP_LSTM_SG", sg)
    _, p1 = build_model(fx["cfg"], fx["w"]).forward(*to_dev(fx["inputs"]), init_states=st)
    assert torch.equal(p0, p1)
    assert np.abs(p1.cpu().numpy() - fx["probs"]).max() <= TOL_TIGHT


def test_philox_states_at_several_batches_are_deterministic_and_match_the_oracle():
    """The h exchange of the LSTM kernels goes through global memory behind a workgroup barrier.  A tiling tried in round
    2 read the initial states back wrong for a few lanes, non-deterministically and only with non-zero initial states at
    several workgroups per CU -- exactly what the zero-state property test cannot see.  So: in-kernel N(0,1) states,
    300k sites (4.6 rounds of workgroups), five repetitions bit-identical, both directions against the oracle on a
    strided sample, and the fp32 path against the split-precision kernels (different code, same states)."""
    torch = _torch()
    from deepsignal_plant_amd import synth
    from deepsignal_plant_amd.models import ModelBiLSTM
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    model = ModelBiLSTM(init_state="randn", seed=3)
    sd = synth.random_state_dict(model, seed=5, scale=2.0)
    model.load_state_dict(sd)
    model.cuda(0)
    n = 300000
    ins = synth.feature_batch(n, device="cuda:0", seed=9)
    p = model.forward(*ins)[1].clone()
    for _ in range(4):
        assert torch.equal(model.forward(*ins)[1], p)
    model.set_precision("bf16x9")
    assert float((model.forward(*ins)[1] - p).abs().max()) <= 5e-6
    model.set_precision("fp32")
    idx = torch.arange(5, n, 1511, device="cuda:0")
    cfg = onp.OracleConfig()
    w = {k: v.numpy() for k, v in sd.items()}
    worst = 0.0
    for i in idx.tolist()[::8]:   # the oracle takes one site offset per call
        one = [t[i:i + 1].cpu().numpy() for t in ins]
        _, po = oc.forward(cfg, w, *one, init_mode="philox", seed=3, site_offset=i)
        worst = max(worst, float(np.abs(p[i].cpu().numpy() - po[0]).max()))
    print("philox full-batch: max|dprob| vs oracle on %d strided sites = %.3e" % (len(idx.tolist()[::8]), worst))
    assert worst <= TOL_TIGHT


@pytest.mark.parametrize("hidden,module,n", [(320, "both_bilstm", 70), (384, "seq_bilstm", 130), (512, "both_bilstm", 70),
                                             (512, "signal_bilstm", 33), (258, "both_bilstm", 200),
                                             (640, "both_bilstm", 70), (768, "seq_bilstm", 40), (1024, "both_bilstm", 40),
                                             (1100, "signal_bilstm", 70), (2048, "seq_bilstm", 33)])
def test_hidden_sizes_above_256(hidden, module, n):
    """hid_rnn > 256 (models.py:103-128 accepts any hidden_size): the hidden state is padded to a multiple of 8 unit
    tiles and a step runs hidden / 256 passes, every wave computing one unit tile per pass, with the cell state in a global
    scratch (dsp_lstm_kernel<.., 0>); the front ends of a both_bilstm model take whichever kernel their half of the hidden
    size selects; against the oracle
    with explicit N(0,1) states, and with in-kernel Philox states (non-zero initial states through the h0 scratch)"""
    torch = _torch()
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    cfg = onp.OracleConfig(hidden_size=hidden, num_layers1=2, module=module)
    w = onp.make_weights(cfg, 77, 2.0)
    ins = onp.make_inputs(cfg, n, 78)
    st = onp.make_init_states(cfg, n, 79)
    m = build_model(cfg, w)
    logits, probs = m.forward(*to_dev(ins), init_states={k: torch.from_numpy(v).cuda(0) for k, v in st.items()})
    _, po = oc.forward(cfg, w, *ins, states=st, init_mode="explicit")
    d = np.abs(probs.cpu().numpy() - po).max()
    m.init_state, m.seed = "randn", 21
    _, pp = m.forward(*to_dev(ins))
    _, pq = oc.forward(cfg, w, *ins, init_mode="philox", seed=21)
    d2 = np.abs(pp.cpu().numpy() - pq).max()
    print("hidden %d %s: max|dprob| = %.3e (explicit states), %.3e (Philox states)" % (hidden, module, d, d2))
    assert d <= TOL_TIGHT and d2 <= TOL_TIGHT
    assert torch.equal(m.forward(*to_dev(ins))[1], pp)


@pytest.mark.parametrize("signal_len,is_base,is_signallen,hidden,module,what", [
    (8, True, True, 128, "both_bilstm", "one live x-part k-group in both front ends (dsp_lstm_kernel<2, 1, 1>)"),
    (16, True, True, 128, "both_bilstm", "seq one, signal two live k-groups (<2, 1, 1> and <2, 1, 2>)"),
    (24, True, True, 64, "both_bilstm", "signal three live k-groups (<2, 1, 3>)"),
    (32, True, True, 64, "both_bilstm", "signal block full: the dense kernel (<0, 1>)"),
    (40, True, True, 64, "both_bilstm", "signal block of 40: padding in the tail k-groups, the tested kernel (<1, 1>)"),
    (16, False, False, 128, "both_bilstm", "seq branch of 2 features (mean, std)"),
    (20, True, True, 320, "signal_bilstm", "hidden 320: two passes, features at the front (<1, 0>)"),
])
def test_front_end_shapes(signal_len, is_base, is_signallen, hidden, module, what):
    """Every instantiation of the LSTM kernel that a front end can select: the features sit at the end of the 32-wide
    padded block when hidden <= 256 (refill-only stages first, known at compile time), at the front otherwise; against
    the oracle with explicit N(0,1) initial states and with Philox states (models.py:178-214 for any signal_len /
    is_base / is_signallen)"""
    torch = _torch()
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    cfg = onp.OracleConfig(signal_len=signal_len, hidden_size=hidden, num_layers1=1, is_base=is_base,
                           is_signallen=is_signallen, module=module)
    n = 150
    w = onp.make_weights(cfg, 91, 2.0)
    ins = onp.make_inputs(cfg, n, 92)
    st = onp.make_init_states(cfg, n, 93)
    m = build_model(cfg, w)
    _, probs = m.forward(*to_dev(ins), init_states={k: torch.from_numpy(v).cuda(0) for k, v in st.items()})
    _, po = oc.forward(cfg, w, *ins, states=st, init_mode="explicit")
    d = np.abs(probs.cpu().numpy() - po).max()
    m.init_state, m.seed = "randn", 5
    _, pp = m.forward(*to_dev(ins))
    _, pq = oc.forward(cfg, w, *ins, init_mode="philox", seed=5)
    d2 = np.abs(pp.cpu().numpy() - pq).max()
    print("%s: max|dprob| = %.3e (explicit states), %.3e (Philox states)" % (what, d, d2))
    assert d <= TOL_TIGHT and d2 <= TOL_TIGHT


@pytest.mark.parametrize("hid", [128, 256])
@pytest.mark.parametrize("precision", ["fp32", "bf16x9", "fp16x3"])
def test_hip_matches_the_reference_on_the_checkpoint_it_trained(precision, hid, monkeypatch):
    """F8 (tests/golden/make_golden_trained.py): weights out of the reference's own optimiser, saved by its own
    torch.save, rows parsed by this build's parser; pinned N(0,1) states and zero states against the reference model's
    outputs.  The one fixture whose weight statistics are a trained model's (tests/test_trained_fixture.py)."""
    torch = _torch()
    from tests.helpers import have_f8, load_f8
    if not have_f8(hid):   # hid_rnn 256 = the reference's default architecture: 18.8 MB, committed since round 4 (tests/helpers.py)
        pytest.skip("no hid_rnn %d checkpoint in tests/golden/" % hid)
    f = load_f8(hid)
    m = build_model(f["cfg"], f["w"])
    m.set_precision(precision)
    ins = to_dev(f["inputs"])
    st = {k: torch.from_numpy(v).cuda(0) for k, v in f["states"].items()}
    logits, probs, labels = m.forward(*ins, init_states=st, want_labels=True)
    m0 = build_model(f["cfg"], f["w"], init_state="zeros")
    m0.set_precision(precision)
    logits0, probs0 = m0.forward(*ins)
    torch.cuda.synchronize()
    dp = np.abs(probs.cpu().numpy() - f["probs"]).max()
    dp0 = np.abs(probs0.cpu().numpy() - f["probs0"]).max()
    print("F8 trained checkpoint hid_rnn %d, %s: HIP vs reference max|dprob| %.2e (pinned N(0,1) states) %.2e (zero states); "
          "reference fp32 vs float64 %.2e; p1 in [%.3g, %.6f]" % (hid, precision, dp, dp0, f["noise"], f["probs"][:, 1].min(),
                                                                   f["probs"][:, 1].max()))
    assert max(dp, dp0) <= TOL_PROB
    assert max(dp, dp0) <= max(TOL_TIGHT, 4.0 * f["noise"])
    sure = np.abs(f["probs"][:, 1] - 0.5) >= 1e-4
    assert np.array_equal(labels.cpu().numpy()[sure], f["probs"].argmax(1)[sure])
    called = (probs0.cpu().numpy()[:, 1] > 0.5).astype(int)
    assert float((called == f["labels"]).mean()) == pytest.approx(float(f["raw"]["accuracy"]), abs=0.006)
    if precision == "fp32":   # 400 sites run the small-batch tiling: the full-batch kernel must give the same bytes
        monkeypatch.setenv("DSP_LSTM_TILING", "0")
        _, probs64 = build_model(f["cfg"], f["w"]).forward(*ins, init_states=st)
        assert torch.equal(probs64, probs)


@pytest.mark.parametrize("kw,label", [
    (dict(), "configs1"),
    (dict(module="seq_bilstm", num_layers1=2), "configs2_seq_only"),
    (dict(hidden_size=128), "hidden128_UT4"),
    (dict(hidden_size=64, num_layers1=2), "hidden64_UT2"),
    (dict(hidden_size=160), "hidden160_UT6"),
    (dict(hidden_size=32, num_layers1=1), "hidden32_UT1_not_eligible"),
])
def test_small_batch_tiling_does_not_change_a_bit(kw, label, monkeypatch):
    """Batches whose 32-site tiles x 2 directions fit the CUs at once (<= 4,096 sites on 256 CUs) run the combined stack
    with <2 unit tiles, 1 site tile> per wave (dsp_lstm21_kernel: twice the workgroups, half the latency of a launch);
    larger ones with the shipped <1, 2> tiling.  Same fragments, same order of every sum: DSP_LSTM_TILING=0 (never) and
    =21 (always) give the bytes of the automatic choice at every size around the switch, with N(0,1) Philox states, and
    the oracle agrees.  (Round 2's build of this tiling returned wrong h0 for some lanes: the store-data hazard,
    profiles/LAB_NOTEBOOK_r1_r3.md section 3a; the guarded stores are what this test now also watches.)"""
    torch = _torch()
    from deepsignal_plant_amd import synth
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    cfg = onp.OracleConfig(**kw)
    w = onp.make_weights(cfg, 91, 2.0)
    sizes = (1, 33, 513, 2000, 4096, 4097, 8200)
    ins = {n: synth.feature_batch(n, device="cuda:0", seed=400 + n) for n in sizes}
    res = {}
    for mode in (None, "0", "21"):
        if mode is None:
            monkeypatch.delenv("DSP_LSTM_TILING", raising=False)
        else:
            monkeypatch.setenv("DSP_LSTM_TILING", mode)
        m = build_model(cfg, w, init_state="randn", seed=17)
        for n in sizes:
            m.site_offset = 10 * n
            res[mode, n] = m.forward(*ins[n])[1].clone()
    torch.cuda.synchronize()
    for n in sizes:
        assert torch.equal(res[None, n], res["0", n]) and torch.equal(res[None, n], res["21", n]), (label, n)
    n = 2000
    sample = [t.cpu().numpy() for t in ins[n]]
    _, po = oc.forward(cfg, w, *sample, init_mode="philox", seed=17, site_offset=10 * n)
    assert np.abs(res[None, n].cpu().numpy() - po).max() <= TOL_TIGHT


@pytest.mark.parametrize("kw,label", [
    (dict(), "configs1_hidden256"),
    (dict(module="seq_bilstm", num_layers1=2), "configs2_seq_only"),
    (dict(hidden_size=200, num_layers1=2), "hidden200_padded_to_UT8"),
    (dict(hidden_size=128, num_layers1=2, num_layers2=2), "hidden128_UT4_two_front_layers"),
    (dict(module="signal_bilstm", num_layers1=1), "signal_only"),
])
def test_small_batch_kernels_do_not_change_a_bit(kw, label, monkeypatch):
    """Round 4: what a batch of <= 4,096 sites runs on.  dsp_lstmc_kernel spreads a (site tile, direction)'s 8 unit tiles x 4
    gates over a cluster of 2 / 4 / 8 workgroups on as many CUs (<= 2,048 / 1,024 / 512 sites; h crosses CUs through memory:
    write-through stores, an arrival counter, sc1 loads), runs 4-unit-tile layers (the front ends) and -- from 2,049 sites --
    8-unit-tile layers as one workgroup of 4 / 8 waves, the two front-end branches go down two streams, the head kernel
    takes one site tile per workgroup, fc_seq and fc_signal share a launch.  Same MFMAs in the same order on the same values: every switch combination gives
    the bytes of the round-3 path (everything off), at every batch size around the switches, for Philox, explicit N(0,1)
    and zero states -- and the oracle agrees."""
    torch = _torch()
    from deepsignal_plant_amd import synth
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    cfg = onp.OracleConfig(**kw)
    w = onp.make_weights(cfg, 92, 2.0)
    sizes = (1, 31, 512, 513, 1024, 1025, 1400, 2048, 2049, 3000, 4096, 4097, 5700, 9001, 14000)   # (pieces: 1,025 = 1,024 + 1; 3,000 = 2,048 + 952; 5,700 = 4,096 + 1,024 + 580; 14,000 = 8,192 + 4,096 + 1,712)
    ins = {n: synth.feature_batch(n, device="cuda:0", seed=500 + n) for n in sizes}
    n_x = 700
    states = {k: torch.from_numpy(v).cuda(0) for k, v in onp.make_init_states(cfg, n_x, 9).items()}
    ins_x = synth.feature_batch(n_x, device="cuda:0", seed=77)
    switches = ("DSP_LSTM_CLUSTER", "DSP_LSTM_LOCAL8", "DSP_TWO_STREAMS", "DSP_HEAD_ST4", "DSP_LSTM_TILING", "DSP_CLUSTER_TIMEOUT",
                "DSP_FC_FUSED", "DSP_LSTM_FRONT_CLUSTER", "DSP_FC_SMALL", "DSP_LSTM_HANDOFF", "DSP_FORWARD_SPLIT", "DSP_RSRC_EXTENTS")
    modes = {"round3": {"DSP_LSTM_CLUSTER": "0", "DSP_LSTM_LOCAL8": "0", "DSP_TWO_STREAMS": "0", "DSP_HEAD_ST4": "1",
                        "DSP_FC_FUSED": "0", "DSP_FC_SMALL": "0"},
             # round 5: arrivals counted per wave and deferred into the next step's x part, the counter requested a block early
             # (auto) -- against round 4's drain + workgroup barrier + one arrival per workgroup, at every cluster size
             "round4_handoff": {"DSP_LSTM_HANDOFF": "0"},
             "round4_handoff_G2_front_G2": {"DSP_LSTM_HANDOFF": "0", "DSP_LSTM_CLUSTER": "2", "DSP_LSTM_FRONT_CLUSTER": "2"},
             # round 5: a call runs its whole 8,192-site rounds, then its remainder as the cheapest sequence of small-batch pieces
             # (4,097 = 4,096 + 1; 9,001 = 8,192 + 809; 3,000 = 2,048 + 952) -- off
             "one_piece": {"DSP_FORWARD_SPLIT": "0"},
             "round4_fc_kernel": {"DSP_FC_SMALL": "0"},   # (auto, round 5: one accumulator tile per wave in the fc projections)
             "fc_small_one_stream": {"DSP_TWO_STREAMS": "0"},   # ... and in the shared fc_seq+fc_signal launch
             # round 5: the front ends (4 unit tiles) clustered too -- off, 1 gate per wave (P = 4), 2 (P = 2), on one stream
             # (a branch may then take all the CUs), with every cluster abandoned to the clean-up launch
             "round4_front_ends_local": {"DSP_LSTM_FRONT_CLUSTER": "0"},
             "front_G1": {"DSP_LSTM_FRONT_CLUSTER": "1"}, "front_G2": {"DSP_LSTM_FRONT_CLUSTER": "2"},
             "front_G1_one_stream": {"DSP_LSTM_FRONT_CLUSTER": "1", "DSP_TWO_STREAMS": "0"},
             "front_G2_one_stream_abandoned": {"DSP_LSTM_FRONT_CLUSTER": "2", "DSP_TWO_STREAMS": "0", "DSP_CLUSTER_TIMEOUT": "0"},
             "fc_launches_apart": {"DSP_FC_FUSED": "0"},   # (auto: fc_seq and fc_signal share a launch when the branches share a stream)
             "auto": {},
             # round 6: the descriptors' real extents (auto) against the 2 GiB windows of rounds 1-5 (range check off): a legitimate
             # access past the end of its allocation would read zeros / be dropped and change the bytes.  (The tight extents and the
             # bounds-recording build: tests/test_gpu_zz_extents.py, the suite's last module.)
             "descriptors_2GiB_windows": {"DSP_RSRC_EXTENTS": "wide"},
             "one_stream": {"DSP_TWO_STREAMS": "0"},
             "two_streams_always": {"DSP_TWO_STREAMS": "1"},
             "G4": {"DSP_LSTM_CLUSTER": "4"}, "G2": {"DSP_LSTM_CLUSTER": "2"}, "G1": {"DSP_LSTM_CLUSTER": "1"},
             "every_cluster_abandoned": {"DSP_CLUSTER_TIMEOUT": "0"},   # the clean-up launch computes all of them
             "no_local8": {"DSP_LSTM_LOCAL8": "0"},
             "no_lstm21": {"DSP_LSTM_TILING": "0"}}
    res = {}
    for name, env in modes.items():
        for k in switches:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = build_model(cfg, w, init_state="randn", seed=18)
        # the clustered launches must really be ON where this test says it compares them (a handle whose XCC probe fails
        # silently falls back to the workgroup-local forms: every comparison would pass without testing anything)
        assert m.query("clustering") == (0 if env.get("DSP_LSTM_CLUSTER") == "0" else 1) and m.query("xcc_probe_failed") == 0, name
        for n in sizes:
            m.site_offset = 11 * n
            res[name, n] = m.forward(*ins[n])[1].clone()
        res[name, "explicit"] = m.forward(*ins_x, init_states=states)[1].clone()
        mz = build_model(cfg, w, init_state="zeros")
        res[name, "zeros"] = mz.forward(*ins[513])[1].clone()
        # twice in a row on one handle: the arrival counters start from zero every time
        res[name, "again"] = m.forward(*ins_x, init_states=states)[1].clone()
    torch.cuda.synchronize()
    for name in modes:
        for key in list(sizes) + ["explicit", "zeros"]:
            assert torch.equal(res[name, key], res["round3", key]), (label, name, key)
        assert torch.equal(res[name, "again"], res["round3", "explicit"]), (label, name)
    for n in (513, 3000):
        sample = [t.cpu().numpy() for t in ins[n]]
        _, po = oc.forward(cfg, w, *sample, init_mode="philox", seed=18, site_offset=11 * n)
        assert np.abs(res["auto", n].cpu().numpy() - po).max() <= TOL_TIGHT, (label, n)


def test_clustered_forwards_under_uneven_load_stay_bit_identical():
    """The members of a cluster wait for each other (arrival counters polled across compute units): the hand-off must not
    depend on the members running in step.  Small clustered forwards are issued while ANOTHER handle keeps the chip busy
    with full batches on its own stream -- cluster members then start late, apart from each other, and share CUs with
    foreign workgroups -- 300 forwards of 512 / 1,024 / 2,048 sites, every output compared bit for bit with the
    unclustered path's."""
    torch = _torch()
    import os
    from deepsignal_plant_amd import synth
    from oracle import forward_np as onp
    cfg = onp.OracleConfig()
    w = onp.make_weights(cfg, 93, 2.0)
    sizes = (512, 1024, 2048)
    ins = {n: synth.feature_batch(n, device="cuda:0", seed=600 + n) for n in sizes}
    old = {k: os.environ.get(k) for k in ("DSP_LSTM_CLUSTER", "DSP_LSTM_LOCAL8", "DSP_TWO_STREAMS", "DSP_HEAD_ST4")}
    try:
        os.environ.update({"DSP_LSTM_CLUSTER": "0", "DSP_LSTM_LOCAL8": "0", "DSP_TWO_STREAMS": "0", "DSP_HEAD_ST4": "1"})
        ref_model = build_model(cfg, w, init_state="randn", seed=19)
        ref = {}
        for n in sizes:
            ref_model.site_offset = 7 * n
            ref[n] = ref_model.forward(*ins[n])[1].clone()
        for k in old:
            os.environ.pop(k, None)
        small = build_model(cfg, w, init_state="randn", seed=19)
        big = build_model(cfg, w, init_state="randn", seed=3)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    big_in = synth.feature_batch(65536, device="cuda:0", seed=1)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    bad = 0
    outs = []
    for it in range(100):
        if it % 10 == 0:
            with torch.cuda.stream(side):     # ~53 ms of full-chip work under the next forwards
                big.forward(*big_in)
        for n in sizes:
            small.site_offset = 7 * n
            outs.append((n, small.forward(*ins[n])[1]))
        if len(outs) >= 60:
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(o, ref[n]) else 1 for n, o in outs)
            outs = []
    torch.cuda.synchronize()
    bad += sum(0 if torch.equal(o, ref[n]) else 1 for n, o in outs)
    assert bad == 0, "%d of 300 clustered forwards differ from the unclustered path under load" % bad


def test_a_workspace_that_cannot_be_had_is_refused_and_leaves_the_handle_usable():
    """dsp_model_reserve beyond the GPU's memory: DSP_ENOMEM with the size in the message -- and the next forward on the
    same handle runs and gives the same bytes as before (the failed hipMalloc must not linger in the runtime's
    last-error slot, where the next launch's check would find it)"""
    torch = _torch()
    from oracle import forward_np as onp
    cfg = onp.OracleConfig()
    m = build_model(cfg, onp.make_weights(cfg, 5, 1.0), init_state="zeros")
    ins = to_dev(onp.make_inputs(cfg, 700, 6))
    ref = m.forward(*ins)[1].clone()
    total = torch.cuda.mem_get_info()[1]
    too_many = int(total / 60000)          # ~73 kB of workspace per site: more sites than fit by a fifth
    with pytest.raises(RuntimeError, match="workspace hipMalloc"):
        m.reserve(too_many)
    assert torch.equal(m.forward(*ins)[1], ref)


@pytest.mark.parametrize("kw,label", [(dict(hidden_size=160, num_layers1=2), "hidden160_no_clustered_forms"),
                                      (dict(module="seq_bilstm", num_layers1=2), "configs2_seq_only"),
                                      (dict(hidden_size=128), "hidden128")])
def test_a_call_cut_into_rounds_and_pieces_gives_the_bits_of_one_piece(kw, label, monkeypatch):
    """dsp_forward runs a call as its whole 8,192-site rounds + its remainder as small-batch pieces (round 5).  For a model
    whose combined stack has no clustered forms (hidden 160: five unit tiles) the remainder is one piece or a round; either
    way the bits are those of the call run in one piece -- Philox states keyed by the global site index, per-site keys too."""
    torch = _torch()
    from deepsignal_plant_amd import synth
    from oracle import forward_np as onp
    cfg = onp.OracleConfig(**kw)
    w = onp.make_weights(cfg, 95, 2.0)
    sizes = (1100, 3000, 5000, 7000, 9001, 12300, 17000)
    ins = {n: synth.feature_batch(n, device="cuda:0", seed=700 + n) for n in sizes}
    keys = torch.arange(3_000_000_000, 3_000_000_000 + 17000, device="cuda:0")
    res = {}
    for split in ("0", "1"):
        monkeypatch.setenv("DSP_FORWARD_SPLIT", split)
        m = build_model(cfg, w, init_state="randn", seed=19)
        for n in sizes:
            m.site_offset = 13 * n
            res[split, n] = m.forward(*ins[n])[1].clone()
        res[split, "keys"] = m.forward(*ins[17000], site_keys=keys)[1].clone()
    torch.cuda.synchronize()
    for n in list(sizes) + ["keys"]:
        assert torch.equal(res["0", n], res["1", n]), (label, n)
