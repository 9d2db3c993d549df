"""GPU parity deep inside large batches, with N(0,1) initial states (VERDICT r2 "weak" 1).

The LSTM kernels exchange h through global memory behind a workgroup barrier (dsp_kernels.hip, "h exchange").  Every
earlier oracle comparison with NON-ZERO initial states either stopped at a few workgroups (fixtures: <= 300 sites) or
looked at a strided sample (25 sites of 300 k).  Here: CONTIGUOUS windows of 2,304 sites -- every lane, wave, SIMD slot
and site group of 36 consecutive workgroups per direction -- at five depths of the batch (the first round of
workgroups, the round whose workgroups share CUs with the first one's on the front-end launches, a middle round, the
last full round, the ragged tail), against the C oracle, for the models of BASELINE.json configs[1] and configs[2], in
the fp32 path and the product-exact split modes; Philox states (oracle: same generator, contiguous site_offset) at
300 k sites and EXPLICIT states (reference layout, drawn on the device) at 65,536 + tail.  Plus the per-site key array
of dsp_init_state (site_keys) and a bit-exact permutation property at full batch under non-zero states.
Reference: models.py:169-176 (init_hidden), :196-228 (the three nn.LSTM calls)."""
import numpy as np
import pytest

from tests.test_gpu_parity import TOL_TIGHT, _torch, build_model

pytestmark = pytest.mark.gpu

WIN = 2304          # 36 site groups of 64 sites
ROUND = 8192        # sites of one round of combined-stack workgroups on 256 CUs (one 8-wave workgroup per CU, 2 directions)


def _cfg(which):
    from oracle import forward_np as onp
    return onp.OracleConfig() if which.startswith("configs1") else onp.OracleConfig(module="seq_bilstm", num_layers1=2)


def _windows(n):
    last_full = (n // ROUND - 1) * ROUND
    mid = (n // ROUND // 2) * ROUND
    starts = [0, ROUND, mid + 4000 + 13, last_full + 1500 + 7, n - WIN + 200]  # the last one runs into the ragged tail
    return [(a, min(n, a + WIN)) for a in starts]


CONFIGS = ["configs1_both_bilstm", "configs2_seq_only_hid256x2"]


@pytest.mark.parametrize("which", CONFIGS)
def test_contiguous_windows_of_a_300k_batch_with_philox_states(which):
    torch = _torch()
    from deepsignal_plant_amd import synth
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    cfg = _cfg(which)
    w = onp.make_weights(cfg, 61, 2.0)
    n = 300000
    ins = synth.feature_batch(n, device="cuda:0", seed=62)
    m = build_model(cfg, w, init_state="randn", seed=11)
    m.site_offset = 123456789012  # a global index beyond 32 bits: both counter words of the generator matter
    res = {}
    for precision in ("fp32", "bf16x9", "fp16x3"):
        m.set_precision(precision)
        res[precision] = m.forward(*ins)[1].clone()
    torch.cuda.synchronize()
    worst = dict.fromkeys(res, 0.0)
    for a, b in _windows(n):
        sample = [t[a:b].cpu().numpy() for t in ins]
        _, po = oc.forward(cfg, w, *sample, init_mode="philox", seed=11, site_offset=m.site_offset + a)
        for precision, p in res.items():
            d = float(np.abs(p[a:b].cpu().numpy() - po).max())
            worst[precision] = max(worst[precision], d)
            assert d <= TOL_TIGHT, (which, precision, a, b, d)
    print(which, "300k sites, Philox states, 5 contiguous windows of %d sites vs the oracle: max|dprob| %s" % (
        WIN, ", ".join("%s %.2e" % kv for kv in worst.items())))


@pytest.mark.parametrize("which", CONFIGS)
def test_contiguous_windows_of_a_full_batch_with_explicit_normal_states(which):
    torch = _torch()
    from deepsignal_plant_amd import synth
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    cfg = _cfg(which)
    w = onp.make_weights(cfg, 71, 2.0)
    n = 65536 + 1234
    ins = synth.feature_batch(n, device="cuda:0", seed=72)
    g = torch.Generator(device="cuda:0").manual_seed(73)
    st = {}
    for k, (layers, hid) in (("seq", (cfg.num_layers2, cfg.nhid_seq)), ("sig", (cfg.num_layers2, cfg.nhid_signal)),
                             ("comb", (cfg.num_layers1, cfg.hidden_size))):
        if hid:  # the reference layout: (2 * layers, n, H), models.py:169-176
            st["h_" + k] = torch.randn((2 * layers, n, hid), device="cuda:0", generator=g)
            st["c_" + k] = torch.randn((2 * layers, n, hid), device="cuda:0", generator=g)
    m = build_model(cfg, w)
    res = {}
    for precision in ("fp32", "bf16x9", "fp16x3"):
        m.set_precision(precision)
        res[precision] = m.forward(*ins, init_states=st)[1].clone()
    torch.cuda.synchronize()
    worst = dict.fromkeys(res, 0.0)
    for a, b in _windows(n):
        sample = [t[a:b].cpu().numpy() for t in ins]
        states = {k: v[:, a:b].contiguous().cpu().numpy() for k, v in st.items()}
        _, po = oc.forward(cfg, w, *sample, states=states, init_mode="explicit")
        for precision, p in res.items():
            d = float(np.abs(p[a:b].cpu().numpy() - po).max())
            worst[precision] = max(worst[precision], d)
            assert d <= TOL_TIGHT, (which, precision, a, b, d)
    print(which, "65,536 + 1,234 sites, explicit N(0,1) states, 5 contiguous windows vs the oracle: max|dprob| %s" % (
        ", ".join("%s %.2e" % kv for kv in worst.items())))


def test_site_keys_name_the_sites_for_the_initial_state_generator():
    """dsp_init_state.site_keys: the Philox counter of a site is its key -- equal to the oracle given the same keys,
    equal to site_offset + index when the keys say so, and (at full batch, N(0,1) states) a permutation of the rows with
    their keys permutes the results bit for bit: no lane, wave or workgroup slot computes a site differently."""
    torch = _torch()
    from deepsignal_plant_amd import synth
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    cfg = onp.OracleConfig()
    w = onp.make_weights(cfg, 81, 2.0)
    m = build_model(cfg, w, init_state="randn", seed=21)
    n = 3000
    ins = synth.feature_batch(n, device="cuda:0", seed=82)
    rng = np.random.default_rng(83)
    keys = rng.integers(0, 1 << 63, n, dtype=np.int64) * 2 + rng.integers(0, 2, n)  # all 64 bits in use (wraps)
    kd = torch.from_numpy(keys).cuda(0)
    p = m.forward(*ins, site_keys=kd)[1]
    _, po = oc.forward(cfg, w, *[t.cpu().numpy() for t in ins], init_mode="philox", seed=21,
                       site_keys=keys.view(np.uint64))
    assert np.abs(p.cpu().numpy() - po).max() <= TOL_TIGHT
    m.site_offset = 777
    assert torch.equal(m.forward(*ins)[1], m.forward(*ins, site_keys=torch.arange(777, 777 + n, device="cuda:0"))[1])
    with pytest.raises(RuntimeError):
        m.forward(*ins, site_keys=kd[:-1])
    with pytest.raises(RuntimeError):
        m.forward(*ins, site_keys=kd.cpu())
    # full batch + tail: permutation equivariance under non-zero states, bit for bit, fp32 and bf16x9
    n = 65536 + 999
    ins = synth.feature_batch(n, device="cuda:0", seed=84)
    kd = torch.arange(5_000_000_000, 5_000_000_000 + n, device="cuda:0")
    perm = torch.randperm(n, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(2))
    for precision in ("fp32", "bf16x9"):
        m.set_precision(precision)
        p = m.forward(*ins, site_keys=kd)[1].clone()
        pp = m.forward(*[t[perm] for t in ins], site_keys=kd[perm])[1]
        assert torch.equal(pp, p[perm]), precision


def test_handles_on_separate_streams_run_concurrently_with_unchanged_results():
    """include/dsp_amd.h "Threading / ownership": one handle per stream for concurrent forwards.  Small batches (the
    reference's default 512, call_modifications.py:147) fill 1/16 of the GPU each; six handles on six streams, issued
    back to back so that their launches overlap on the device, must each return exactly what the same batch gives alone
    (own scratch per handle, nothing shared but the read-only inputs) -- and against the oracle for one of them."""
    torch = _torch()
    from deepsignal_plant_amd import synth
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    cfg = onp.OracleConfig()
    w = onp.make_weights(cfg, 71, 2.0)
    nh, n = 6, 512 + 37
    models = [build_model(cfg, w, init_state="randn", seed=5) for _ in range(nh)]
    batches = [synth.feature_batch(n, device="cuda:0", seed=300 + i) for i in range(nh)]
    alone = []
    for i in range(nh):
        models[0].site_offset = 1000 * i
        alone.append(models[0].forward(*batches[i])[1].clone())
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream("cuda:0") for _ in range(nh)]
    for _ in range(5):   # several rounds: overlap is a matter of timing
        outs = [None] * nh
        for i in range(nh):
            with torch.cuda.stream(streams[i]):
                models[i].site_offset = 1000 * i
                outs[i] = models[i].forward(*batches[i])[1]
        torch.cuda.synchronize()
        for i in range(nh):
            assert torch.equal(outs[i], alone[i]), i
    sample = [t.cpu().numpy() for t in batches[3]]
    _, po = oc.forward(cfg, w, *sample, init_mode="philox", seed=5, site_offset=3000)
    assert float(np.abs(outs[3].cpu().numpy() - po).max()) <= TOL_TIGHT


_STRESS = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
from oracle import forward_np as onp
cfg = onp.OracleConfig()
w = onp.make_weights(cfg, 72, 2.0)
def build():
    m = ModelBiLSTM(13, 16, 3, 1, 2, 0, 256, 16, 4, True, True, module="both_bilstm", device=0, init_state="randn", seed=6)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return m.cuda(0).eval()
nh, n = 16, 500
os.environ["DSP_LSTM_CLUSTER"] = "0"
ref_model = build()
del os.environ["DSP_LSTM_CLUSTER"]
models = [build() for _ in range(nh)]
batches = [synth.feature_batch(n, device="cuda:0", seed=700 + i) for i in range(nh)]
alone = []
for i in range(nh):
    ref_model.site_offset = 1000 * i
    alone.append(ref_model.forward(*batches[i])[1].clone())
torch.cuda.synchronize()
streams = [torch.cuda.Stream("cuda:0") for _ in range(nh)]
bad = 0
for _ in range(12):
    outs = [None] * nh
    for rep in range(3):          # three forwards per handle back to back: the queues stay full
        for i in range(nh):
            with torch.cuda.stream(streams[i]):
                models[i].site_offset = 1000 * i
                outs[i] = models[i].forward(*batches[i])[1]
    torch.cuda.synchronize()
    bad += sum(0 if torch.equal(outs[i], alone[i]) else 1 for i in range(nh))
print("stress: %%d of %%d concurrent clustered forwards differ" %% (bad, 12 * nh))
sys.exit(1 if bad else 0)
"""


@pytest.mark.parametrize("queues", ["4", "32"])
def test_sixteen_clustered_forwards_at_once_neither_hang_nor_change_a_bit(queues):
    """Round 4: a forward of <= 512 sites spreads every (site tile, direction) of the combined stack over a cluster of 8
    workgroups that wait for each other (dsp_lstmc_kernel) -- which needs all 8 resident.  Sixteen handles on sixteen streams
    ask for 16 x 256 such workgroups at once; with 32 hardware queues (GPU_MAX_HW_QUEUES, read when the runtime starts: a child
    process) the dispatcher shares the CUs among many launches and can leave clusters incomplete (five launches sharing an
    XCD's 32 slots hold 6 members of 8 each -- without a way out nobody would ever finish).  The members of a cluster that does
    not assemble in time abandon it and the clean-up launch computes it (cluster_admit): every forward completes, and
    returns exactly what the same batch gives alone on the unclustered path."""
    import os
    import subprocess
    import sys
    from tests.helpers import ROOT
    r = subprocess.run([sys.executable, "-c", _STRESS % {"root": ROOT}], cwd=ROOT, capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, GPU_MAX_HW_QUEUES=queues))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "stress: 0 of 192" in r.stdout


@pytest.mark.parametrize("n", [500, 2048, 4096, 9000])
def test_forward_captures_into_a_hip_graph_and_replays_the_same_bits(n):
    """`dsp_forward` is asynchronous on the caller's stream, allocates nothing once reserved and never synchronises: a caller
    may capture it into a HIP graph (here: torch.cuda.CUDAGraph).  The capture takes in the handle's side stream (the signal
    branch forked from and joined back into the captured stream), the clustered launches with their clean-up launches and the
    counter zeroing of the first kernel; a replay on new input contents gives the bits of an eager forward.  (The Philox
    site offset is a kernel argument: a replay keeps the one it was captured with.)"""
    torch = _torch()
    from deepsignal_plant_amd import synth
    from oracle import forward_np as onp
    cfg = onp.OracleConfig()
    m = build_model(cfg, onp.make_weights(cfg, 81, 2.0), init_state="randn", seed=4)
    m.reserve(n)
    static = [t.clone() for t in synth.feature_batch(n, device="cuda:0", seed=11)]
    other = synth.feature_batch(n, device="cuda:0", seed=12)
    m.site_offset = 123
    eager = [m.forward(*static)[1].clone(), m.forward(*other)[1].clone()]
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        m.forward(*static)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m.forward(*static)[1]
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager[0])
    for a, b in zip(static, other):
        a.copy_(b)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager[1])


def test_small_batches_really_take_the_clustered_forms(tmp_path):
    """The small-batch forms are chosen silently (a handle whose XCC probe fails, a shape a form does not take: the launch
    falls back to a slower kernel with the same bits) -- round 5's first probe switched clustering off for every handle but a
    process's first and every bit-identity test still passed.  So: what DSP_DEBUG_LSTM prints for three model shapes at 512
    sites, in a fresh process -- gates per wave of every LSTM launch (CG: 1 = a cluster of UT workgroups; 4 = one workgroup)."""
    import os
    import subprocess
    import sys
    from tests.helpers import ROOT
    script = tmp_path / "shapes.py"
    script.write_text('''
import sys
sys.path.insert(0, %r)
import torch
from deepsignal_plant_amd import synth
from deepsignal_plant_amd.models import ModelBiLSTM
for kw in (dict(), dict(module="seq_bilstm", num_layers1=2), dict(hidden_size=128), dict(module="signal_bilstm", num_layers1=1)):
    for rep in range(2):   # a process's SECOND handle of a shape too
        m = ModelBiLSTM(init_state="randn", seed=3, **kw)
        m.load_state_dict(synth.random_state_dict(m, seed=5)); m.cuda(0)
        assert m.query("clustering") == 1 and m.query("xcc_probe_failed") == 0
        sys.stderr.write("== %%s %%d\\n" %% (sorted(kw.items()), rep)); sys.stderr.flush()
        m(*synth.feature_batch(512, device="cuda:0", seed=9)); torch.cuda.synchronize()
''' % ROOT)
    env = dict(os.environ, DSP_DEBUG_LSTM="1")
    for k in ("DSP_LSTM_CLUSTER", "DSP_LSTM_FRONT_CLUSTER", "DSP_TWO_STREAMS", "DSP_LSTM_TILING"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    runs, cur = {}, None
    for line in r.stderr.splitlines():
        if line.startswith("== "):
            cur = line[3:]
            runs[cur] = []
        elif line.startswith("[lstm] ") and " CG=" in line and cur is not None:
            w = line.split()
            runs[cur].append((w[1], int([x for x in w if x.startswith("CG=")][0][3:]), int([x for x in w if x.startswith("UT=")][0][3:])))
    assert len(runs) == 8, list(runs)
    for name, launches in runs.items():
        assert launches, name
        for lstm, cg, ut in launches:
            if ut in (4, 8):   # every layer of 4 or 8 unit tiles of these shapes has a clustered form at 512 sites
                assert cg == 1, (name, launches)
    default = [v for k, v in runs.items() if k.startswith("[] ")][0]
    assert [x[0] for x in default] == ["lstm_seq", "lstm_signal", "lstm_comb", "lstm_comb", "lstm_comb"] and all(x[1] == 1 for x in default)
