"""CPU: the KERNELS' OWN SOURCE run on the host (round 6; GPU access was closed for the whole round).

tests/native/emu is a SIMT interpreter for the test-suite: csrc/dsp_kernels.hip + csrc/dsp_capi.cpp compiled for the host
(-DDSP_EMU, a hip/ header pair in front of the real one), every GPU thread a fiber, 64 of them a wave, MFMAs / readfirstlane /
shuffles / the workgroup barrier carried out when the wave (the workgroup) has arrived, the workgroups of a launch scheduled
concurrently -- the clustered launches' members really wait for each other --, buffer descriptors with the hardware's
range-check semantics (a load past num_records returns zeros, a store past it is dropped).  It checks the kernels' LOGIC --
indexing, extents, the hand-off protocol's control flow, the clean-up path, the piece cut -- never their speed or the memory
model (sequentially consistent here: tests/native/cluster_model.cpp is where store buffers live).

What is held, all through the C ABI of that library (dsp_model_create / dsp_forward, bound here by hand: the package's loader
REFUSES the library -- it answers dsp_abi_version() with 1003 -- so it can never stand in for the product):
  * parity of the kernels' source with the REFERENCE's outputs on F1 fixtures (1e-6; the GPU suite's bound is 2e-5);
  * every kernel form gives the same bytes: the A/B switch matrix of test_small_batch_kernels_do_not_change_a_bit on a small model
    (full-batch kernels, <2 unit tiles, 1 site tile>, clusters of every size, round-4 hand-off, every cluster abandoned to the
    clean-up launch, one stream, one-tile fc / head off) and an adversarial wave schedule (DSP_EMU_SEED);
  * the round-6 extents: region / tight / wide bit-identical; the BOUNDS-RECORDING build (-DDSP_BOUNDS) runs every form without
    a record -- the tight extents fit every access the kernels make -- and names the operand when an extent is shortened;
  * Philox states against the C oracle's same generator; ragged sizes; a call cut into pieces; the many-pass kernel (hidden
    320); the split-precision kernels against the fp32 path.
The interpreter runs about 10^7 lane-instructions a second: by default this module takes about a minute and runs a
representative cut of everything above; DSP_EMU_LONG=1 runs all of it -- the whole switch matrix on two models with three
state modes, the default architecture (hidden 256 x 3 layers, clusters of 8: a minute per forward), every bounds-build case --
in a quarter of an hour (profiles/r6/kernel_emu_long.txt is that run's log)."""
import contextlib
import ctypes
import os
import subprocess

import numpy as np
import pytest

from deepsignal_plant_amd import _native as nat   # (the ctypes structures only: the emulated library is bound by hand below)
from oracle import c_oracle as oc
from oracle import forward_np as onp
from tests import bgjobs
from tests.helpers import ROOT, load_f1

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
CSRC = os.path.join(ROOT, "deepsignal_plant_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "native", "emu")
SWITCHES = ("DSP_LSTM_CLUSTER", "DSP_LSTM_LOCAL8", "DSP_TWO_STREAMS", "DSP_HEAD_ST4", "DSP_LSTM_TILING", "DSP_CLUSTER_TIMEOUT", "DSP_FC_FUSED",
            "DSP_LSTM_FRONT_CLUSTER", "DSP_FC_SMALL", "DSP_LSTM_HANDOFF", "DSP_FORWARD_SPLIT", "DSP_RSRC_EXTENTS", "DSP_EMU_SEED", "DSP_BOUNDS_TEST_SHRINK",
            "DSP_PRECISION", "EMU_CUS", "DSP_LSTM_XAHEAD", "DSP_LSTM_XAHEAD_TILES", "DSP_LSTM_XAHEAD_RING")

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="the image's clang++ builds the interpreter")
LONG = bool(os.environ.get("DSP_EMU_LONG"))
# DSP_EMU_SANITIZE=1: the extraction and call_freq interpreter libraries are built with AddressSanitizer + UBSan (run with the shared
# ASan runtime preloaded: test_the_extraction_and_call_freq_kernels_under_sanitizers does)
SAN = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-shared-libsan"] if os.environ.get("DSP_EMU_SANITIZE") else []


def _cache_dir(*flags):
    """a build directory keyed by the sources it is built from (tests/native/_build/, git-ignored): a second run of the suite on
    unchanged sources does not compile the interpreter again"""
    import hashlib
    h = hashlib.sha256(" ".join(flags).encode())
    for f in sorted(os.listdir(CSRC)) + sorted(os.path.join("..", "..", "tests", "native", "emu", x) for x in ("hip_emu.cpp", "hip/hip_runtime.h", "hip/hip_runtime_api.h")) + \
            [os.path.join("..", "..", "tests", "native", x) for x in ("parse_dev_host.cpp",)] + [os.path.join("..", "..", "include", "dsp_amd.h")]:
        path = os.path.join(CSRC, f)
        if os.path.isfile(path) and path.endswith((".hip", ".h", ".cpp")):
            h.update(open(path, "rb").read())
    d = os.path.join(ROOT, "tests", "native", "_build", h.hexdigest()[:16])
    os.makedirs(d, exist_ok=True)
    return d


def _build(out, *flags):
    if not os.path.exists(out):
        _compile(out, *flags)
    L = ctypes.CDLL(out)
    L.dsp_last_error.restype = ctypes.c_char_p
    L.dsp_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                              ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(nat.InitState), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.dsp_model_destroy.argtypes = [ctypes.c_void_p]
    L.dsp_model_set_precision.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    assert L.dsp_abi_version() == nat.ABI_VERSION + 1000
    return L


def _compile(out, *flags):
    tmp = out + ".tmp%d" % os.getpid()
    cmd = [CLANG, "-std=c++17", "-O2", "-march=native", "-fPIC", "-shared", "-DDSP_EMU", "-Wno-unused-value", "-I", EMU, "-I", os.path.join(ROOT, "include"),
           "-I", CSRC, "-x", "c++", os.path.join(CSRC, "dsp_kernels.hip"), os.path.join(CSRC, "dsp_capi.cpp"), os.path.join(EMU, "hip_emu.cpp"), "-o", tmp,
           "-pthread"] + list(flags)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    os.replace(tmp, out)


def _compile_emu_lib(name, *flags):
    out = os.path.join(_cache_dir("emu"), name)
    if not os.path.exists(out):
        _compile(out, *flags)
    return out


# the two interpreter builds are background jobs (tests/bgjobs.py) that any selected test of this file starts at collection end:
# a minute of compilation each, side by side, next to the tests in front of this module
bgjobs.job("emu_lib")(lambda: _compile_emu_lib("libdsp_amd_emu.so"))
bgjobs.job("emu_lib_bounds")(lambda: _compile_emu_lib("libdsp_amd_emu_bounds.so", "-DDSP_BOUNDS"))
bgjobs.module_uses("test_kernel_emu.py", ["emu_lib", "emu_lib_bounds"])


@pytest.fixture(scope="module")
def libs():
    """the interpreter build of the library and its bounds-recording twin"""
    return _build(bgjobs.result("emu_lib")), _build(bgjobs.result("emu_lib_bounds"))


@pytest.fixture(scope="module")
def emu(libs):
    return libs[0]


@pytest.fixture(scope="module")
def emu_bounds(libs):
    return libs[1]


@contextlib.contextmanager
def env(**kw):
    saved = {k: os.environ.get(k) for k in SWITCHES}
    for k in SWITCHES:
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in kw.items()})
    try:
        yield
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


class Model(object):
    """dsp_model_create / dsp_forward of the emulated library on host arrays (by hand)"""

    def __init__(self, L, cfg, w, precision=None):
        self.L, self.cfg = L, cfg
        mod = {"both_bilstm": 0, "seq_bilstm": 1, "signal_bilstm": 2}[cfg.module]
        c = nat.ModelCfg(cfg.seq_len, cfg.signal_len, cfg.num_layers1, cfg.num_layers2, cfg.num_classes, cfg.hidden_size, cfg.vocab_size,
                         cfg.embedding_size, int(cfg.is_base), int(cfg.is_signallen), mod)
        ws = [np.ascontiguousarray(w[k], dtype=np.float32) for k, _ in onp.state_dict_spec(cfg)]
        self.h = ctypes.c_void_p()
        rc = L.dsp_model_create(ctypes.byref(c), (ctypes.c_void_p * len(ws))(*[x.ctypes.data for x in ws]), (ctypes.c_int64 * len(ws))(*[x.size for x in ws]),
                                len(ws), 0, ctypes.byref(self.h))
        assert rc == 0, L.dsp_last_error()
        if precision:
            assert L.dsp_model_set_precision(self.h, nat.PRECISION[precision]) == 0, L.dsp_last_error()

    def forward(self, inputs, states=None, philox=None, expect_rc=0):
        """states: the reference layout (explicit); philox: (seed, site_offset); neither: zero states -> (probs, logits, labels)"""
        n = int(inputs[0].shape[0])
        a = [np.ascontiguousarray(x, dtype=np.float32) for x in inputs]
        keep = []
        if states is not None:
            st = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in states.items()}
            keep.append(st)
            init = nat.InitState(1, 0, 0, *[st[k].ctypes.data if k in st else None for k in ("h_seq", "c_seq", "h_sig", "c_sig", "h_comb", "c_comb")], None)
        elif philox is not None:
            init = nat.InitState(2, philox[0], philox[1], None, None, None, None, None, None, None)
        else:
            init = nat.InitState(0, 0, 0, None, None, None, None, None, None, None)
        C = self.cfg.num_classes
        logits, probs, labels = np.zeros((n, C), np.float32), np.zeros((n, C), np.float32), np.zeros(n, np.uint8)
        rc = self.L.dsp_forward(self.h, None, n, a[0].ctypes.data, 0, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, 0, a[4].ctypes.data,
                                ctypes.byref(init), logits.ctypes.data, probs.ctypes.data, labels.ctypes.data)
        if expect_rc:
            assert rc == expect_rc, (rc, self.L.dsp_last_error())
            return self.L.dsp_last_error().decode()
        assert rc == 0, self.L.dsp_last_error()
        return probs, logits, labels

    def activation(self, which, n):
        """dsp_debug_read_activation: 0 = the combined stack's input (relu(fc) of both branches), 1 = its output; [n, T, features]"""
        feats = self.cfg.hidden_size if which == 0 else 2 * self.cfg.hidden_size
        out = np.zeros((n, self.cfg.seq_len, feats), np.float32)
        rc = self.L.dsp_debug_read_activation(self.h, None, ctypes.c_int32(which), ctypes.c_int64(n), ctypes.c_void_p(out.ctypes.data))
        assert rc == 0, self.L.dsp_last_error()
        return out

    def close(self):
        self.L.dsp_model_destroy(self.h)


# ---- parity of the kernels' source with the reference's outputs ------------------------------------------------------------------

@pytest.mark.parametrize("name", ["tiny_h64_l2", "nobase_h128"] + (["nosiglen_h128", "both_default", "signal_only", "seq_cfg3"] if LONG else []))
def test_the_kernels_source_reproduces_the_reference_fixture(emu, name):
    f = load_f1(name)
    inter = {}
    with env():
        m = Model(emu, f["cfg"], f["w"])
        probs, logits, labels = m.forward(f["inputs"], states=f["states"])
        # the reference's intermediates (forward hooks, first 8 sites) against the K4 buffers read back through the C ABI
        if "lstm_comb" in f["inter"]:
            inter["lstm_comb"] = (m.activation(1, f["n"])[: f["inter"]["lstm_comb"].shape[0]], f["inter"]["lstm_comb"])
        if f["cfg"].module == "both_bilstm" and "relu_seq" in f["inter"]:
            inter["relu_fc"] = (m.activation(0, f["n"])[: f["inter"]["relu_seq"].shape[0]], np.concatenate((f["inter"]["relu_seq"], f["inter"]["relu_signal"]), axis=2))
        m.close()
    for k, (got, ref) in inter.items():
        assert np.abs(got - ref).max() <= 5e-6, (name, k, float(np.abs(got - ref).max()))
    dp = float(np.abs(probs - f["probs"]).max())
    print(name, "interpreted kernels vs the reference: max|dprob| %.2e (n = %d)" % (dp, f["n"]))
    assert dp <= 1e-6 and np.abs(logits - f["logits"]).max() <= 2e-5
    sure = np.abs(f["probs"][:, 1] - 0.5) >= 1e-4 if f["probs"].shape[1] == 2 else np.ones(f["n"], bool)
    assert np.array_equal(labels[sure], f["probs"].argmax(1)[sure])


# ---- a small model with every kernel form: front ends of 4 unit tiles (hidden 128 each), a combined stack of 8 ---------------------

T_SMALL = 5 if LONG else 1
SMALL = dict(seq_len=T_SMALL, signal_len=8, hidden_size=256, num_layers1=1, num_layers2=1)     # both_bilstm: hseq = hsig = 128 (UT 4), combined 256 (UT 8)
SMALL4 = dict(seq_len=T_SMALL, signal_len=8, hidden_size=128, num_layers1=2, num_layers2=1)    # combined stack of 4 unit tiles (the clustered dense4 forms), front ends of 2
SHORT = ("auto", "full_batch_kernels", "every_cluster_abandoned", "adversarial_schedule_2_G2", "descriptors_tight")


def _case(kw, n, seed=3):
    cfg = onp.OracleConfig(**kw)
    return cfg, onp.make_weights(cfg, 50 + seed, 2.0), onp.make_inputs(cfg, n, 60 + seed), onp.make_init_states(cfg, n, 70 + seed)


MODES = {
    "auto": {},
    "full_batch_kernels": {"DSP_LSTM_CLUSTER": "0", "DSP_LSTM_TILING": "0", "DSP_LSTM_LOCAL8": "0", "DSP_TWO_STREAMS": "0", "DSP_HEAD_ST4": "1", "DSP_FC_FUSED": "0",
                           "DSP_FC_SMALL": "0"},
    "lstm21": {"DSP_LSTM_CLUSTER": "0", "DSP_LSTM_TILING": "21"},
    "G4": {"DSP_LSTM_CLUSTER": "4"}, "G2": {"DSP_LSTM_CLUSTER": "2"}, "G1": {"DSP_LSTM_CLUSTER": "1"},
    "front_G1_one_stream": {"DSP_LSTM_FRONT_CLUSTER": "1", "DSP_TWO_STREAMS": "0"}, "front_G2": {"DSP_LSTM_FRONT_CLUSTER": "2"},
    "front_local": {"DSP_LSTM_FRONT_CLUSTER": "0"},
    "round4_handoff": {"DSP_LSTM_HANDOFF": "0"},
    "every_cluster_abandoned": {"DSP_CLUSTER_TIMEOUT": "0"},
    "every_cluster_abandoned_round4_handoff_G2": {"DSP_CLUSTER_TIMEOUT": "0", "DSP_LSTM_HANDOFF": "0", "DSP_LSTM_CLUSTER": "2", "DSP_LSTM_FRONT_CLUSTER": "2"},
    "adversarial_schedule_1": {"DSP_EMU_SEED": "1"}, "adversarial_schedule_2_G2": {"DSP_EMU_SEED": "2", "DSP_LSTM_CLUSTER": "2"},
    "descriptors_tight": {"DSP_RSRC_EXTENTS": "tight"}, "descriptors_2GiB_windows": {"DSP_RSRC_EXTENTS": "wide"},
    "descriptors_tight_abandoned": {"DSP_RSRC_EXTENTS": "tight", "DSP_CLUSTER_TIMEOUT": "0"},
    "64_compute_units": {"EMU_CUS": "64"},
}


@pytest.mark.parametrize("kw,label", [(SMALL, "UT8_stack_UT4_front_ends")] + ([(SMALL4, "UT4_stack_UT2_front_ends")] if LONG else []))
def test_every_kernel_form_gives_the_same_bytes(emu, kw, label):
    cfg, w, ins, st = _case(kw, 45)
    want = oc.forward(cfg, w, *ins, states=st)[1]
    first = None
    for mode, sw in MODES.items():
        if not LONG and mode not in SHORT:
            continue
        with env(**sw):
            m = Model(emu, cfg, w)
            probs = m.forward(ins, states=st)[0]
            pz = m.forward(ins)[0] if LONG or first is None else first[1]
            pp = m.forward(ins, philox=(11, 1000))[0] if LONG or first is None or mode == "every_cluster_abandoned" else first[2]
            m.close()
        if first is None:
            first = (probs, pz, pp)
            assert np.abs(probs - want).max() <= 1e-6, (label, mode)
            assert np.abs(pz - oc.forward(cfg, w, *ins, init_mode="zeros")[1]).max() <= 1e-6
            assert np.abs(pp - oc.forward(cfg, w, *ins, init_mode="philox", seed=11, site_offset=1000)[1]).max() <= 2e-6
        else:
            for got, ref, what in zip((probs, pz, pp), first, ("explicit", "zeros", "philox")):
                assert np.array_equal(got, ref), (label, mode, what, float(np.abs(got - ref).max()))


def test_x_ahead_gives_the_same_bytes(emu, emu_bounds):
    """DSP_LSTM_XAHEAD=1 (round 6, opt-in): dsp_xahead_kernel sums the k-groups [0, xs) of every step's x part ahead of the
    recurrence, dsp_lstmc_kernel<.., XA> starts each step's accumulators from those sums and keeps one ring of x part; tiles
    without a live site are skipped.  Same MFMAs in the same order: the bytes of every cluster size, hand-off, schedule, of
    abandoned clusters (computed whole by the clean-up launch) and of both x-part lengths (32 and 64 k-groups: layers 0 and 1+)
    are those of the undivided launches; the bounds build records nothing; above the live-tile limit the form is not taken.
    (DSP_LSTM_XAHEAD_RING=8: the one-gate-per-wave form with rings 8 deep, keeping 8 k-groups of x part.)
    (DSP_EMU_LONG=1: all ten switch sets on two sizes, the seq-only shape; the default run a cut of it.)"""
    kw = dict(seq_len=3 if LONG else 2, signal_len=8, hidden_size=256, num_layers1=2, num_layers2=1)
    modes = [{}, {"DSP_LSTM_CLUSTER": "2"}, {"DSP_LSTM_CLUSTER": "4"}, {"DSP_LSTM_HANDOFF": "0"}, {"DSP_CLUSTER_TIMEOUT": "0"},
             {"DSP_EMU_SEED": "2"}, {"DSP_RSRC_EXTENTS": "tight"}, {"DSP_EMU_SEED": "3", "DSP_LSTM_HANDOFF": "0", "DSP_LSTM_CLUSTER": "2"},
             {"DSP_LSTM_XAHEAD_RING": "8"}, {"DSP_LSTM_XAHEAD_RING": "8", "DSP_LSTM_HANDOFF": "0", "DSP_EMU_SEED": "4"}]
    if not LONG:
        modes = [{}, {"DSP_CLUSTER_TIMEOUT": "0", "DSP_LSTM_HANDOFF": "0", "DSP_LSTM_CLUSTER": "2", "DSP_EMU_SEED": "3"}, {"DSP_LSTM_XAHEAD_RING": "8"}]
    for n in ((45, 200, 1, 256) if LONG else (45,)):      # (256 sites = the 8 live tiles up to which the form is taken)
        cfg, w, ins, st = _case(kw, n)
        with env():
            m = Model(emu, cfg, w)
            want = (m.forward(ins, states=st)[0], m.forward(ins, philox=(11, 1000))[0])
            m.close()
        assert np.abs(want[0] - oc.forward(cfg, w, *ins, states=st)[1]).max() <= 1e-6
        for k, sw in enumerate(modes if n == 45 else modes[:1]):
            with env(DSP_LSTM_XAHEAD="1", **sw):
                m = Model(emu, cfg, w)
                got = m.forward(ins, states=st)[0]
                assert np.array_equal(got, want[0]), (n, sw)
                if LONG or k == 0:
                    assert np.array_equal(m.forward(ins, philox=(11, 1000))[0], want[1]), (n, sw)
                m.close()
        with env(DSP_LSTM_XAHEAD="1"):
            m = Model(emu_bounds, cfg, w)      # (tight extents, every access compared: a record would come back as DSP_EBOUNDS)
            got = m.forward(ins, states=st)[0]
            m.close()
        assert np.array_equal(got, want[0])
        if LONG:
            with env(DSP_LSTM_XAHEAD="1", DSP_LSTM_XAHEAD_TILES="1"):   # (two live tiles or more: not taken)
                m = Model(emu, cfg, w)
                got = m.forward(ins, states=st)[0]
                m.close()
            assert np.array_equal(got, want[0])
    # hidden 128: dense layers of 4 unit tiles on clusters of 4 (2 with DSP_LSTM_CLUSTER=2), rings 4 deep
    cfg, w, ins, st = _case(dict(SMALL4, seq_len=2), 45)
    out = []
    for sw in (({}, {"DSP_LSTM_XAHEAD": "1"}, {"DSP_LSTM_XAHEAD": "1", "DSP_LSTM_CLUSTER": "2", "DSP_LSTM_HANDOFF": "0"}, {"DSP_LSTM_XAHEAD": "1", "DSP_CLUSTER_TIMEOUT": "0"})
               if LONG else ({}, {"DSP_LSTM_XAHEAD": "1"})):
        with env(**sw):
            m = Model(emu, cfg, w)
            out.append(m.forward(ins, states=st)[0])
            m.close()
        assert np.array_equal(out[-1], out[0]), sw
    assert np.abs(out[0] - oc.forward(cfg, w, *ins, states=st)[1]).max() <= 1e-6
    if not LONG:
        return
    # the seq-only shape of BASELINE configs[2], with a second front-end layer: dense layers of 8 unit tiles in both stacks
    cfg, w, ins, st = _case(dict(seq_len=2, signal_len=8, hidden_size=256, num_layers1=2, num_layers2=2, module="seq_bilstm"), 40)
    out = []
    for xa in ("0", "1"):
        with env(DSP_LSTM_XAHEAD=xa):
            m = Model(emu, cfg, w)
            out.append(m.forward(ins, states=st)[0])
            m.close()
    assert np.array_equal(out[0], out[1])
    assert np.abs(out[0] - oc.forward(cfg, w, *ins, states=st)[1]).max() <= 1e-6


def test_ragged_sizes_and_a_cut_call(emu):
    """1, 33, 513 sites (tile tails, two classes of cluster sizes) and 1,100 sites = 1,024 + 76: the cut changes no bit"""
    kw = dict(seq_len=3 if LONG else 2, signal_len=8, hidden_size=128, num_layers1=1, num_layers2=1)
    cfg = onp.OracleConfig(**kw)
    w = onp.make_weights(cfg, 5, 2.0)
    for n in ((1, 33, 513) if LONG else (33,)):
        ins = onp.make_inputs(cfg, n, 100 + n)
        with env():
            m = Model(emu, cfg, w)
            pp = m.forward(ins, philox=(7, 5 * n))[0]
            m.close()
        assert np.abs(pp - oc.forward(cfg, w, *ins, init_mode="philox", seed=7, site_offset=5 * n)[1]).max() <= 2e-6, n
    ins = onp.make_inputs(cfg, 1100, 9)
    out = {}
    for split in ("1", "0"):
        with env(DSP_FORWARD_SPLIT=split):
            m = Model(emu, cfg, w)
            out[split] = m.forward(ins, philox=(7, 0))[0]
            m.close()
    assert np.array_equal(out["1"], out["0"])
    assert np.abs(out["1"] - oc.forward(cfg, w, *ins, init_mode="philox", seed=7, site_offset=0)[1]).max() <= 2e-6


@pytest.mark.skipif(not LONG, reason="DSP_EMU_LONG=1: up to 9,001 sites through the interpreter, five minutes")
def test_the_forms_each_batch_size_takes_by_itself(emu):
    """no switch set: 513 sites (clusters of 4), 1,025 (of 2), 2,049 (one eight-wave workgroup per site tile and direction),
    4,097 (4,096 + 1: a cut), 9,001 (a round of 8,192 sites on the full-batch kernels + 809) of a hidden-256 model, Philox
    states, against the C oracle"""
    cfg = onp.OracleConfig(seq_len=2, signal_len=8, hidden_size=256, num_layers1=2, num_layers2=1)
    w = onp.make_weights(cfg, 5, 2.0)
    for n in (513, 1025, 2049, 4097, 9001):
        ins = onp.make_inputs(cfg, n, 100 + n)
        with env():
            m = Model(emu, cfg, w)
            pp = m.forward(ins, philox=(7, 5 * n))[0]
            m.close()
        d = float(np.abs(pp - oc.forward(cfg, w, *ins, init_mode="philox", seed=7, site_offset=5 * n)[1]).max())
        print("%5d sites: max|dprob| vs the C oracle %.2e" % (n, d))
        assert d <= 2e-6, n


def test_the_many_pass_kernel_and_padded_shapes(emu):
    """hidden 320 (two passes per step, the cell state in the scratch behind a descriptor), hidden 100 (padded unit tiles), a
    signal window wider than 32 features, no k-mer / no lengths"""
    cases = (dict(seq_len=3, signal_len=8, hidden_size=320, num_layers1=1), dict(seq_len=3, signal_len=8, hidden_size=100, num_layers1=2),
             dict(seq_len=3, signal_len=40, hidden_size=64, num_layers1=1), dict(seq_len=3, signal_len=8, hidden_size=64, is_base=False, is_signallen=False))
    for kw in (cases if LONG else (dict(seq_len=1, signal_len=8, hidden_size=320, num_layers1=1), cases[2])):
        cfg, w, ins, st = _case(kw, 37)
        with env():
            m = Model(emu, cfg, w)
            probs = m.forward(ins, states=st)[0]
            m.close()
        assert np.abs(probs - oc.forward(cfg, w, *ins, states=st)[1]).max() <= 1e-6, kw


def test_split_precision_kernels_against_the_fp32_path(emu):
    cfg, w, ins, st = _case(dict(seq_len=3 if LONG else 1, signal_len=16, hidden_size=256, num_layers1=1), 40)
    with env():
        m = Model(emu, cfg, w)
        ref = m.forward(ins, states=st)[0]
        m.close()
        for precision, tol in ((("bf16x9", 2e-7), ("bf16x6", 2e-6), ("fp16x3", 2e-6)) if LONG else (("bf16x9", 2e-7),)):
            m = Model(emu, cfg, w, precision=precision)
            got = m.forward(ins, states=st)[0]
            m.close()
            assert np.abs(got - ref).max() <= tol, (precision, float(np.abs(got - ref).max()))


# ---- the bounds-recording build: the TIGHT extents fit every access the kernels make -----------------------------------------------

def test_the_bounds_build_records_nothing_over_the_kernel_forms(emu, emu_bounds):
    cfg, w, ins, st = _case(SMALL, 45)
    with env():
        m = Model(emu, cfg, w)
        want = m.forward(ins, states=st)[0]
        m.close()
    for mode in (("auto", "full_batch_kernels", "lstm21", "G4", "G2", "front_G1_one_stream", "front_local", "round4_handoff", "every_cluster_abandoned") if LONG
                 else ("auto", "every_cluster_abandoned")):
        with env(**MODES[mode]):
            m = Model(emu_bounds, cfg, w)
            got = m.forward(ins, states=st)[0]        # (a record would come back as DSP_EBOUNDS: the assert inside forward())
            gz = m.forward(ins, philox=(3, 9))[0]
            m.close()
        assert np.array_equal(got, want), mode
        assert np.isfinite(gz).all()
    for kw in ((SMALL4, dict(seq_len=3, signal_len=8, hidden_size=320, num_layers1=1), dict(seq_len=3, signal_len=40, hidden_size=64, num_layers1=1)) if LONG
               else (dict(seq_len=2, signal_len=8, hidden_size=128, num_layers1=1), dict(seq_len=1, signal_len=8, hidden_size=320, num_layers1=1))):
        cfg, w, ins, st = _case(kw, 70)
        for precision in ((None, "bf16x9", "fp16x3") if LONG else ((None, "bf16x9") if kw["hidden_size"] == 128 else (None,))):
            with env():
                m = Model(emu_bounds, cfg, w, precision=precision)
                m.forward(ins, states=st)
                m.close()
    # a cut call: whole round + pieces, every piece under its own tight extents
    cfg = onp.OracleConfig(seq_len=3 if LONG else 2, signal_len=8, hidden_size=64, num_layers1=1)
    w = onp.make_weights(cfg, 5, 2.0)
    with env(EMU_CUS="32"):
        m = Model(emu_bounds, cfg, w)
        m.forward(onp.make_inputs(cfg, 1100, 9), philox=(7, 0))
        m.close()


def test_the_bounds_build_names_an_access_past_a_shortened_extent(emu_bounds):
    cfg, w, ins, st = _case(SMALL, 45)
    with env(DSP_BOUNDS_TEST_SHRINK="4096"):
        m = Model(emu_bounds, cfg, w)
        msg = m.forward(ins, states=st, expect_rc=-7)     # DSP_EBOUNDS
        m.close()
    print(msg)
    assert "out of range" in msg and "operand K4 input" in msg and "dsp_kernels.hip:" in msg


def test_the_range_check_drops_what_lies_past_an_extent(emu):
    """the interpreter's descriptors behave as dsp_debug_range_probe expects of the hardware (tests/test_gpu_zz_extents.py)"""
    emu.dsp_debug_range_probe.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]
    out = (ctypes.c_int32 * 4)()
    assert emu.dsp_debug_range_probe(0, out) == 0
    assert tuple(out) == (16, 48, 64, 1024)


def test_the_package_refuses_the_interpreter_as_its_library(emu, tmp_path):
    """no way to run the product on the CPU through it: the loader checks the ABI version the library answers with"""
    import sys
    r = subprocess.run([sys.executable, "-c", "from deepsignal_plant_amd import _native; _native.lib()"], cwd=ROOT, capture_output=True, text=True,
                       env=dict(os.environ, DSP_AMD_LIB=emu._name), timeout=300)
    assert r.returncode != 0 and "implements C-ABI version 1003" in r.stderr, r.stderr[-2000:]


@pytest.mark.skipif(not LONG, reason="DSP_EMU_LONG=1: two minutes of compilation, a quarter of an hour of interpreted forwards under sanitizers")
def test_the_kernels_under_address_and_ub_sanitizers(tmp_path):
    """GPU sanitizers are not available on this pool -- but the interpreter is a host build of the kernels whose every "device"
    allocation is a malloc block: tests/native/emu_asan_driver.cpp runs every kernel form with DSP_RSRC_EXTENTS=wide (nothing
    between a kernel's offsets and AddressSanitizer) and again with the default extents, bit-identical, without a report."""
    exe = str(tmp_path / "emu_asan")
    cmd = [CLANG, "-std=c++17", "-O1", "-g", "-march=native", "-Wno-psabi", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-DDSP_EMU", "-Wno-unused-value", "-I", EMU, "-I", os.path.join(ROOT, "include"), "-I", CSRC, os.path.join(ROOT, "tests", "native", "emu_asan_driver.cpp"),
           "-x", "c++", os.path.join(CSRC, "dsp_kernels.hip"), os.path.join(CSRC, "dsp_capi.cpp"), os.path.join(EMU, "hip_emu.cpp"), "-o", exe, "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-4000:]
    with env():
        r = subprocess.run([exe], capture_output=True, text=True, timeout=5400,
                           env=dict(os.environ, ASAN_OPTIONS="detect_stack_use_after_return=0:detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    print(r.stdout)
    assert r.returncode == 0 and "emu_asan_driver: ok" in r.stdout, (r.stdout[-2000:], r.stderr[-6000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-6000:]


def _parse_kernels_job(kernel):
    """build (once per state of the sources) and run tests/native/parse_dev_host.cpp -DPARSE_THROUGH_KERNELS under ASan + UBSan"""
    from tests.helpers import cached_build
    exe = cached_build([CLANG, "-std=c++17", "-O1", "-g", "-march=native", "-Wno-psabi", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined", "-ffp-contract=off", "-DDSP_EMU", "-DPARSE_THROUGH_KERNELS", "-Wno-unused-value", "-Wno-unused-function",
                        "-I", EMU, "-I", os.path.join(ROOT, "include"), "-I", CSRC, os.path.join(ROOT, "tests", "native", "parse_dev_host.cpp"),
                        os.path.join(CSRC, "dsp_text.cpp"), "-x", "c++", os.path.join(CSRC, "dsp_parse_dev.hip"), os.path.join(EMU, "hip_emu.cpp"), "-pthread"],
                       "parse_kernels_asan")
    n, m = ("200000", "30000") if LONG else ("5000", "800")
    e = {k: v for k, v in os.environ.items() if k not in ("DSP_PARSE_KERNEL", "DSP_PARSE_RB")}
    e.update(ASAN_OPTIONS="detect_stack_use_after_return=0", UBSAN_OPTIONS="print_stacktrace=1")
    if kernel == "rows":
        e["DSP_PARSE_KERNEL"] = "rows"
    return subprocess.run([exe, n, m], capture_output=True, text=True, timeout=3000, env=e)


for _k in ("tokens", "rows"):
    bgjobs.job("parse_kernels_" + _k)(lambda k=_k: _parse_kernels_job(k))


@pytest.mark.parametrize("kernel", ["tokens", "rows"])
@bgjobs.uses(lambda p: ["parse_kernels_" + p["kernel"]])
def test_the_row_parsers_kernels_under_sanitizers_against_the_host_parser(kernel):
    """csrc/dsp_parse_dev.hip -- the token-parallel kernel and the thread-per-row pair -- compiled for the host by the interpreter
    and run under ASan + UBSan over random float spellings, the writer's grammar and byte-mutated blocks
    (tests/native/parse_dev_host.cpp -DPARSE_THROUGH_KERNELS): every array the kernels touch -- the staged text and its 64 bytes
    of slack, the row offsets, the segment table, the outputs -- is exactly sized, so a cursor that runs past a row, a table
    entry followed out of the block, a token stored one too far is a report.  Every accepted row equals the host parser's bit for
    bit, every row it rejects is flagged.  (Round 5's one unexplained death of a GPU-suite run fell between test_gpu_parse.py and
    the eight-rank bench: this is those kernels with a sanitizer on.)  A background job (tests/bgjobs.py)."""
    r = bgjobs.result("parse_kernels_" + kernel)
    print(r.stdout)
    assert r.returncode == 0 and "parse_dev_host: ok (through the interpreted kernels" in r.stdout, (r.stdout[-2000:], r.stderr[-6000:])
    assert ("thread-per-row" in r.stdout) == (kernel == "rows")
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-6000:]
    assert "never accepted what the host rejects" in r.stdout


def test_random_model_shapes_through_the_interpreter(emu):
    """tools/parity_sweep.py's idea at the interpreter's scale: random model shapes -- every flag of the reference's constructor,
    hidden sizes 20 ... 200 (padded unit tiles, 1 ... 8 of them), 1 ... 2 layers, odd k-mer lengths, any signal window, 2 ... 5
    classes -- x random batch sizes with tile tails, explicit N(0,1) states against the C oracle"""
    rng = np.random.default_rng(2026)
    worst = 0.0
    for case in range(40 if LONG else 6):
        module = ["both_bilstm", "seq_bilstm", "signal_bilstm"][int(rng.integers(0, 3))]
        hidden = int(rng.choice([20, 32, 50, 64, 96, 100, 128, 160, 200]))
        if module == "both_bilstm" and hidden % 2:
            hidden += 1
        cfg = onp.OracleConfig(seq_len=int(rng.choice([1, 2, 3, 5])), signal_len=int(rng.choice([4, 8, 12, 16, 24, 40])), num_layers1=int(rng.integers(1, 3)),
                               num_layers2=int(rng.integers(1, 3)), num_classes=int(rng.choice([2, 2, 3, 5])), hidden_size=hidden, vocab_size=int(rng.choice([5, 16])),
                               embedding_size=int(rng.choice([2, 4, 6])), is_base=bool(rng.integers(0, 2)), is_signallen=bool(rng.integers(0, 2)), module=module)
        n = int(rng.choice([1, 5, 31, 32, 33, 64, 65, 100]))
        w = onp.make_weights(cfg, 10_000 + case, float(rng.choice([1.0, 2.0, 3.0])))
        ins = onp.make_inputs(cfg, n, 20_000 + case, wide_alphabet=cfg.vocab_size == 16)
        if cfg.vocab_size < 16:
            ins = (np.minimum(ins[0], cfg.vocab_size - 1),) + tuple(ins[1:])
        st = onp.make_init_states(cfg, n, 30_000 + case)
        with env():
            m = Model(emu, cfg, w)
            probs = m.forward(ins, states=st)[0]
            m.close()
        d = float(np.abs(probs - oc.forward(cfg, w, *ins, states=st)[1]).max())
        worst = max(worst, d)
        assert d <= 2e-6, (case, cfg, n, d)
    print("worst max|dprob| of the interpreted kernels against the C oracle over the random shapes: %.2e" % worst)


def test_issued_matrix_work_against_the_flops_a_forward_is_credited_with(emu):
    """bench.py's roofline credits a forward with dsp_flops_per_site x sites (the reference's multiply-adds); the kernels ISSUE
    whole 32x32x2 MFMAs over padded tiles.  Counted by the interpreter for 512 sites (whole tiles, so that only the padding of the
    SHAPES shows): issued / credited is 1 plus the front ends' zero-padded input features -- the combined stack issues exactly
    its credited work."""
    emu.emu_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
    emu.dsp_flops_per_site.restype = ctypes.c_int64
    out = (ctypes.c_ulonglong * 4)()
    cases = ((dict(seq_len=1, signal_len=16, hidden_size=256, num_layers1=1), 1.0, 1.02),
             (dict(seq_len=2, signal_len=16, hidden_size=256, num_layers1=1, module="seq_bilstm"), 1.0, 1.02))
    for kw, lo, hi in (cases if LONG else cases[:1]):
        cfg = onp.OracleConfig(**kw)
        w = onp.make_weights(cfg, 3, 1.0)
        ins = onp.make_inputs(cfg, 512, 4)
        with env():
            m = Model(emu, cfg, w)
            emu.emu_stats(out)
            m.forward(ins)
            emu.emu_stats(out)
            m.close()
        issued = out[0] * 32 * 32 * 2 * 2
        mod = {"both_bilstm": 0, "seq_bilstm": 1, "signal_bilstm": 2}[cfg.module]
        c = nat.ModelCfg(cfg.seq_len, cfg.signal_len, cfg.num_layers1, cfg.num_layers2, cfg.num_classes, cfg.hidden_size, cfg.vocab_size, cfg.embedding_size,
                         int(cfg.is_base), int(cfg.is_signallen), mod)
        credited = int(emu.dsp_flops_per_site(ctypes.byref(c))) * 512
        print(kw, "issued %.3f GFLOP in %d fp32 MFMAs over %d launches; credited %.3f GFLOP; ratio %.3f" % (issued / 1e9, out[0], out[2], credited / 1e9, issued / credited))
        assert lo <= issued / credited <= hi, (issued, credited)


# ---- the extraction kernels (csrc/dsp_extract.hip: SURVEY.md 8(f)-3) through the interpreter -------------------------------------

@pytest.fixture(scope="module")
def emu_extract():
    """dsp_extract.hip (its static __shared__ variables as `static`, its workgroups one after another: -DDSP_EMU_STATIC_LDS) +
    the host-side site enumerator, for the host; float64 arithmetic in numpy's evaluation order: -ffp-contract=off as in the product"""
    d = _cache_dir("extract", *SAN)
    out = os.path.join(d, "libdsp_extract_emu.so")
    if not os.path.exists(out):
        stub = os.path.join(d, "err_stub.cpp")
        with open(stub, "w") as f:
            f.write('#include <string>\nstatic std::string g;\nextern "C" void dsp_set_error_(const char* m) { g = m ? m : ""; }\n'
                    'extern "C" const char* dsp_last_error(void) { return g.c_str(); }\n')
        tmp = out + ".tmp%d" % os.getpid()
        cmd = [CLANG, "-std=c++17", "-O2", "-march=native", "-Wno-psabi", "-fPIC", "-shared", "-DDSP_EMU", "-DDSP_EMU_STATIC_LDS", "-ffp-contract=off", "-Wno-unused-value",
               "-I", EMU, "-I", os.path.join(ROOT, "include"), "-I", CSRC, stub, os.path.join(EMU, "hip_emu.cpp"), "-x", "c++", os.path.join(CSRC, "dsp_extract.hip"),
               "-o", tmp, "-pthread"] + SAN
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-4000:]
        os.replace(tmp, out)
    L = ctypes.CDLL(out)
    L.dsp_last_error.restype = ctypes.c_char_p
    return L


def _extract_through_the_interpreter(L, fx, reads, first_read_uid=0):
    """FeatureExtractor.stage + .launch (deepsignal_plant_amd/extract_features.py) on host arrays: the host half is the product's own
    code (_select_reads, _sites = csrc/dsp_sites.cpp), the three kernel entry points are the interpreted ones"""
    from deepsignal_plant_amd.extract_features import ReadBatchC
    uid_of = {id(r): first_read_uid + i for i, r in enumerate(reads)}
    sel, lo, hi = fx._select_reads(reads)
    R = len(sel)
    i64 = lambda xs: np.concatenate([[0], np.cumsum(xs)]).astype(np.int64)
    raw_off, ev_off = i64([r.raw.shape[0] for r in sel]), i64([r.ev_base.shape[0] for r in sel])
    cat = lambda xs, dt: np.ascontiguousarray(np.concatenate(xs) if xs else np.zeros(0), dtype=dt)
    ev_base = cat([r.ev_base for r in sel], np.uint8)
    site_read, site_loc, info, row_off, info_len, _, _ = fx._sites(sel, ev_base, ev_off, lo, hi) if R else (np.zeros(0, np.int32),) * 2 + (None,) * 5
    n, E = int(site_read.shape[0]), int(ev_off[-1])
    raw = cat([r.raw for r in sel], np.int16)
    scaling, offset = np.array([r.scaling for r in sel], np.float64), np.array([r.offset for r in sel], np.float64)
    ev_start, ev_len = cat([r.ev_start for r in sel], np.int64), cat([r.ev_len for r in sel], np.int64)
    uid = np.array([uid_of[id(r)] for r in sel], np.uint64)
    site_read, site_loc = np.ascontiguousarray(site_read, np.int32), np.ascontiguousarray(site_loc, np.int32)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    batch = ReadBatchC(R, int(raw_off[-1]), E, raw.ctypes.data, raw_off.ctypes.data, scaling.ctypes.data, offset.ctypes.data, ev_start.ctypes.data, ev_len.ctypes.data,
                       ev_base.ctypes.data, ev_off.ctypes.data)
    shift, scale = np.empty(R), np.empty(R)
    base_mean, base_std = np.empty(E), np.empty(E)
    base_len, base_lo, blk_off = np.empty(E, np.int32), np.empty(E, np.int64), np.empty(R + 1, np.int64)
    out = dict(kmer=np.empty((n, fx.L), np.uint8), means=np.empty((n, fx.L), np.float32), stds=np.empty((n, fx.L), np.float32), lens=np.empty((n, fx.L), np.int32),
               signals=np.empty((n, fx.L, fx.S), np.float32))
    ok = lambda rc: (_ for _ in ()).throw(AssertionError(L.dsp_last_error())) if rc else None
    ok(L.dsp_extract_normalize(None, ctypes.byref(batch), fx.method, p(shift), p(scale)))
    ok(L.dsp_extract_base_stats(None, ctypes.byref(batch), p(shift), p(scale), p(blk_off), p(base_mean), p(base_std), p(base_len), p(base_lo)))
    if n:
        ok(L.dsp_extract_gather(None, ctypes.byref(batch), p(shift), p(scale), p(base_mean), p(base_std), p(base_len), p(base_lo), ctypes.c_int64(n), p(site_read),
                                p(site_loc), fx.L, fx.S, int(fx.round_stats), ctypes.c_uint64(fx.seed & ((1 << 64) - 1)), p(uid), p(out["kmer"]), p(out["means"]),
                                p(out["stds"]), p(out["lens"]), p(out["signals"])))
    out["sampleinfo"] = [bytes(info[int(row_off[i]):int(row_off[i]) + int(info_len[i])]).decode() for i in range(n)]
    return out


def test_the_extraction_kernels_match_the_oracle_bit_for_bit(emu_extract):
    """tests/test_gpu_extract.py's first test on the host: read statistics (MAD by code histograms / radix select, z-score by
    numpy's pairwise sums), per-base means and stds, the window gather and the seeded subset draw of long bases -- float64 in
    numpy's evaluation order, compared to the last bit with oracle/extract_np.py (itself pinned by F6, the reference's own output)"""
    from deepsignal_plant_amd import extract_features as ef
    from oracle import extract_np as ox
    from tests.test_extract_oracle import CASES, case_inputs
    for c in (CASES if LONG else CASES[:1] + CASES[4:]):
        rs, motif_seqs, chrom2len, region, positions, g = case_inputs(c)
        for round_stats in ((False, True) if LONG else (False,)):
            fx = ef.FeatureExtractor(motifs=c["motifs"], mod_loc=c["mod_loc"], seq_len=c["k"], signal_len=c["s"], normalize_method=c["method"], chrom2len=chrom2len,
                                     positions=positions, region=region, seed=77, round_stats=round_stats)
            got = _extract_through_the_interpreter(emu_extract, fx, rs, first_read_uid=5)
            feats = ox.extract_features(rs, c["method"], motif_seqs, c["mod_loc"], chrom2len, c["k"], c["s"], 1, positions, region, sampler="hash", seed=77, first_read_uid=5)
            want = ox.features_to_arrays(feats, c["k"], c["s"], round_stats=round_stats)
            assert got["sampleinfo"] == want["sampleinfo"] and len(want["sampleinfo"]) == int(g("n_sites")), c["name"]
            for k in ("kmer", "means", "stds", "lens", "signals"):
                a, b = np.asarray(got[k]), np.asarray(want[k])
                assert a.shape == b.shape and a.tobytes() == b.astype(a.dtype).tobytes(), (c["name"], k)


# ---- the call_freq kernels (csrc/dsp_freq_dev.hip: SURVEY.md 8(f)-1) through the interpreter -------------------------------------

def test_the_call_freq_kernels_reproduce_the_references_tables():
    """call_mods_freq.DeviceSiteFrequency's device half on the host: the encode / iota / gather / count / reduce kernels of
    csrc/dsp_freq_dev.hip interpreted (rocPRIM's radix sort of (key, index) as the stable sort it is), the host half -- block
    keys, the site table, the formatter -- the product's own (csrc/dsp_freq.cpp): byte-identical to the reference's call_freq
    outputs (F5: tsv, sorted bedMethyl, prob_cf 0 / 0.2 / 0.5), in one block and in seven"""
    from deepsignal_plant_amd import call_mods_freq as cf
    from tests.helpers import GOLDEN
    from tests.test_call_freq import CALLS, _rows_from_calls
    d = _cache_dir("freq", *SAN)
    out = os.path.join(d, "libdsp_freq_emu.so")
    if not os.path.exists(out):
        stub = os.path.join(d, "err_stub.cpp")
        with open(stub, "w") as f:
            f.write('#include <string>\nstatic std::string g;\nextern "C" void dsp_set_error_(const char* m) { g = m ? m : ""; }\n'
                    'extern "C" const char* dsp_last_error(void) { return g.c_str(); }\n')
        tmp = out + ".tmp%d" % os.getpid()
        cmd = [CLANG, "-std=c++17", "-O2", "-march=native", "-Wno-psabi", "-fPIC", "-shared", "-DDSP_EMU", "-ffp-contract=off", "-Wno-unused-value", "-I", EMU,
               "-I", os.path.join(ROOT, "include"), "-I", CSRC, stub, os.path.join(EMU, "hip_emu.cpp"), "-x", "c++", os.path.join(CSRC, "dsp_freq_dev.hip"), "-o", tmp, "-pthread"] + SAN
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-4000:]
        os.replace(tmp, out)
    K = ctypes.CDLL(out)
    K.dsp_last_error.restype = ctypes.c_char_p
    H = nat.lib()                      # the host half: the product library's own host functions
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    ok = lambda rc: (_ for _ in ()).throw(AssertionError(K.dsp_last_error())) if rc else None
    lines = open(CALLS).read().splitlines()
    r, probs, labels = _rows_from_calls(lines)
    for tag, kw in (("tsv", {}), ("bed_sorted", dict(bed=True, sort=True)), ("tsv_cf0", dict(prob_cf=0.0)), ("bed_cf02", dict(bed=True, prob_cf=0.2))):
        for blocks in (1, 7):
            cfv = kw.get("prob_cf", 0.5)
            agg = cf.SiteFrequency(cfv, nthreads=2)
            keys, packs, piss, rows = [], [], [], []
            cuts = np.linspace(0, r.n, blocks + 1).astype(int)
            for a, b in zip(cuts[:-1], cuts[1:]):      # DeviceSiteFrequency.add_block
                n = int(b - a)
                key, pis, meta = np.empty(n, np.int64), np.empty(n, np.int64), np.empty(n, np.uint32)
                assert H.dsp_freq_block_keys(agg._h, p(r.text), p(r.row_off[a:b]), p(r.info_len[a:b]), p(np.ascontiguousarray(r.kmer[a:b])), 5, n, p(key), p(pis), p(meta)) == n
                key_o, packed = np.empty(n, np.int64), np.empty(n, np.int64)
                pr, lb = np.ascontiguousarray(probs[a:b]), np.ascontiguousarray(labels[a:b])
                ok(K.dsp_freq_dev_encode(None, ctypes.c_int64(n), p(pr), 2, p(lb), p(key), p(meta), ctypes.c_double(cfv), p(key_o), p(packed)))
                keys.append(key_o); packs.append(packed); piss.append(pis); rows.append(np.arange(a, b, dtype=np.int64))
            key, packed, pis, row = (np.concatenate(x) for x in (keys, packs, piss, rows))      # DeviceSiteFrequency.finish, one rank
            live = key != 0x7fffffffffffffff
            key, packed, pis, row = (np.ascontiguousarray(x[live]) for x in (key, packed, pis, row))
            n = int(key.size)
            outs = [np.empty(n, np.int64) for _ in range(4)]
            need = ctypes.c_size_t(0)
            args = [None, ctypes.c_int64(n)] + [p(t) for t in (key, packed, pis, row)] + [p(t) for t in outs]
            ok(K.dsp_freq_dev_sort_records(*args, None, ctypes.byref(need)))
            tmpbuf = np.empty(max(need.value, 1), np.uint8)
            ok(K.dsp_freq_dev_sort_records(*args, p(tmpbuf), ctypes.byref(need)))
            key, packed, pis, row = outs
            cnt = np.zeros(1, np.int64)
            ok(K.dsp_freq_dev_count_sites(None, ctypes.c_int64(n), p(key), p(cnt)))
            ns = int(cnt[0])
            oi = [np.empty(ns, np.int64) for _ in range(6)]
            od = [np.empty(ns, np.float64) for _ in range(2)]
            cnt[0] = 0
            ok(K.dsp_freq_dev_reduce(None, ctypes.c_int64(n), p(key), p(packed), p(pis), p(row), p(cnt), ctypes.c_int64(ns), p(oi[0]), p(oi[1]), p(oi[2]), p(oi[3]),
                                     p(od[0]), p(od[1]), p(oi[4]), p(oi[5])))
            order = np.argsort(oi[1], kind="stable")
            cols = [np.ascontiguousarray(c[order]) for c in (oi[0], oi[1], oi[2], oi[3], od[0], od[1], oi[4], oi[5])]
            table = cf.SiteFrequency(cfv)
            for i in range(H.dsp_freq_chrom_count(agg._h)):
                k = int(H.dsp_freq_chrom_name(agg._h, i, None, 0))
                buf = ctypes.create_string_buffer(max(k, 1))
                H.dsp_freq_chrom_name(agg._h, i, buf, k)
                assert H.dsp_freq_intern_chrom(table._h, buf.raw[:k], k) == i
            assert H.dsp_freq_add_sites(table._h, ns, *[p(c) for c in cols]) == ns
            H.dsp_freq_add_counts(table._h, r.n)
            assert table.format(kw.get("sort", False), kw.get("bed", False)) == open(os.path.join(GOLDEN, "f5_freq_%s.txt" % tag), "rb").read(), (tag, blocks)


@pytest.mark.skipif(not LONG or bool(SAN), reason="DSP_EMU_LONG=1: the two tests above again, their interpreter libraries built with ASan + UBSan")
def test_the_extraction_and_call_freq_kernels_under_sanitizers():
    """csrc/dsp_extract.hip and csrc/dsp_freq_dev.hip -- every hand-written kernel file besides the forward's and the parser's,
    which have their own sanitizer runs above -- interpreted under AddressSanitizer + UBSan: the same two tests in a child
    process with the shared ASan runtime preloaded; every "device" array is an exactly sized numpy array or malloc block."""
    import sys
    rt = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        pytest.skip("no shared ASan runtime in this image")
    e = dict(os.environ, DSP_EMU_SANITIZE="1", LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=0",
             UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-p", "no:cacheprovider", "-k",
                        "extraction_kernels_match or call_freq_kernels_reproduce"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=3000)
    print(r.stdout[-1500:])
    assert r.returncode == 0 and "2 passed" in r.stdout, (r.stdout[-3000:], r.stderr[-6000:])
    assert "AddressSanitizer" not in r.stdout + r.stderr and "runtime error" not in r.stdout + r.stderr, (r.stdout + r.stderr)[-6000:]
