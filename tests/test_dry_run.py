"""CPU: DRY RUNS of dsp_forward (round 6).  GPU access was closed for the whole round, and the round changed the host half of
every launch: buffer descriptors now carry real extents (a pointer without the end of its allocation, or a launch reaching
past it, is REFUSED by the launch wrappers), the cut of a call into pieces is planned from the model, the launch geometry was
factored into shape_lstm / pick_form.  dsp_debug_dry_run runs that host half -- the very code of dsp_forward: workspace layout,
piece plan, every launch's arguments, the wrappers' checks and grid / block / LDS arithmetic -- on a handle whose allocations
are made-up addresses, and NOTES each launch instead of making it.  What is held here:
  * no launch of any model shape / batch size / precision / initial-state mode / extents mode is refused;
  * which kernel form a batch size takes (the GPU suite reads the same off DSP_DEBUG_LSTM: test_small_batches_really_take_...);
  * residency of the clustered launches: never more workgroups than compute units, the two front-end branches together too;
  * the launch sequence of a cut call is the concatenation of its pieces' sequences."""
import ctypes
import re

import pytest

from deepsignal_plant_amd import _native as nat

FP32, BF16X6, BF16X9, FP16X3 = 0, 6, 9, 3
ZEROS, EXPLICIT, PHILOX = 0, 1, 2


def cfg(**kw):
    d = dict(seq_len=13, signal_len=16, num_layers1=3, num_layers2=1, num_classes=2, hidden_size=256, vocab_size=16,
             embedding_size=4, is_base=1, is_signallen=1, module=0)
    d.update(kw)
    return nat.ModelCfg(*[d[k] for k, _ in nat.ModelCfg._fields_])


def dry(c, n, cus=256, init=PHILOX, prec=FP32, extents=b"region"):
    buf = ctypes.create_string_buffer(1 << 17)
    k = int(nat.lib().dsp_debug_dry_run(ctypes.byref(c), cus, n, init, prec, extents, buf, 1 << 17))
    assert k >= 0, (n, cus, init, prec, extents, nat.last_error())
    lines = buf.value.decode().splitlines()
    assert len(lines) == k
    out = []
    for ln in lines:
        m = re.match(r"^\(?(\w+)(?:<([^>]*)>)?\)? grid (\d+),(\d+) block (\d+) lds (\d+)$", ln)
        assert m, ln
        out.append((m.group(1), tuple(x.strip() for x in (m.group(2) or "").split(",")) if m.group(2) else (), int(m.group(3)) * int(m.group(4)),
                    int(m.group(5)), int(m.group(6))))
    return out


CASES = [   # tools/extents_sweep.py's shapes (the GPU twin of this test)
    ("default", dict(), (1, 33, 300, 512, 1024, 1100, 2048, 3000, 4096, 4097, 9001, 16500, 65536, 300000), (FP32, BF16X9, FP16X3)),
    ("cfg3_seq_only", dict(module=1, num_layers1=2), (31, 512, 1024, 2048, 4097, 9001), (FP32, BF16X6)),
    ("signal_only", dict(module=2, num_layers1=1), (33, 513, 4100), (FP32,)),
    ("hid128", dict(hidden_size=128, num_layers1=2, num_layers2=2), (1, 512, 1024, 2048, 5000), (FP32, BF16X9)),
    ("hid200_padded", dict(hidden_size=200, num_layers1=2), (65, 1025, 4097), (FP32,)),
    ("hid100_ut4_padded", dict(hidden_size=100), (64, 700), (FP32,)),
    ("hid64_ut2", dict(hidden_size=64, num_layers2=2, is_base=0), (33, 2000), (FP32,)),
    ("hid320_many_pass", dict(hidden_size=320, num_layers1=2), (40, 600, 4200), (FP32,)),
    ("hid640_three_classes", dict(hidden_size=640, num_layers1=1, num_classes=3, is_signallen=0), (50, 300), (FP32,)),
    ("hid2048", dict(hidden_size=2048, num_layers1=1), (70,), (FP32,)),
    ("k9_s24", dict(seq_len=9, signal_len=24, hidden_size=96), (100, 1500), (FP32,)),
    ("s40_wide_window", dict(signal_len=40, hidden_size=256, num_layers1=1), (90, 600), (FP32,)),
]


@pytest.mark.parametrize("label,kw,sizes,precisions", CASES, ids=[c[0] for c in CASES])
def test_no_launch_is_refused(label, kw, sizes, precisions):
    c = cfg(**kw)
    for prec in precisions:
        for n in sizes:
            launches = dry(c, n, prec=prec)
            assert launches[0][0] == "dsp_pack_kernel" and launches[-1][0] == "dsp_head_kernel", (label, n)
        n = sizes[min(1, len(sizes) - 1)]
        for init in (ZEROS, EXPLICIT):
            dry(c, n, init=init, prec=prec)
    for extents in (b"tight", b"wide"):
        for n in sizes:
            dry(c, n, extents=extents)
    for cus in (304, 128, 64):                    # another device: the class sizes move, nothing is refused
        for n in sizes[:4]:
            dry(c, n, cus=cus)


def names(launches):
    return ["%s<%s>" % (k, ", ".join(t)) if t else k for k, t, *_ in launches]


def test_the_default_model_takes_the_forms_the_design_says():
    c = cfg()
    f512 = names(dry(c, 512))
    # front ends clustered (4 workgroups per site tile and direction, dead k-groups 3 / 2), each behind its clean-up launch, the
    # one-tile fc kernel, the combined stack on clusters of 8 CUs (1 gate per wave, rings 16 deep), the one-tile head
    assert f512 == ["dsp_pack_kernel", "dsp_lstmc_kernel<1, 4, false, 3, 4, true>", "dsp_lstmc_kernel<4, 4, true, 3>", "dsp_linear1_kernel",
                    "dsp_lstmc_kernel<1, 4, false, 2, 4, true>", "dsp_lstmc_kernel<4, 4, true, 2>", "dsp_linear1_kernel"] + \
                   ["dsp_lstmc_kernel<1, 16>", "dsp_lstmc_kernel<4, 4, true, 0, 8>"] * 3 + ["dsp_head_kernel<1>"]
    assert names(dry(c, 1024)).count("dsp_lstmc_kernel<2, 8>") == 3
    assert names(dry(c, 2048)).count("dsp_lstmc_kernel<4, 4>") == 3
    f4096 = names(dry(c, 4096))
    assert f4096.count("dsp_lstmc_kernel<4, 4, true, 0, 8>") == 3 and "dsp_linear1_kernel" in f4096 and not any("false" in x for x in f4096)
    f65536 = dry(c, 65536)
    assert names(f65536) == ["dsp_pack_kernel", "dsp_lstm_kernel<2, 1, 1>", "dsp_lstm_kernel<2, 1, 2>", "dsp_linear_kernel",
                             "dsp_lstm_kernel<0, 1>", "dsp_lstm_kernel<0, 1>", "dsp_lstm_kernel<0, 1>", "dsp_head_kernel<4>"]
    assert [g for k, t, g, b, l in f65536 if k == "dsp_lstm_kernel" and t == ("0", "1")] == [2048] * 3   # one 8-wave workgroup per 64 sites and direction
    assert [b for k, t, g, b, l in f65536 if k == "dsp_lstm_kernel" and t == ("0", "1")] == [512] * 3
    # explicit states / split precision: no cut, the split kernels on the combined stack
    assert names(dry(c, 9001, init=EXPLICIT)).count("dsp_pack_kernel") == 1
    assert names(dry(c, 9001, prec=BF16X9)).count("dsp_lstm6_kernel<9>") == 5 and names(dry(c, 9001, prec=FP16X3)).count("dsp_lstm6_kernel<3>") == 3


def test_other_models_small_batch_forms():
    hid128 = cfg(hidden_size=128)
    assert names(dry(hid128, 512)).count("dsp_lstmc_kernel<1, 4>") == 3      # dense layers of 4 unit tiles: clusters of 4
    assert names(dry(hid128, 2048)).count("dsp_lstmc_kernel<2, 4>") == 3
    cfg3 = cfg(module=1, num_layers1=2)
    f = dry(cfg3, 512)
    assert names(f)[1] == "dsp_lstmc_kernel<1, 4, false, 3, 4, true>" and f[1][2] == 256      # the seq front end of 8 unit tiles on clusters of 8
    assert names(dry(cfg3, 2048)).count("dsp_pack_kernel") == 2                                 # planned as 1,024 + 1,024
    big = names(dry(cfg(hidden_size=320, num_layers1=2), 600))
    assert "dsp_lstm_kernel<0, 0>" in big or "dsp_lstm_kernel<1, 0>" in big                      # hidden > 256: the many-pass kernel


@pytest.mark.parametrize("cus", [256, 304, 128])
def test_clustered_launches_fit_the_compute_units(cus):
    """members of a cluster wait for each other: the host may only pick a cluster size whose whole grid is resident at once,
    one workgroup per CU -- and the two front-end branches that run side by side must fit TOGETHER"""
    for kw in (dict(), dict(hidden_size=128), dict(module=1, num_layers1=2), dict(hidden_size=200, num_layers1=2)):
        c = cfg(**kw)
        for n in (1, 300, 512, 600, 1024, 1500, 2048, 2500, 4096):
            launches = dry(c, n, cus=cus)
            clustered = [(k, t, g) for k, t, g, b, l in launches if k == "dsp_lstmc_kernel" and len(t) >= 3 and t[2] == "false" or
                         (k == "dsp_lstmc_kernel" and len(t) == 2)]
            for k, t, g in clustered:
                assert g <= cus, (kw, n, cus, t, g)
            fronts = [g for k, t, g in clustered if len(t) == 6 and t[5] == "true"]
            if len(fronts) == 2 and kw.get("module", 0) == 0:
                assert sum(fronts) <= cus, (kw, n, cus, fronts)


def test_a_cut_call_is_the_concatenation_of_its_pieces():
    c = cfg()
    whole = names(dry(c, 14000))                   # 8,192 + 4,096 + 1,712
    parts = names(dry(c, 8192)) + names(dry(c, 4096)) + names(dry(c, 1712))
    assert whole == parts
    assert names(dry(c, 9000, cus=304)).count("dsp_pack_kernel") == 1      # one round of 9,728 sites there


def test_x_ahead_is_taken_by_small_calls_only_and_never_refused(monkeypatch):
    """DSP_LSTM_XAHEAD=1 (opt-in, round 6): calls of <= 8 live site tiles run dsp_xahead_kernel in front of every clustered dense
    layer of 8 unit tiles -- live clusters x T x 8 workgroups -- and the XA instantiation of the recurrent launch; nothing
    else changes, nothing is refused in any extents mode, and without the switch no launch of the form is ever made"""
    c = cfg()
    plain = {n: names(dry(c, n)) for n in (1, 45, 256, 257, 512, 1100, 9001)}
    assert not any("xahead" in x or x.endswith("false, true>") for f in plain.values() for x in f)
    monkeypatch.setenv("DSP_LSTM_XAHEAD", "1")
    for n in (1, 45, 256):
        f = dry(c, n)
        xa = [(k, t, g, b) for k, t, g, b, l in f if k == "dsp_xahead_kernel"]
        live = (n + 31) // 32 * 2
        assert [g for k, t, g, b in xa] == [live * 13 * 8] * 3 and all(b == 256 for k, t, g, b in xa), (n, xa)
        assert names(f).count("dsp_lstmc_kernel<1, 16, false, 0, 4, false, true>") == 3
        # the rest of the sequence is the plain call's
        assert [x for x in names(f) if x != "dsp_xahead_kernel"] == [x.replace("<1, 16>", "<1, 16, false, 0, 4, false, true>") for x in plain[n]]
    for n in (257, 512):
        assert names(dry(c, n)) == plain[n]
    # a cut call: 1,100 = 1,024 + 76 -- the 76-site piece is a small call of its own
    f = names(dry(c, 1100))
    assert f.count("dsp_pack_kernel") == 2 and f.count("dsp_xahead_kernel") == 3 and f.index("dsp_xahead_kernel") > f.index("dsp_lstmc_kernel<2, 8>")
    monkeypatch.setenv("DSP_LSTM_XAHEAD_TILES", "16")
    assert names(dry(c, 512)).count("dsp_xahead_kernel") == 3
    monkeypatch.setenv("DSP_LSTM_XAHEAD_TILES", "2")
    assert names(dry(c, 64)).count("dsp_xahead_kernel") == 3 and names(dry(c, 65)).count("dsp_xahead_kernel") == 0
    monkeypatch.delenv("DSP_LSTM_XAHEAD_TILES")
    for label, kw, sizes, precisions in CASES:
        cc = cfg(**kw)
        for extents in (b"region", b"tight"):
            for n in (33, 256):
                dry(cc, n, extents=extents)
        dry(cc, 100, init=EXPLICIT)
        dry(cc, 100, cus=64)
    # the seq-only shape (BASELINE configs[2]): both dense layers of its combined stack
    assert names(dry(cfg(module=1, num_layers1=2), 100)).count("dsp_xahead_kernel") == 2
    h128 = names(dry(cfg(hidden_size=128), 100))      # dense layers of 4 unit tiles: clusters of 4, rings 4 deep
    assert h128.count("dsp_xahead_kernel") == 3 and h128.count("dsp_lstmc_kernel<1, 4, false, 0, 4, false, true>") == 3
    assert names(dry(cfg(hidden_size=64), 100)).count("dsp_xahead_kernel") == 0       # 2 unit tiles: no clustered form
    # split precision: its own kernels, no clustered form, no x ahead
    assert names(dry(c, 100, prec=BF16X9)).count("dsp_xahead_kernel") == 0
