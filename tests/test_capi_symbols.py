"""CPU: the C-ABI library loads and exports every symbol include/dsp_amd.h declares (no compute calls),
and its metadata entry points agree with the oracle's spec."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "dsp_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dsp_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from deepsignal_plant_amd import _native
    lib = _native.lib()
    syms = declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), "libdsp_amd.so does not export %s" % s
    assert lib.dsp_abi_version() == 3


def test_weight_spec_and_flops_match_oracle_spec():
    from deepsignal_plant_amd.models import ModelBiLSTM
    from oracle import forward_np as onp
    for kw in (dict(), dict(module="seq_bilstm", num_layers1=2), dict(module="signal_bilstm", hidden_size=100),
               dict(hidden_size=64, num_layers2=2, is_base=False), dict(is_signallen=False, num_classes=3)):
        cfg = onp.OracleConfig(**kw)
        m = ModelBiLSTM(cfg.seq_len, cfg.signal_len, cfg.num_layers1, cfg.num_layers2, cfg.num_classes, 0,
                        cfg.hidden_size, cfg.vocab_size, cfg.embedding_size, cfg.is_base, cfg.is_signallen,
                        module=cfg.module)
        assert [(k, tuple(s)) for k, s in m._spec] == [(k, tuple(s)) for k, s in onp.state_dict_spec(cfg)]
        assert m.flops_per_site() == onp.flops_per_site(cfg)


def test_no_cpu_fallback_and_reference_error_behaviour():
    import torch
    from deepsignal_plant_amd.models import ModelBiLSTM
    with pytest.raises(ValueError, match="--model_type is not right!"):
        ModelBiLSTM(module="cnn")
    m = ModelBiLSTM()
    sd = m.state_dict()
    assert len(sd) == 49
    del sd["fc2.bias"]
    with pytest.raises(RuntimeError, match="Missing key"):
        m.load_state_dict(sd)
    sd = m.state_dict()
    sd["extra.weight"] = torch.zeros(1)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        m.load_state_dict(sd)
    if not torch.cuda.is_available():
        # without a GPU the product path must fail loudly, never compute on the CPU
        with pytest.raises(RuntimeError):
            m.cuda(0)
        with pytest.raises(RuntimeError):
            m(torch.zeros(2, 13), torch.zeros(2, 13), torch.zeros(2, 13), torch.zeros(2, 13), torch.zeros(2, 13, 16))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "deepsignal_plant_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("# oracle-free", ""), "%s mentions the oracle" % f


def test_a_missing_library_is_an_error_not_a_fallback(monkeypatch, tmp_path):
    """no libdsp_amd.so: every entry of the product path raises and says how to build it -- nothing computes on the CPU"""
    from deepsignal_plant_amd import _native as nat
    monkeypatch.setattr(nat, "_lib", None)
    monkeypatch.setattr(nat, "LIB_PATH", str(tmp_path / "libdsp_amd.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nat.lib()
    from deepsignal_plant_amd import textio
    from deepsignal_plant_amd.models import ModelBiLSTM
    with pytest.raises(RuntimeError, match="libdsp_amd.so not found"):
        ModelBiLSTM()
    with pytest.raises(RuntimeError, match="libdsp_amd.so not found"):
        textio.parse_rows(b"x\n", 13, 16)


def test_init_state_file_is_validated_on_the_host(tmp_path):
    """--init_state file:<npz> (round 4): explicit initial states of every input row in init_hidden's layout
    (models.py:169-176).  Shapes, missing arrays and row counts are checked before anything runs (no GPU needed)."""
    import argparse
    import numpy as np
    from deepsignal_plant_amd import call_modifications as cm
    from deepsignal_plant_amd.models import ModelBiLSTM
    assert cm.init_state_mode(argparse.Namespace(init_state="randn")) == ("randn", None)
    assert cm.init_state_mode(argparse.Namespace(init_state="file:/x/y.npz")) == ("file", "/x/y.npz")
    with pytest.raises(ValueError):
        cm.init_state_mode(argparse.Namespace(init_state="random"))
    model = ModelBiLSTM()    # both_bilstm, hid 256, 3 + 1 layers
    golden = os.path.join(ROOT, "tests", "golden", "f1_randn_capture.npz")
    st = cm.FileInitStates(golden, model)          # the fixtures' `state_` prefix is accepted
    assert st.rows == 4 and sorted(st.arrays) == ["c_comb", "c_seq", "c_sig", "h_comb", "h_seq", "h_sig"]
    part = st.for_rows(1, 2)
    assert tuple(part["h_comb"].shape) == (6, 2, 256) and tuple(part["c_seq"].shape) == (2, 2, 128)
    assert np.array_equal(part["h_sig"].numpy(), np.load(golden)["state_h_sig"][:, 1:3])
    with pytest.raises(ValueError, match="holds the states of 4 rows"):
        st.for_rows(3, 2)
    d = {k: np.zeros(s, np.float32) for k, s in (("h_seq", (2, 5, 128)), ("c_seq", (2, 5, 128)), ("h_sig", (2, 5, 128)),
                                                  ("c_sig", (2, 5, 128)), ("h_comb", (6, 5, 256)), ("c_comb", (6, 5, 256)))}
    p = str(tmp_path / "s.npz")
    np.savez(p, **d)
    assert cm.FileInitStates(p, model).rows == 5
    np.savez(p, **dict(d, h_comb=np.zeros((4, 5, 256), np.float32)))
    with pytest.raises(ValueError, match="the model needs"):
        cm.FileInitStates(p, model)
    np.savez(p, **dict(d, c_sig=np.zeros((2, 6, 128), np.float32)))
    with pytest.raises(ValueError, match="holds 6 rows"):
        cm.FileInitStates(p, model)
    del d["c_comb"]
    np.savez(p, **d)
    with pytest.raises(ValueError, match="holds no array c_comb"):
        cm.FileInitStates(p, model)
    seq_only = ModelBiLSTM(module="seq_bilstm", num_layers1=2)
    np.savez(p, h_seq=np.zeros((2, 3, 256), np.float32), c_seq=np.zeros((2, 3, 256), np.float32),
             h_comb=np.zeros((4, 3, 256), np.float32), c_comb=np.zeros((4, 3, 256), np.float32))
    assert sorted(cm.FileInitStates(p, seq_only).arrays) == ["c_comb", "c_seq", "h_comb", "h_seq"]
