"""CPU: F8, the checkpoint TRAINED by the reference (tests/golden/make_golden_trained.py).

The published checkpoints are not in the reference tree, so parity on trained weights was pinned nowhere; this fixture is
a model the reference's own `train` command optimised (train.py:22-186) on a synthetic task it can learn, saved by its
own torch.save, plus the reference model's outputs on 400 rows.  Here: the checkpoint is what it claims to be (the 49
keys of the boundary, weights that have left their initial range, a caller that calls), and both CPU restatements
reproduce the reference's outputs from the rows as THIS build's parser reads them."""
import numpy as np
import pytest

from tests.helpers import have_f8, load_f8


@pytest.fixture(scope="module", params=[128, 256], ids=["hid128", "hid256_default_architecture"])
def f8(request):
    if not have_f8(request.param):
        pytest.skip("tests/golden/ holds no hid_rnn %d checkpoint (make_golden_trained.py --hid_rnn %d)" % (request.param, request.param))
    return load_f8(request.param)


def test_the_checkpoint_is_a_trained_one(f8):
    from oracle import forward_np as onp
    cfg, w = f8["cfg"], f8["w"]
    assert [k for k in w] == [k for k, _ in onp.state_dict_spec(cfg)]
    for k, shape in onp.state_dict_spec(cfg):
        assert w[k].shape == tuple(shape), k
    bound = 1.0 / np.sqrt(cfg.hidden_size)          # torch's initial range of every lstm_comb tensor
    moved = [k for k in w if k.startswith("lstm_comb") and np.abs(w[k]).max() > 1.5 * bound]
    assert len(moved) >= 12, moved                    # the optimiser has pushed weights well outside it
    acc = float(f8["raw"]["accuracy"])
    assert acc >= 0.85                                # ... and the model calls the task it was trained on
    assert np.array_equal(f8["row_labels"], f8["labels"])
    p1 = f8["probs0"][:, 1]
    assert float(((p1 > 0.5).astype(int) == f8["labels"]).mean()) == pytest.approx(acc)
    assert p1.min() < 0.05 and p1.max() > 0.95        # confident calls on both sides, like a trained caller's


@pytest.mark.parametrize("which", ["pinned_states", "zero_states"])
def test_oracles_reproduce_the_reference_on_the_trained_checkpoint(f8, which):
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    cfg, w, ins = f8["cfg"], f8["w"], f8["inputs"]
    st = f8["states"] if which == "pinned_states" else {k: np.zeros_like(v) for k, v in f8["states"].items()}
    want_l, want_p = (f8["logits"], f8["probs"]) if which == "pinned_states" else (f8["logits0"], f8["probs0"])
    tol = max(1e-6, 2.5 * f8["noise"])                # another fp32 summation order may sit on the other side of float64
    lo, po = onp.forward(cfg, w, *ins, st, dtype=np.float32)
    assert np.abs(po - want_p).max() <= tol
    lc, pc = oc.forward(cfg, w, *ins, states=st)
    print("F8 %s: numpy fp32 %.2e, C oracle %.2e (reference fp32 vs float64: %.2e)" % (
        which, np.abs(po - want_p).max(), np.abs(pc - want_p).max(), f8["noise"]))
    assert np.abs(pc - want_p).max() <= tol
    assert np.abs(lc - want_l).max() <= 50 * tol
    l64, p64 = onp.forward(cfg, w, *ins, st, dtype=np.float64)
    assert np.abs(p64 - want_p).max() <= max(1e-6, 1.01 * f8["noise"]) or which == "zero_states"
