"""The oracle (CPU restatement) against golden vectors produced by the reference itself
(tools/gen_golden.py).  fp32 CPU: identical op sequence, so the bar is tight (1e-6)."""
import numpy as np
import pytest
import torch

from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_
from oracle import layers as olayers
from oracle import mcd, metrics
from oracle.resnet18 import ResNet18EarlyExit, ResNet18MC, ResNet18MCEarlyExit
from tests.helpers import build_seeded, golden_kwargs, load_golden, state_checksum

CASES = ["exit_only", "block_exit", "block_noexit", "layer_exit", "mask4_block_exit", "mask8_exit_c100", "block_exit_p02",
         "layer_exit_p256"]


@pytest.mark.parametrize("name", CASES)
def test_resnet18_mc_early_exit(name):
    g = load_golden(f"resnet18_{name}.npz")
    kw = golden_kwargs(g)
    B, T, seed = int(g["B"]), int(g["T"]), int(g["seed"])
    model = build_seeded(ResNet18MCEarlyExit, kw)
    # same torch seed + same construction order => same initial weights and same state_dict keys
    assert state_checksum(model.state_dict()) == str(g["init_checksum"])
    synthetic_weights_(model, 0)
    assert state_checksum(model.state_dict()) == str(g["weights_checksum"])
    for k in g.files:
        if k.startswith("mask__"):
            assert np.array_equal(model.state_dict()[k[6:]].numpy().astype(np.uint8), g[k])
    x = synthetic_images(B, seed=1234)
    r = mcd.mcd_predict(model, x, T, seed)
    np.testing.assert_allclose(r["logits"], g["logits"], rtol=0, atol=1e-6)
    # the reference's own T-loop (_get_output 5-tuple)
    np.testing.assert_allclose(r["logit_mean"], g["go_output"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(r["mean"], g["go_output_sm"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(r["mean"], g["go_output_sm_np"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(r["ensemble_logit_mean"], g["go_ensemble_output"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(r["ensemble_mean"], g["go_ensemble_output_sm"], rtol=0, atol=1e-6)
    # build-defined variance: ddof=0 over the T passes
    np.testing.assert_allclose(r["var"], np.var(r["probs"], axis=0), rtol=0, atol=0)


def test_masksembles_cycle_persists_across_calls():
    """cnt persists across batches (SA/utils.py:166-168): T=10, M=4 -> masks weighted 3/3/2/2."""
    g = load_golden("resnet18_mask4_block_exit.npz")
    model = build_seeded(ResNet18MCEarlyExit, golden_kwargs(g))
    synthetic_weights_(model, 0)
    x = synthetic_images(int(g["B"]), seed=1234)
    a = mcd.mcd_passes(model, x, 10, 42)[0]
    np.testing.assert_allclose(a, g["logits"], atol=1e-6)
    assert model.exit_dropout.cnt == 10 % 4
    b = mcd.mcd_passes(model, x, 2, 42)[0]          # continues from cnt = 2
    np.testing.assert_allclose(b[0], a[2], atol=1e-6)
    np.testing.assert_allclose(a[4], a[0], atol=1e-6)  # period M


def test_resnet18mc_single_exit():
    g = load_golden("resnet18mc_block_exit.npz")
    m = build_seeded(ResNet18MC, dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10))
    assert state_checksum(m.state_dict()) == str(g["init_checksum"])
    synthetic_weights_(m, 0)
    x = synthetic_images(int(g["B"]), seed=1234)
    np.testing.assert_allclose(mcd.mcd_passes(m, x, int(g["T"]), int(g["seed"]))[0], g["logits"], atol=1e-6)


def test_resnet18_early_exit_deterministic():
    g = load_golden("resnet18_early_exit.npz")
    m = build_seeded(ResNet18EarlyExit, dict(out_dim=10))
    assert state_checksum(m.state_dict()) == str(g["init_checksum"])
    synthetic_weights_(m, 0)
    m.eval()
    with torch.no_grad():
        out = np.stack([o.numpy() for o in m(synthetic_images(int(g["B"]), seed=1234))])
    np.testing.assert_allclose(out, g["logits"], atol=1e-6)


def test_layer_mask_mode_is_broken_like_the_reference():
    with pytest.raises(UnboundLocalError):
        ResNet18MCEarlyExit(dropout="layer", mask_type="mask", out_dim=10)


def test_masksembles_layers():
    g = load_golden("masksembles.npz")
    np.random.seed(3)
    m2 = olayers.Masksembles2D(16, 4, 2.0).eval()
    m1 = olayers.Masksembles1D(32, 4, 2.0).eval()
    assert np.array_equal(m2.masks.numpy(), g["masks2"])
    assert np.array_equal(m1.masks.numpy(), g["masks1"])
    x2, x1 = torch.from_numpy(g["x2"]), torch.from_numpy(g["x1"])
    y2 = np.stack([m2(x2).numpy() for _ in range(8)])
    y1 = np.stack([m1(x1).numpy() for _ in range(8)])
    assert np.array_equal(y2, g["y2"]) and np.array_equal(y1, g["y1"])
    # no rescale, one mask for the whole batch, period n
    assert np.array_equal(y1[0], g["x1"] * g["masks1"][0][None])
    assert np.array_equal(y1[4], y1[0])


def test_generation_wrapper_properties():
    import hashlib
    g = load_golden("masksembles.npz")
    for (c, n, s, rows, cols, ones, equal, digest) in g["props"]:
        np.random.seed(11)
        mk = olayers.generation_wrapper(int(c), int(n), float(s))
        assert mk.shape == (rows, cols) == (n, c)
        assert set(np.unique(mk)) <= {0.0, 1.0}
        assert int(mk.sum(1)[0]) == ones and bool((mk.sum(1) == ones).all()) == bool(equal)
        assert hashlib.sha256(mk.astype(np.uint8).tobytes()).hexdigest() == digest
    with pytest.raises(ValueError):
        olayers.generation_wrapper(8, 4, 2.0)
    with pytest.raises(ValueError):
        olayers.generation_wrapper(64, 4, 6.5)


def test_metrics():
    g = load_golden("metrics.npz")
    p, onehot = g["p"], g["onehot"]
    assert metrics.ece_hist_binary(p, onehot) == pytest.approx(float(g["ece_hist"]), abs=1e-7)
    nll, mse, acc = metrics.nll_mse_acc(p, onehot)
    assert nll == pytest.approx(float(g["nll"]), rel=1e-12)
    assert mse == pytest.approx(float(g["mse"]), rel=1e-12)
    assert acc == float(g["acc"])
    logits = [torch.from_numpy(l) for l in g["logits"]]
    y = torch.from_numpy(g["y"])
    np.testing.assert_allclose(metrics.multi_exit_accuracy(logits, y, 4), g["acc_vec4"], atol=1e-7)
    np.testing.assert_allclose(metrics.multi_exit_accuracy(logits, y, 1), g["acc_vec1"], atol=1e-7)
