"""VGG-19 multi-exit family: oracle and model mirror against reference-produced
golden vectors on CPU; the HIP path against the same vectors under -m gpu."""
import numpy as np
import pytest
import torch

from bayesnn_fpga_amd.engine import CompiledGraph
from bayesnn_fpga_amd.models import get_network
from bayesnn_fpga_amd.models.vgg19 import vgg19 as bvgg
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_
from oracle import mcd
from oracle import vgg19 as ovgg
from tests.helpers import build_seeded, golden_kwargs, load_golden, state_checksum

pytestmark = pytest.mark.usefixtures("fp16_engine_default")      # (tests/conftest.py: these tests pin the fp16 kernels)

CASES = ["exit_mc", "exit_mask4"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_vgg_matches_reference(name):
    g = load_golden(f"vgg19_{name}.npz")
    kw = golden_kwargs(g)
    m = build_seeded(ovgg.VGG19MCEarlyExit, kw)
    assert state_checksum(m.state_dict()) == str(g["init_checksum"])
    synthetic_weights_(m, 0)
    assert state_checksum(m.state_dict()) == str(g["weights_checksum"])
    x = synthetic_images(int(g["B"]), seed=1234)
    np.testing.assert_allclose(mcd.mcd_passes(m, x, int(g["T"]), int(g["seed"]))[0], g["logits"], atol=1e-6)


def test_oracle_vgg_single_exit_and_broken_modes():
    g = load_golden("vgg19mc_exit.npz")
    m = build_seeded(ovgg.VGG19MC, dict(dropout_exit=True, dropout_p=0.5, out_dim=10))
    assert state_checksum(m.state_dict()) == str(g["init_checksum"])
    synthetic_weights_(m, 0)
    x = synthetic_images(int(g["B"]), seed=1234)
    np.testing.assert_allclose(mcd.mcd_passes(m, x, int(g["T"]), int(g["seed"]))[0], g["logits"], atol=1e-6)
    assert list(load_golden("vgg19_broken_modes.npz")["errors"]) == ["AttributeError"] * 4
    for mod in (ovgg, bvgg):
        for cls in (mod.VGG19MC, mod.VGG19MCEarlyExit):
            for mode in ("block", "layer"):
                with pytest.raises(AttributeError):
                    cls(dropout=mode, dropout_exit=True, out_dim=10)


@pytest.mark.parametrize("name", CASES)
def test_mirror_vgg_init_keys_and_graph(name):
    g = load_golden(f"vgg19_{name}.npz")
    kw = golden_kwargs(g)
    m = build_seeded(bvgg.VGG19MCEarlyExit, kw)
    assert state_checksum(m.state_dict()) == str(g["init_checksum"])
    synthetic_weights_(m, 0)
    assert state_checksum(m.state_dict()) == str(g["weights_checksum"])
    cg = CompiledGraph(m, "cpu", 8, 2)
    assert cg.n_exits == 5 and cg.n_suffix_ops == 5            # exit-only dropout: everything but the 5 heads is prefix
    if kw["out_dim"] == 100:
        assert cg.prefix_macs + cg.suffix_macs == 426698752
    with pytest.raises(RuntimeError, match="no CPU"):
        m(torch.zeros(1, 3, 32, 32))


def test_vgg_factory():
    hp = dict(call="VGG19", resnet_type="mc_early_exit", load_model=None, out_dim=100, image_size=32, dropout=None,
              dropout_exit=True, dropout_p=0.25, n_exits=5, mask_type="mc", num_masks=4, mask_scale=4.0)
    assert type(get_network(hp)) is bvgg.VGG19MCEarlyExit
    assert type(get_network(dict(hp, resnet_type="mc", n_exits=1))) is bvgg.VGG19MC
    assert type(get_network(dict(hp, resnet_type="early_exit"))) is bvgg.VGG19EarlyExit
    assert type(get_network(dict(hp, resnet_type=None))) is bvgg.VGG19
    with pytest.raises(ValueError):
        get_network(dict(hp, resnet_type="bogus"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_vgg_against_reference_golden(name):
    g = load_golden(f"vgg19_{name}.npz")
    kw = golden_kwargs(g)
    B, T, seed = int(g["B"]), int(g["T"]), int(g["seed"])
    m = build_seeded(bvgg.VGG19MCEarlyExit, kw)
    synthetic_weights_(m, 0)
    m = m.to("cuda:0").eval()
    m.mc_seed = seed
    x = synthetic_images(B, seed=1234).to("cuda:0")
    passes = np.stack([np.stack([o.cpu().numpy() for o in m(x)]) for _ in range(T)])
    np.testing.assert_allclose(passes, g["logits"], rtol=0, atol=2e-2)
    ref_probs = torch.softmax(torch.from_numpy(g["logits"]), -1).numpy().astype(np.float64)
    r = m.engine(x.device, max_batch=B).predict(x, T, seed=seed)
    np.testing.assert_allclose(r["mean"].cpu().numpy(), ref_probs.mean(0), rtol=0, atol=1e-3)
    np.testing.assert_allclose(r["var"].cpu().numpy(), ref_probs.var(0), rtol=0, atol=1e-3)


@pytest.mark.gpu
def test_gpu_vgg_single_exit():
    g = load_golden("vgg19mc_exit.npz")
    m = build_seeded(bvgg.VGG19MC, dict(dropout_exit=True, dropout_p=0.5, out_dim=10))
    synthetic_weights_(m, 0)
    m = m.to("cuda:0").eval()
    m.mc_seed = int(g["seed"])
    x = synthetic_images(int(g["B"]), seed=1234).to("cuda:0")
    passes = np.stack([np.stack([o.cpu().numpy() for o in m(x)]) for _ in range(int(g["T"]))])
    np.testing.assert_allclose(passes, g["logits"], rtol=0, atol=2e-2)


@pytest.mark.gpu
def test_gpu_full_analysis_collates_every_exit_the_vgg_returns():
    """``VGG19MCEarlyExit`` built with the constructor's default ``n_exits`` (4) still returns FIVE logits tensors (SA/models/vgg19/vgg19.py:327-382; the
    reference's entry point passes n_exits = 5 for VGG, SA/train/hyperparameters.py:94-98): the FullAnalysis mirror sizes by what the forward returns
    — round 6: `tools/loop_bench.py --workload vgg19_me` crashed on a [4, ...] array for five exits."""
    from bayesnn_fpga_amd.synthetic import synthetic_labels
    from bayesnn_fpga_amd.train.results_analyzer import FullAnalysis
    m = synthetic_weights_(build_seeded(bvgg.VGG19MCEarlyExit, dict(dropout_exit=True, dropout=None, dropout_p=0.25, out_dim=10)), 0).to("cuda:0").eval()
    x, y = synthetic_images(24, seed=3), synthetic_labels(24, 10, seed=4)
    loader = [(x[:12], y[:12]), (x[12:], y[12:])]
    fa = FullAnalysis(m, loader, gpu=0, mc_dropout=True, mc_passes=4, ece="hist")
    assert fa.preds.shape == (5, 24, 10) and fa.ensemble_preds.shape == (5, 24, 10)
    assert np.allclose(fa.preds.sum(-1), 1.0, atol=1e-6) and sorted(fa.layer_correct) == [0, 1, 2, 3, 4]
    rows = fa.all_experiments("v", write=False)
    assert [r[0] for r in rows] == ["0", "1", "2", "3", "4"] + [f"Ensemble{i}" for i in range(5)]
