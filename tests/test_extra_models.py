"""Build-defined BASELINE configs with no PyTorch reference model: LeNet-5 (config 1, CPU
plumbing only), VGG-11-BN with 3 dropout sites (config 2) and ResNet-50 multi-exit (config 5).
PARITY UNPINNED BY THE REFERENCE: there is no reference forward to generate golden vectors from, so these
compare the HIP path against the fp32 CPU restatement in oracle/extra_models.py (same mask convention,
same synthetic weights) — tolerance 1e-3 on mean/variance as BASELINE.json states."""
import numpy as np
import pytest
import torch

from bayesnn_fpga_amd.engine import CompiledGraph
from bayesnn_fpga_amd.models import extra as bx
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_
from oracle import extra_models as ox
from oracle import mcd
from tests.helpers import build_seeded, state_checksum

pytestmark = pytest.mark.usefixtures("fp16_engine_default")      # (tests/conftest.py: these tests pin the fp16 kernels)

PAIRS = [
    (bx.VGG11MC, ox.VGG11MC, dict(num_bayes_layer=3, dropout_p=0.25, out_dim=10)),
    (bx.VGG11MC, ox.VGG11MC, dict(num_bayes_layer=7, dropout_p=0.25, out_dim=10)),
    (bx.ResNet50MCEarlyExit, ox.ResNet50MCEarlyExit, dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)),
    (bx.ResNet50MCEarlyExit, ox.ResNet50MCEarlyExit, dict(dropout_exit=True, dropout=None, dropout_p=0.25, out_dim=100)),
]


@pytest.mark.parametrize("cls,ocls,kw", PAIRS)
def test_mirror_matches_oracle_init_and_compiles(cls, ocls, kw):
    a, b = build_seeded(cls, kw), build_seeded(ocls, kw)
    assert list(a.state_dict()) == list(b.state_dict())
    assert state_checksum(a.state_dict()) == state_checksum(b.state_dict())
    cg = CompiledGraph(a, "cpu", 8, 2)
    assert cg.n_exits == a.n_exits
    with pytest.raises(RuntimeError, match="no CPU"):
        a(torch.zeros(1, 3, 32, 32))


def test_vgg11_site_counter_rule():
    """models.py:211-287 — sites sit at locations >= 7 - num_bayes_layer of (pool1..4, flatten, dense1, dense2)."""
    for nb, sites in ((0, 0), (1, 1), (3, 3), (7, 7)):
        m = ox.VGG11MC(num_bayes_layer=nb)
        assert sum(isinstance(x, torch.nn.Dropout) for x in m.modules()) == sites
    cg = CompiledGraph(build_seeded(bx.VGG11MC, dict(num_bayes_layer=3)), "cpu", 8, 2)
    assert cg.n_suffix_ops == 4          # mask(flatten), dense1+site, dense2+site, head — all 8 convs are prefix
    assert cg.prefix_macs == 152764416 - 0 and cg.suffix_macs == 512 * 512 * 2 + 512 * 10


def test_lenet5_mcd_plumbing_on_cpu():
    """BASELINE config 1: MC-dropout LeNet-5 on 28x28x1, T=10 — runs on the oracle only (no GPU config)."""
    torch.manual_seed(0)
    m = ox.LeNet5MC(dropout_p=0.25)
    x = torch.randn(32, 1, 28, 28)
    r = mcd.mcd_predict(m, x, 10, seed=5)
    assert r["mean"].shape == (1, 32, 10) and r["var"].shape == (1, 32, 10)
    np.testing.assert_allclose(r["mean"].sum(-1), 1.0, atol=1e-6)
    assert r["var"].max() > 0                                   # dropout actually samples
    r2 = mcd.mcd_predict(m, x, 10, seed=5)
    np.testing.assert_array_equal(r["probs"], r2["probs"])      # counter-based masks: reproducible
    a = mcd.mcd_passes(m, x, 4, seed=5, t_begin=6)[1]
    np.testing.assert_array_equal(a, r["probs"][6:10])          # and addressable by sample index


@pytest.mark.gpu
@pytest.mark.parametrize("cls,ocls,kw", PAIRS)
def test_gpu_extra_model_against_oracle(cls, ocls, kw):
    B, T, seed = 6, 5, 77
    m, o = build_seeded(cls, kw), build_seeded(ocls, kw)
    synthetic_weights_(m, 0)
    synthetic_weights_(o, 0)
    x = synthetic_images(B, seed=1234)
    ref = mcd.mcd_predict(o, x, T, seed)
    m = m.to("cuda:0").eval()
    m.mc_seed = seed
    xd = x.to("cuda:0")
    passes = np.stack([np.stack([t.cpu().numpy() for t in m(xd)]) for _ in range(T)])
    scale = max(1.0, float(np.abs(ref["logits"]).max()))
    np.testing.assert_allclose(passes, ref["logits"], rtol=0, atol=5e-3 * scale)     # fp16 activations vs fp32 oracle
    r = m.engine(xd.device, max_batch=B).predict(xd, T, seed=seed)
    np.testing.assert_allclose(r["mean"].cpu().numpy(), ref["mean"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(r["var"].cpu().numpy(), ref["var"], rtol=0, atol=1e-3)


@pytest.mark.gpu
def test_gpu_splitk_prefix_convs_match_unsplit():
    """bmi_plan gives the skinny deterministic 3x3 convs (VGG-11's 256+ channel convs on 8x8 ... 2x2 maps at a small batch: a few
    tiles on 256 CUs) a split-K launch: one workgroup per (tile, tap), fp32 partial sums in workspace scratch, a finishing
    pass.  Same function as the unsplit launch up to the fp32 summation order over K."""
    from bayesnn_fpga_amd import _lib
    B, T, seed = 6, 4, 3
    x = synthetic_images(B, seed=7).to("cuda:0")
    out, ws = {}, {}
    for sk in (0, 1):
        _lib.set_option("splitk", sk)
        try:
            m = synthetic_weights_(build_seeded(bx.VGG11MC, dict(num_bayes_layer=3, dropout_p=0.25, out_dim=10)), 0).to("cuda:0").eval()
            eng = m.engine(x.device, max_batch=B)
            ws[sk] = eng.workspace_bytes
            out[sk] = eng.predict(x, T, seed=seed)["mean"].cpu().numpy()
        finally:
            _lib.set_option("splitk", 1)
    assert ws[1] > ws[0]                                  # the partial-sum scratch was planned, i.e. the split path ran
    np.testing.assert_allclose(out[1], out[0], rtol=0, atol=5e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("dt,B", [("f16x2", 6), ("bf16x3", 250)])
def test_gpu_split_engines_splitk_and_narrow_tiles_match(dt, B):
    """The split engines on VGG-11's B-image prefix (a few dozen 256-channel tiles on 256 CUs): conv_split narrows its channel tile on
    small grids ("split_tile": the same K order per accumulator, so the same bits) and bmi_plan gives the 3x3 convs with 256+ input
    channels contiguous K ranges per tile (split-K: raw fp32 sums in workspace scratch, a finishing pass that encodes pair32; the same
    function up to the fp32 summation order over K)."""
    from bayesnn_fpga_amd import _lib
    T, seed = 3, 3
    x = synthetic_images(B, seed=7).to("cuda:0")
    out, ws = {}, {}
    for arm, (sk, tile) in {"both": (1, 1), "wide": (1, 0), "unsplit": (0, 1), "neither": (0, 0)}.items():
        _lib.set_option("splitk", sk)
        _lib.set_option("split_tile", tile)
        try:
            m = synthetic_weights_(build_seeded(bx.VGG11MC, dict(num_bayes_layer=3, dropout_p=0.25, out_dim=10)), 0).to("cuda:0").eval()
            eng = m.engine(x.device, max_batch=B, dtype=dt)
            ws[arm] = eng.workspace_bytes
            out[arm] = eng.predict(x, T, seed=seed)["mean"].cpu().numpy()
        finally:
            _lib.set_option("splitk", 1)
            _lib.set_option("split_tile", 1)
    assert ws["both"] > ws["unsplit"]                    # the partial-sum scratch was planned, i.e. the split path ran
    assert np.array_equal(out["unsplit"], out["neither"]) and np.array_equal(out["both"], out["wide"])      # tile width: the same bits
    np.testing.assert_allclose(out["both"], out["unsplit"], rtol=0, atol=2e-6 if dt == "f16x2" else 2e-5)
