"""-m gpu: confidence-threshold early exiting ON the device (bmi_forward_mcd_exit / MCDEngine.predict_with_exit) against
the reference's post-hoc rule applied to a FULL run (train/confidence_exiting.py, itself pinned to
FullAnalysis.confidence_exiting / is_confident by tests/golden/confidence_exiting.npz): same exit per image, and the
prediction of every image at its exit equals the full run's — the compacted stages compute the same values bit for bit
(same masks: the Philox element index keeps the ORIGINAL image index; tile composition does not enter an MFMA result)."""
import numpy as np
import pytest
import torch

from bayesnn_fpga_amd.models.resnet18.resnet18 import ResNet18MCEarlyExit
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_
from bayesnn_fpga_amd.train import confidence_exiting as cex
from tests.helpers import build_seeded

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("fp16_engine_default")]
DEV = "cuda:0"

KWS = {
    "mc_block_exit": dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10),
    "mc_layer_exit": dict(dropout_exit=True, dropout="layer", dropout_p=0.25, out_dim=10),
    "masksembles": dict(dropout_exit=True, dropout="block", mask_type="mask", num_masks=4, mask_scale=4.0, out_dim=10),
}


@pytest.mark.parametrize("dt", ["f16", "bf16", "f16x2", "bf16x3"])
@pytest.mark.parametrize("name", sorted(KWS))
def test_device_exit_equals_posthoc_rule_on_the_full_run(name, dt):
    """Every engine that runs the path at speed — the 16-bit ones and, round 6, the split engines (conv_split's IMAP instantiations: the
    engine ``engine_dtype="auto"`` falls back to must not lose the feature) — against the reference's post-hoc rule on ITS OWN full run."""
    B, T, seed = 45, 6, 11
    m = build_seeded(ResNet18MCEarlyExit, KWS[name])
    synthetic_weights_(m, 0)
    eng = m.to(DEV).eval().engine(torch.device(DEV), max_batch=B, dtype=dt)
    x = synthetic_images(B, seed=21).to(DEV)
    full = eng.predict(x, T, seed=seed)
    p_full = full["mean"].cpu().numpy()
    conf = p_full.max(-1)                                        # [E, B]
    for thr in (float(np.median(conf[1])), float(np.quantile(conf[2], 0.3)), 0.0, 1.0):
        want = cex.exit_layer(p_full.copy(), thr)
        r = eng.predict_with_exit(x, T, thr, seed=seed)
        got = r["exit_layer"].cpu().numpy()
        np.testing.assert_array_equal(got, want)
        best = r["best_preds"].cpu().numpy()
        np.testing.assert_allclose(best, p_full[want, np.arange(B)], rtol=0, atol=1e-13)
        np.testing.assert_allclose(best.sum(-1), 1.0, atol=1e-6)
        # the stages really ran on fewer images: an image that left at exit e has no samples in the later exits
        mean = r["mean"].cpu().numpy()
        for e in range(1, 4):
            gone = got < e
            assert np.all(mean[e][gone] == 0.0)
            np.testing.assert_allclose(mean[e][~gone], p_full[e][~gone], rtol=0, atol=1e-13)
        act = r["active_after"]
        assert act[0] == B and act[1] == int((got > 1).sum()) and (act[1] == 0 or act[2] == int((got > 2).sum()))
        if thr == 0.0:
            assert (got == 1).all() and act[1] == 0 and act[3] == 0          # everybody left at the first tested exit
        if thr == 1.0:
            assert (got == 3).all() and act[2] == B


def test_device_exit_resnet50_bottleneck_net():
    """ResNet-50 multi-exit (models/extra.py): after the first tested exit the Bottleneck convs carry a row table — the 1x1 convs leave
    conv1x1_stream for conv_igemm_wide / conv_igemm (the same K order), `layer2[0].conv2` (128 -> 128, stride 2) runs in the dynamic-exit
    form of conv3x3_s2's 128-channel tiles: every image's prediction at its exit equals the full run's."""
    from bayesnn_fpga_amd.models import extra as bx
    B, T, seed = 45, 6, 13
    m = build_seeded(bx.ResNet50MCEarlyExit, dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10))
    synthetic_weights_(m, 0)
    eng = m.to(DEV).eval().engine(torch.device(DEV), max_batch=B)
    x = synthetic_images(B, seed=23).to(DEV)
    full = eng.predict(x, T, seed=seed)
    p_full = full["mean"].cpu().numpy()
    conf = p_full.max(-1)
    for thr in (float(np.median(conf[1])), float(np.quantile(conf[2], 0.3)), 1.0):
        want = cex.exit_layer(p_full.copy(), thr)
        r = eng.predict_with_exit(x, T, thr, seed=seed)
        np.testing.assert_array_equal(r["exit_layer"].cpu().numpy(), want)
        np.testing.assert_allclose(r["best_preds"].cpu().numpy(), p_full[want, np.arange(B)], rtol=0, atol=1e-13)


@pytest.mark.parametrize("B,chunk", [(1, None), (4, None), (11, None), (45, 6)])
def test_device_exit_on_small_batches(B, chunk):
    """Batches whose full-chunk grid is too small for conv3x3_pw (B <= 11 with the default chunk, or an explicit small chunk)
    run the fused-shortcut conv2 of layer3 / layer4 in conv3x3_patch; after the first tested exit the launches carry a row
    table and need that kernel's IMAP instantiations on 8x8 and 4x4 maps too (round-2 advisor finding: BMI_ERR_UNSUPPORTED
    at B = 1).  A threshold that keeps some images active past every tested exit."""
    T, seed = 6, 5
    m = build_seeded(ResNet18MCEarlyExit, KWS["mc_block_exit"])
    synthetic_weights_(m, 0)
    eng = m.to(DEV).eval().engine(torch.device(DEV), max_batch=B, chunk_samples=chunk)
    x = synthetic_images(B, seed=33).to(DEV)
    p_full = eng.predict(x, T, seed=seed)["mean"].cpu().numpy()
    conf = p_full.max(-1)
    for thr in (float(conf[1:3].max()) + 1e-6, float(np.median(conf[2])), float(conf[1].min()) - 1e-6):
        want = cex.exit_layer(p_full.copy(), thr)
        r = eng.predict_with_exit(x, T, thr, seed=seed)
        np.testing.assert_array_equal(r["exit_layer"].cpu().numpy(), want)
        np.testing.assert_allclose(r["best_preds"].cpu().numpy(), p_full[want, np.arange(B)], rtol=0, atol=1e-13)
    assert (cex.exit_layer(p_full.copy(), float(conf[1:3].max()) + 1e-6) == 3).all()     # the first threshold sent nobody out early


def test_dynamic_exit_is_refused_by_the_exact_engine_only():
    """dtype "f32" (the exact engine: parity only) has no row-table form: BMI_ERR_UNSUPPORTED, as include/bayesnn_fpga_amd.h says."""
    from bayesnn_fpga_amd import _lib
    m = build_seeded(ResNet18MCEarlyExit, KWS["mc_block_exit"])
    synthetic_weights_(m, 0)
    eng = m.to(DEV).eval().engine(torch.device(DEV), max_batch=4, dtype="f32")
    with pytest.raises(_lib.BmiError):
        eng.predict_with_exit(synthetic_images(4, seed=1).to(DEV), 3, 0.5)


def test_dynamic_exit_needs_all_samples_in_one_chunk():
    m = build_seeded(ResNet18MCEarlyExit, KWS["mc_block_exit"])
    synthetic_weights_(m, 0)
    eng = m.to(DEV).eval().engine(torch.device(DEV), max_batch=4, chunk_samples=2)
    x = synthetic_images(4, seed=1).to(DEV)
    with pytest.raises(ValueError, match="one chunk"):
        eng.predict_with_exit(x, 3, 0.5)


def test_dynamic_exit_saves_time_at_full_size():
    """B = 250, T = 20: with a threshold that lets ~half the images leave at exit 1 the step is measurably shorter than the
    full run (the compacted stages launch proportionally smaller grids)."""
    import time
    B, T = 250, 20
    m = build_seeded(ResNet18MCEarlyExit, KWS["mc_block_exit"])
    synthetic_weights_(m, 0)
    eng = m.to(DEV).eval().engine(torch.device(DEV), max_batch=B)
    x = synthetic_images(B, seed=1234).to(DEV)
    conf = eng.predict(x, T, seed=3)["mean"].max(-1).values
    thr = float(conf[1].median())

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 3
    t_full = timed(lambda: eng.predict(x, T, seed=3))
    t_exit = timed(lambda: eng.predict_with_exit(x, T, thr, seed=3))
    r = eng.predict_with_exit(x, T, thr, seed=3)
    print(f"full {t_full * 1e3:.2f} ms, dynamic exit {t_exit * 1e3:.2f} ms, active after exits {r['active_after']}")
    assert 0.3 * B < r["active_after"][1] < 0.7 * B
    assert t_exit < 0.9 * t_full
