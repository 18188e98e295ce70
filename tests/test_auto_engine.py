"""engine_dtype = "auto" (the product default), the non-finite guard, and parity at the headline's OWN size.

Round-5 review, weak #1: the fast fp16 engine holds north_star's 1e-3 on near-uniform predictive distributions and misses it on
trained-like, peaky ones, "and nothing in the product notices"; and no HIP-vs-oracle run had ever happened at B = 250 x T = 100.
Here: (a) BASELINE configs[2] at B = 250, T = 100 on fp16 and f16x2 against the CPU oracle — the synthetic model AND the peaky twin
(classifiers x 24) from ONE oracle walk —, with the error printed for T in {4, 10, 30, 100}; (b) the calibration keeps fp16 on the
synthetic model and rejects it on the peaky one; (c) what the calibration batch cannot see (other batches) stays inside the bar;
(d) non-finite moment sums are an error, not a number.
"""
import warnings

import numpy as np
import pytest
import torch

from bayesnn_fpga_amd.models.resnet18.resnet18 import ResNet18MCEarlyExit
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_labels, synthetic_weights_
from tests.helpers import build_seeded

DEV = "cuda:0"
KW = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)      # BASELINE configs[2]
HEADS = ("ex1linear", "ex2linear", "ex3linear", "linear")
GAIN = 24.0


def _peaky_(model, gain=GAIN):
    """Trained-like logits: every exit's classifier weights x gain (tests/test_split_engine.py's stress model)."""
    with torch.no_grad():
        for name in HEADS:
            getattr(model, name).weight.mul_(gain)
    return model


def _mirror(peaky=False):
    m = synthetic_weights_(build_seeded(ResNet18MCEarlyExit, KW), 0)
    if peaky:
        _peaky_(m)
    return m.to(DEV).eval()


def test_auto_is_the_default_everywhere():
    """Host-only: the mirrors, FullAnalysis and evaluate take ``engine_dtype`` from the model, and the model's default is "auto"."""
    from bayesnn_fpga_amd.converter.pytorch.nn2bnn import MCDropout
    from bayesnn_fpga_amd.models._engine_mixin import AUTO_CANDIDATES, AUTO_TOL, EngineModelMixin
    assert EngineModelMixin.engine_dtype == "auto" and AUTO_CANDIDATES == ("f16", "f16x2") and AUTO_TOL <= 5e-4
    m = build_seeded(ResNet18MCEarlyExit, KW)
    assert m.engine_dtype == "auto" and m._auto == {}
    assert issubclass(MCDropout, EngineModelMixin)
    assert m.resolve_engine_dtype("cuda:0", "bf16") == "bf16"              # an explicit dtype never calibrates
    m.engine_dtype = "f16x2"
    assert m.resolve_engine_dtype("cuda:0") == "f16x2"


@pytest.mark.gpu
def test_auto_keeps_fp16_on_the_synthetic_model_and_rejects_it_on_the_peaky_one():
    """The calibration (first batch, 8 samples, fp16 vs f16x2 on the same masks) keeps fp16 where it agrees with the split engine to AUTO_TOL
    and switches — saying so once — where it does not: the stress model of tests/test_split_engine.py (fp16 5.0e-3 against the oracle)."""
    from oracle import mcd
    from oracle import resnet18 as oresnet
    B = 250
    x = synthetic_images(B, seed=1234).to(DEV)
    m = _mirror()
    with warnings.catch_warnings():
        warnings.simplefilter("error")                        # keeping fp16 is silent
        eng = m.engine(torch.device(DEV), max_batch=B, calib=x)
    rec = m._auto[DEV]
    print(f"synthetic headline model: auto -> {eng.dtype}; fp16 vs f16x2 on the calibration batch: mean {rec['dmean']:.2e} var {rec['dvar']:.2e} (tol {rec['tol']:.0e})")
    assert eng.dtype == "f16" and rec["dtype"] == "f16" and rec["dmean"] <= rec["tol"] and rec["nonfinite"] == {"f16": 0, "f16x2": 0}
    assert rec["calibrated_on"].startswith("the first 250 images")
    assert m.engine(torch.device(DEV), max_batch=B) is eng            # decided once per (weights, device)
    m.invalidate_engine()
    assert m._auto == {}                                               # new weights -> a new calibration

    p = _mirror(peaky=True)
    with pytest.warns(UserWarning, match="keeps the split engine 'f16x2'"):
        engp = p.engine(torch.device(DEV), max_batch=B, calib=x)
    recp = p._auto[DEV]
    print(f"peaky headline model: auto -> {engp.dtype}; fp16 vs f16x2: mean {recp['dmean']:.2e} var {recp['dvar']:.2e}")
    assert engp.dtype == "f16x2" and recp["dmean"] > recp["tol"]
    # ... and what auto runs holds north_star's bar against the oracle (T = 4: seconds of host time)
    o = _peaky_(synthetic_weights_(build_seeded(oresnet.ResNet18MCEarlyExit, KW), 0))
    ref = mcd.mcd_predict(o, x.cpu(), 4, 42)
    r = engp.predict(x, 4, seed=42)
    em, ev = np.abs(r["mean"].cpu().numpy() - ref["mean"]).max(), np.abs(r["var"].cpu().numpy() - ref["var"]).max()
    print(f"peaky model through auto vs oracle: mean {em:.2e} var {ev:.2e}")
    assert em <= 1e-3 and ev <= 1e-3
    # model(x) and the converter wrapper go through the same choice
    p2 = _mirror(peaky=True)
    with pytest.warns(UserWarning):
        p2(x[:64])
    assert p2._auto[DEV]["dtype"] == "f16x2" and "64 images" in p2._auto[DEV]["calibrated_on"]
    # without a batch to look at: synthetic calibration images, and it says so
    p3 = _mirror(peaky=True)
    with pytest.warns(UserWarning):
        p3.engine(torch.device(DEV), max_batch=8)
    assert "synthetic" in p3._auto[DEV]["calibrated_on"]


@pytest.mark.gpu
def test_auto_choice_holds_on_the_batches_the_calibration_did_not_see():
    """AUTO_TOL is half of north_star's 1e-3: the other half is for the batches of a loader the first one cannot speak for.  Measured here:
    fp16 vs f16x2 (same masks) on four further batches of other images and other seeds, T = 10 (the reference's mc_dropout_passes)."""
    from bayesnn_fpga_amd.models._engine_mixin import AUTO_TOL
    B, T = 250, 10
    m = _mirror()
    x0 = synthetic_images(B, seed=1234).to(DEV)
    assert m.resolve_engine_dtype(torch.device(DEV), None, calib=x0) == "f16"
    e16 = m.engine(torch.device(DEV), max_batch=B, dtype="f16")
    e32 = m.engine(torch.device(DEV), max_batch=B, dtype="f16x2")
    worst = 0.0
    for k in range(1, 5):
        x = synthetic_images(B, seed=1234 + 17 * k).to(DEV)
        a, b = e16.predict(x, T, seed=k), e32.predict(x, T, seed=k)
        d = max(float((a[q] - b[q]).abs().max()) for q in ("mean", "var"))
        worst = max(worst, d)
    print(f"fp16 vs f16x2 on 4 unseen batches (T = {T}): worst {worst:.2e}; calibration saw {m._auto[DEV]['dmean']:.2e} (tol {AUTO_TOL:.0e})")
    assert worst <= 1e-3 - 5e-5          # (f16x2 itself sits within 5e-5 of the oracle: tests/test_split_engine.py)


@pytest.mark.gpu
def test_headline_size_parity_b250_t100_and_error_versus_T():
    """BASELINE configs[2] at ITS OWN size — B = 250 images x T = 100 samples — on fp16 and f16x2 against the CPU oracle (one walk of
    25 000 image-samples, ~75-150 s of host time: once per suite), predictive mean AND variance within north_star's 1e-3; the error is
    printed for T in {4, 10, 30, 100} (the first T samples of the same streams) so that it is known whether fp16's error falls with T
    (independent rounding per sample) or has a floor (weight rounding and the prefix are common to all samples).  The peaky twin
    (classifiers x 24) comes from the SAME walk: forward pre-hooks keep the classifier inputs of every pass, and the twin's logits are its
    own ``F.linear`` on them — exactly what its oracle would compute behind an identical trunk."""
    import torch.nn.functional as F
    from oracle import mcd
    from oracle import resnet18 as oresnet
    B, T, seed = 250, 100, 42
    Ts = (4, 10, 30, 100)
    x = synthetic_images(B, seed=1234)
    o = synthetic_weights_(build_seeded(oresnet.ResNet18MCEarlyExit, KW), 0)
    feats = {n: [] for n in HEADS}
    hooks = [getattr(o, n).register_forward_pre_hook(lambda mod, inp, n=n: feats[n].append(inp[0].detach().clone())) for n in HEADS]
    n_thr = torch.get_num_threads()
    torch.set_num_threads(min(32, n_thr))        # (ATen's CPU convs at batch 250 do not scale past ~32 threads on the GPU box's host: bench.py's sweep)
    try:
        logits, probs = mcd.mcd_passes(o, x, T, seed)                  # float64 [T, E, B, C]
    finally:
        torch.set_num_threads(n_thr)
    for h in hooks:
        h.remove()
    assert all(len(v) == T for v in feats.values())
    with torch.no_grad():
        pk_logits = np.stack([np.stack([F.linear(feats[n][t], getattr(o, n).weight * GAIN, getattr(o, n).bias).numpy() for n in HEADS])
                              for t in range(T)]).astype(np.float64)
        pk_probs = np.stack([np.stack([F.softmax(F.linear(feats[n][t], getattr(o, n).weight * GAIN, getattr(o, n).bias), dim=1).numpy()
                                       for n in HEADS]) for t in range(T)]).astype(np.float64)
    # (the hooks' features reproduce the walk's own logits: the twin's are then its oracle's)
    with torch.no_grad():
        again = np.stack([F.linear(feats[n][0], getattr(o, n).weight, getattr(o, n).bias).numpy() for n in HEADS])
    np.testing.assert_array_equal(again.astype(np.float64), logits[0])
    conf = pk_probs.mean(0)[-1].max(-1)
    print(f"peaky twin: max|logit| {np.abs(pk_logits).max():.0f}, final-exit max prob >= 0.99 on {100 * float((conf >= 0.99).mean()):.0f} % of the images")
    xd = x.to(DEV)
    table = {}
    for tag, model, P in (("synthetic", _mirror(), probs), ("peaky x24", _mirror(peaky=True), pk_probs)):
        for dt in ("f16", "f16x2"):
            eng = model.engine(torch.device(DEV), max_batch=B, dtype=dt)
            for t in Ts:
                r = eng.predict(xd, t, seed=seed)
                em = float(np.abs(r["mean"].cpu().numpy() - P[:t].mean(0)).max())
                ev = float(np.abs(r["var"].cpu().numpy() - P[:t].var(0)).max())
                table[(tag, dt, t)] = (em, ev)
            eng.check_finite()
            eng.close()
            eng.workspace = None
            model.invalidate_engine()
    print("max |HIP - oracle| at B = 250 (mean / variance) versus T:")
    for tag in ("synthetic", "peaky x24"):
        for dt in ("f16", "f16x2"):
            print(f"  {tag:10s} {dt:6s} " + "   ".join(f"T={t}: {table[(tag, dt, t)][0]:.2e} / {table[(tag, dt, t)][1]:.2e}" for t in Ts))
    # the headline configuration at its own size: both engines inside north_star's bar on the synthetic model ...
    for dt in ("f16", "f16x2"):
        assert table[("synthetic", dt, 100)][0] <= 1e-3 and table[("synthetic", dt, 100)][1] <= 1e-3, (dt, table[("synthetic", dt, 100)])
    assert table[("synthetic", "f16x2", 100)][0] <= 1e-4
    # ... and on trained-like logits the engine auto picks (f16x2) holds it at every T; what fp16 does there is printed, not asserted
    for t in Ts:
        assert table[("peaky x24", "f16x2", t)][0] <= 1e-3 and table[("peaky x24", "f16x2", t)][1] <= 1e-3, (t, table[("peaky x24", "f16x2", t)])


@pytest.mark.gpu
def test_non_finite_moment_sums_are_an_error_not_a_number():
    """An fp16 activation past 65 504 becomes inf, then NaN in the softmax; ``bmi_finalize_checked`` counts the non-finite sums on the
    device and the mirrors raise where they read results (FullAnalysis) — on every engine.  Provoked here at the last step of that chain
    (a NaN in one classifier's bias: every logit sum and softmax of that exit is non-finite), which does not depend on how an overflow
    happens to travel through ReLUs (``fmaxf(NaN, 0)`` is 0) and masks on the way."""
    from bayesnn_fpga_amd.train.results_analyzer import FullAnalysis
    m = synthetic_weights_(build_seeded(ResNet18MCEarlyExit, KW), 0)
    with torch.no_grad():
        m.ex2linear.bias[3] = float("nan")
    m = m.to(DEV).eval()
    x = synthetic_images(8, seed=5).to(DEV)
    for dt in ("f16", "f16x2"):
        eng = m.engine(torch.device(DEV), max_batch=8, dtype=dt)
        r = eng.predict(x, 3, seed=1)
        assert not bool(torch.isfinite(r["mean"]).all())
        assert eng.nonfinite_count(reset=False) > 0
        with pytest.raises(FloatingPointError, match="non-finite"):
            eng.check_finite()
        assert eng.nonfinite_count() == 0                     # the check resets the counter
    # ... behind a hipGraph replay too (the counter is engine-owned device memory allocated at construction, outside any capture: every replay adds to it)
    from bayesnn_fpga_amd.engine import BatchesInFlight
    pipe = BatchesInFlight(m, torch.device(DEV), n=1, max_batch=8, dtype="f16")
    for k in range(3):                                        # capture, replay, replay
        pipe.predict_graphed(x, 3, seed=1)
        pipe.last_stream.synchronize()
        assert pipe.engines[0].nonfinite_count() > 0, k
    pipe.close()
    good = _mirror()
    eng = good.engine(torch.device(DEV), max_batch=8, dtype="f16")
    eng.predict(x, 3, seed=1)
    eng.check_finite()                                        # finite results: no error
    m.engine_dtype = "f16"
    loader = [(x.cpu(), synthetic_labels(8, 10, seed=6))]
    with pytest.raises(FloatingPointError):
        FullAnalysis(m, loader, gpu=0, mc_dropout=True, mc_passes=3)
    # auto on such a model: both candidates overflow -> the safe one is kept and the walk still raises instead of returning NaNs
    m.engine_dtype = "auto"
    m.invalidate_engine()
    with pytest.warns(UserWarning, match="non-finite"):
        assert m.resolve_engine_dtype(torch.device(DEV), None, calib=x) == "f16x2"


# ---- the other configurations at THEIR own size, through the product's default engine choice ------------------------------------------------
def _own_size_cases():
    from bayesnn_fpga_amd.models import extra as bx
    from oracle import extra_models as ox
    from oracle import resnet18 as oresnet
    return {
        # BASELINE configs[3]: Masksembles M = 8, C = 100, T = M = 8 (one mask per GPU on eight ranks; here all eight on one)
        "config4_masksembles_m8_T8": (ResNet18MCEarlyExit, oresnet.ResNet18MCEarlyExit,
                                      dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=100, mask_type="mask", num_masks=8, mask_scale=4.0), 8),
        # what every run of the paper uses: exit-only dropout, C = 100, T = 10 (journal_script.sh:10-63, hyperparameters.py:111-114)
        "paper_exit_only_c100_T10": (ResNet18MCEarlyExit, oresnet.ResNet18MCEarlyExit, dict(dropout_exit=True, dropout=None, dropout_p=0.25, out_dim=100), 10),
        # BASELINE configs[1]: VGG-11, 3 dropout layers, T = 30
        "config2_vgg11_T30": (bx.VGG11MC, ox.VGG11MC, dict(num_bayes_layer=3, dropout_p=0.25, out_dim=10), 30),
    }


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["config4_masksembles_m8_T8", "paper_exit_only_c100_T10", "config2_vgg11_T30"])
def test_configs_at_their_own_size_through_the_default_engine(name):
    """B = 250 at each configuration's own T, HIP vs the CPU oracle, through ``engine_dtype = "auto"`` — whatever it picks must hold north_star's
    1e-3 on mean and variance (what it picked and what the other candidate measures are printed).  VGG-11 is the case where the choice matters on
    SYNTHETIC weights: plain fp16 sits at 5-8e-4 of the oracle at B = 250 (tests/test_full_batch.py), the calibration may keep or reject it."""
    from oracle import mcd
    cls, ocls, kw, T = _own_size_cases()[name]
    B, seed = 250, 42
    m = synthetic_weights_(build_seeded(cls, kw), 0).to(DEV).eval()
    o = synthetic_weights_(build_seeded(ocls, kw), 0)
    x = synthetic_images(B, seed=1234)
    n_thr = torch.get_num_threads()
    torch.set_num_threads(min(32, n_thr))
    try:
        ref = mcd.mcd_predict(o, x, T, seed)
    finally:
        torch.set_num_threads(n_thr)
    xd = x.to(DEV)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        eng = m.engine(torch.device(DEV), max_batch=B, calib=xd)
    rec = m._auto[DEV]
    r = eng.predict(xd, T, seed=seed)
    eng.check_finite()
    em, ev = float(np.abs(r["mean"].cpu().numpy() - ref["mean"]).max()), float(np.abs(r["var"].cpu().numpy() - ref["var"]).max())
    other = rec["safe"] if eng.dtype == rec["fast"] else rec["fast"]
    ro = m.engine(torch.device(DEV), max_batch=B, dtype=other).predict(xd, T, seed=seed)
    om, ov = float(np.abs(ro["mean"].cpu().numpy() - ref["mean"]).max()), float(np.abs(ro["var"].cpu().numpy() - ref["var"]).max())
    print(f"{name}: B = {B}, T = {T}: auto -> {eng.dtype} (calibration: {rec['dmean']:.1e} / {rec['dvar']:.1e}, tol {rec['tol']:.0e}): "
          f"mean {em:.2e} var {ev:.2e}   |   {other}: mean {om:.2e} var {ov:.2e}")
    assert em <= 1e-3 and ev <= 1e-3
