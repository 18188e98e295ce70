"""-m gpu: the EXACT engine (bmi_model_desc.dtype = BMI_DTYPE_F32, csrc/conv_exact.hip: fp32 activations, fp32 weights, every
product on v_mfma_f32_32x32x2_f32) — the arithmetic of the reference's fp32 CPU path on the device (SURVEY.md §7 hard part 2:
"keep an fp32-MFMA path for parity tests").  What it is for: a parity test whose fp16 tolerance had to be wider than
north_star's 1e-3 (per-pass logits, the converter goldens) gets a twin here at fp32 summation-order tolerances, so "green"
can tell rounding from a bug.  Same graph builder, same site numbering, same Philox indices, same head kernel as the fp16
engine: only the conv / mask / pool kernels and the element size differ.

Tolerances: fp32 vs fp32 in another summation order.  Per-pass logits 2e-4 (logits of O(10) through 20 layers), probabilities,
predictive mean and variance 2e-5 — 50x inside north_star's 1e-3."""
import ctypes as C

import numpy as np
import pytest
import torch

from bayesnn_fpga_amd import _lib
from bayesnn_fpga_amd.converter.pytorch import MCDropout
from bayesnn_fpga_amd.engine import CompiledGraph
from bayesnn_fpga_amd.models import extra as bx
from bayesnn_fpga_amd.models.resnet18.resnet18 import ResNet18Base, ResNet18MCEarlyExit
from bayesnn_fpga_amd.models.vgg19 import vgg19 as bvgg
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_
from oracle import extra_models as ox
from oracle import mcd
from tests import gpu_helpers as gh
from tests.helpers import build_seeded, converter_cnn, golden_kwargs, load_golden

pytestmark = pytest.mark.usefixtures("fp16_engine_default")      # (tests/conftest.py: these tests pin the fp16 kernels)

DEV = "cuda:0"
LOGIT_TOL, PROB_TOL = 2e-4, 2e-5


# ---- CPU side: the graph of the exact engine -----------------------------------------------------------------------------------
def test_exact_graph_has_no_speed_only_features():
    """dtype='f32' compiles on a CPU-only box too (host-only bmi_create / bmi_plan): fp32 conv weights, no fused shortcut (in2),
    4-byte activations in the plan; an unknown dtype is rejected."""
    kw = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
    m = synthetic_weights_(build_seeded(ResNet18MCEarlyExit, kw), 0)
    c16, c32 = CompiledGraph(m, "cpu", 8, 2), CompiledGraph(m, "cpu", 8, 2, dtype="f32")
    assert any(op.get("in2", -1) >= 0 for op in c16.graph.ops) and not any(op.get("in2", -1) >= 0 for op in c32.graph.ops)
    assert all(op["weight"].dtype == torch.float32 for op in c32.graph.ops)
    assert c32.prefix_macs + 8 * c32.suffix_macs == c16.prefix_macs + 8 * c16.suffix_macs       # the same arithmetic, re-grouped
    assert c32.workspace_bytes > 1.5 * c16.workspace_bytes
    with pytest.raises(ValueError):
        CompiledGraph(m, "cpu", 8, 2, dtype="f64")


# ---- the kernels, through the single-kernel entry points under unit_entry_dtype = F32 ----------------------------------------
@pytest.fixture
def f32_entries():
    _lib.set_option("unit_entry_dtype", _lib.DTYPE_F32)
    yield
    _lib.set_option("unit_entry_dtype", _lib.DTYPE_F16)


SHAPES = {"S1": (64, 64, 32, 3, 1, 1), "D2": (64, 128, 32, 3, 2, 1), "P3": (128, 256, 16, 1, 2, 0), "S3": (256, 256, 8, 3, 1, 1),
          "S4": (512, 512, 4, 3, 1, 1), "V5": (96, 192, 6, 5, 1, 2)}


def _conv64(x, w, scale, bias, res, relu, stride, pad, n, in_mod, res_mod):
    xi = x.double().cpu().permute(0, 3, 1, 2)[torch.arange(n) % in_mod]
    y = torch.nn.functional.conv2d(xi, w.double().cpu().permute(0, 3, 1, 2), stride=stride, padding=pad)
    y = y * scale.double().cpu()[None, :, None, None] + bias.double().cpu()[None, :, None, None]
    if res is not None:
        y = y + res.double().cpu().permute(0, 3, 1, 2)[torch.arange(n) % res_mod]
    return torch.relu(y) if relu else y


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SHAPES))
def test_exact_conv_against_float64(name, f32_entries):
    """conv + BN + residual + ReLU in fp32 against torch float64 on the SAME fp32 operands: 1e-5 relative to the output scale
    (K up to 4608 fp32 products).  n = 3 images: ragged last 64-pixel tile on the 4x4 / 6x6 maps; broadcast input (n % in_mod)."""
    cin, cout, H, k, s, p = SHAPES[name]
    g = torch.Generator().manual_seed(11)
    B, tc = 3, 2
    x = torch.randn(B, H, H, cin, generator=g).to(DEV)
    w = (torch.randn(cout, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(DEV)
    scale, bias = (0.5 + torch.rand(cout, generator=g)).to(DEV), (0.2 * torch.randn(cout, generator=g)).to(DEV)
    ho = (H + 2 * p - k) // s + 1
    res = torch.randn(B * tc, ho, ho, cout, generator=g).to(DEV)
    out = gh.run_conv(x, w, scale, bias, res, True, s, p, B * tc, B, B * tc, batch=B, out_dtype=torch.float32)
    ref = _conv64(x, w, scale, bias, res, True, s, p, B * tc, B, B * tc)
    got = out.double().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["elementwise", "channel", "masksemble"])
def test_exact_conv_fused_site_is_bit_exact_on_the_mask(kind, f32_entries):
    cin, cout, H, k, s, p = SHAPES["S3"]
    B, tc, t0, seed, cnt0 = 3, 3, 5, (7 << 32) + 42, 2
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B * tc, H, H, cin, generator=g).to(DEV)
    w = (torch.randn(cout, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(DEV)
    scale, bias = (0.5 + torch.rand(cout, generator=g)).to(DEV), (0.2 * torch.randn(cout, generator=g)).to(DEV)
    if kind == "elementwise":
        site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=4, p=0.25)
    elif kind == "channel":
        site = dict(kind=_lib.SITE_CHANNEL, site_id=2, p=0.5)
    else:
        site = dict(kind=_lib.SITE_MASKSEMBLE, site_id=1, masks=(np.random.RandomState(0).rand(4, cout) < 0.4).astype(np.float32))
    out = gh.run_conv(x, w, scale, bias, None, True, s, p, B * tc, B * tc, 1, site=site, batch=B, t0=t0, seed=seed, cnt0=cnt0,
                      out_dtype=torch.float32)
    mult = gh.folded_site_mask(site, B, cout, H, H, tc, t0, seed, cnt0).double()
    ref = _conv64(x, w, scale, bias, None, True, s, p, B * tc, B * tc, 1) * mult
    got = out.double().cpu().permute(0, 3, 1, 2)
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    dropped = mult == 0
    assert dropped.any() and torch.equal(got[dropped], torch.zeros(int(dropped.sum()), dtype=torch.float64))


@pytest.mark.gpu
def test_exact_conv_rejects_what_it_does_not_build(f32_entries):
    lib = _lib.lib()
    x = torch.zeros(1, 8, 8, 48, device=DEV)
    w = torch.zeros(64, 3, 3, 48, device=DEV)
    o = torch.zeros(1, 8, 8, 64, device=DEV)
    rc = lib.bmi_conv_igemm_fwd(gh.ptr(x), None, 1.0, gh.ptr(w), None, None, None, gh.ptr(o), 1, 1, 1, 8, 8, 48, 64, 3, 1, 1, 0, None, 1,
                                0, 0, 0, gh.stream())
    assert rc == -95                                   # Cin % 32
    bits = torch.zeros(64, dtype=torch.uint8, device=DEV)
    x = torch.zeros(1, 8, 8, 64, device=DEV)
    w = torch.zeros(64, 3, 3, 64, device=DEV)
    rc = lib.bmi_conv_igemm_fwd(gh.ptr(x), gh.ptr(bits), 1.0, gh.ptr(w), None, None, None, gh.ptr(o), 1, 1, 1, 8, 8, 64, 64, 3, 1, 1, 0, None,
                                1, 0, 0, 0, gh.stream())
    assert rc == -95                                   # input-side keep bits are a speed feature of the 16-bit engines


@pytest.mark.gpu
def test_exact_stem_mask_and_maxpool(f32_entries):
    lib = _lib.lib()
    g = torch.Generator().manual_seed(3)
    n = 5
    x = torch.randn(n, 3, 32, 32, generator=g).to(DEV)
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.3
    scale, bias = 0.5 + torch.rand(64, generator=g), 0.2 * torch.randn(64, generator=g)
    out = torch.empty(n, 32, 32, 64, device=DEV)
    wd, sd, bd = w.to(DEV), scale.to(DEV), bias.to(DEV)
    _lib.check(lib.bmi_stem_conv_fwd(gh.ptr(x), gh.ptr(wd), gh.ptr(sd), gh.ptr(bd), gh.ptr(out), n, 3, 32, 32, 64, 3, 1, 1, 0, gh.stream()),
               "bmi_stem_conv_fwd")
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().permute(0, 3, 1, 2), padding=1) * scale.double()[None, :, None, None] + \
        bias.double()[None, :, None, None]
    assert float((out.double().cpu().permute(0, 3, 1, 2) - ref).abs().max()) <= 1e-5
    # stand-alone site on an fp32 tensor: expands B deterministic images to the folded batch, bit-exact (one fp32 product)
    B, tc, H, Cc, t0, seed = 3, 4, 6, 64, 2, 99
    xs = torch.randn(B, H, H, Cc, generator=g).to(DEV)
    site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=0, p=0.375)
    keep = []
    s = gh.site_struct(site, keep)
    om = torch.empty(B * tc, H, H, Cc, device=DEV)
    _lib.check(lib.bmi_mask_apply(gh.ptr(xs), gh.ptr(om), B * tc, B, H * H, Cc, C.byref(s), B, t0, seed, 0, gh.stream()), "bmi_mask_apply")
    torch.cuda.synchronize()
    mult = gh.folded_site_mask(site, B, Cc, H, H, tc, t0, seed)
    assert torch.equal(om.cpu().permute(0, 3, 1, 2), xs.cpu().permute(0, 3, 1, 2).repeat(tc, 1, 1, 1) * mult)
    xp = torch.randn(3, 8, 8, 64, generator=g).to(DEV)
    op = torch.empty(3, 4, 4, 64, device=DEV)
    _lib.check(lib.bmi_maxpool2(gh.ptr(xp), gh.ptr(op), 3, 8, 8, 64, gh.stream()), "bmi_maxpool2")
    torch.cuda.synchronize()
    assert torch.equal(op.cpu().permute(0, 3, 1, 2), torch.nn.functional.max_pool2d(xp.cpu().permute(0, 3, 1, 2), 2))


# ---- the whole path on the exact engine, against the reference's own outputs ----------------------------------------------------
def _exact(model):
    model = model.to(DEV).eval()
    model.engine_dtype = "f32"            # model(x) and engine() without a dtype now compile the exact engine
    return model


RESNET_CASES = ["exit_only", "block_exit", "block_noexit", "layer_exit", "mask4_block_exit", "mask8_exit_c100", "block_exit_p02",
                "layer_exit_p256"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", RESNET_CASES)
def test_exact_engine_against_reference_golden(name):
    """The eight ResNet-18 goldens of tests/test_gpu_model.py (per-pass logits of the reference's own ResNet18MCEarlyExit and
    its _get_output 5-tuple) at fp32 tolerances: logits 2e-4 (fp16 engine: 2e-2), mean / var 2e-5 (fp16: 1e-3)."""
    g = load_golden(f"resnet18_{name}.npz")
    kw = golden_kwargs(g)
    B, T, seed = int(g["B"]), int(g["T"]), int(g["seed"])
    model = _exact(synthetic_weights_(build_seeded(ResNet18MCEarlyExit, kw), 0))
    model.mc_seed = seed
    x = synthetic_images(B, seed=1234).to(DEV)
    passes = np.stack([np.stack([o.cpu().numpy() for o in model(x)]) for _ in range(T)])
    np.testing.assert_allclose(passes, g["logits"], rtol=0, atol=LOGIT_TOL)
    ref_probs = torch.softmax(torch.from_numpy(g["logits"]), -1).numpy().astype(np.float64)
    eng = model.engine(x.device, max_batch=B)
    assert eng.dtype == "f32"
    r = eng.predict(x, T, seed=seed, cnt0=0)
    np.testing.assert_allclose(r["mean"].cpu().numpy(), g["go_output_sm"], rtol=0, atol=PROB_TOL)
    np.testing.assert_allclose(r["var"].cpu().numpy(), ref_probs.var(0), rtol=0, atol=PROB_TOL)
    np.testing.assert_allclose(r["logit_mean"].cpu().numpy(), g["go_output"], rtol=0, atol=LOGIT_TOL)
    # chunking invariance holds for this engine too (per-sample values do not depend on the chunk)
    e1 = model.engine(x.device, max_batch=B, chunk_samples=1)
    S1 = e1.accumulate(x, e1.new_moments(B), 0, T, seed).cpu()
    e3 = model.engine(x.device, max_batch=B, chunk_samples=3)
    torch.testing.assert_close(e3.accumulate(x, e3.new_moments(B), 0, T, seed).cpu(), S1, rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["exit_mc", "exit_mask4"])
def test_exact_engine_vgg19_against_reference_golden(name):
    g = load_golden(f"vgg19_{name}.npz")
    kw = golden_kwargs(g)
    B, T, seed = int(g["B"]), int(g["T"]), int(g["seed"])
    m = _exact(synthetic_weights_(build_seeded(bvgg.VGG19MCEarlyExit, kw), 0))
    m.mc_seed = seed
    x = synthetic_images(B, seed=1234).to(DEV)
    passes = np.stack([np.stack([o.cpu().numpy() for o in m(x)]) for _ in range(T)])
    np.testing.assert_allclose(passes, g["logits"], rtol=0, atol=LOGIT_TOL)
    ref_probs = torch.softmax(torch.from_numpy(g["logits"]), -1).numpy().astype(np.float64)
    r = m.engine(x.device, max_batch=B).predict(x, T, seed=seed)
    np.testing.assert_allclose(r["mean"].cpu().numpy(), ref_probs.mean(0), rtol=0, atol=PROB_TOL)
    np.testing.assert_allclose(r["var"].cpu().numpy(), ref_probs.var(0), rtol=0, atol=PROB_TOL)


def _converted_case(fixture, net):
    g = load_golden(fixture)
    B, T, seed, p = int(g["B"]), int(g["T"]), int(g["seed"]), float(g["p"])
    synthetic_weights_(net, 0)
    m = MCDropout(net, nSamples=T, p=p).to(DEV)
    m.mc_seed = seed
    x = synthetic_images(B, seed=1234).to(DEV)
    ref = g["logits"]
    ref_probs = torch.softmax(torch.from_numpy(ref), -1).numpy().astype(np.float64)
    out = {}
    for dt in ("f32", "f16"):
        r = m.engine(x.device, max_batch=B, dtype=dt).predict(x, T, seed=seed)
        out[dt] = (float(np.abs(r["mean"].cpu().numpy() - ref_probs.mean(0)).max()), float(np.abs(r["var"].cpu().numpy() - ref_probs.var(0)).max()),
                   float(np.abs(r["logit_mean"].cpu().numpy() - ref.mean(0)).max()))
    print(f"{fixture}: max|logit| {np.abs(ref).max():.1f}; exact engine mean {out['f32'][0]:.2e} var {out['f32'][1]:.2e} logit_mean {out['f32'][2]:.2e}"
          f" | fp16 engine mean {out['f16'][0]:.2e} var {out['f16'][1]:.2e} logit_mean {out['f16'][2]:.2e}")
    m.engine_dtype = "f32"
    m.train()                              # training mode: one stochastic pass per call
    return g, m, x, out, ref


@pytest.mark.gpu
@pytest.mark.parametrize("fixture,make,logit_scale", [
    ("converter_cnn.npz", lambda: (torch.manual_seed(0), converter_cnn())[1], 1.0),
    ("converter_resnet18base.npz", lambda: (torch.manual_seed(0), ResNet18Base(n_exits=1, out_dim=10))[1], 1.0),
    ("converter_vgg19.npz", lambda: (torch.manual_seed(0), bvgg.VGG19(n_exits=1, out_dim=10))[1], 5.0),
])
def test_exact_engine_converter_goldens_mean_and_var_within_1e3(fixture, make, logit_scale):
    """The reference's own converter on its own networks (nn2bnn.py:32-45 + Dropouts.py:25-56 applied to resnet18.py:189-204 and
    vgg19.py:186-192, and to the toy CNN): predictive mean AND variance within north_star's 1e-3 — here 2e-5 — on the exact engine,
    per-pass logits to fp32 summation order, the exact zero pattern of the dropped logits.  The fp16 engine's error on the same inputs
    is printed beside it (its own assertions live in tests/test_converter.py)."""
    g, m, x, out, ref = _converted_case(fixture, make())
    T = int(g["T"])
    assert out["f32"][0] <= PROB_TOL and out["f32"][1] <= PROB_TOL, out
    assert out["f32"][2] <= LOGIT_TOL * logit_scale, out
    m.mc_pass = 0
    passes = []
    for _ in range(T):
        o = m(x)
        passes.append((o[0] if isinstance(o, list) else o).cpu().numpy()[None])
    passes = np.stack(passes)
    assert passes.shape == ref.shape
    np.testing.assert_allclose(passes, ref, rtol=0, atol=LOGIT_TOL * logit_scale)
    zero = ref == 0
    assert zero.any() and np.array_equal(passes == 0, zero)


@pytest.mark.gpu
@pytest.mark.parametrize("cls,ocls,kw,B,T", [
    (bx.VGG11MC, ox.VGG11MC, dict(num_bayes_layer=3, dropout_p=0.25, out_dim=10), 250, 3),
    (bx.ResNet50MCEarlyExit, ox.ResNet50MCEarlyExit, dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10), 32, 2),
])
def test_exact_engine_on_the_unpinned_configs(cls, ocls, kw, B, T):
    """VGG-11 (config 2, at the reference's batch of 250) and ResNet-50 multi-exit (config 5): no reference model exists, the oracle is
    the build's own fp32 restatement; the fp16 engine sits at 0.78e-3 / 0.77e-3 of the 1e-3 bar against it (tests/test_full_batch.py).
    The exact engine agrees with the same oracle to 2e-5: the rest IS fp16 rounding."""
    seed = 42
    m, o = build_seeded(cls, kw), build_seeded(ocls, kw)
    synthetic_weights_(m, 0)
    synthetic_weights_(o, 0)
    x = synthetic_images(B, seed=1234)
    ref = mcd.mcd_predict(o, x, T, seed)
    r = _exact(m).engine(torch.device(DEV), max_batch=B).predict(x.to(DEV), T, seed=seed)
    err_m = float(np.abs(r["mean"].cpu().numpy() - ref["mean"]).max())
    err_v = float(np.abs(r["var"].cpu().numpy() - ref["var"]).max())
    print(f"{cls.__name__} exact engine: B={B} T={T} max|mean-oracle|={err_m:.2e} max|var-oracle|={err_v:.2e}")
    assert err_m <= PROB_TOL and err_v <= PROB_TOL
    assert float(np.abs(r["logit_mean"].cpu().numpy() - ref["logit_mean"]).max()) <= LOGIT_TOL * max(1.0, float(np.abs(ref["logits"]).max()) / 10)


@pytest.mark.gpu
@pytest.mark.parametrize("kw,T,cnt0", [
    (dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10), 3, 0),                                          # BASELINE configs[2]
    (dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=100, mask_type="mask", num_masks=8, mask_scale=4.0), 3, 6),   # configs[3], mask wrap-around
], ids=["resnet18_block_exit", "resnet18_masksembles_m8_c100"])
def test_exact_engine_at_the_reference_batch(kw, T, cnt0):
    """The two reference-pinned BASELINE configs at the reference's REAL test batch (B = 250, hyperparameters.py:265-266) on the exact engine
    against the fp32 CPU oracle: the twins of tests/test_full_batch.py::test_real_batch_against_oracle (fp16: 3.6e-4 / 1.1e-5) at 2e-5."""
    from oracle import resnet18 as oresnet
    B, seed = 250, 42
    m, o = build_seeded(ResNet18MCEarlyExit, kw), build_seeded(oresnet.ResNet18MCEarlyExit, kw)
    synthetic_weights_(m, 0)
    synthetic_weights_(o, 0)
    if cnt0:
        for mod in o.modules():
            if hasattr(mod, "cnt") and hasattr(mod, "masks"):
                mod.cnt = cnt0
    x = synthetic_images(B, seed=1234)
    ref = mcd.mcd_predict(o, x, T, seed)
    r = _exact(m).engine(torch.device(DEV), max_batch=B).predict(x.to(DEV), T, seed=seed, cnt0=cnt0)
    err_m = float(np.abs(r["mean"].cpu().numpy() - ref["mean"]).max())
    err_v = float(np.abs(r["var"].cpu().numpy() - ref["var"]).max())
    print(f"exact engine, B=250 T={T}: max|mean-oracle|={err_m:.2e} max|var-oracle|={err_v:.2e}")
    assert err_m <= PROB_TOL and err_v <= PROB_TOL


@pytest.mark.gpu
def test_layer_trace_reads_both_workspaces():
    """tools/layer_trace.py (bmi_tensor_info + "ws_no_reuse"): every activation tensor of the fp16 engine against the exact engine on the converted
    toy CNN — the relative rms error of each layer is fp16 rounding (< 2e-3), the first layers' well under 1e-3, nothing jumps."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "layer_trace.py"), "--model", "converter_cnn", "--batch", "5", "--T", "3"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [ln.split("|")[1].split() for ln in r.stdout.splitlines() if re.match(r"\s+\d+ (stem|conv|mask|maxpool|dense)", ln)]
    assert len(rows) >= 6
    rel = [float(x[-1]) for x in rows]
    assert max(rel) < 2e-3 and rel[0] < 1e-3 and all(v > 0 for v in rel)
    m = re.search(r"# mean: max\|f16 - exact\| = ([0-9.e+-]+)", r.stdout)
    assert m and float(m.group(1)) < 1e-3
