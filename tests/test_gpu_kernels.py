"""-m gpu: every HIP kernel, through the C ABI, against the CPU oracle / a torch fp32 reference
of the same op on the same (fp16-rounded) operands."""
import ctypes as C

import numpy as np
import pytest
import torch

from bayesnn_fpga_amd import _lib
from oracle import philox
from tests import gpu_helpers as gh

pytestmark = pytest.mark.gpu
DEV = gh.DEV


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def pytest_generate_tests(metafunc):
    """Every conv test runs under both MFMA instruction shapes (v_mfma_f32_32x32x16_f16 / 16x16x32_f16: different
    accumulator layouts, hence different main-loop addressing and epilogue code in conv3x3_patch / conv_igemm_wide), and the
    16x16x32 shape with and without the lite epilogue (BN + residual + ReLU + 2-bit elementwise site on the registers)."""
    if metafunc.function.__name__.startswith("test_conv"):
        metafunc.fixturenames.append("mfma_shape")
        metafunc.parametrize("mfma_shape", [(32, 1), (16, 1), (16, 0)], indirect=True, ids=["mfma32", "mfma16", "mfma16-general"])


@pytest.fixture
def mfma_shape(request):
    shape, lite = request.param
    for k in ("mfma_shape_patch", "mfma_shape_wide"):
        _lib.set_option(k, shape)
    _lib.set_option("epilogue_lite", lite)
    yield shape
    for k in ("mfma_shape_patch", "mfma_shape_wide"):
        _lib.set_option(k, 0)
    _lib.set_option("epilogue_lite", 1)


@pytest.mark.parametrize("seed,site,t,p,n", [(42, 0, 0, 0.25, 4096), (7, 3, 99, 0.5, 1003), ((1 << 40) + 5, 6, 5, 0.125, 64),
                                              (1, 1, 1, 0.0, 16), (1, 1, 1, 1.0, 16), (3, 2, 17, 0.2, 777)])
def test_philox_mask_bit_exact(seed, site, t, p, n):
    lib = _lib.lib()
    keep = torch.zeros(n, dtype=torch.uint8, device=DEV)
    _lib.check(lib.bmi_philox_mask(gh.ptr(keep), n, seed, site, t, p, gh.stream()), "bmi_philox_mask")
    torch.cuda.synchronize()
    assert np.array_equal(keep.cpu().numpy().astype(bool), philox.keep_bits(n, seed, site, t, p))


# (Cin, Cout, H, k, stride, pad) — one per GEMM shape class of SURVEY.md Appendix A
SHAPES = {
    "S1": (64, 64, 32, 3, 1, 1), "D2": (64, 128, 32, 3, 2, 1), "P2": (64, 128, 32, 1, 2, 0),
    "S2": (128, 128, 16, 3, 1, 1), "D3": (128, 256, 16, 3, 2, 1), "P3": (128, 256, 16, 1, 2, 0),
    "S3": (256, 256, 8, 3, 1, 1), "D4": (256, 512, 8, 3, 2, 1), "P4": (256, 512, 8, 1, 2, 0),
    "S4": (512, 512, 4, 3, 1, 1),
}


def _conv_inputs(cin, cout, H, k, n_in, seed, with_res, n_res=None):
    g = _gen(seed)
    x = (torch.randn(n_in, H, H, cin, generator=g)).to(torch.float16).to(DEV)
    w = (torch.randn(cout, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(torch.float16).to(DEV)
    scale = (0.5 + torch.rand(cout, generator=g)).to(DEV)
    bias = (0.2 * torch.randn(cout, generator=g)).to(DEV)
    return x, w, scale, bias, g


@pytest.mark.parametrize("name", list(SHAPES))
def test_conv_shape_classes(name):
    cin, cout, H, k, s, p = SHAPES[name]
    n = 3                                        # 3 images: M is never a multiple of the 128-pixel tile for the 4x4 maps
    x, w, scale, bias, g = _conv_inputs(cin, cout, H, k, n, 11, False)
    ho = (H + 2 * p - k) // s + 1
    res = torch.randn(n, ho, ho, cout, generator=g).to(torch.float16).to(DEV)
    out = gh.run_conv(x, w, scale, bias, res, True, s, p, n, n, n)
    ref = gh.conv_ref(x, w, scale, bias, res, True, s, p, n, n, n)
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-3)


def test_conv_no_epilogue_terms_and_no_relu():
    cin, cout, H, k, s, p = SHAPES["S2"]
    x, w, _, _, _ = _conv_inputs(cin, cout, H, k, 2, 5, False)
    out = gh.run_conv(x, w, None, None, None, False, s, p, 2, 2, 1)
    ref = gh.conv_ref(x, w, None, None, None, False, s, p, 2, 2, 1)
    assert (ref < 0).any()
    torch.testing.assert_close(out.float().cpu().permute(0, 3, 1, 2), ref, rtol=2e-3, atol=2e-3)


def test_conv_broadcast_input_over_samples():
    """suffix launch reading a deterministic tensor: image n reads input n % B, residual n % B."""
    cin, cout, H, k, s, p = SHAPES["D3"]
    B, tc = 3, 4
    x, w, scale, bias, g = _conv_inputs(cin, cout, H, k, B, 9, True)
    res = torch.randn(B, 8, 8, cout, generator=g).to(torch.float16).to(DEV)
    out = gh.run_conv(x, w, scale, bias, res, True, s, p, B * tc, B, B, batch=B)
    ref = gh.conv_ref(x, w, scale, bias, res, True, s, p, B * tc, B, B)
    torch.testing.assert_close(out.float().cpu().permute(0, 3, 1, 2), ref, rtol=2e-3, atol=2e-3)
    assert torch.equal(out[:B], out[B:2 * B])


@pytest.mark.parametrize("kind", ["elementwise", "channel", "masksemble"])
def test_conv_fused_site(kind):
    """fused mask + MFMA conv: the epilogue's site must reproduce the oracle's mask bit for bit
    (every dropped element is exactly 0, every kept one is scaled)."""
    cin, cout, H, k, s, p = SHAPES["S3"]
    B, tc, t0, seed, cnt0 = 3, 3, 5, (7 << 32) + 42, 2
    x, w, scale, bias, g = _conv_inputs(cin, cout, H, k, B * tc, 21, False)
    if kind == "elementwise":
        site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=4, p=0.25)
    elif kind == "channel":
        site = dict(kind=_lib.SITE_CHANNEL, site_id=2, p=0.5)
    else:
        rng = np.random.RandomState(0)
        site = dict(kind=_lib.SITE_MASKSEMBLE, site_id=1, masks=(rng.rand(4, cout) < 0.4).astype(np.float32))
    out = gh.run_conv(x, w, scale, bias, None, True, s, p, B * tc, B * tc, 1, site=site, batch=B, t0=t0, seed=seed, cnt0=cnt0)
    ref = gh.conv_ref(x, w, scale, bias, None, True, s, p, B * tc, B * tc, 1)
    mult = gh.folded_site_mask(site, B, cout, 8, 8, tc, t0, seed, cnt0)
    got = out.float().cpu().permute(0, 3, 1, 2)
    torch.testing.assert_close(got, ref * mult, rtol=2e-3, atol=2e-3)
    assert torch.equal(got[mult == 0], torch.zeros_like(got[mult == 0]))
    assert (mult == 0).any() and (mult != 0).any()


def test_conv_rejects_unsupported_shapes():
    lib = _lib.lib()
    x = torch.zeros(1, 8, 8, 48, dtype=torch.float16, device=DEV)
    w = torch.zeros(64, 3, 3, 48, dtype=torch.float16, device=DEV)
    o = torch.zeros(1, 8, 8, 64, dtype=torch.float16, device=DEV)
    rc = lib.bmi_conv_igemm_fwd(gh.ptr(x), None, 1.0, gh.ptr(w), None, None, None, gh.ptr(o), 1, 1, 1, 8, 8, 48, 64, 3, 1, 1, 0, None, 1,
                                0, 0, 0, gh.stream())
    assert rc == -95


def test_stem_conv():
    lib = _lib.lib()
    g = _gen(3)
    n = 5
    x = torch.randn(n, 3, 32, 32, generator=g).to(DEV)
    w = (torch.randn(64, 3, 3, 3, generator=g) * 0.3)           # [Cout][ky][kx][Cin]
    scale = (0.5 + torch.rand(64, generator=g))
    bias = 0.2 * torch.randn(64, generator=g)
    out = torch.empty(n, 32, 32, 64, dtype=torch.float16, device=DEV)
    wd, sd, bd = w.to(DEV), scale.to(DEV), bias.to(DEV)       # keep the device copies alive across the launch
    _lib.check(lib.bmi_stem_conv_fwd(gh.ptr(x), gh.ptr(wd), gh.ptr(sd), gh.ptr(bd), gh.ptr(out),
                                     n, 3, 32, 32, 64, 3, 1, 1, 0, gh.stream()), "bmi_stem_conv_fwd")
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.cpu(), w.permute(0, 3, 1, 2), padding=1) * scale[None, :, None, None] + bias[None, :, None, None]
    assert (ref < 0).any()
    torch.testing.assert_close(out.float().cpu().permute(0, 3, 1, 2), ref, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("kind", ["elementwise", "masksemble"])
def test_mask_apply_expands_and_masks(kind):
    lib = _lib.lib()
    B, tc, H, Cc, t0, seed, cnt0 = 3, 4, 6, 64, 2, 99, 3
    g = _gen(4)
    x = torch.randn(B, H, H, Cc, generator=g).to(torch.float16).to(DEV)
    if kind == "elementwise":
        site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=0, p=0.375)
    else:
        site = dict(kind=_lib.SITE_MASKSEMBLE, site_id=0, masks=(np.random.RandomState(1).rand(4, Cc) < 0.5).astype(np.float32))
    keep = []
    s = gh.site_struct(site, keep)
    out = torch.empty(B * tc, H, H, Cc, dtype=torch.float16, device=DEV)
    _lib.check(lib.bmi_mask_apply(gh.ptr(x), gh.ptr(out), B * tc, B, H * H, Cc, C.byref(s), B, t0, seed, cnt0, gh.stream()),
               "bmi_mask_apply")
    torch.cuda.synchronize()
    mult = gh.folded_site_mask(site, B, Cc, H, H, tc, t0, seed, cnt0)
    xin = x.float().cpu().permute(0, 3, 1, 2).repeat(tc, 1, 1, 1)
    ref = (xin * mult).to(torch.float16).float()
    assert torch.equal(out.float().cpu().permute(0, 3, 1, 2), ref)        # one rounding, bit-exact


@pytest.mark.parametrize("p", [0.0, 0.25, 0.5, 0.75, 1.0])
def test_two_bit_fast_paths_cover_every_threshold(p):
    """2 bits per element (p * 4 an integer): thresholds 0..3 and drop-all.  The fast kernels — mask_apply_lb1 (whole 4096-element
    super-blocks) and mask_bits_call<1> — must equal the oracle's mask for each, p = 0 (threshold 0: keep all) included."""
    lib = _lib.lib()
    B, tc, H, Cc, t0, seed = 4, 3, 8, 64, 1, 5
    x = torch.randn(B, H, H, Cc, generator=_gen(6)).to(torch.float16).to(DEV)
    site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=2, p=p)
    keep = []
    s = gh.site_struct(site, keep)
    out = torch.empty(B * tc, H, H, Cc, dtype=torch.float16, device=DEV)
    _lib.check(lib.bmi_mask_apply(gh.ptr(x), gh.ptr(out), B * tc, B, H * H, Cc, C.byref(s), B, t0, seed, 0, gh.stream()), "bmi_mask_apply")
    bits = torch.zeros(B * tc * H * H * Cc // 8, dtype=torch.uint8, device=DEV)
    _lib.check(lib.bmi_mask_bits(gh.ptr(bits), B * tc, H * H, Cc, C.byref(s), B, t0, seed, gh.stream()), "bmi_mask_bits")
    torch.cuda.synchronize()
    mult = gh.folded_site_mask(site, B, Cc, H, H, tc, t0, seed)
    ref = (x.float().cpu().permute(0, 3, 1, 2).repeat(tc, 1, 1, 1) * mult).to(torch.float16).float()
    assert torch.equal(out.float().cpu().permute(0, 3, 1, 2), ref)
    want = np.concatenate([philox.keep_bits(B * H * H * Cc, seed, 2, t0 + tl, p) for tl in range(tc)])
    assert np.array_equal(np.unpackbits(bits.cpu().numpy(), bitorder="little").astype(bool), want)
    if p == 0.0:
        assert want.all()
    if p == 1.0:
        assert not want.any()


def _head_ref(x_nhwk, in_mod, B, tc, t0, seed, w, b, out_dim, site, site_logits, cnt0=0):
    """float64 reference of one exit head over tc samples: returns (S1, S2, SL) [B, out_dim] and per-sample logits."""
    N = B * tc
    K = x_nhwk.shape[-1]
    pooled = torch.relu(x_nhwk.float().cpu()).mean(dim=(1, 2))[torch.arange(N) % in_mod]
    if site is not None:
        pooled = pooled * gh.folded_site_mask(site, B, K, 1, 1, tc, t0, seed, cnt0).reshape(N, K)
    logits = pooled.double() @ w[:out_dim].double().T + b.double()
    lmult = None
    if site_logits is not None:
        lmult = gh.folded_site_mask(site_logits, B, out_dim, 1, 1, tc, t0, seed).reshape(N, out_dim).double()
        logits = logits * lmult
    p = torch.softmax(logits, 1).reshape(tc, B, out_dim)
    l = logits.reshape(tc, B, out_dim)
    return p.sum(0), (p * p).sum(0), l.sum(0), lmult


@pytest.mark.parametrize("out_dim,K,HW,tc,site_kind,in_kind,det", [
    (10, 512, 16, 3, "elementwise", "f16", False),       # the ResNet-18 exit heads (4x4 maps, exit dropout)
    (100, 512, 16, 37, "masksemble", "f16", False),      # C = 100 (4 class tiles), two sample groups (32 + 5), Masksembles1D
    (10, 512, 1, 70, "elementwise", "f32", False),       # VGG-11: fp32 input from a dense layer, HW = 1, three groups
    (10, 2048, 16, 5, None, "f16", True),                # ResNet-50 final head: K in four LDS chunks; deterministic input (in_mod = B)
    (100, 256, 4, 33, "channel", "bf16", False),         # bf16 bits, 2x2 maps, K < one chunk
    (10, 96, 1, 2, "logits", "f16", False),              # dropout on the LOGITS (converter/pytorch rule); K = 96: 7-chunk swizzle
])
def test_head_fused(out_dim, K, HW, tc, site_kind, in_kind, det):
    """bmi_head_fused = pool + site + Linear + softmax + float64 moment sums in one launch, against a float64 reference;
    the site masks are bit-exact (a dropped logit is exactly 0), the call accumulates (+=)."""
    lib = _lib.lib()
    B, t0, seed, cnt0 = 5, 4, (3 << 32) + 1234, 2
    N = B * tc
    in_mod = B if det else N
    g = _gen(8)
    h = int(HW ** 0.5)
    x = torch.randn(in_mod, h, h, K, generator=g)
    tdt = dict(f16=torch.float16, bf16=torch.bfloat16, f32=torch.float32)[in_kind]
    xd = x.to(tdt).to(DEV)
    w = torch.zeros((out_dim + 31) // 32 * 32, K)
    w[:out_dim] = (2.0 / K ** 0.5) * torch.randn(out_dim, K, generator=g)
    b = 0.2 * torch.randn(out_dim, generator=g)
    wd, bd = w.to(DEV), b.to(DEV)
    site = site_logits = None
    if site_kind == "elementwise":
        site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=6, p=0.25)
    elif site_kind == "channel":
        site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=3, p=0.5)            # on a [B, K] tensor channel-wise IS elementwise
    elif site_kind == "masksemble":
        site = dict(kind=_lib.SITE_MASKSEMBLE, site_id=1, masks=(torch.rand(4, K, generator=g) < 0.5).float().numpy())
    elif site_kind == "logits":
        site_logits = dict(kind=_lib.SITE_ELEMENTWISE, site_id=7, p=0.25)
    keep = []
    s1 = gh.site_struct(dict(site, kind=_lib.SITE_CHANNEL) if site_kind == "channel" else site, keep)
    s2 = gh.site_struct(site_logits, keep)
    S = torch.zeros(3, B, out_dim, dtype=torch.float64, device=DEV)
    if in_kind == "bf16":
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_BF16)
    try:
        for _ in range(2):   # accumulates (+=)
            _lib.check(lib.bmi_head_fused(gh.ptr(xd), int(in_kind == "f32"), in_mod, HW, K, gh.ptr(wd), gh.ptr(bd), out_dim,
                                          C.byref(s1) if s1 is not None else None, C.byref(s2) if s2 is not None else None, B, t0, tc,
                                          seed, cnt0, gh.ptr(S[0]), gh.ptr(S[1]), gh.ptr(S[2]), gh.stream()), "bmi_head_fused")
        torch.cuda.synchronize()
    finally:
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_F16)
    r1, r2, rl, lmult = _head_ref(xd, in_mod, B, tc, t0, seed, w, b, out_dim, site, site_logits, cnt0)
    torch.testing.assert_close(S[0].cpu(), 2 * r1, rtol=2e-5, atol=2e-6 * tc)      # fp32 pool / Linear / softmax vs float64
    torch.testing.assert_close(S[1].cpu(), 2 * r2, rtol=2e-5, atol=2e-6 * tc)
    torch.testing.assert_close(S[2].cpu(), 2 * rl, rtol=2e-5, atol=2e-5 * tc)
    torch.testing.assert_close(S[0].sum(-1).cpu(), torch.full((B,), 2.0 * tc, dtype=torch.float64), rtol=0, atol=1e-5 * tc)
    if tc == 2 and lmult is not None:      # every sample of an image dropped the logit -> its sum is exactly 0
        dead = (lmult.reshape(tc, B, out_dim) == 0).all(0)
        assert dead.any() and torch.equal(S[2].cpu()[dead], torch.zeros(int(dead.sum()), dtype=torch.float64))
    assert lib.bmi_head_fused(gh.ptr(xd), int(in_kind == "f32"), in_mod, HW, 100, gh.ptr(wd), gh.ptr(bd), out_dim, None, None, B, t0, tc,
                              seed, cnt0, gh.ptr(S[0]), gh.ptr(S[1]), gh.ptr(S[2]), gh.stream()) == -95       # K % 32 != 0


def test_head_fused_is_chunking_invariant_and_finalize():
    """Per-sample values do not depend on how the samples are grouped: one call over 40 samples equals 3 + 32 + 5 to the
    float64 summation order; bmi_finalize turns the sums into mean / variance (ddof = 0) / mean logit."""
    lib = _lib.lib()
    B, K, out_dim, tc, seed = 3, 512, 10, 40, 9
    g = _gen(5)
    x = torch.randn(B * tc, 4, 4, K, generator=g).half().to(DEV)
    w = torch.zeros(32, K)
    w[:out_dim] = 0.1 * torch.randn(out_dim, K, generator=g)
    wd, bd = w.to(DEV), (0.1 * torch.randn(out_dim, generator=g)).to(DEV)
    keep = []
    s = gh.site_struct(dict(kind=_lib.SITE_ELEMENTWISE, site_id=2, p=0.25), keep)

    def run(spans):
        S = torch.zeros(3, B, out_dim, dtype=torch.float64, device=DEV)
        for lo, n in spans:
            xs = x[lo * B:(lo + n) * B]
            _lib.check(lib.bmi_head_fused(gh.ptr(xs), 0, B * n, 16, K, gh.ptr(wd), gh.ptr(bd), out_dim, C.byref(s), None, B, lo, n, seed, 0,
                                          gh.ptr(S[0]), gh.ptr(S[1]), gh.ptr(S[2]), gh.stream()), "bmi_head_fused")
        torch.cuda.synchronize()
        return S
    Sa, Sb = run([(0, 40)]), run([(0, 3), (3, 32), (35, 5)])
    torch.testing.assert_close(Sa.cpu(), Sb.cpu(), rtol=1e-13, atol=1e-13)
    torch.testing.assert_close(run([(0, 40)]).cpu(), Sa.cpu(), rtol=1e-14, atol=1e-14)   # (f64 atomics: order of <= 2 adds varies)
    out = torch.empty(3, B, out_dim, dtype=torch.float64, device=DEV)
    _lib.check(lib.bmi_finalize(B * out_dim, tc, gh.ptr(Sa[0]), gh.ptr(Sa[1]), gh.ptr(Sa[2]), gh.ptr(out[0]), gh.ptr(out[1]),
                                gh.ptr(out[2]), gh.stream()), "bmi_finalize")
    torch.cuda.synchronize()
    torch.testing.assert_close(out[0].cpu(), Sa[0].cpu() / tc, rtol=1e-14, atol=0)
    torch.testing.assert_close(out[1].cpu(), (Sa[1].cpu() / tc - (Sa[0].cpu() / tc) ** 2).clamp(min=0), rtol=1e-9, atol=1e-15)
    torch.testing.assert_close(out[2].cpu(), Sa[2].cpu() / tc, rtol=1e-14, atol=0)


def test_maxpool2():
    lib = _lib.lib()
    x = torch.randn(3, 8, 8, 64, generator=_gen(2)).to(torch.float16).to(DEV)
    out = torch.empty(3, 4, 4, 64, dtype=torch.float16, device=DEV)
    _lib.check(lib.bmi_maxpool2(gh.ptr(x), gh.ptr(out), 3, 8, 8, 64, gh.stream()), "bmi_maxpool2")
    torch.cuda.synchronize()
    ref = torch.nn.functional.max_pool2d(x.float().cpu().permute(0, 3, 1, 2), 2)
    assert torch.equal(out.float().cpu().permute(0, 3, 1, 2), ref)


def test_mask_bits_and_masked_input_conv():
    """Input-side MC-dropout: keep bits (bit-exact vs the oracle mask) + a stride-2 conv that reads the
    un-expanded deterministic tensor and applies the bits while staging == conv on the materialised
    x * keep / (1 - p)."""
    lib = _lib.lib()
    B, tc, H, cin, cout, t0, seed, p = 3, 3, 16, 64, 128, 4, (5 << 32) + 9, 0.25
    N = B * tc
    site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=0, p=p)
    keep = []
    s = gh.site_struct(site, keep)
    bits = torch.zeros(N * H * H * cin // 8, dtype=torch.uint8, device=DEV)
    _lib.check(lib.bmi_mask_bits(gh.ptr(bits), N, H * H, cin, C.byref(s), B, t0, seed, gh.stream()), "bmi_mask_bits")
    torch.cuda.synchronize()
    got_bits = np.unpackbits(bits.cpu().numpy(), bitorder="little").astype(bool)
    want = np.concatenate([philox.keep_bits(B * H * H * cin, seed, 0, t0 + tl, p) for tl in range(tc)])
    assert np.array_equal(got_bits, want)

    x, w, scale, bias, g = _conv_inputs(cin, cout, H, 3, B, 31, False)
    out = gh.run_conv(x, w, scale, bias, None, True, 2, 1, N, B, 1, batch=B, in_bits=bits,
                      out_mul=float(philox.drop_scale(p)))
    mult = gh.folded_site_mask(site, B, cin, H, H, tc, t0, seed)                 # keep * 1/(1-p), [N,C,H,W]
    xin = (x.float().cpu().permute(0, 3, 1, 2).repeat(tc, 1, 1, 1) * mult)
    ref = torch.nn.functional.conv2d(xin, w.float().cpu().permute(0, 3, 1, 2), stride=2, padding=1)
    ref = torch.relu(ref * scale.cpu()[None, :, None, None] + bias.cpu()[None, :, None, None])
    torch.testing.assert_close(out.float().cpu().permute(0, 3, 1, 2), ref, rtol=2e-3, atol=2e-3)
    # 1x1 stride-2 (the downsample conv) takes the same path
    w1 = (torch.randn(cout, 1, 1, cin, generator=g) * 0.2).to(torch.float16).to(DEV)
    out1 = gh.run_conv(x, w1, scale, bias, None, False, 2, 0, N, B, 1, batch=B, in_bits=bits, out_mul=float(philox.drop_scale(p)))
    ref1 = torch.nn.functional.conv2d(xin, w1.float().cpu().permute(0, 3, 1, 2), stride=2)
    ref1 = ref1 * scale.cpu()[None, :, None, None] + bias.cpu()[None, :, None, None]
    torch.testing.assert_close(out1.float().cpu().permute(0, 3, 1, 2), ref1, rtol=2e-3, atol=2e-3)


@pytest.fixture(params=[0, 2], ids=["patch", "pw"])
def pw_mode(request):
    _lib.set_option("conv_pw", request.param)
    yield request.param
    _lib.set_option("conv_pw", 1)


@pytest.mark.parametrize("name,cin2", [("S2", 64), ("S3", 128), ("S4", 256)])
def test_conv3x3_with_fused_shortcut(name, cin2, pw_mode):
    """BasicBlock tail with downsample in ONE launch: relu(conv3x3(a) + conv1x1_s2(x) + bias); in conv3x3_patch and (8x8 /
    4x4 maps, Cout % 256 == 0) in conv3x3_pw."""
    lib = _lib.lib()
    cin, cout, H, _, _, _ = SHAPES[name]
    n = 5 if pw_mode == 0 else 21
    g = _gen(77)
    a_in = torch.randn(n, H, H, cin, generator=g).to(torch.float16).to(DEV)
    x_in = torch.randn(n, 2 * H, 2 * H, cin2, generator=g).to(torch.float16).to(DEV)
    w = (torch.randn(cout, 3, 3, cin, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(torch.float16).to(DEV)
    w2 = (torch.randn(cout, cin2, generator=g) * (1.0 / cin2) ** 0.5).to(torch.float16).to(DEV)
    bias = (0.2 * torch.randn(cout, generator=g)).to(DEV)
    out = torch.full((n, H, H, cout), float("nan"), dtype=torch.float16, device=DEV)
    _lib.check(lib.bmi_conv3x3_shortcut_fwd(gh.ptr(a_in), gh.ptr(w), gh.ptr(x_in), gh.ptr(w2), gh.ptr(bias), gh.ptr(out), n, H, H,
                                            cin, cout, cin2, 1, gh.stream()), "bmi_conv3x3_shortcut_fwd")
    torch.cuda.synchronize()
    F_ = torch.nn.functional
    ref = F_.conv2d(a_in.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), padding=1)
    ref = ref + F_.conv2d(x_in.float().cpu().permute(0, 3, 1, 2), w2.float().cpu()[:, :, None, None], stride=2)
    ref = torch.relu(ref + bias.cpu()[None, :, None, None])
    torch.testing.assert_close(out.float().cpu().permute(0, 3, 1, 2), ref, rtol=2e-3, atol=3e-3)


@pytest.mark.parametrize("name,n", [("D3", 801), ("D4", 1601), ("P4", 1601), ("D3", 5)])
def test_conv_wide_tile_kernel_tails_residual_and_site(name, n):
    """Cout % 256 == 0 shapes with at least 192 tiles run in the 256 x 256-tile LDS-DMA kernel (conv_igemm_wide.hip;
    smaller grids fall back to conv_igemm's 128 x 128 tiles: the n = 5 case): ragged pixel tiles (M = n*Ho*Wo is not a
    multiple of 256), residual, ReLU and a fused elementwise site, bit-exact mask."""
    cin, cout, H, k, s, p = SHAPES[name]
    B, tc, t0, seed = n, 1, 3, 99
    x, w, scale, bias, g = _conv_inputs(cin, cout, H, k, n, 31, True)
    ho = (H + 2 * p - k) // s + 1
    assert (n * ho * ho) % 256 != 0
    res = torch.randn(n, ho, ho, cout, generator=g).to(torch.float16).to(DEV)
    site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=5, p=0.25)
    out = gh.run_conv(x, w, scale, bias, res, True, s, p, n, n, n, site=site, batch=B, t0=t0, seed=seed)
    ref = gh.conv_ref(x, w, scale, bias, res, True, s, p, n, n, n)
    mult = gh.folded_site_mask(site, B, cout, ho, ho, tc, t0, seed)
    got = out.float().cpu().permute(0, 3, 1, 2)
    torch.testing.assert_close(got, ref * mult, rtol=2e-3, atol=3e-3)
    assert torch.equal(got[mult == 0], torch.zeros_like(got[mult == 0]))


@pytest.mark.parametrize("cin,ca,cb,H,k,s,p,n,in_mod", [
    (64, 128, 128, 32, 3, 2, 1, 3, 3),        # layer2[0].conv1 + ex1conv1
    (128, 256, 256, 16, 3, 2, 1, 6, 2),       # layer3[0].conv1 + ex2conv1, input broadcast over 3 samples
    (256, 512, 512, 8, 3, 2, 1, 5, 5),        # layer4[0].conv1 + ex3conv1
    (64, 128, 384, 8, 1, 1, 0, 9, 9),         # unequal halves, 1x1
])
def test_conv_pair_one_launch_two_outputs(cin, ca, cb, H, k, s, p, n, in_mod):
    lib = _lib.lib()
    g = _gen(5)
    x = torch.randn(in_mod, H, H, cin, generator=g).to(torch.float16).to(DEV)
    ho = (H + 2 * p - k) // s + 1
    ws, scs, bis, outs = [], [], [], []
    for c in (ca, cb):
        ws.append((torch.randn(c, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(torch.float16).to(DEV))
        scs.append((0.5 + torch.rand(c, generator=g)).to(DEV))
        bis.append((0.2 * torch.randn(c, generator=g)).to(DEV))
        outs.append(torch.full((n, ho, ho, c), float("nan"), dtype=torch.float16, device=DEV))
    rc = lib.bmi_conv_pair_fwd(gh.ptr(x), gh.ptr(ws[0]), gh.ptr(scs[0]), gh.ptr(bis[0]), gh.ptr(outs[0]), gh.ptr(ws[1]), gh.ptr(scs[1]),
                               gh.ptr(bis[1]), gh.ptr(outs[1]), n, in_mod, H, H, cin, ca, cb, k, s, p, 1, gh.stream())
    _lib.check(rc, "bmi_conv_pair_fwd")
    torch.cuda.synchronize()
    for i in range(2):
        ref = gh.conv_ref(x, ws[i], scs[i], bis[i], None, True, s, p, n, in_mod, 1)
        torch.testing.assert_close(outs[i].float().cpu().permute(0, 3, 1, 2), ref, rtol=2e-3, atol=3e-3)
        # and identical to the same conv launched alone
        alone = gh.run_conv(x, ws[i], scs[i], bis[i], None, True, s, p, n, in_mod, 1)
        assert torch.equal(alone, outs[i])


@pytest.mark.parametrize("name,n,shortcut,dt", [("S3", 1031, False, "f16"), ("S4", 4101, False, "f16"), ("S3", 1500, True, "f16"), ("S4", 4500, True, "f16"),
                                                ("S3", 2050, False, "bf16"), ("S4", 5000, True, "bf16"), ("S4", 37, False, "f16")])
def test_persistent_pw_equals_the_per_tile_kernel_bit_for_bit(name, n, shortcut, dt):
    """conv3x3_pwp (round 4): conv3x3_pw's plain-epilogue launches as ONE persistent workgroup per CU whose main loop continues across tiles (the
    last chunk of a tile prefetches the next tile's first weight stages and sub-patch, the epilogue runs in two 32 KB rounds beside them).  Same K
    order and arithmetic: identical bits — several tiles per CU (1031 / 4 = 258 tiles ... 4500 / 16 x 2), ragged last tiles, with and without the
    fused shortcut, fp16 and bf16, and fewer tiles than CUs (n = 37)."""
    lib = _lib.lib()
    cin, cout, H, k, s, p = SHAPES[name]
    g = _gen(17)
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    x = torch.randn(n, H, H, cin, generator=g).to(tdt).to(DEV)
    w = (torch.randn(cout, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(tdt).to(DEV)
    scale, bias = (0.5 + torch.rand(cout, generator=g)).to(DEV), (0.2 * torch.randn(cout, generator=g)).to(DEV)
    x2 = torch.randn(n, 2 * H, 2 * H, cin // 2, generator=g).to(tdt).to(DEV) if shortcut else None
    w2 = (torch.randn(cout, cin // 2, generator=g) * (2.0 / cin) ** 0.5).to(tdt).to(DEV) if shortcut else None
    if dt == "bf16":
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_BF16)
    _lib.set_option("conv_pw", 2)                       # (no minimum-grid rule: n = 37 too)
    outs = []
    try:
        for persist in (0, 1):
            _lib.set_option("pw_persist", persist)
            out = torch.full((n, H, H, cout), float("nan"), dtype=tdt, device=DEV)
            if shortcut:
                _lib.check(lib.bmi_conv3x3_shortcut_fwd(gh.ptr(x), gh.ptr(w), gh.ptr(x2), gh.ptr(w2), gh.ptr(bias), gh.ptr(out), n, H, H, cin, cout, cin // 2, 1,
                                                        gh.stream()), "bmi_conv3x3_shortcut_fwd")
            else:
                _lib.check(lib.bmi_conv_igemm_fwd(gh.ptr(x), None, 1.0, gh.ptr(w), gh.ptr(scale), gh.ptr(bias), None, gh.ptr(out), n, n, n, H, H, cin, cout, k, s, p, 1,
                                                  None, n, 0, 0, 0, gh.stream()), "bmi_conv_igemm_fwd")
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        _lib.set_option("pw_persist", 1)
        _lib.set_option("conv_pw", 1)
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_F16)
    assert torch.isfinite(outs[0].float()).all() and float(outs[0].float().abs().max()) > 0
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("name,n,dt", [("S3", 1031, "f16"), ("S4", 4101, "f16"), ("S3", 2050, "bf16"), ("S4", 5000, "bf16"), ("S4", 37, "f16")])
@pytest.mark.parametrize("use_site", [0, 1, 2])
def test_persistent_pw_basicblock_tails_equal_the_lite_epilogue_bit_for_bit(name, n, dt, use_site):
    """conv3x3_pwp with EPIK = LITE_RES / LITE_RES_MC (round 4): the BasicBlock tails (BN + residual + ReLU [+ the 2-bit elementwise site]) on the
    persistent walk, finished straight from the registers in the channel-permuted accumulator layout (a lane fetches its own residual runs, the
    Philox words are exchanged inside a pixel column) — against conv3x3_pw_kernel's lite epilogue ("pw_persist" = 0): identical bits, ragged last
    tiles, several tiles per CU and fewer tiles than CUs, fp16 and bf16; and against the general epilogue ("epilogue_lite" = 0)."""
    cin, cout, H, k, s, p = SHAPES[name]
    g = _gen(23)
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    x = torch.randn(n, H, H, cin, generator=g).to(tdt).to(DEV)
    w = (torch.randn(cout, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(tdt).to(DEV)
    scale, bias = (0.5 + torch.rand(cout, generator=g)).to(DEV), (0.2 * torch.randn(cout, generator=g)).to(DEV)
    res = torch.randn(n, H, H, cout, generator=g).to(tdt).to(DEV)
    site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=2, p=0.25) if use_site == 1 else None
    if use_site == 2:      # Masksembles2D: per-channel multipliers of mask (cnt0 + t) mod M, t = image // batch
        site = dict(kind=_lib.SITE_MASKSEMBLE, site_id=1, masks=(torch.rand(4, cout, generator=g) < 0.6).float().numpy() * 1.5)
    if dt == "bf16":
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_BF16)
    _lib.set_option("conv_pw", 2)
    outs = []
    try:
        for persist, lite in ((0, 1), (1, 1), (0, 0)):
            _lib.set_option("pw_persist", persist)
            _lib.set_option("epilogue_lite", lite)
            outs.append(gh.run_conv(x, w, scale, bias, res, True, s, p, n, n, n, site=site, batch=7, t0=3, seed=9, cnt0=2, out_dtype=tdt))
    finally:
        _lib.set_option("pw_persist", 1)
        _lib.set_option("epilogue_lite", 1)
        _lib.set_option("conv_pw", 1)
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_F16)
    assert torch.isfinite(outs[0].float()).all() and float(outs[0].float().abs().max()) > 0
    if use_site:
        assert float((outs[0] == 0).float().mean()) > 0.25            # dropped elements (and ReLU zeros)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    assert torch.equal(outs[0].view(torch.int16), outs[2].view(torch.int16))


def test_conv_pair_rejects_bad_splits():
    lib = _lib.lib()
    z = torch.zeros(1, 8, 8, 64, dtype=torch.float16, device=DEV)
    w = torch.zeros(192, 3, 3, 64, dtype=torch.float16, device=DEV)
    o = torch.zeros(1, 8, 8, 192, dtype=torch.float16, device=DEV)
    f = torch.zeros(192, device=DEV)
    args = lambda ca, cb: (gh.ptr(z), gh.ptr(w), gh.ptr(f), gh.ptr(f), gh.ptr(o), gh.ptr(w), gh.ptr(f), gh.ptr(f), gh.ptr(o), 1, 1, 8, 8, 64,
                           ca, cb, 3, 1, 1, 1, gh.stream())
    assert lib.bmi_conv_pair_fwd(*args(64, 192)) == -22        # first half must be a multiple of 128 channels
    assert lib.bmi_conv_pair_fwd(*args(128, 64)) == -95        # total not a multiple of the 256-channel tile


@pytest.mark.parametrize("name,kwargs", [("S2", dict(with_res=True, with_site=True)), ("S2", dict(with_res=False, with_site=False)),
                                         ("D3", dict(with_res=False, with_site=False)), ("S4", dict(with_res=True, with_site=False))])
def test_conv_kernels_are_not_pathologically_slow(name, kwargs):
    """Coarse speed floor (a 10x regression guard, not a benchmark): the epilogue once fell into scratch memory and
    the parity tests still passed.  Healthy kernels run these shapes at 500-1000 TFLOP/s; the floor is 150."""
    cin, cout, H, k, s, p = SHAPES[name]
    n = 4000
    x, w, scale, bias, g = _conv_inputs(cin, cout, H, k, n, 3, False)
    ho = (H + 2 * p - k) // s + 1
    res = torch.randn(n, ho, ho, cout, generator=g).to(torch.float16).to(DEV) if kwargs["with_res"] else None
    site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=1, p=0.25) if kwargs["with_site"] else None
    run = lambda: gh.run_conv(x, w, scale, bias, res, True, s, p, n, n, n, site=site, batch=250, seed=1)
    run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record()
    torch.cuda.synchronize()
    tflops = 3 * 2.0 * n * ho * ho * cout * k * k * cin / (e0.elapsed_time(e1) * 1e-3) / 1e12
    assert tflops > 150, f"{name} {kwargs}: {tflops:.0f} TFLOP/s"


@pytest.mark.parametrize("p,in_stoch", [(0.25, False), (0.25, True), (0.5, False), (0.375, False), (1 / 256, True), (0.2, False)])
def test_mask_apply_lane_shared_philox(p, in_stoch):
    """Sites with 2 / 4 / 8 bits per element run the kernel in which a wave's lanes share Philox calls (one call masks
    64 / 32 / 16 elements); p = 0.2 (16 bits) and ragged sizes take the per-item kernel.  Bit-exact against the oracle
    either way, for an expanding launch (input [B]) and a same-size one (input [tc*B])."""
    lib = _lib.lib()
    B, tc, H, Cc, t0, seed = 4, 3, 8, 64, 5, (9 << 32) + 1            # 4*64*64 = 16384 elements per sample
    g = _gen(6)
    n_in = B * tc if in_stoch else B
    x = torch.randn(n_in, H, H, Cc, generator=g).to(torch.float16).to(DEV)
    site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=3, p=p)
    keep = []
    s = gh.site_struct(site, keep)
    out = torch.full((B * tc, H, H, Cc), float("nan"), dtype=torch.float16, device=DEV)
    _lib.check(lib.bmi_mask_apply(gh.ptr(x), gh.ptr(out), B * tc, n_in, H * H, Cc, C.byref(s), B, t0, seed, 0, gh.stream()), "bmi_mask_apply")
    torch.cuda.synchronize()
    mult = gh.folded_site_mask(site, B, Cc, H, H, tc, t0, seed)
    xin = x.float().cpu().permute(0, 3, 1, 2)
    if not in_stoch:
        xin = xin.repeat(tc, 1, 1, 1)
    ref = (xin * mult).to(torch.float16).float()
    assert torch.equal(out.float().cpu().permute(0, 3, 1, 2), ref)
    assert 0 < float((mult == 0).float().mean()) < 1


@pytest.mark.parametrize("name,n", [("D3", 2111), ("D4", 4500), ("P4", 9001)])
def test_conv_wide_persistent_kernel(name, n, request):
    """More than 2 tiles per CU: the plain wide-tile launches run the persistent kernel (one workgroup per CU walks over
    the tiles, the next tile's first K-step lands during the epilogue).  Ragged last tile, bit-exact repeatability."""
    cin, cout, H, k, s, p = SHAPES[name]
    _lib.set_option("conv_s2", 0)                # (the plain 3x3 stride-2 launches are conv3x3_s2's by default: tests/test_conv3x3_s2.py)
    request.addfinalizer(lambda: _lib.set_option("conv_s2", 1))
    x, w, scale, bias, g = _conv_inputs(cin, cout, H, k, n, 17, False)
    ho = (H + 2 * p - k) // s + 1
    tiles = -(-n * ho * ho // 256) * (cout // 256)
    assert tiles > 512 and (n * ho * ho) % 256 != 0
    out = gh.run_conv(x, w, scale, bias, None, True, s, p, n, n, 1)
    ref = gh.conv_ref(x, w, scale, bias, None, True, s, p, n, n, 1)
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=3e-3)
    assert torch.equal(out, gh.run_conv(x, w, scale, bias, None, True, s, p, n, n, 1))


def test_conv_pair_persistent_kernel(request):
    lib = _lib.lib()
    _lib.set_option("conv_s2", 0)                # conv_igemm_wide's pair mode (conv3x3_s2's: tests/test_conv3x3_s2.py)
    request.addfinalizer(lambda: _lib.set_option("conv_s2", 1))
    cin, ca, cb, H, k, s, p, n = 64, 128, 128, 32, 3, 2, 1, 600          # 600 tiles of 256 pixels
    g = _gen(8)
    x = torch.randn(n, H, H, cin, generator=g).to(torch.float16).to(DEV)
    ws, scs, bis, outs = [], [], [], []
    for c in (ca, cb):
        ws.append((torch.randn(c, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(torch.float16).to(DEV))
        scs.append((0.5 + torch.rand(c, generator=g)).to(DEV))
        bis.append((0.2 * torch.randn(c, generator=g)).to(DEV))
        outs.append(torch.full((n, 16, 16, c), float("nan"), dtype=torch.float16, device=DEV))
    rc = lib.bmi_conv_pair_fwd(gh.ptr(x), gh.ptr(ws[0]), gh.ptr(scs[0]), gh.ptr(bis[0]), gh.ptr(outs[0]), gh.ptr(ws[1]), gh.ptr(scs[1]),
                               gh.ptr(bis[1]), gh.ptr(outs[1]), n, n, H, H, cin, ca, cb, k, s, p, 1, gh.stream())
    _lib.check(rc, "bmi_conv_pair_fwd")
    torch.cuda.synchronize()
    for i in range(2):
        ref = gh.conv_ref(x, ws[i], scs[i], bis[i], None, True, s, p, n, n, 1)
        torch.testing.assert_close(outs[i].float().cpu().permute(0, 3, 1, 2), ref, rtol=2e-3, atol=3e-3)


@pytest.mark.parametrize("in_f32,kind,relu", [(0, "none", 1), (0, _lib.SITE_ELEMENTWISE, 1), (1, _lib.SITE_ELEMENTWISE, 1),
                                              (1, _lib.SITE_MASKSEMBLE, 0), (0, _lib.SITE_CHANNEL, 1)])
def test_dense_f32(in_f32, kind, relu):
    """BMI_OP_DENSE (hidden Dense of the VGG-11 classifier stack) in fp32 on the exact-f32 MFMA: fp16 or fp32 input,
    deterministic input broadcast over the samples (in_mod = B), bit-exact site mask, ragged N (not a multiple of 32)."""
    lib = _lib.lib()
    B, tc, K, Cout, t0, seed, cnt0 = 7, 5, 512, 256, 3, (9 << 32) + 1, 2
    N = B * tc
    g = _gen(11)
    in_mod = N if in_f32 else B
    x = torch.randn(in_mod, K, generator=g)
    xd = x.to(DEV) if in_f32 else x.half().to(DEV)
    xr = xd.cpu().double()
    w = 0.05 * torch.randn(Cout, K, generator=g)
    b = 0.3 * torch.randn(Cout, generator=g)
    wd, bd = w.to(DEV), b.to(DEV)
    out = torch.full((N, Cout), float("nan"), device=DEV)
    site = None
    if kind == _lib.SITE_MASKSEMBLE:
        site = dict(kind=kind, site_id=4, masks=(torch.rand(4, Cout, generator=g) < 0.6).float().numpy())
    elif kind != "none":
        site = dict(kind=kind, site_id=4, p=0.25)
    keep = []
    s = gh.site_struct(site, keep)
    _lib.check(lib.bmi_dense_f32(gh.ptr(xd), in_f32, gh.ptr(wd), gh.ptr(bd), gh.ptr(out), N, in_mod, K, Cout, relu,
                                 C.byref(s) if s is not None else None, B, t0, seed, cnt0, gh.stream()), "bmi_dense_f32")
    torch.cuda.synchronize()
    ref = xr[torch.arange(N) % in_mod] @ w.double().T + b.double()
    if relu:
        ref = torch.relu(ref)
    if site is not None:
        # a per-(image, channel) draw on a [B, C] tensor is the elementwise draw
        ms = dict(site, kind=_lib.SITE_ELEMENTWISE) if kind == _lib.SITE_CHANNEL else site
        mult = gh.folded_site_mask(ms, B, Cout, 1, 1, tc, t0, seed, cnt0).reshape(N, Cout).double()
        ref = ref * mult
        assert torch.equal((out.cpu() == 0) | (ref == 0), (mult == 0) | (ref == 0)) and (mult == 0).any()
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=2e-5)
    assert lib.bmi_dense_f32(gh.ptr(xd), in_f32, gh.ptr(wd), gh.ptr(bd), gh.ptr(out), N, in_mod, K, 200, relu, None, B, t0,
                             seed, cnt0, gh.stream()) == -95


@pytest.fixture
def bf16_entries():
    _lib.set_option("unit_entry_dtype", _lib.DTYPE_BF16)
    yield
    _lib.set_option("unit_entry_dtype", _lib.DTYPE_F16)


@pytest.mark.parametrize("name", ["S1", "S2", "S3", "S4", "D3", "D4", "P4"])
def test_bf16_conv_shape_classes(name, bf16_entries):
    """The bf16 instantiations (v_mfma_f32_16x16x32_bf16 in conv3x3_patch / conv_igemm_wide, 32x32x16_bf16 in conv_igemm) on
    bf16-rounded operands against the fp32 reference: only the output rounding (2^-9 relative) separates them.  With
    residual + ReLU + a fused elementwise site (general epilogue) for the S classes, plain epilogue for the others."""
    cin, cout, H, k, s, p = SHAPES[name]
    B, tc, t0, seed = 3, 2, 1, 77
    n = B * tc
    g = _gen(13)
    x = torch.randn(n, H, H, cin, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(cout, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(torch.bfloat16).to(DEV)
    scale, bias = (0.5 + torch.rand(cout, generator=g)).to(DEV), (0.2 * torch.randn(cout, generator=g)).to(DEV)
    ho = (H + 2 * p - k) // s + 1
    general = name.startswith("S")
    res = torch.randn(n, ho, ho, cout, generator=g).to(torch.bfloat16).to(DEV) if general else None
    site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=3, p=0.25) if general else None
    out = gh.run_conv(x, w, scale, bias, res, True, s, p, n, n, n, site=site, batch=B, t0=t0, seed=seed, out_dtype=torch.bfloat16)
    ref = gh.conv_ref(x, w, scale, bias, res, True, s, p, n, n, n)
    if site is not None:
        mult = gh.folded_site_mask(site, B, cout, ho, ho, tc, t0, seed)
        ref = ref * mult
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got, ref, rtol=1e-2, atol=1e-2)
    if site is not None:
        assert torch.equal(got[mult == 0], torch.zeros_like(got[mult == 0])) and (mult == 0).any()


def test_bf16_elementwise_kernels(bf16_entries):
    """mask_apply (lane-shared Philox path), maxpool2 and dense_f32 reading / writing bfloat16 bits (the head: test_head_fused)."""
    lib = _lib.lib()
    B, tc, H, Cc, t0, seed = 2, 3, 16, 64, 0, 5
    N = B * tc
    g = _gen(4)
    x = torch.randn(B, H, H, Cc, generator=g).to(torch.bfloat16).to(DEV)
    site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=0, p=0.25)
    keep = []
    s = gh.site_struct(site, keep)
    out = torch.empty(N, H, H, Cc, dtype=torch.bfloat16, device=DEV)
    _lib.check(lib.bmi_mask_apply(gh.ptr(x), gh.ptr(out), N, B, H * H, Cc, C.byref(s), B, t0, seed, 0, gh.stream()), "bmi_mask_apply")
    mult = gh.folded_site_mask(site, B, Cc, H, H, tc, t0, seed).permute(0, 2, 3, 1)
    want = (x.float().cpu()[torch.arange(N) % B] * mult).to(torch.bfloat16)
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), want)
    mp = torch.empty(N, H // 2, H // 2, Cc, dtype=torch.bfloat16, device=DEV)
    _lib.check(lib.bmi_maxpool2(gh.ptr(out), gh.ptr(mp), N, H, H, Cc, gh.stream()), "bmi_maxpool2")
    torch.cuda.synchronize()
    assert torch.equal(mp.cpu().float(), torch.nn.functional.max_pool2d(out.cpu().float().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1))
    flat = out.reshape(N, -1)[:, :512].contiguous()
    w = 0.05 * torch.randn(128, 512, generator=g)
    wd, bd = w.to(DEV), torch.zeros(128, device=DEV)        # (named: a temporary's memory is recycled by the next allocation)
    d = torch.empty(N, 128, device=DEV)
    _lib.check(lib.bmi_dense_f32(gh.ptr(flat), 0, gh.ptr(wd), gh.ptr(bd), gh.ptr(d), N, N, 512, 128, 0, None, B, 0, 0, 0,
                                 gh.stream()), "bmi_dense_f32")
    torch.cuda.synchronize()
    torch.testing.assert_close(d.cpu().double(), flat.cpu().double() @ w.double().T, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("name,n,general,dt", [("S3", 5, False, "f16"), ("S3", 37, True, "f16"), ("S4", 3, False, "f16"), ("S4", 50, True, "f16"),
                                              ("S3", 9, True, "bf16"), ("S4", 17, False, "bf16")])
def test_conv3x3_pw_kernel(name, n, general, dt):
    _lib.set_option("conv_pw", 2)            # no minimum-grid rule: these image counts are far below one tile per CU
    """conv3x3_pw (256 x 256 tile, 32-channel double-buffered sub-patches, rotated 64-byte LDS rows) on the 8x8 and 4x4 map
    classes: ragged image counts (tiles with missing images), plain and general epilogue (residual + ReLU + fused
    elementwise site with bit-exact mask), fp16 and bf16 — against the fp32 reference on the rounded operands."""
    cin, cout, H, k, s, p = SHAPES[name]
    B, seed, t0 = n, 5, 2
    tdt = torch.float16 if dt == "f16" else torch.bfloat16
    g = _gen(31)
    x = torch.randn(n, H, H, cin, generator=g).to(tdt).to(DEV)
    w = (torch.randn(cout, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(tdt).to(DEV)
    scale, bias = (0.5 + torch.rand(cout, generator=g)).to(DEV), (0.2 * torch.randn(cout, generator=g)).to(DEV)
    res = torch.randn(n, H, H, cout, generator=g).to(tdt).to(DEV) if general else None
    site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=3, p=0.25) if general else None
    if dt == "bf16":
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_BF16)
    try:
        out = gh.run_conv(x, w, scale, bias, res, True, s, p, n, n, n, site=site, batch=B, t0=t0, seed=seed, out_dtype=tdt)
        _lib.set_option("conv_pw", 0)
        out_patch = gh.run_conv(x, w, scale, bias, res, True, s, p, n, n, n, site=site, batch=B, t0=t0, seed=seed, out_dtype=tdt)
    finally:
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_F16)
        _lib.set_option("conv_pw", 1)
    ref = gh.conv_ref(x, w, scale, bias, res, True, s, p, n, n, n)
    if site is not None:
        mult = gh.folded_site_mask(site, B, cout, H, H, 1, t0, seed)
        ref = ref * mult
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    tol = 2e-3 if dt == "f16" else 1e-2
    torch.testing.assert_close(got, ref, rtol=tol, atol=tol)
    # same values as the patch kernel up to the fp32 summation order over K (different chunking of the channels)
    torch.testing.assert_close(got, out_patch.float().cpu().permute(0, 3, 1, 2), rtol=tol, atol=tol)
    if site is not None:
        assert torch.equal(got[mult == 0], torch.zeros_like(got[mult == 0])) and (mult == 0).any()


@pytest.mark.parametrize("name,n,k1,dt", [("S2", 37, False, "f16"), ("S3", 300, False, "f16"), ("S4", 1100, False, "f16"), ("S3", 900, True, "f16"),
                                          ("S2", 9, False, "bf16"), ("S4", 1050, False, "bf16")])
@pytest.mark.parametrize("use_res,use_site", [(1, 0), (0, 1), (1, 1), (1, 2), (0, 2)])
def test_lite_epilogue_equals_general_bit_for_bit(name, n, k1, dt, use_res, use_site):
    """epilogue_lite (BN on the accumulator registers, residual DMA'd into the LDS output image, results written back in
    place) against epilogue_coalesced (fp32 rounds through LDS) on BasicBlock tails: conv3x3_patch (S2), conv3x3_pw (S3 /
    S4 above the minimum grid) and the wide 1x1 kernel — the same bits, so that a launch that has no lite instantiation
    (per-tap fallback, 32x32x16 shape, dynamic-exit patch variants) still agrees with one that has."""
    cin, cout, H, k, s, p = SHAPES[name]
    if k1:
        k, p = 1, 0
    tdt = torch.float16 if dt == "f16" else torch.bfloat16
    g = _gen(77)
    x = torch.randn(n, H, H, cin, generator=g).to(tdt).to(DEV)
    w = (torch.randn(cout, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(tdt).to(DEV)
    scale, bias = (0.5 + torch.rand(cout, generator=g)).to(DEV), (0.2 * torch.randn(cout, generator=g)).to(DEV)
    res = torch.randn(n, H, H, cout, generator=g).to(tdt).to(DEV) if use_res else None
    site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=1, p=0.25) if use_site == 1 else None
    if use_site == 2:      # Masksembles2D: per-channel multipliers of mask (cnt0 + t) mod M, t = image // batch
        site = dict(kind=_lib.SITE_MASKSEMBLE, site_id=1, masks=(torch.rand(4, cout, generator=g) < 0.6).float().numpy() * 1.5)
    outs = []
    if dt == "bf16":
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_BF16)
    try:
        for lite in (0, 1, 2):     # general | lite with the residual / ReLU / site kind compiled in where such a form exists (round 4) | lite, run-time terms
            _lib.set_option("epilogue_lite", lite)
            outs.append(gh.run_conv(x, w, scale, bias, res, True, s, p, n, n, n, site=site, batch=7, t0=3, seed=9, cnt0=2, out_dtype=tdt))
    finally:
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_F16)
        _lib.set_option("epilogue_lite", 1)
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    assert torch.equal(outs[0].view(torch.int16), outs[2].view(torch.int16))


@pytest.mark.parametrize("cin,cout,H,stride,n,dt", [(128, 512, 16, 1, 37, "f16"), (256, 1024, 8, 1, 131, "f16"), (64, 256, 32, 1, 5, "bf16"),
                                                    (256, 512, 8, 2, 77, "f16"), (512, 256, 4, 1, 700, "f16")])
@pytest.mark.parametrize("use_res,use_site", [(0, 0), (1, 0), (1, 1)])
def test_conv1x1_stream_kernel(cin, cout, H, stride, n, dt, use_res, use_site):
    """conv1x1_stream (HBM-bound Bottleneck 1x1 convs: 128 x 256 tile, single-buffered K-steps, two workgroups per CU) against the
    fp32 reference, and bit for bit against the kernels it replaces (same K order, same epilogue code): ragged pixel counts,
    stride 2, plain / residual / residual + elementwise site."""
    tdt = torch.float16 if dt == "f16" else torch.bfloat16
    g = _gen(41)
    x = torch.randn(n, H, H, cin, generator=g).to(tdt).to(DEV)
    w = (torch.randn(cout, 1, 1, cin, generator=g) * (2.0 / cin) ** 0.5).to(tdt).to(DEV)
    scale, bias = (0.5 + torch.rand(cout, generator=g)).to(DEV), (0.2 * torch.randn(cout, generator=g)).to(DEV)
    ho = (H - 1) // stride + 1
    res = torch.randn(n, ho, ho, cout, generator=g).to(tdt).to(DEV) if use_res else None
    site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=2, p=0.25) if use_site else None
    if dt == "bf16":
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_BF16)
    try:
        _lib.set_option("conv_stream", 2)          # no minimum grid, plain launches too
        out = gh.run_conv(x, w, scale, bias, res, True, stride, 0, n, n, n, site=site, batch=n, t0=1, seed=3, out_dtype=tdt)
        _lib.set_option("conv_stream", 0)
        out_other = gh.run_conv(x, w, scale, bias, res, True, stride, 0, n, n, n, site=site, batch=n, t0=1, seed=3, out_dtype=tdt)
    finally:
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_F16)
        _lib.set_option("conv_stream", 1)
    ref = gh.conv_ref(x, w, scale, bias, res, True, stride, 0, n, n, n)
    if site is not None:
        ref = ref * gh.folded_site_mask(site, n, cout, ho, ho, 1, 1, 3)
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    tol = 2e-3 if dt == "f16" else 1e-2
    torch.testing.assert_close(got, ref, rtol=tol, atol=tol)
    assert torch.equal(out.view(torch.int16), out_other.view(torch.int16))


@pytest.mark.parametrize("cmid,cw,cn,H,n,dt", [(128, 512, 128, 16, 37, "f16"), (256, 1024, 256, 8, 131, "f16"), (128, 512, 128, 5, 7, "f16"),
                                               (64, 256, 128, 16, 3, "bf16"), (256, 1024, 256, 3, 25, "bf16"), (512, 2048, 512, 4, 40, "f16"),
                                               (128, 512, 128, 16, 9, "bf16"), (128, 1024, 128, 7, 11, "f16")])
def test_conv1x1_seam_equals_the_two_launches(cmid, cw, cn, H, n, dt):
    """conv1x1_seam (expand conv + residual + ReLU of one Bottleneck and the reduce conv of the next in one launch, the wide tensor fed to
    the second GEMM from LDS) against the chain it replaces: both tensors bit for bit (the same epilogue code for the wide tensor, the same
    K order for the narrow one), ragged pixel counts (H x H x n not a multiple of the 128-pixel tile), and the fp32 reference.  The last
    shape (512 output channels) is outside the kernel: the entry point runs the two launches itself."""
    tdt = torch.float16 if dt == "f16" else torch.bfloat16
    g = _gen(43)
    m = torch.randn(n, H, H, cmid, generator=g).to(tdt).to(DEV)
    res = torch.randn(n, H, H, cw, generator=g).to(tdt).to(DEV)
    w3 = (torch.randn(cw, 1, 1, cmid, generator=g) * (2.0 / cmid) ** 0.5).to(tdt).to(DEV)
    w1 = (torch.randn(cn, 1, 1, cw, generator=g) * (2.0 / cw) ** 0.5).to(tdt).to(DEV)
    s3, b3 = (0.5 + torch.rand(cw, generator=g)).to(DEV), (0.2 * torch.randn(cw, generator=g)).to(DEV)
    s1, b1 = (0.5 + torch.rand(cn, generator=g)).to(DEV), (0.2 * torch.randn(cn, generator=g)).to(DEV)
    y = torch.full((n, H, H, cw), float("nan"), dtype=tdt, device=DEV)
    z = torch.full((n, H, H, cn), float("nan"), dtype=tdt, device=DEV)
    lib = _lib.lib()
    if dt == "bf16":
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_BF16)
    try:
        _lib.set_option("conv_seam", 2)            # no minimum grid
        _lib.check(lib.bmi_conv1x1_seam_fwd(gh.ptr(m), gh.ptr(w3), gh.ptr(s3), gh.ptr(b3), gh.ptr(res), gh.ptr(y), gh.ptr(w1), gh.ptr(s1), gh.ptr(b1),
                                            gh.ptr(z), n, H, H, cmid, cw, cn, 1, gh.stream()), "bmi_conv1x1_seam_fwd")
        torch.cuda.synchronize()
        y2 = gh.run_conv(m, w3, s3, b3, res, True, 1, 0, n, n, n, out_dtype=tdt)
        z2 = gh.run_conv(y2, w1, s1, b1, None, True, 1, 0, n, n, n, out_dtype=tdt)
    finally:
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_F16)
        _lib.set_option("conv_seam", 1)
    assert torch.equal(y.view(torch.int16), y2.view(torch.int16))
    assert torch.equal(z.view(torch.int16), z2.view(torch.int16))
    ref_y = gh.conv_ref(m, w3, s3, b3, res, True, 1, 0, n, n, n)
    ref_z = gh.conv_ref(ref_y.permute(0, 2, 3, 1).to(tdt), w1, s1, b1, None, True, 1, 0, n, n, n)
    tol = 2e-3 if dt == "f16" else 1e-2
    torch.testing.assert_close(y.float().cpu().permute(0, 3, 1, 2), ref_y, rtol=tol, atol=tol)
    torch.testing.assert_close(z.float().cpu().permute(0, 3, 1, 2), ref_z, rtol=4 * tol, atol=4 * tol)


def test_conv1x1_seam_full_grid_repeats_bit_for_bit():
    """Race screen for conv1x1_seam's DMA / barrier choreography (three rotating weight slots, counted vmcnt, the residual landing in the
    buffer stage B has just read): a grid of 6000 tiles (every CU holds two workgroups at different phases for dozens of rounds), five
    launches, each bit for bit the two-launch chain."""
    cmid, cw, cn, H, n = 128, 512, 128, 16, 3000
    g = _gen(47)
    m = torch.randn(n, H, H, cmid, generator=g).to(torch.float16).to(DEV)
    res = torch.randn(n, H, H, cw, generator=g).to(torch.float16).to(DEV)
    w3 = (torch.randn(cw, 1, 1, cmid, generator=g) * (2.0 / cmid) ** 0.5).to(torch.float16).to(DEV)
    w1 = (torch.randn(cn, 1, 1, cw, generator=g) * (2.0 / cw) ** 0.5).to(torch.float16).to(DEV)
    s3, b3 = (0.5 + torch.rand(cw, generator=g)).to(DEV), (0.2 * torch.randn(cw, generator=g)).to(DEV)
    s1, b1 = (0.5 + torch.rand(cn, generator=g)).to(DEV), (0.2 * torch.randn(cn, generator=g)).to(DEV)
    lib = _lib.lib()
    _lib.set_option("conv_seam", 0)
    try:
        y2 = gh.run_conv(m, w3, s3, b3, res, True, 1, 0, n, n, n)
        z2 = gh.run_conv(y2, w1, s1, b1, None, True, 1, 0, n, n, n)
    finally:
        _lib.set_option("conv_seam", 1)
    for rep in range(5):
        y = torch.full((n, H, H, cw), float("nan"), dtype=torch.float16, device=DEV)
        z = torch.full((n, H, H, cn), float("nan"), dtype=torch.float16, device=DEV)
        _lib.check(lib.bmi_conv1x1_seam_fwd(gh.ptr(m), gh.ptr(w3), gh.ptr(s3), gh.ptr(b3), gh.ptr(res), gh.ptr(y), gh.ptr(w1), gh.ptr(s1), gh.ptr(b1),
                                            gh.ptr(z), n, H, H, cmid, cw, cn, 1, gh.stream()), "bmi_conv1x1_seam_fwd")
        torch.cuda.synchronize()
        assert torch.equal(y.view(torch.int16), y2.view(torch.int16)), rep
        assert torch.equal(z.view(torch.int16), z2.view(torch.int16)), rep


@pytest.mark.parametrize("in_f32", [0, 1])
def test_dense_split_fp16_is_fp32_equivalent(in_f32):
    """The default dense kernel multiplies fp16 head + tail pairs on the fp16 MFMA (fp32 accumulation); against the exact-f32
    MFMA kernel (bmi_set_option "dense_exact") the outputs agree to a few 1e-7 of the operand scale, far inside the 1e-5 the
    float64 reference test allows."""
    lib = _lib.lib()
    N, K, Cout = 777, 512, 512
    g = _gen(23)
    x = torch.randn(N, K, generator=g) * 3.0
    xd = x.to(DEV) if in_f32 else x.half().to(DEV)
    w = 0.05 * torch.randn(Cout, K, generator=g)
    b = 0.3 * torch.randn(Cout, generator=g)
    wd, bd = w.to(DEV), b.to(DEV)
    outs = []
    try:
        for exact in (1, 0):
            _lib.set_option("dense_exact", exact)
            out = torch.full((N, Cout), float("nan"), device=DEV)
            _lib.check(lib.bmi_dense_f32(gh.ptr(xd), in_f32, gh.ptr(wd), gh.ptr(bd), gh.ptr(out), N, N, K, Cout, 0, None, N, 0, 0, 0,
                                         gh.stream()), "bmi_dense_f32")
            torch.cuda.synchronize()
            outs.append(out.cpu().double())
    finally:
        _lib.set_option("dense_exact", 0)
    ref = xd.cpu().double() @ w.double().T + b.double()
    scale = float((xd.cpu().double().abs() @ w.double().abs().T).max())          # sum |x||w|: what a relative error multiplies
    assert float((outs[0] - ref).abs().max()) < 2e-6 * scale                      # exact-f32 chain vs float64
    assert float((outs[1] - ref).abs().max()) < 2e-6 * scale                      # split-fp16 vs float64: the same class
    assert float((outs[1] - outs[0]).abs().max()) < 1e-6 * scale


@pytest.mark.parametrize("cin,cout,H,n", [(256, 256, 8, 21), (512, 512, 4, 37), (64, 256, 8, 1030)])
@pytest.mark.parametrize("use_res,use_site", [(False, False), (True, True)])
def test_conv3x3_pw4_equals_conv3x3_pw(cin, cout, H, n, use_res, use_site):
    """conv3x3_pw4 ("conv_pw" = 4: four waves, accumulators in AGPRs, fragments of the next K-step read under the MFMAs of this one,
    one barrier per K-step) sums K in conv3x3_pw's order and finishes through the same epilogue code: the same bits."""
    x, w, scale, bias, g = _conv_inputs(cin, cout, H, 3, n, 500 + n, False)
    res = torch.randn(n, H, H, cout, generator=g).to(torch.float16).to(DEV) if use_res else None
    site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=2, p=0.25) if use_site else None
    outs = {}
    try:
        for mode in (2, 4):
            _lib.set_option("conv_pw", mode)
            outs[mode] = gh.run_conv(x, w, scale, bias, res, True, 1, 1, n, n, n, site=site, batch=n, t0=1, seed=3)
    finally:
        _lib.set_option("conv_pw", 1)
    assert torch.equal(outs[2].view(torch.int16), outs[4].view(torch.int16))
    ref = gh.conv_ref(x, w, scale, bias, res, True, 1, 1, n, n, n)
    if site is not None:
        ref = ref * gh.folded_site_mask(site, n, cout, H, H, 1, 1, 3)
    torch.testing.assert_close(outs[4].float().cpu().permute(0, 3, 1, 2), ref, rtol=2e-3, atol=4e-3)


@pytest.mark.parametrize("kind", ["none", "elementwise", "elementwise_p02", "channel", "masksemble"])
@pytest.mark.parametrize("bf16", [False, True], ids=["f16", "bf16"])
def test_patch_kernel_64_channel_tile(kind, bf16):
    """conv3x3_patch's 64-channel tile (round 6; BCT = 64: the 64 -> 64 BasicBlocks behind the stem on 32x32 maps — per sample behind a "layer"
    site, once per batch everywhere else) against a torch fp32 reference on the same 16-bit operands: every site kind bit-exact on the mask
    (the 2-bit elementwise site takes the shared-Philox form: one call per pixel for the tile's 64 channels; p = 0.2 draws 16 bits per element and
    takes epilogue_quad), residual, broadcast input (n % in_mod), a ragged batch; and close to the per-tap conv_igemm ("conv_patch64" = 0), which
    sums K in 32-channel chunks."""
    cin, cout, H, k, s, p = SHAPES["S1"]
    B, tc, t0, seed, cnt0 = 3, 3, 5, (7 << 32) + 42, 2
    t16 = torch.bfloat16 if bf16 else torch.float16
    x, w, scale, bias, g = _conv_inputs(cin, cout, H, k, B, 21, False)
    res = torch.randn(B * tc, H, H, cout, generator=g)
    if bf16:
        x, w = x.float().to(t16), w.float().to(t16)
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_BF16)
    res = res.to(t16).to(DEV)
    site = None
    if kind.startswith("elementwise"):
        site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=4, p=0.2 if kind.endswith("p02") else 0.25)
    elif kind == "channel":
        site = dict(kind=_lib.SITE_CHANNEL, site_id=2, p=0.5)
    elif kind == "masksemble":
        site = dict(kind=_lib.SITE_MASKSEMBLE, site_id=1, masks=(np.random.RandomState(0).rand(4, cout) < 0.4).astype(np.float32))
    try:
        out = {}
        for arm in (1, 0):
            _lib.set_option("conv_patch64", arm)
            out[arm] = gh.run_conv(x, w, scale, bias, res, True, s, p, B * tc, B, B * tc, site=site, batch=B, t0=t0, seed=seed, cnt0=cnt0,
                                   out_dtype=t16).float().cpu().permute(0, 3, 1, 2)
    finally:
        _lib.set_option("conv_patch64", 1)
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_F16)
    ref = gh.conv_ref(x, w, scale, bias, res, True, s, p, B * tc, B, B * tc)
    mult = gh.folded_site_mask(site, B, cout, H, H, tc, t0, seed, cnt0) if site is not None else torch.ones_like(ref)
    tol = 2e-2 if bf16 else 2e-3
    assert torch.isfinite(out[1]).all()
    torch.testing.assert_close(out[1], ref * mult, rtol=tol, atol=tol)
    torch.testing.assert_close(out[1], out[0], rtol=tol, atol=tol)
    if site is not None:
        assert torch.equal(out[1][mult == 0], torch.zeros_like(out[1][mult == 0])) and (mult == 0).any() and (mult != 0).any()


@pytest.mark.parametrize("form", ["plain", "plain_norelu", "shortcut", "res", "res_site", "res_broadcast"])
@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_patch_register_epilogue_equals_the_lds_epilogues_bit_for_bit(form, dt):
    """conv3x3_patch's 16x16 class with "patch_direct" (round 6): the MFMA rows of a wave are a permutation of its 64 channels, so a lane's accumulators of a
    pixel are two runs of 8 consecutive channels and the launch finishes on the registers — BN, residual, ReLU, the 2-bit elementwise site — with 16-byte
    stores straight to HBM.  Same arithmetic in the same order as epilogue_plain / epilogue_lite ("patch_direct" = 0): IDENTICAL bits, for the plain launch
    (with and without ReLU), the fused shortcut (its weight rows follow the same permutation), the residual tails with and without the site, fp16 and bf16;
    a residual of fewer rows than the output (res_mod < N) is not the specialised tail and takes the LDS epilogue in both arms."""
    lib = _lib.lib()
    cin, cout, H, k, s, p = SHAPES["S2"]
    n, B, t0, seed = 6, 3, 4, (5 << 32) + 7
    g = _gen(23)
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    x = torch.randn(n, H, H, cin, generator=g).to(tdt).to(DEV)
    w = (torch.randn(cout, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(tdt).to(DEV)
    scale, bias = (0.5 + torch.rand(cout, generator=g)).to(DEV), (0.2 * torch.randn(cout, generator=g)).to(DEV)
    res = torch.randn(B if form == "res_broadcast" else n, H, H, cout, generator=g).to(tdt).to(DEV)
    x2 = torch.randn(n, 2 * H, 2 * H, cin // 2, generator=g).to(tdt).to(DEV)
    w2 = (torch.randn(cout, cin // 2, generator=g) * (2.0 / cin) ** 0.5).to(tdt).to(DEV)
    site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=3, p=0.25) if form == "res_site" else None
    if dt == "bf16":
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_BF16)
    outs = {}
    try:
        for arm in (1, 0):
            _lib.set_option("patch_direct", 2 * arm)          # 2: the register epilogue for the plain launches too (default 1: the tails only)
            if form == "shortcut":
                out = torch.full((n, H, H, cout), float("nan"), dtype=tdt, device=DEV)
                _lib.check(lib.bmi_conv3x3_shortcut_fwd(gh.ptr(x), gh.ptr(w), gh.ptr(x2), gh.ptr(w2), gh.ptr(bias), gh.ptr(out), n, H, H, cin, cout, cin // 2, 1,
                                                        gh.stream()), "bmi_conv3x3_shortcut_fwd")
                torch.cuda.synchronize()
            else:
                r = res if form.startswith("res") else None
                out = gh.run_conv(x, w, scale, bias, r, form != "plain_norelu", s, p, n, n, B if form == "res_broadcast" else n, site=site, batch=B, t0=t0,
                                  seed=seed, out_dtype=tdt)
            outs[arm] = out
    finally:
        _lib.set_option("patch_direct", 1)
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_F16)
    assert torch.isfinite(outs[1].float()).all()
    assert torch.equal(outs[1].view(torch.int16), outs[0].view(torch.int16))
    if form != "shortcut":                      # ... and against the fp32 reference on the same 16-bit operands
        r = res if form.startswith("res") else None
        ref = gh.conv_ref(x, w, scale, bias, r, form != "plain_norelu", s, p, n, n, B if form == "res_broadcast" else n)
        if site is not None:
            ref = ref * gh.folded_site_mask(site, B, cout, H, H, n // B, t0, seed, 0)
        tol = 3e-2 if dt == "bf16" else 3e-3
        torch.testing.assert_close(outs[1].float().cpu().permute(0, 3, 1, 2), ref, rtol=tol, atol=tol)
