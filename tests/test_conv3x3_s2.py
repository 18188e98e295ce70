"""-m gpu: conv3x3_s2 — the 3x3 stride-2 convs (BasicBlock conv1 of layers 2-4, every exit-head conv:
SA/models/resnet18/resnet18.py:280-299, :306-329) with the input patch resident in LDS as four parity planes — through the
C ABI against a torch fp32 reference of the same op on the same fp16-rounded operands, and against conv_igemm_wide (the kernel
it replaces: equal to fp32 summation order, i.e. to fp16 rounding of the outputs)."""
import pytest
import torch

from bayesnn_fpga_amd import _lib
from tests import gpu_helpers as gh
from tests.test_gpu_kernels import _conv_inputs, _gen

pytestmark = pytest.mark.gpu
DEV = gh.DEV


@pytest.fixture
def s2_always():
    _lib.set_option("conv_s2", 2)        # no minimum-grid rule: small image counts run in conv3x3_s2 too
    yield
    _lib.set_option("conv_s2", 1)


def _wide(fn):
    _lib.set_option("conv_s2", 0)
    try:
        return fn()
    finally:
        _lib.set_option("conv_s2", 2)


# (Cin, Cout, H): 32x32 -> 16x16 (one map per tile), 16x16 -> 8x8 (4 maps), 8x8 -> 4x4 (16 maps); Cout = one to four channel tiles
CASES = [(64, 256, 32, 1), (64, 256, 32, 5), (128, 256, 16, 3), (128, 512, 16, 10), (256, 512, 8, 37), (256, 1024, 8, 16),
         (64, 256, 8, 50),
         # Cout % 256 != 0 on the 16x16 output maps: the 128-channel tiles (round 4; ResNet-50's 256 -> 128 and 128 -> 128 stride-2 convs)
         (256, 128, 32, 3), (128, 128, 32, 7), (64, 384, 32, 2)]


@pytest.mark.parametrize("cin,cout,H,n", CASES)
@pytest.mark.parametrize("relu", [True, False])
def test_s2_small_and_ragged_tiles(s2_always, cin, cout, H, n, relu):
    """Image counts that are not a multiple of the maps per tile (the last tile's missing images load zeros through the
    buffer descriptor's bounds check and are not stored), one or several channel tiles, with and without ReLU / BN."""
    x, w, scale, bias, g = _conv_inputs(cin, cout, H, 3, n, 100 + n, False)
    out = gh.run_conv(x, w, scale, bias, None, relu, 2, 1, n, n, 1)
    ref = gh.conv_ref(x, w, scale, bias, None, relu, 2, 1, n, n, 1)
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-3)
    wide = _wide(lambda: gh.run_conv(x, w, scale, bias, None, relu, 2, 1, n, n, 1))
    torch.testing.assert_close(out.float(), wide.float(), rtol=2e-3, atol=2e-3)
    assert not torch.equal(out, torch.zeros_like(out))
    if not relu:
        assert (ref < 0).any()
    out2 = gh.run_conv(x, w, None, None, None, relu, 2, 1, n, n, 1)      # no BN vectors
    ref2 = gh.conv_ref(x, w, None, None, None, relu, 2, 1, n, n, 1)
    torch.testing.assert_close(out2.float().cpu().permute(0, 3, 1, 2), ref2, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("cin,cout,H,n", [(64, 256, 32, 700), (128, 256, 16, 2111), (256, 512, 8, 4500), (128, 512, 16, 1203), (128, 128, 32, 777)])
def test_s2_persistent_walk(cin, cout, H, n):
    """More tiles than CUs under the DEFAULT selection rule: every workgroup walks several tiles, the next tile's first
    weight stages and plane A / B pieces land under the epilogue.  Ragged last tile; repeat launches agree bit for bit."""
    x, w, scale, bias, g = _conv_inputs(cin, cout, H, 3, n, 17, False)
    imgs = 256 // ((H // 2) ** 2)
    assert -(-n // imgs) * (cout // (256 if cout % 256 == 0 else 128)) > 512 and (imgs == 1 or n % imgs != 0)
    out = gh.run_conv(x, w, scale, bias, None, True, 2, 1, n, n, 1)
    ref = gh.conv_ref(x, w, scale, bias, None, True, 2, 1, n, n, 1)
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=3e-3)
    for _ in range(3):
        assert torch.equal(out, gh.run_conv(x, w, scale, bias, None, True, 2, 1, n, n, 1))


@pytest.mark.parametrize("cin,ca,cb,H,n", [(64, 128, 128, 32, 3), (128, 256, 256, 16, 9), (256, 512, 512, 8, 21), (128, 128, 384, 16, 6)])
def test_s2_pair_mode(s2_always, cin, ca, cb, H, n):
    """Two convs on the same input in one launch (layerN[0].conv1 + the exit head's first conv): channel tiles below `split`
    take the first conv's weights / BN / output tensor.  Equal to the reference, and — where a conv alone is a whole number
    of 256-channel tiles — bit for bit the same conv launched alone."""
    lib = _lib.lib()
    g = _gen(5)
    x = torch.randn(n, H, H, cin, generator=g).to(torch.float16).to(DEV)
    ho = H // 2
    ws, scs, bis, outs = [], [], [], []
    for c in (ca, cb):
        ws.append((torch.randn(c, 3, 3, cin, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(torch.float16).to(DEV))
        scs.append((0.5 + torch.rand(c, generator=g)).to(DEV))
        bis.append((0.2 * torch.randn(c, generator=g)).to(DEV))
        outs.append(torch.full((n, ho, ho, c), float("nan"), dtype=torch.float16, device=DEV))
    rc = lib.bmi_conv_pair_fwd(gh.ptr(x), gh.ptr(ws[0]), gh.ptr(scs[0]), gh.ptr(bis[0]), gh.ptr(outs[0]), gh.ptr(ws[1]), gh.ptr(scs[1]),
                               gh.ptr(bis[1]), gh.ptr(outs[1]), n, n, H, H, cin, ca, cb, 3, 2, 1, 1, gh.stream())
    _lib.check(rc, "bmi_conv_pair_fwd")
    torch.cuda.synchronize()
    for i, c in enumerate((ca, cb)):
        ref = gh.conv_ref(x, ws[i], scs[i], bis[i], None, True, 2, 1, n, n, 1)
        torch.testing.assert_close(outs[i].float().cpu().permute(0, 3, 1, 2), ref, rtol=2e-3, atol=3e-3)
        if c % 256 == 0:
            assert torch.equal(outs[i], gh.run_conv(x, ws[i], scs[i], bis[i], None, True, 2, 1, n, n, 1))


@pytest.mark.parametrize("cin,cout,H", [(128, 256, 16), (128, 128, 32)])
def test_s2_bf16(s2_always, cin, cout, H):
    _lib.set_option("unit_entry_dtype", _lib.DTYPE_BF16)
    try:
        g = _gen(3)
        n = 9
        x = torch.randn(n, H, H, cin, generator=g).to(torch.bfloat16).to(DEV)
        w = (torch.randn(cout, 3, 3, cin, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(torch.bfloat16).to(DEV)
        scale, bias = (0.5 + torch.rand(cout, generator=g)).to(DEV), (0.2 * torch.randn(cout, generator=g)).to(DEV)
        out = gh.run_conv(x, w, scale, bias, None, True, 2, 1, n, n, 1, out_dtype=torch.bfloat16)
        ref = gh.conv_ref(x, w, scale, bias, None, True, 2, 1, n, n, 1)
        torch.testing.assert_close(out.float().cpu().permute(0, 3, 1, 2), ref, rtol=1.6e-2, atol=1.6e-2)
    finally:
        _lib.set_option("unit_entry_dtype", _lib.DTYPE_F16)


def test_s2_leaves_what_it_does_not_take_to_the_wide_kernel(s2_always):
    """A residual / site epilogue, a deterministic input broadcast over samples (in_mod < N) and Cout % 256 != 0 are not
    this kernel's: the launch must still come out right (conv_igemm_wide / conv_igemm take it)."""
    cin, cout, H, n = 128, 256, 16, 6
    x, w, scale, bias, g = _conv_inputs(cin, cout, H, 3, 2, 9, True)
    out = gh.run_conv(x, w, scale, bias, None, True, 2, 1, n, 2, 1, batch=2)                 # in_mod = 2 < N = 6
    ref = gh.conv_ref(x, w, scale, bias, None, True, 2, 1, n, 2, 1)
    torch.testing.assert_close(out.float().cpu().permute(0, 3, 1, 2), ref, rtol=2e-3, atol=2e-3)
    x6, w6, scale6, bias6, g = _conv_inputs(cin, cout, H, 3, n, 10, True)
    res = torch.randn(n, 8, 8, cout, generator=g).to(torch.float16).to(DEV)
    out = gh.run_conv(x6, w6, scale6, bias6, res, True, 2, 1, n, n, n)
    ref = gh.conv_ref(x6, w6, scale6, bias6, res, True, 2, 1, n, n, n)
    torch.testing.assert_close(out.float().cpu().permute(0, 3, 1, 2), ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("cin,cout,B,tc", [(64, 256, 3, 5), (64, 256, 7, 41), (128, 512, 2, 3), (64, 256, 25, 40), (256, 128, 5, 3), (256, 128, 9, 61)])
def test_s2_keep_bits_on_the_input_equal_the_materialised_mask(s2_always, cin, cout, B, tc):
    """ConvArgs::in_bits on the 32x32 -> 16x16 class: the patch pieces come from the B deterministic images (pre-scaled by 1/(1-p)
    in fp16), their keep bits ride along as a second DMA and the issuing thread clears the dropped elements in LDS — bit for bit
    the conv on the tensor bmi_mask_apply materialises (one image per tile up to several tiles per workgroup: 7 x 41 and
    25 x 40 images on 256 CUs)."""
    import ctypes as C
    from oracle import philox
    lib = _lib.lib()
    H, N, t0, seed, p = 32, B * tc, 3, (7 << 32) + 5, 0.25
    x, w, scale, bias, g = _conv_inputs(cin, cout, H, 3, B, 300 + B, False)
    x = torch.relu(x)                                                  # (post-ReLU like the tensor the site sits on)
    site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=1, p=p)
    keep = []
    s = gh.site_struct(site, keep)
    bits = torch.zeros(N * H * H * cin // 8, dtype=torch.uint8, device=DEV)
    _lib.check(lib.bmi_mask_bits(gh.ptr(bits), N, H * H, cin, C.byref(s), B, t0, seed, gh.stream()), "bmi_mask_bits")
    masked = torch.empty(N, H, H, cin, dtype=torch.float16, device=DEV)
    _lib.check(lib.bmi_mask_apply(gh.ptr(x), gh.ptr(masked), N, B, H * H, cin, C.byref(s), B, t0, seed, 0, gh.stream()), "bmi_mask_apply")
    torch.cuda.synchronize()
    xs = (x.float() * float(philox.drop_scale(p))).to(torch.float16)
    got = gh.run_conv(xs, w, scale, bias, None, True, 2, 1, N, B, 1, batch=B, in_bits=bits)
    want = gh.run_conv(masked, w, scale, bias, None, True, 2, 1, N, N, 1, batch=B)
    assert torch.isfinite(got.float()).all() and float(got.float().abs().max()) > 0
    assert torch.equal(got, want)
    for _ in range(3):                                                 # (the in-LDS masking sits in the counted-wait schedule: repeats agree)
        assert torch.equal(gh.run_conv(xs, w, scale, bias, None, True, 2, 1, N, B, 1, batch=B, in_bits=bits), got)
    ref = gh.conv_ref(masked, w, scale, bias, None, True, 2, 1, N, N, 1)
    torch.testing.assert_close(got.float().cpu().permute(0, 3, 1, 2), ref, rtol=2e-3, atol=2e-3)
