"""The SPLIT engines (bmi_model_desc.dtype = BMI_DTYPE_F16X2 / BMI_DTYPE_BF16X3, csrc/conv_split.hip): fp32 activations, every conv
operand a 16-bit head + tail pair, w.x = w_lo.x_hi + w_hi.x_lo + w_hi.x_hi on the fp16 / bf16 matrix pipe — the reference's fp32
arithmetic (SA/models/resnet18/resnet18.py:32-48, :302-346) at 3 MFMAs per K-step instead of the exact engine's 16, so that
north_star's 1e-3 holds at speed where plain fp16 / bf16 do not: the converted VGG19EarlyExit (logits up to 61,
Hardware_Artifact/converter/pytorch/nn2bnn.py:32-45 on SA/models/vgg19/vgg19.py:256-324; fp16: 1.7e-3), BASELINE configs[1] as written
("bf16": 3.3e-3 in plain bf16), trained-like nets with peaky softmaxes.

Tolerances.  f16x2 carries 22 significant bits per operand: the exact engine's bars (kernel 2e-5 of the output scale, logits 2e-4,
probabilities 2e-5).  bf16x3 carries 16: kernel 2e-4, logits 2e-3, probabilities 1e-4 — ten times inside north_star's 1e-3."""
import ctypes as C

import numpy as np
import pytest
import torch

from bayesnn_fpga_amd import _lib
from bayesnn_fpga_amd.converter.pytorch import MCDropout
from bayesnn_fpga_amd.engine import CompiledGraph
from bayesnn_fpga_amd.models import extra as bx
from bayesnn_fpga_amd.models.resnet18.resnet18 import ResNet18MCEarlyExit
from bayesnn_fpga_amd.models.vgg19 import vgg19 as bvgg
from bayesnn_fpga_amd.synthetic import synthetic_images, synthetic_weights_
from oracle import extra_models as ox
from oracle import mcd
from oracle import resnet18 as oresnet
from tests import gpu_helpers as gh
from tests.helpers import build_seeded, golden_kwargs, load_golden
from tests.test_exact_engine import SHAPES, _conv64

DEV = "cuda:0"
TOLS = {"f16x2": dict(kernel=2e-5, logit=2e-4, prob=2e-5), "bf16x3": dict(kernel=2e-4, logit=2e-3, prob=1e-4)}
TORCH16 = {"f16x2": torch.float16, "bf16x3": torch.bfloat16}


def split_planes(w, dt):
    """[2][...]: rn16(w), rn16(w - rn16(w)) — what GraphBuilder.conv_weight hands the C ABI."""
    hi = w.float().to(TORCH16[dt])
    return torch.stack([hi, (w.float() - hi.float()).to(TORCH16[dt])]).contiguous()


# ---- CPU side ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", ["f16x2", "bf16x3"])
def test_split_graph_layout(dt, monkeypatch):
    """Host-only (bmi_create / bmi_plan run on a CPU box): head / tail weight planes, 4-byte (pair32) activations in the plan, the same MACs
    as the fp16 graph; head + tail reproduce the fp32 weight to the split's precision; the three downsample paths of ResNet-18 ride in
    their block's conv2 as extra K-steps (in2 / weight2 planes, BN scale folded) unless BMI_FUSE_SHORTCUT = 0."""
    kw = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
    m = synthetic_weights_(build_seeded(ResNet18MCEarlyExit, kw), 0)
    c16, c32, cs = CompiledGraph(m, "cpu", 8, 2), CompiledGraph(m, "cpu", 8, 2, dtype="f32"), CompiledGraph(m, "cpu", 8, 2, dtype=dt)
    fused = [op for op in cs.graph.ops if op.get("in2", -1) >= 0]
    assert len(fused) == 3 and not any(op.get("in2", -1) >= 0 for op in c32.graph.ops)
    for op in fused:
        cout, cin2 = op["weight"].shape[1], cs.graph.tensors[op["in2"]][2]
        assert op["weight2"].dtype == TORCH16[dt] and tuple(op["weight2"].shape) == (2, cout, cin2)
        # (round 6: `scale` carries the inverse of the host's per-channel power-of-two weight lift — the only scale a split-engine conv with in2 takes)
        k = torch.log2(op["scale"])
        assert op["scale"] is not None and torch.equal(k, k.round()) and float(op["weight"][0].float().abs().amax()) < 256.0
    assert abs(cs.workspace_bytes - c32.workspace_bytes) <= 0.1 * c32.workspace_bytes   # (no shortcut tensors, but a block's input lives until its conv2)
    assert cs.prefix_macs + 8 * cs.suffix_macs == c16.prefix_macs + 8 * c16.suffix_macs
    monkeypatch.setenv("BMI_FUSE_SHORTCUT", "0")
    cu = CompiledGraph(m, "cpu", 8, 2, dtype=dt)
    assert not any(op.get("in2", -1) >= 0 for op in cu.graph.ops)
    assert abs(cu.workspace_bytes - c32.workspace_bytes) <= 0.1 * c32.workspace_bytes    # (4-byte elements; pair launches move two live ranges)
    convs = [(a, b) for a, b in zip(cu.graph.ops, c32.graph.ops) if a["kind"] == _lib.OP_CONV]
    assert convs
    rel = 2.0 ** -21 if dt == "f16x2" else 2.0 ** -16
    for a, b in convs:
        assert a["weight"].dtype == TORCH16[dt] and tuple(a["weight"].shape) == (2,) + tuple(b["weight"].shape)
        # head + tail = weight x lift, lift = a power of two per output channel folded back into the epilogue scale (engine.GraphBuilder.channel_lift):
        # every channel's largest weight sits in [2^7, 2^8), so its tails are normal fp16 numbers
        lift = b["scale"] / a["scale"]
        assert torch.equal(torch.log2(lift), torch.log2(lift).round())
        rec = (a["weight"][0].float() + a["weight"][1].float()) / lift[:, None, None, None]
        assert float((rec - b["weight"]).abs().max()) <= rel * float(b["weight"].abs().max())
        amax = a["weight"][0].float().abs().reshape(a["weight"].shape[1], -1).amax(dim=1)
        assert bool(((amax >= 127.9) & (amax <= 256.0)).all())
    assert all(op["weight"].dtype == torch.float32 for op in cs.graph.ops if op["kind"] == _lib.OP_STEM)


# ---- the kernel, through bmi_conv_igemm_fwd under unit_entry_dtype = F16X2 / BF16X3 ------------------------------------------------
@pytest.fixture(params=["f16x2", "bf16x3"])
def split_entries(request):
    _lib.set_option("unit_entry_dtype", _lib.DTYPES[request.param])
    yield request.param
    _lib.set_option("unit_entry_dtype", _lib.DTYPE_F16)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SHAPES))
def test_split_conv_against_float64(name, split_entries):
    """conv + BN + residual + ReLU on split operands against torch float64 on the SAME fp32 operands: the six shape classes of the
    exact-engine test (64 / 128 / 256-channel tiles, 1x1 / 3x3 / 5x5, stride 2, ragged last pixel tile, broadcast input)."""
    dt = split_entries
    cin, cout, H, k, s, p = SHAPES[name]
    g = torch.Generator().manual_seed(11)
    B, tc = 3, 2
    x = torch.randn(B, H, H, cin, generator=g).to(DEV)
    w = (torch.randn(cout, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(DEV)
    scale, bias = (0.5 + torch.rand(cout, generator=g)).to(DEV), (0.2 * torch.randn(cout, generator=g)).to(DEV)
    ho = (H + 2 * p - k) // s + 1
    res = torch.randn(B * tc, ho, ho, cout, generator=g).to(DEV)
    out = _run_split(x, w, dt, scale, bias, res, True, s, p, B * tc, B, B * tc, batch=B)
    ref = _conv64(x, w, scale, bias, res, True, s, p, B * tc, B, B * tc)
    got = out.double().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    print(f"{dt} {name}: max err / max|ref| = {err:.2e}")
    assert err <= TOLS[dt]["kernel"]


def _run_split(x, w, dt, scale, bias, res, relu, stride, pad, n, in_mod, res_mod, **kw):
    """gh.run_conv with the weight handed over as head / tail planes (cout and k are read from the fp32 weight's shape)."""
    lib = _lib.lib()
    wp = split_planes(w, dt)
    n_in, H, W, cin = x.shape
    cout, k = w.shape[0], w.shape[1]
    ho, wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    # activations cross the C ABI in the split engines' pair32 layout (16-bit head + tail per element, 32-channel blocks)
    x = gh.pair32_encode(x, TORCH16[dt])
    res = gh.pair32_encode(res, TORCH16[dt]) if res is not None else None
    out = torch.full((n, ho, wo, cout // 32, 2, 32), float("nan"), dtype=TORCH16[dt], device=DEV)
    keep = []
    s = gh.site_struct(kw.get("site"), keep)
    rc = lib.bmi_conv_igemm_fwd(gh.ptr(x), None, 1.0, gh.ptr(wp), gh.ptr(scale), gh.ptr(bias), gh.ptr(res), gh.ptr(out), n, in_mod, res_mod, H, W,
                                cin, cout, k, stride, pad, int(relu), C.byref(s) if s is not None else None, kw.get("batch", n),
                                kw.get("t0", 0), kw.get("seed", 0), kw.get("cnt0", 0), gh.stream())
    _lib.check(rc, "bmi_conv_igemm_fwd")
    torch.cuda.synchronize()
    return gh.pair32_decode(out)


@pytest.mark.gpu
def test_split_stem_mask_maxpool_head_and_dense_on_pair32(split_entries):
    """The split engines' other kernels on pair32 tensors, through the unit entry points: the stem writes the encoding of its fp32 result;
    the stand-alone site and the max-pool read and write it (a dropped element is exactly 0, a kept one the encoding of x / (1 - p); the
    maximum is one of its four inputs); the exit head and a hidden dense layer read it (against their fp32-input forms on the decoded
    tensor: the same numbers)."""
    dt = split_entries
    t16 = TORCH16[dt]
    eps = 2.0 ** -20 if dt == "f16x2" else 2.0 ** -15
    lib = _lib.lib()
    g = torch.Generator().manual_seed(3)
    n = 5
    x = torch.randn(n, 3, 32, 32, generator=g).to(DEV)
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.3
    scale, bias = 0.5 + torch.rand(64, generator=g), 0.2 * torch.randn(64, generator=g)
    out = torch.empty(n, 32, 32, 2, 2, 32, dtype=t16, device=DEV)
    wd, sd, bd = w.to(DEV), scale.to(DEV), bias.to(DEV)
    _lib.check(lib.bmi_stem_conv_fwd(gh.ptr(x), gh.ptr(wd), gh.ptr(sd), gh.ptr(bd), gh.ptr(out), n, 3, 32, 32, 64, 3, 1, 1, 0, gh.stream()), "bmi_stem_conv_fwd")
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().permute(0, 3, 1, 2), padding=1) * scale.double()[None, :, None, None] + \
        bias.double()[None, :, None, None]
    got = gh.pair32_decode(out).double().cpu().permute(0, 3, 1, 2)
    assert float((got - ref).abs().max()) <= 1e-5 + eps * float(ref.abs().max())
    # stand-alone site: B deterministic images -> the folded batch (the second shape and p = 0.25: the one-call-per-64-elements kernel, a
    # sample's 8-channel items a multiple of 512)
    for B, tc, H, Cc, t0, seed, p in ((3, 4, 6, 64, 2, 99, 0.375), (4, 3, 8, 64, 1, 7, 0.25), (4, 2, 8, 128, 0, 5, 0.75)):
        xs = torch.randn(B, H, H, Cc, generator=g).to(DEV)
        xs = gh.pair32_decode(gh.pair32_encode(xs, t16))            # (values the layout holds exactly)
        site = dict(kind=_lib.SITE_ELEMENTWISE, site_id=0, p=p)
        keep = []
        s = gh.site_struct(site, keep)
        om = torch.empty(B * tc, H, H, Cc // 32, 2, 32, dtype=t16, device=DEV)
        _lib.check(lib.bmi_mask_apply(gh.ptr(gh.pair32_encode(xs, t16)), gh.ptr(om), B * tc, B, H * H, Cc, C.byref(s), B, t0, seed, 0, gh.stream()), "bmi_mask_apply")
        torch.cuda.synchronize()
        mult = gh.folded_site_mask(site, B, Cc, H, H, tc, t0, seed)
        want = xs.cpu().permute(0, 3, 1, 2).repeat(tc, 1, 1, 1) * mult
        gotm = gh.pair32_decode(om).cpu().permute(0, 3, 1, 2)
        assert torch.equal(gotm == 0, want == 0) and float((gotm - want).abs().max()) <= eps * float(want.abs().max()), (B, tc, H, Cc, p)
    # max-pool: exact (the maximum is one of the encoded inputs)
    xp = gh.pair32_decode(gh.pair32_encode(torch.randn(3, 8, 8, 64, generator=g).to(DEV), t16))
    op = torch.empty(3, 4, 4, 2, 2, 32, dtype=t16, device=DEV)
    _lib.check(lib.bmi_maxpool2(gh.ptr(gh.pair32_encode(xp, t16)), gh.ptr(op), 3, 8, 8, 64, gh.stream()), "bmi_maxpool2")
    torch.cuda.synchronize()
    assert torch.equal(gh.pair32_decode(op).cpu().permute(0, 3, 1, 2), torch.nn.functional.max_pool2d(xp.cpu().permute(0, 3, 1, 2), 2))
    # exit head and hidden dense layer: pair32 input against the fp32-input form on the decoded tensor
    Bh, tch, HW, K, Co = 4, 3, 16, 128, 10
    xh = gh.pair32_decode(gh.pair32_encode(torch.randn(Bh * tch, 4, 4, K, generator=g).abs().to(DEV), t16))
    wl = torch.zeros(32, K)
    wl[:Co] = torch.randn(Co, K, generator=g) * 0.1
    wl, bl = wl.to(DEV), (0.1 * torch.randn(Co, generator=g)).to(DEV)
    outs = []
    for is_f32, buf in ((1, xh.contiguous()), (0, gh.pair32_encode(xh, t16))):
        S = torch.zeros(3, Bh, Co, dtype=torch.float64, device=DEV)
        _lib.check(lib.bmi_head_fused(gh.ptr(buf), is_f32, Bh * tch, HW, K, gh.ptr(wl), gh.ptr(bl), Co, None, None, Bh, 0, tch, 7, 0,
                                      gh.ptr(S[0]), gh.ptr(S[1]), gh.ptr(S[2]), gh.stream()), "bmi_head_fused")
        torch.cuda.synchronize()
        outs.append(S.clone())
    assert torch.equal(outs[0], outs[1])
    xd = gh.pair32_decode(gh.pair32_encode(torch.randn(7, 1, 1, 64, generator=g).to(DEV), t16))
    wdn, bdn = (torch.randn(64, 64, generator=g) * 0.2).to(DEV), (0.1 * torch.randn(64, generator=g)).to(DEV)
    od = []
    for is_f32, buf in ((1, xd.contiguous()), (0, gh.pair32_encode(xd, t16))):
        o = torch.empty(7, 64, device=DEV)
        _lib.check(lib.bmi_dense_f32(gh.ptr(buf), is_f32, gh.ptr(wdn), gh.ptr(bdn), gh.ptr(o), 7, 7, 64, 64, 1, None, 7, 0, 0, 0, gh.stream()), "bmi_dense_f32")
        torch.cuda.synchronize()
        od.append(o.clone())
    assert torch.equal(od[0], od[1])


@pytest.mark.gpu
@pytest.mark.parametrize("cin,ca,cb,H,k,stride,n", [(64, 128, 128, 32, 3, 2, 21), (128, 128, 128, 16, 3, 2, 9), (64, 128, 384, 8, 1, 1, 70)])
def test_split_pair_launch_equals_the_two_launches(split_entries, cin, ca, cb, H, k, stride, n):
    """Pair mode of conv_split (two plain convs on one input in one launch: layerN[0].conv1 and the first conv of the exit head in front of
    it, on the 256-channel tile instead of two 128-channel ones): both outputs bit for bit the single launches' (the same K order per
    accumulator), through bmi_conv_pair_fwd under the split unit dtypes."""
    dt = split_entries
    t16 = TORCH16[dt]
    lib = _lib.lib()
    g = torch.Generator().manual_seed(11)
    pad = k // 2
    x = torch.randn(n, H, H, cin, generator=g).to(DEV)
    ho = (H + 2 * pad - k) // stride + 1
    outs = {}
    ws, ss, bs = [], [], []
    for co in (ca, cb):
        w = (torch.randn(co, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5)
        ws.append(w)
        ss.append((0.5 + torch.rand(co, generator=g)).to(DEV))
        bs.append((0.2 * torch.randn(co, generator=g)).to(DEV))
    single = [_run_split(x, ws[i].to(DEV), dt, ss[i], bs[i], None, True, stride, pad, n, n, n) for i in range(2)]
    xp = gh.pair32_encode(x, t16)
    wp = [split_planes(w, dt).to(DEV) for w in ws]
    oa = torch.full((n, ho, ho, ca // 32, 2, 32), float("nan"), dtype=t16, device=DEV)
    ob = torch.full((n, ho, ho, cb // 32, 2, 32), float("nan"), dtype=t16, device=DEV)
    _lib.check(lib.bmi_conv_pair_fwd(gh.ptr(xp), gh.ptr(wp[0]), gh.ptr(ss[0]), gh.ptr(bs[0]), gh.ptr(oa), gh.ptr(wp[1]), gh.ptr(ss[1]), gh.ptr(bs[1]),
                                     gh.ptr(ob), n, n, H, H, cin, ca, cb, k, stride, pad, 1, gh.stream()), "bmi_conv_pair_fwd")
    torch.cuda.synchronize()
    assert torch.equal(gh.pair32_decode(oa), single[0]) and torch.equal(gh.pair32_decode(ob), single[1])


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,cin2,H,n", [(128, 128, 64, 16, 5), (256, 256, 128, 8, 37), (512, 512, 256, 4, 70), (64, 64, 32, 6, 3)])
def test_split_conv_fused_shortcut_against_float64(split_entries, cin, cout, cin2, H, n):
    """The BasicBlock downsample path as extra K-steps of conv_split (ConvArgs::in2 / wgt2: out = relu(conv3x3(x; w) + conv1x1_stride2(x2; w2) +
    bias), both BN scales folded into the weight planes) against float64 on the same fp32 operands, ragged pixel counts included."""
    dt = split_entries
    t16 = TORCH16[dt]
    lib = _lib.lib()
    g = torch.Generator().manual_seed(13)
    x = torch.randn(n, H, H, cin, generator=g)
    x2 = torch.randn(n, 2 * H, 2 * H, cin2, generator=g)
    w = torch.randn(cout, 3, 3, cin, generator=g) * (2.0 / (9 * cin)) ** 0.5
    w2 = torch.randn(cout, cin2, generator=g) * (2.0 / cin2) ** 0.5
    bias = 0.2 * torch.randn(cout, generator=g)
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1) + \
        torch.nn.functional.conv2d(x2.double().permute(0, 3, 1, 2), w2.double()[:, :, None, None], stride=2) + bias.double()[None, :, None, None]
    ref = torch.relu(ref)
    out = torch.full((n, H, H, cout // 32, 2, 32), float("nan"), dtype=t16, device=DEV)
    xp, x2p = gh.pair32_encode(x.to(DEV), t16), gh.pair32_encode(x2.to(DEV), t16)
    wp, w2p, bd = split_planes(w, dt).to(DEV), split_planes(w2, dt).to(DEV), bias.to(DEV)
    _lib.check(lib.bmi_conv3x3_shortcut_fwd(gh.ptr(xp), gh.ptr(wp), gh.ptr(x2p), gh.ptr(w2p), gh.ptr(bd), gh.ptr(out), n, H, H, cin, cout, cin2, 1,
                                            gh.stream()), "bmi_conv3x3_shortcut_fwd")
    torch.cuda.synchronize()
    got = gh.pair32_decode(out).double().cpu().permute(0, 3, 1, 2)
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    print(f"{dt} fused shortcut {cin}->{cout} + {cin2}: max err / max|ref| = {err:.2e}")
    assert torch.isfinite(got).all() and err <= (4e-6 if dt == "f16x2" else 4e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,H,n,res", [(128, 128, 16, 5, True), (256, 256, 8, 37, False), (512, 512, 4, 70, True), (64, 64, 32, 3, False),
                                              (128, 256, 2, 300, True), (64, 128, 6, 9, False), (32, 64, 1, 700, True)])
def test_split_conv_shared_pixel_slot_is_bit_for_bit(split_entries, cin, cout, H, n, res):
    """conv_split's 3x3 stride-1 launches fetch the pixel tile of a tap row ONCE and read it shifted by one pixel for the kx = 0 / 2 taps
    (edge lanes cleared in registers): bit for bit the loop that fetches it per tap ("split_shx" 1 / 0) — 16x16 ... 1x1 maps (every lane an
    edge), 32x32 (four tiles per image), ragged last tiles, a 6x6 map (256 % 6 != 0: the shared form is not taken), with and without a
    residual; and the fused-shortcut form, whose extra K-steps read their own slots."""
    dt = split_entries
    g = torch.Generator().manual_seed(17)
    x = torch.randn(n, H, H, cin, generator=g).to(DEV)
    w = (torch.randn(cout, 3, 3, cin, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(DEV)
    scale, bias = (0.5 + torch.rand(cout, generator=g)).to(DEV), (0.2 * torch.randn(cout, generator=g)).to(DEV)
    r = torch.randn(n, H, H, cout, generator=g).to(DEV) if res else None
    out = {}
    for shx in (1, 2, 0):           # 1: a zero cell in front of every map row where rows are multiples of 8 pixels, else edge lanes cleared; 2: cleared everywhere
        _lib.set_option("split_shx", shx)
        try:
            out[shx] = _run_split(x, w, dt, scale, bias, r, True, 1, 1, n, n, n)
        finally:
            _lib.set_option("split_shx", 1)
    assert torch.isfinite(out[1]).all() and torch.equal(out[1], out[0]) and torch.equal(out[2], out[0])
    if H in (16, 8, 4):                                   # the fused shortcut on top (in2 = the block input at twice the map size)
        lib = _lib.lib()
        t16 = TORCH16[dt]
        cin2 = cin // 2
        x2 = torch.randn(n, 2 * H, 2 * H, cin2, generator=g).to(DEV)
        w2 = (torch.randn(cout, cin2, generator=g) * (2.0 / cin2) ** 0.5)
        xp, x2p = gh.pair32_encode(x, t16), gh.pair32_encode(x2, t16)
        wp, w2p = split_planes(w.cpu(), dt).to(DEV), split_planes(w2, dt).to(DEV)
        outs = {}
        for shx in (1, 2, 0):
            _lib.set_option("split_shx", shx)
            try:
                o = torch.full((n, H, H, cout // 32, 2, 32), float("nan"), dtype=t16, device=DEV)
                _lib.check(lib.bmi_conv3x3_shortcut_fwd(gh.ptr(xp), gh.ptr(wp), gh.ptr(x2p), gh.ptr(w2p), gh.ptr(bias), gh.ptr(o), n, H, H, cin, cout, cin2, 1,
                                                        gh.stream()), "bmi_conv3x3_shortcut_fwd")
                torch.cuda.synchronize()
                outs[shx] = o
            finally:
                _lib.set_option("split_shx", 1)
        assert torch.equal(outs[1].view(torch.int16), outs[0].view(torch.int16)) and torch.equal(outs[2].view(torch.int16), outs[0].view(torch.int16))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["elementwise", "channel", "masksemble"])
def test_split_conv_fused_site_is_bit_exact_on_the_mask(kind, split_entries):
    dt = split_entries
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
    out = _run_split(x, w, dt, scale, bias, None, True, s, p, B * tc, B * tc, 1, site=site, batch=B, t0=t0, seed=seed, cnt0=cnt0)
    mult = gh.folded_site_mask(site, B, cout, H, H, tc, t0, seed, cnt0).double()
    ref = _conv64(x, w, scale, bias, None, True, s, p, B * tc, B * tc, 1) * mult
    got = out.double().cpu().permute(0, 3, 1, 2)
    assert float((got - ref).abs().max()) <= 2 * TOLS[dt]["kernel"] * float(ref.abs().max())
    dropped = mult == 0
    assert dropped.any() and torch.equal(got[dropped], torch.zeros(int(dropped.sum()), dtype=torch.float64))


@pytest.mark.gpu
def test_split_conv_f16x2_is_far_closer_than_fp16(split_entries):
    """What the split buys, on one 3x3 conv with K = 2304: the same fp32 operands through the plain 16-bit kernel (operands rounded
    to 16 bits) and through the split kernel, both against float64."""
    dt = split_entries
    cin, cout, H, k, s, p = SHAPES["S3"]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, H, H, cin, generator=g).to(DEV)
    w = (torch.randn(cout, k, k, cin, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(DEV)
    one, zero = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
    ref = _conv64(x, w, one, zero, None, False, s, p, 4, 4, 1)
    e_split = float((_run_split(x, w, dt, one, zero, None, False, s, p, 4, 4, 1).double().cpu().permute(0, 3, 1, 2) - ref).abs().max())
    t16 = TORCH16[dt]
    _lib.set_option("unit_entry_dtype", _lib.DTYPE_F16 if dt == "f16x2" else _lib.DTYPE_BF16)
    try:
        o16 = gh.run_conv(x.to(t16), w.to(t16), one, zero, None, False, s, p, 4, 4, 1, out_dtype=t16)
    finally:
        _lib.set_option("unit_entry_dtype", _lib.DTYPES[dt])
    e16 = float((o16.double().cpu().permute(0, 3, 1, 2) - ref).abs().max())
    print(f"{dt}: split {e_split:.2e}, plain 16-bit {e16:.2e} (max|ref| {float(ref.abs().max()):.2f})")
    assert e_split * 50 < e16


# ---- the whole path on the split engines ---------------------------------------------------------------------------------------------
def _on(model, dt):
    model = model.to(DEV).eval()
    model.engine_dtype = dt
    return model


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("name", ["block_exit", "layer_exit", "mask8_exit_c100", "block_exit_p02"])
def test_split_engine_against_reference_golden(name, dt):
    """ResNet-18 goldens of the reference (per-pass logits of its own ResNet18MCEarlyExit and its _get_output 5-tuple) on the split
    engines, at the exact engine's tolerances (f16x2) / ten times inside 1e-3 (bf16x3)."""
    tol = TOLS[dt]
    g = load_golden(f"resnet18_{name}.npz")
    kw = golden_kwargs(g)
    B, T, seed = int(g["B"]), int(g["T"]), int(g["seed"])
    model = _on(synthetic_weights_(build_seeded(ResNet18MCEarlyExit, kw), 0), dt)
    model.mc_seed = seed
    x = synthetic_images(B, seed=1234).to(DEV)
    passes = np.stack([np.stack([o.cpu().numpy() for o in model(x)]) for _ in range(T)])
    np.testing.assert_allclose(passes, g["logits"], rtol=0, atol=tol["logit"])
    ref_probs = torch.softmax(torch.from_numpy(g["logits"]), -1).numpy().astype(np.float64)
    eng = model.engine(x.device, max_batch=B)
    assert eng.dtype == dt
    r = eng.predict(x, T, seed=seed, cnt0=0)
    np.testing.assert_allclose(r["mean"].cpu().numpy(), g["go_output_sm"], rtol=0, atol=tol["prob"])
    np.testing.assert_allclose(r["var"].cpu().numpy(), ref_probs.var(0), rtol=0, atol=tol["prob"])
    e1 = model.engine(x.device, max_batch=B, chunk_samples=1)
    S1 = e1.accumulate(x, e1.new_moments(B), 0, T, seed).cpu()
    e3 = model.engine(x.device, max_batch=B, chunk_samples=3)
    torch.testing.assert_close(e3.accumulate(x, e3.new_moments(B), 0, T, seed).cpu(), S1, rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f16x2", "bf16x3"])
def test_split_engine_vgg19_against_reference_golden(dt):
    tol = TOLS[dt]
    g = load_golden("vgg19_exit_mc.npz")
    kw = golden_kwargs(g)
    B, T, seed = int(g["B"]), int(g["T"]), int(g["seed"])
    m = _on(synthetic_weights_(build_seeded(bvgg.VGG19MCEarlyExit, kw), 0), dt)
    m.mc_seed = seed
    x = synthetic_images(B, seed=1234).to(DEV)
    passes = np.stack([np.stack([o.cpu().numpy() for o in m(x)]) for _ in range(T)])
    np.testing.assert_allclose(passes, g["logits"], rtol=0, atol=tol["logit"])
    ref_probs = torch.softmax(torch.from_numpy(g["logits"]), -1).numpy().astype(np.float64)
    r = m.engine(x.device, max_batch=B).predict(x, T, seed=seed)
    np.testing.assert_allclose(r["mean"].cpu().numpy(), ref_probs.mean(0), rtol=0, atol=tol["prob"])
    np.testing.assert_allclose(r["var"].cpu().numpy(), ref_probs.var(0), rtol=0, atol=tol["prob"])


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f16x2", "bf16x3"])
def test_converted_vgg19_early_exit_within_1e3_at_speed(dt):
    """THE golden the fp16 engine misses (tests/test_converter.py asserts it at 3e-3 there, measured 1.7e-3): the reference's converter on
    its own VGG19EarlyExit, 32 sites per pass, logits up to 61.  Predictive mean AND variance within north_star's 1e-3 — asserted at
    the split engines' own bars, 2e-5 / 1e-4 — per-pass logits to 1e-5 / 1e-4 of their scale, the exact zero pattern of the dropped logits."""
    from bayesnn_fpga_amd.models.vgg19.vgg19 import VGG19EarlyExit
    tol = TOLS[dt]
    g = load_golden("converter_vgg19ee.npz")
    B, T, seed, p = int(g["B"]), int(g["T"]), int(g["seed"]), float(g["p"])
    torch.manual_seed(0)
    net = synthetic_weights_(VGG19EarlyExit(n_exits=5, out_dim=10), 0)
    m = MCDropout(net, nSamples=T, p=p).to(DEV)
    m.mc_seed = seed
    x = synthetic_images(B, seed=1234).to(DEV)
    ref = g["logits"]
    ref_probs = torch.softmax(torch.from_numpy(ref), -1).numpy().astype(np.float64)
    scale = float(np.abs(ref).max())
    m.engine_dtype = dt
    m.train()
    m.mc_pass = 0
    passes = np.stack([np.stack([o.cpu().numpy() for o in m(x)]) for _ in range(T)])
    np.testing.assert_allclose(passes, ref, rtol=0, atol=tol["logit"] * max(1.0, scale / 10))
    zero = ref == 0
    assert zero.any() and np.array_equal(passes == 0, zero)
    r = m.engine(x.device, max_batch=B, dtype=dt).predict(x, T, seed=seed)
    em, ev = np.abs(r["mean"].cpu().numpy() - ref_probs.mean(0)).max(), np.abs(r["var"].cpu().numpy() - ref_probs.var(0)).max()
    print(f"converter_vgg19ee {dt}: max|logit| {scale:.1f}  mean {em:.2e}  var {ev:.2e}")
    assert em <= 1e-3 and ev <= 1e-3            # north_star
    assert em <= 5 * tol["prob"] and ev <= 5 * tol["prob"]


@pytest.mark.gpu
def test_vgg11_config_as_written_in_bf16x3():
    """BASELINE configs[1]: "VGG-11 CIFAR-10, 3 dropout layers, T=30, 1xMI355X bf16" at the reference's batch of 250 (T = 4 here so the
    oracle finishes in seconds) ON THE BF16 MATRIX PIPE within north_star's 1e-3: plain bf16 measures 3.3e-3 on it."""
    B, T, seed = 250, 4, 42
    kw = dict(num_bayes_layer=3, dropout_p=0.25, out_dim=10)
    m, o = build_seeded(bx.VGG11MC, kw), build_seeded(ox.VGG11MC, kw)
    synthetic_weights_(m, 0)
    synthetic_weights_(o, 0)
    x = synthetic_images(B, seed=1234)
    ref = mcd.mcd_predict(o, x, T, seed)
    out = {}
    for dt in ("bf16x3", "bf16"):
        r = _on(m, dt).engine(torch.device(DEV), max_batch=B, dtype=dt).predict(x.to(DEV), T, seed=seed)
        out[dt] = (float(np.abs(r["mean"].cpu().numpy() - ref["mean"]).max()), float(np.abs(r["var"].cpu().numpy() - ref["var"]).max()))
    print(f"VGG-11 B=250 T={T}: bf16x3 mean {out['bf16x3'][0]:.2e} var {out['bf16x3'][1]:.2e} | plain bf16 mean {out['bf16'][0]:.2e} var {out['bf16'][1]:.2e}")
    assert out["bf16x3"][0] <= 1e-3 and out["bf16x3"][1] <= 1e-3
    assert out["bf16x3"][0] <= 1e-4


def _peaky_(model, oracle_model, gain):
    """Trained-like logits: the classifiers' weights (every exit) x gain on both models, so that the softmax saturates (max prob >= 0.99 on
    most images) — where one 16-bit ulp of a large logit moves a probability by more than 1e-3."""
    with torch.no_grad():
        for mdl in (model, oracle_model):
            for name in ("ex1linear", "ex2linear", "ex3linear", "linear"):
                getattr(mdl, name).weight.mul_(gain)


@pytest.mark.gpu
def test_peaky_headline_model_fp16_vs_split_vs_oracle():
    """Stress: the headline model (BASELINE configs[2]) with its classifiers scaled until the predictive distribution is trained-like
    (max prob >= 0.99 on more than half of the images), B = 250, T = 4, HIP engines vs the fp32 CPU oracle on the same inputs and masks.
    The errors of all four engines are printed; the split engines must hold north_star's 1e-3 (their own bars: 2e-5 / 2e-4 x the gain's
    amplification), whatever plain fp16 / bf16 do here."""
    B, T, seed = 250, 4, 42
    kw = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
    m, o = build_seeded(ResNet18MCEarlyExit, kw), build_seeded(oresnet.ResNet18MCEarlyExit, kw)
    synthetic_weights_(m, 0)
    synthetic_weights_(o, 0)
    _peaky_(m, o, 24.0)
    x = synthetic_images(B, seed=1234)
    ref = mcd.mcd_predict(o, x, T, seed)
    conf = ref["mean"][-1].max(-1)
    peaky = float((conf >= 0.99).mean())
    out = {}
    for dt in ("f16", "bf16", "f16x2", "bf16x3"):
        r = _on(m, dt).engine(torch.device(DEV), max_batch=B, dtype=dt).predict(x.to(DEV), T, seed=seed)
        out[dt] = (float(np.abs(r["mean"].cpu().numpy() - ref["mean"]).max()), float(np.abs(r["var"].cpu().numpy() - ref["var"]).max()))
    print(f"peaky headline (max|logit| {float(np.abs(ref['logits']).max()):.0f}, final-exit max prob >= 0.99 on {100 * peaky:.0f} % of the images): "
          + " | ".join(f"{dt} mean {e[0]:.2e} var {e[1]:.2e}" for dt, e in out.items()))
    assert peaky >= 0.5
    assert out["f16x2"][0] <= 1e-3 and out["f16x2"][1] <= 1e-3
    assert out["bf16x3"][0] <= 1e-3 and out["bf16x3"][1] <= 1e-3
    assert out["f16x2"][0] <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f16x2", "bf16x3"])
def test_split_engine_repeats_bit_for_bit(dt):
    """Race screen: the headline model at B = 250, T = 3 five times over — every run bit for bit the first (the kernel hides its weight
    LDS-DMA from the compiler and reuses the staging LDS for its epilogue: a DMA still in flight there, or a missed barrier in the
    ping-pong loop, shows up as run-to-run differences at this size; the first LDS-epilogue build failed exactly so)."""
    kw = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
    m = _on(synthetic_weights_(build_seeded(ResNet18MCEarlyExit, kw), 0), dt)
    x = synthetic_images(250, seed=1234).to(DEV)
    eng = m.engine(torch.device(DEV), max_batch=250)
    first = eng.accumulate(x, eng.new_moments(250), 0, 3, seed=9).clone()
    for _ in range(4):
        again = eng.accumulate(x, eng.new_moments(250), 0, 3, seed=9)
        assert torch.equal(again, first)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f16x2", "bf16x3"])
def test_split_engine_resnet50_multi_exit(dt):
    """BASELINE configs[4]'s model (Bottleneck blocks: 1x1 convs with 64-2048 channels, every channel-tile width of the split kernel)
    against the build's fp32 oracle: the fp16 engine sits at 7.7e-4 of the 1e-3 bar there (tests/test_full_batch.py)."""
    B, T, seed = 32, 2, 42
    kw = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
    m, o = build_seeded(bx.ResNet50MCEarlyExit, kw), build_seeded(ox.ResNet50MCEarlyExit, kw)
    synthetic_weights_(m, 0)
    synthetic_weights_(o, 0)
    x = synthetic_images(B, seed=1234)
    ref = mcd.mcd_predict(o, x, T, seed)
    r = _on(m, dt).engine(torch.device(DEV), max_batch=B, dtype=dt).predict(x.to(DEV), T, seed=seed)
    err_m = float(np.abs(r["mean"].cpu().numpy() - ref["mean"]).max())
    err_v = float(np.abs(r["var"].cpu().numpy() - ref["var"]).max())
    print(f"ResNet-50 multi-exit {dt}: B={B} T={T} max|mean-oracle|={err_m:.2e} max|var-oracle|={err_v:.2e}")
    assert err_m <= 5 * TOLS[dt]["prob"] and err_v <= 5 * TOLS[dt]["prob"]


def _shrink_(model, f):
    """The same function with every (conv, BatchNorm) pair's conv output f times smaller: conv weights x f, the BN's running mean x f, running
    variance and eps x f^2 — BN(f y) with those statistics is BN(y) exactly.  What changes is the MAGNITUDE of the weights the engines must
    represent (f = 2^-11: He-init weights of ~0.02 become ~1e-5, under fp16's smallest normal number)."""
    from torch import nn
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, nn.Conv2d):
                mod.weight.mul_(f)
            elif isinstance(mod, nn.BatchNorm2d):
                mod.running_mean.mul_(f)
                mod.running_var.mul_(f * f)
                mod.eps = mod.eps * f * f
    return model


@pytest.mark.gpu
def test_small_magnitude_weights_keep_their_bits():
    """Round-5 advisor (low): the fp16 tail rn16(w - hi) of a weight below 2^-3 is an fp16 subnormal, so un-scaled BN-folded weights of 1e-3
    would keep 15 bits instead of 22 (and fp16 itself loses bits below 6.1e-5).  The host lifts every output channel's weights by an exact
    power of two before rounding / splitting and folds it back in the fp32 epilogue scale (engine.GraphBuilder.channel_lift): the headline
    model with ALL conv weights 2^-11 times smaller (the same function: the BatchNorms absorb the factor) must come out like the unshrunk
    model does — f16x2 at its usual 1e-5 of the oracle, fp16 at its usual few 1e-4."""
    B, T, seed = 64, 3, 42
    kw = dict(dropout_exit=True, dropout="block", dropout_p=0.25, out_dim=10)
    x = synthetic_images(B, seed=1234)
    out = {}
    for tag, f in (("plain", 1.0), ("shrunk", 2.0 ** -11)):
        m = _shrink_(synthetic_weights_(build_seeded(ResNet18MCEarlyExit, kw), 0), f)
        o = _shrink_(synthetic_weights_(build_seeded(oresnet.ResNet18MCEarlyExit, kw), 0), f)
        ref = mcd.mcd_predict(o, x, T, seed)
        for dt in ("f16x2", "f16"):
            r = _on(m, dt).engine(torch.device(DEV), max_batch=B, dtype=dt).predict(x.to(DEV), T, seed=seed)
            out[(tag, dt)] = float(np.abs(r["mean"].cpu().numpy() - ref["mean"]).max())
    print("max|mean - oracle|: " + " | ".join(f"{k[0]} {k[1]} {v:.2e}" for k, v in out.items()))
    assert out[("shrunk", "f16x2")] <= 5e-5 and out[("plain", "f16x2")] <= 5e-5
    assert out[("shrunk", "f16")] <= 1e-3
    assert out[("shrunk", "f16")] <= 2 * out[("plain", "f16")] + 1e-5          # a power of two commutes with every rounding on the way
