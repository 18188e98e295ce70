"""Helpers for the -m gpu parity tests: every call goes through the C ABI (ctypes)."""
import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

from bayesnn_fpga_amd import _lib
from oracle import philox

DEV = "cuda:0"


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def site_struct(site, keep):
    """site: None | dict(kind, site_id, p) | dict(kind, site_id, masks=np [M,C])."""
    if site is None:
        return None
    if site["kind"] == _lib.SITE_MASKSEMBLE:
        m = torch.from_numpy(np.ascontiguousarray(site["masks"], dtype=np.float32)).to(DEV)
        keep.append(m)
        return _lib.make_site(site["kind"], site["site_id"], 0.0, m.shape[0], m.data_ptr())
    return _lib.make_site(site["kind"], site["site_id"], site["p"])


def site_mask_ref(site, shape_bchw, seed, t, cnt0=0):
    """float32 multiplier [B,C,H,W] (or [B,C]) the site applies for sample t, from the oracle."""
    if site is None:
        return None
    if site["kind"] == _lib.SITE_ELEMENTWISE:
        return philox.elementwise_mask(shape_bchw, seed, site["site_id"], t, site["p"]) * float(philox.drop_scale(site["p"]))
    if site["kind"] == _lib.SITE_CHANNEL:
        m = philox.channel_mask(shape_bchw, seed, site["site_id"], t, site["p"]) * float(philox.drop_scale(site["p"]))
        return np.broadcast_to(m, shape_bchw).copy()
    row = np.asarray(site["masks"], dtype=np.float32)[(cnt0 + t) % len(site["masks"])]
    return np.broadcast_to(row.reshape((1, -1) + (1,) * (len(shape_bchw) - 2)), shape_bchw).copy()


def folded_site_mask(site, B, C_, H, W, tc, t0, seed, cnt0=0):
    """[tc*B, C, H, W] multiplier for a folded batch n = t_local*B + b."""
    if site is None:
        return None
    return torch.from_numpy(np.concatenate([site_mask_ref(site, (B, C_, H, W), seed, t0 + tl, cnt0) for tl in range(tc)]))


def conv_ref(x, w, scale, bias, res, relu, stride, pad, n, in_mod, res_mod):
    """fp32 CPU reference on the fp16-rounded operands.  x [n_in,H,W,Cin] fp16, w [Cout,k,k,Cin] fp16."""
    xi = x.float().cpu().permute(0, 3, 1, 2)[torch.arange(n) % in_mod]
    y = F.conv2d(xi, w.float().cpu().permute(0, 3, 1, 2), stride=stride, padding=pad)
    if scale is not None:
        y = y * scale.cpu()[None, :, None, None]
    if bias is not None:
        y = y + bias.cpu()[None, :, None, None]
    if res is not None:
        y = y + res.float().cpu().permute(0, 3, 1, 2)[torch.arange(n) % res_mod]
    if relu:
        y = torch.relu(y)
    return y        # NCHW fp32


def run_conv(x, w, scale, bias, res, relu, stride, pad, n, in_mod, res_mod, site=None, batch=None, t0=0, seed=0, cnt0=0,
             in_bits=None, out_mul=1.0, out_dtype=torch.float16):
    lib = _lib.lib()
    n_in, H, W, cin = x.shape
    cout, k = w.shape[0], w.shape[1]
    ho, wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    out = torch.full((n, ho, wo, cout), float("nan"), dtype=out_dtype, device=DEV)
    keep = []
    s = site_struct(site, keep)
    rc = lib.bmi_conv_igemm_fwd(ptr(x), ptr(in_bits), float(out_mul), ptr(w), ptr(scale), ptr(bias), ptr(res), ptr(out), n, in_mod, res_mod, H, W, cin,
                                cout, k, stride, pad, int(relu), C.byref(s) if s is not None else None,
                                batch if batch is not None else n, t0, seed, cnt0, stream())
    _lib.check(rc, "bmi_conv_igemm_fwd")
    torch.cuda.synchronize()
    return out


# ---- the split engines' activation layout ("pair32", csrc/conv_epilogue.h): per pixel, 32-channel blocks [32 heads | 32 tails] ---------
def pair32_encode(x, dt16):
    """fp32 [N, H, W, C] (C % 32 == 0) -> 16-bit [N, H, W, C/32, 2, 32]: hi = rn16(v), lo = rn16(v - hi)."""
    n, h, w, c = x.shape
    b = x.float().reshape(n, h, w, c // 32, 32)
    hi = b.to(dt16)
    lo = (b - hi.float()).to(dt16)
    return torch.stack([hi, lo], dim=4).contiguous()


def pair32_decode(p):
    """the inverse: 16-bit [N, H, W, C/32, 2, 32] -> fp32 [N, H, W, C]."""
    n, h, w, cb = p.shape[:4]
    return (p[..., 0, :].float() + p[..., 1, :].float()).reshape(n, h, w, cb * 32)
