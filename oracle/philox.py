"""Philox4x32-10 and the dropout-mask convention (TEST ORACLE — see oracle/__init__.py).

Algorithm: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3"
(SC'11), Random123 ``philox4x32_R(10, ctr, key)``; restated from the paper, pinned by
the Random123 known-answer vectors in ``tests/test_philox.py``.

Mask convention shared by oracle and HIP kernels (bayesnn_fpga_amd/csrc/philox.h):

  bits     k = the fewest of {2, 4, 8, 16} bits per element with fl32(p) * 2**k an integer, else 16
             (p = 0.25 or 0.5 -> 2, 0.125 / 0.375 -> 4, 0.1 -> 16: quantised to 1/65536)
  key      = (seed & 0xffffffff, seed >> 32)
  counter  = (g & 0xffffffff, g >> 32, t, site)          g = element_index // (128 // k)
  element  e uses field f = e % (128 // k) of the call's 128 output bits: bits [f*k, (f+1)*k) of r[0] | r[1]<<32 | ...
             (k = 16: the 16-bit half (e & 1) of word r[(e % 8) // 2], low half first)
  keep(e)  = field >= thresh,   thresh = min(floor(fl32(p) * 2**k + 0.5), 2**k)      P(drop) = thresh / 2**k
  out      = x * keep * fl32(1 / fl32(1 - p))            (MCDropout: F.dropout, always on,
                                                          SA/models/resnet18/resnet18.py:209-210)
The field width follows p because the Philox call is what the GPU pays for (v_mad_u64_u32 issues at a quarter rate):
at p = 0.25 one call masks 64 elements instead of 8.

``element_index`` is the NHWC-linear index inside ONE Monte-Carlo sample's activation of
logical shape [B, C, H, W]:  ((b*H + h)*W + w)*C + c  ([B, C] tensors have H = W = 1), so
a group of 8 consecutive channels of a pixel shares one Philox call.  ``t`` is the global
Monte-Carlo sample index and ``site`` the call-order index of the stochastic layer inside
one forward.  Channel-wise sites (``F.dropout2d`` semantics of
``Hardware_Artifact/converter/pytorch/Dropouts.py:25-56``) use element_index = b*C + c.
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  Counter words are array-likes (broadcast together),
    key words are scalars.  Returns 4 uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(
        np.asarray(c0, dtype=np.uint64), np.asarray(c1, dtype=np.uint64),
        np.asarray(c2, dtype=np.uint64), np.asarray(c3, dtype=np.uint64))
    c0 = c0 & _MASK32; c1 = c1 & _MASK32; c2 = c2 & _MASK32; c3 = c3 & _MASK32
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = PHILOX_M0 * c0            # 32x32 -> 64 bit products, exact in uint64
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0)
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return (c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32))


def site_bits(p):
    """Bits drawn per element for drop probability p: 2, 4, 8 when fl32(p) * 2**k is an integer, else 16."""
    pf = float(np.float32(p))
    for k in (2, 4, 8):
        if pf * (1 << k) == np.floor(pf * (1 << k)):
            return k
    return 16


def drop_threshold(p, k=None):
    """k-bit threshold (k = site_bits(p) by default): keep iff field >= thresh.  P(drop) = thresh / 2**k."""
    k = site_bits(p) if k is None else k
    return min(int(np.floor(float(np.float32(p)) * float(1 << k) + 0.5)), 1 << k)


def drop_scale(p):
    """fl32(1 / fl32(1 - p)); 0 when p >= 1 (everything is dropped anyway)."""
    q = np.float32(1.0) - np.float32(p)
    if q <= 0:
        return np.float32(0.0)
    return np.float32(1.0) / q


def keep_bits(n_elems, seed, site, t, p):
    """Boolean keep-mask for elements 0..n_elems-1 of one (seed, site, t) stream."""
    k = site_bits(p)
    per_call = 128 // k
    n_groups = (n_elems + per_call - 1) // per_call
    g = np.arange(n_groups, dtype=np.uint64)
    r = philox4x32_10(g & _MASK32, g >> np.uint64(32), np.uint64(t), np.uint64(site),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.stack(r, axis=1).astype(np.uint32)                        # [groups, 4]
    f = np.arange(per_call, dtype=np.uint32) * np.uint32(k)              # bit position of field f inside the call
    fields = (words[:, f >> np.uint32(5)] >> (f & np.uint32(31))) & np.uint32((1 << k) - 1)     # [groups, per_call]
    thr = drop_threshold(p, k)
    if thr >= (1 << k):
        return np.zeros(n_elems, dtype=bool)
    return (fields.reshape(-1)[:n_elems] >= thr)


def elementwise_mask(shape, seed, site, t, p):
    """float32 keep-mask (0/1) with the logical NCHW ``shape`` ([B,C,H,W] or [B,C]);
    generated in NHWC-linear element order (see module docstring)."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape))
    bits = keep_bits(n, seed, site, t, p).astype(np.float32)
    if len(shape) == 4:
        b, c, h, w = shape
        return np.ascontiguousarray(bits.reshape(b, h, w, c).transpose(0, 3, 1, 2))
    if len(shape) == 2:
        return bits.reshape(shape)
    raise ValueError(f"unsupported dropout site shape {shape}")


def channel_mask(shape, seed, site, t, p):
    """float32 keep-mask of shape [B,C,1,1] (dropout2d semantics), element = b*C + c."""
    b, c = int(shape[0]), int(shape[1])
    bits = keep_bits(b * c, seed, site, t, p).astype(np.float32)
    return bits.reshape(b, c, *([1] * (len(shape) - 2)))
