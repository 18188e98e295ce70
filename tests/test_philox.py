"""Philox4x32-10 known-answer tests (Random123 kat_vectors) and the mask convention."""
import numpy as np
import pytest

from oracle import philox
from tests.helpers import load_golden

# Random123 `kat_vectors`: philox4x32 10  ctr[4] key[2] -> out[4]
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox_kat():
    for ctr, key, out in KAT:
        r = philox.philox4x32_10(*ctr, *key)
        assert tuple(int(v) for v in r) == out


def test_philox_vectorised_matches_scalar():
    g = np.arange(37, dtype=np.uint64)
    r = philox.philox4x32_10(g, 0, 5, 2, 123, 456)
    for i in (0, 1, 17, 36):
        s = philox.philox4x32_10(i, 0, 5, 2, 123, 456)
        assert all(int(a[i]) == int(b) for a, b in zip(r, s))


def test_threshold_and_scale():
    assert [philox.site_bits(p) for p in (0.0, 0.25, 0.5, 0.75, 1.0, 0.125, 0.375, 0.0625, 1 / 256, 0.1, 0.2, 0.3)] == \
        [2, 2, 2, 2, 2, 4, 4, 4, 8, 16, 16, 16]
    assert philox.drop_threshold(0.0) == 0
    assert philox.drop_threshold(0.25) == 1                         # 2 bits per element: drop iff field == 0
    assert philox.drop_threshold(0.5) == 2
    assert philox.drop_threshold(1.0) == 4                          # == 2**k: everything dropped
    assert philox.drop_threshold(0.375) == 6 and philox.drop_threshold(0.125) == 2
    assert philox.drop_threshold(0.2) == 13107                      # 16 bits: quantised to 1/65536
    assert philox.drop_scale(0.25) == np.float32(1.0) / np.float32(0.75)
    assert philox.drop_scale(1.0) == 0.0


@pytest.mark.parametrize("p", [0.25, 0.375, 1 / 256, 0.2])
def test_mask_layout_rule(p):
    """element index is NHWC-linear: ((b*H+h)*W+w)*C+c; 128 // k consecutive elements per call, k bits each
    (k = 2, 4, 8, 16 by p), fields in little-endian bit order over r[0..3]."""
    seed, site, t = 42, 3, 5
    B, C, H, W = 2, 24, 3, 5
    k = philox.site_bits(p)
    m = philox.elementwise_mask((B, C, H, W), seed, site, t, p)
    thr = philox.drop_threshold(p)
    for (b, c, h, w) in [(0, 0, 0, 0), (1, 5, 2, 1), (0, 23, 1, 0), (1, 3, 0, 4), (1, 17, 2, 4)]:
        e = ((b * H + h) * W + w) * C + c
        r = philox.philox4x32_10(e // (128 // k), 0, t, site, seed, 0)
        bits128 = sum(int(r[i]) << (32 * i) for i in range(4))
        field = (bits128 >> ((e % (128 // k)) * k)) & ((1 << k) - 1)
        assert m[b, c, h, w] == float(field >= thr)


def test_mask_statistics():
    m = philox.elementwise_mask((64, 512), 1, 0, 0, 0.25)
    assert abs(m.mean() - 0.75) < 0.01
    assert philox.elementwise_mask((1, 8), 0, 0, 0, 0.0).all()
    assert not philox.elementwise_mask((1, 8), 9, 2, 1, 1.0).any()


def test_golden_masks():
    g = load_golden("philox_masks.npz")
    for case in g["cases"]:
        m = philox.elementwise_mask(case["shape"], case["seed"], case["site"], case["t"], case["p"])
        assert np.array_equal(m.astype(np.uint8), case["mask"])


def test_streams_disjoint():
    a = philox.elementwise_mask((4, 512), 42, 0, 0, 0.5)
    assert not np.array_equal(a, philox.elementwise_mask((4, 512), 42, 0, 1, 0.5))
    assert not np.array_equal(a, philox.elementwise_mask((4, 512), 42, 1, 0, 0.5))
    assert not np.array_equal(a, philox.elementwise_mask((4, 512), 43, 0, 0, 0.5))
