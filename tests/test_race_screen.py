"""-m gpu: race screen of the LDS-DMA conv kernels.  The ping-pong / persistent main loops of conv_igemm_wide and the
patch kernel order their LDS-DMA writes and fragment reads by counted s_waitcnt and raw s_barrier only (hazard table in
csrc/conv_igemm_wide.hip); a misplaced wait shows up as RARE wrong tiles that depend on memory timing, which a single
parity check does not catch.  Every shape is launched 24 times beside 0.5 GB of concurrent HBM traffic on a second
stream; all 24 outputs must equal the first launch bit for bit.  Sizes are chosen so that every main-loop variant runs:
wide per-tile (3/4 n_cu <= tiles <= n_cu), wide persistent (tiles > n_cu), 1x1 taps, the three patch geometries — under
both MFMA shapes."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

SHAPES = [  # cin, cout, H, k, s, p, images
    (128, 256, 16, 3, 2, 1, 1000),   # D3: 250 tiles  -> wide kernel, one workgroup per tile
    (128, 256, 16, 3, 2, 1, 3000),   # D3: 750 tiles  -> persistent wide kernel
    (256, 512, 8, 3, 2, 1, 1800),    # D4: 226 tiles, two channel tiles
    (256, 512, 8, 3, 2, 1, 5000),    # D4: 626 tiles  -> persistent
    (256, 512, 8, 1, 2, 0, 5000),    # P4: 1x1 taps (4 K-steps per tile), persistent
    (128, 128, 16, 3, 1, 1, 257), (256, 256, 8, 3, 1, 1, 515), (512, 512, 4, 3, 1, 1, 1031),   # patch kernel S2 / S3 / S4
    (256, 256, 8, 3, 1, 1, 1030), (512, 512, 4, 3, 1, 1, 4100),                                # conv3x3_pw (>= 3/4 tile per CU)
    # BasicBlock tails (residual + ReLU + p = 0.25 elementwise site): the lite epilogue DMAs the residual into the LDS image the
    # results are written back to in place — patch S2, conv3x3_pw S3 / S4, wide 1x1
    (128, 128, 16, 3, 1, 1, 259, 1), (256, 256, 8, 3, 1, 1, 1027, 1), (512, 512, 4, 3, 1, 1, 4099, 1), (256, 512, 8, 1, 1, 0, 3001, 1),
    # conv1x1_stream (single-buffered K-steps refilled behind a barrier, two workgroups per CU): Bottleneck tail and a plain
    # Cout = 128 reduce conv
    (128, 512, 16, 1, 1, 0, 1203, 1), (512, 128, 16, 1, 1, 0, 777),
]


@pytest.mark.parametrize("mfma_shape", [16, 32])
def test_repeat_launches_are_bit_identical_under_memory_traffic(mfma_shape):
    import race_screen
    from bayesnn_fpga_amd import _lib
    for k in ("mfma_shape_patch", "mfma_shape_wide"):
        _lib.set_option(k, mfma_shape)
    _lib.set_option("conv_pw", 1)
    _lib.set_option("conv_s2", 0)                # the plain stride-2 shapes of this list screen conv_igemm_wide; conv3x3_s2 below
    try:
        bad = race_screen.screen(SHAPES, rounds=24, verbose=True)
    finally:
        for k in ("mfma_shape_patch", "mfma_shape_wide"):
            _lib.set_option(k, 0)
        _lib.set_option("conv_s2", 1)
    assert bad == 0, f"{bad} launches differed from the first launch of the same inputs"


# conv3x3_s2 (persistent, patch planes refilled in place behind counted waits): fewer tiles than CUs, a few tiles per CU, many;
# one to four channel tiles; 2, 4 and 8 channel chunks
S2_SHAPES = [
    (64, 256, 32, 3, 2, 1, 230), (64, 256, 32, 3, 2, 1, 1500), (128, 256, 16, 3, 2, 1, 1000), (128, 512, 16, 3, 2, 1, 3001),
    (256, 512, 8, 3, 2, 1, 1800), (256, 1024, 8, 3, 2, 1, 5003), (256, 512, 8, 3, 2, 1, 12000),
    (128, 128, 32, 3, 2, 1, 300), (256, 128, 32, 3, 2, 1, 1300),      # the 128-channel tiles (round 4): a tile or five per CU
]


def test_s2_repeat_launches_are_bit_identical_under_memory_traffic():
    import race_screen
    bad = race_screen.screen(S2_SHAPES, rounds=24, verbose=True)
    assert bad == 0, f"{bad} launches differed from the first launch of the same inputs"
