"""Build-time guard for the conv kernels (no GPU needed: hipcc cross-compiles): every product instantiation must keep
its accumulators in registers — ScratchSize 0 and no VGPR spill.  A correct-but-10x-slower build slipped through the
parity tests once (an extra inlined Philox expansion pushed the epilogue's item loop past the full-unroll budget and
the accumulator arrays moved to scratch), which is what this pins."""
import os
import re
import shutil
import subprocess

import pytest

from bayesnn_fpga_amd import _build

FILES = ["conv3x3_patch.hip", "conv3x3_pw.hip", "conv3x3_s2.hip", "conv_igemm_wide.hip", "conv_igemm.hip", "conv1x1_stream.hip", "head_fused.hip"]


@pytest.mark.parametrize("src", FILES)
def test_conv_kernels_have_no_scratch_and_no_spills(src):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    r = subprocess.run([hipcc, *_build.FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(_build.CSRC, src),
                        "-o", os.devnull], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    kernels = re.findall(r"Function Name: (\S+).*?VGPRs Spill: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)", r.stderr, flags=re.S)
    assert kernels, "no kernel-resource-usage remarks in the hipcc output"
    # conv3x3_s2's dynamic-exit instantiations (IMAP = true, the third template argument of five) keep a few tile-setup values in scratch
    # OUTSIDE the main loop (stored before it, reloaded behind it: checked in the ISA when the kernel was written — between the first
    # and the last v_mfma there is no scratch instruction): tolerated up to 256 bytes per lane, nothing else is
    # (conv3x3_pw4, the four-wave measurement reference: 256 accumulator AGPRs + 256 VGPRs, 2-5 values spilled around the epilogue)
    allowed = lambda n: (256 if re.fullmatch(r"_Z17conv3x3_s2_kernelILi\d+ELb[01]ELb1ELb0ELi\d+EEv8ConvArgsi", n) else
                         32 if n.startswith("_Z18conv3x3_pw4_kernel") else 0)
    bad = [(n, sp, sc) for n, sp, sc in kernels if int(sc) > allowed(n) or (int(sp) and not allowed(n))]
    assert not bad, f"kernels with VGPR spills / scratch: {bad}"
