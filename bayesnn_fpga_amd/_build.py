"""Builds libbayesnn_fpga_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so
travels to the GPU box with the snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbayesnn_fpga_amd.so")
SOURCES = ["conv_igemm.hip", "conv_igemm_wide.hip", "conv3x3_patch.hip", "conv3x3_pw.hip", "conv3x3_s2.hip", "conv1x1_stream.hip", "conv1x1_seam.hip", "conv_exact.hip", "conv_split.hip", "misc_kernels.hip", "dense_f32.hip", "head_fused.hip", "engine.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build the gfx950 library")
    return exe


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "bayesnn_fpga_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compiles what is stale: an object is rebuilt when its source or any header is newer than it; objects of sources that no
    longer exist are removed (csrc/build/ travels to the GPU box with the snapshot)."""
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    wanted = {src.replace(".hip", ".o") for src in SOURCES}
    for f in os.listdir(objdir):
        if f.endswith(".o") and f not in wanted:
            os.remove(os.path.join(objdir, f))
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "bayesnn_fpga_amd.h")]
    t_hdr = max(os.path.getmtime(h) for h in headers)

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(t_hdr, os.path.getmtime(path)):
            return obj
        cmd = [hipcc, *FLAGS, "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(cc, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
