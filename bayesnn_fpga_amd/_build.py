"""Builds libbayesnn_fpga_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so
travels to the GPU box with the snapshot (it is git-ignored, not gpurun-ignored).
"""
import hashlib
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


def _sha(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "bayesnn_fpga_amd.h")]


def _want(src, hdr_sha):
    """Content key of one object: its source, every header, the flags (NOT mtimes: a snapshot copied to the GPU box, or a checkout, gives
    every file a fresh timestamp in arbitrary order, and an object that was not recompiled could look newer than its edited source)."""
    return _sha([os.path.join(CSRC, src)], hdr_sha + " ".join(FLAGS))


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _lib_key(keys):
    return hashlib.sha256("".join(keys).encode()).hexdigest()


def _stale():
    if not os.path.exists(LIB):
        return True
    hdr_sha = _sha(_headers())
    return _read(LIB + ".sha256") != _lib_key([_want(src, hdr_sha) for src in SOURCES])


def build(force=False, verbose=False):
    """Compiles what is stale BY CONTENT: beside every object sits the sha256 of (its source + all headers + the flags) it was compiled
    from, beside the library the digest of those keys; anything whose key differs is rebuilt, objects of sources that no longer exist are
    removed (csrc/build/ and the stamps travel to the GPU box with the snapshot)."""
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    wanted = {src.replace(".hip", ".o") for src in SOURCES}
    for f in os.listdir(objdir):
        if (f.endswith(".o") and f not in wanted) or (f.endswith(".o.sha256") and f[:-len(".sha256")] not in wanted):
            os.remove(os.path.join(objdir, f))
    hdr_sha = _sha(_headers())

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        key = _want(src, hdr_sha)
        if not force and os.path.exists(obj) and _read(obj + ".sha256") == key:
            return obj, key
        cmd = [hipcc, *FLAGS, "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        with open(obj + ".sha256", "w") as f:
            f.write(key)
        return obj, key

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        done = list(ex.map(cc, SOURCES))
    objs = [o for o, _ in done]
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(LIB + ".sha256", "w") as f:
        f.write(_lib_key([k for _, k in done]))
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
