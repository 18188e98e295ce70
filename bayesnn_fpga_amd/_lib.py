"""ctypes binding of libbayesnn_fpga_amd.so (C ABI in include/bayesnn_fpga_amd.h).

There is NO CPU fallback: if the library is missing or cannot be loaded, every product entry
point raises.  Build it with ``python -m bayesnn_fpga_amd._build`` (or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libbayesnn_fpga_amd.so")

BMI_OK = 0
SITE_NONE, SITE_ELEMENTWISE, SITE_CHANNEL, SITE_MASKSEMBLE = 0, 1, 2, 3
SITE_POS_OUTER, SITE_POS_INNER = 0, 1
DTYPE_F16, DTYPE_BF16, DTYPE_F32, DTYPE_F16X2, DTYPE_BF16X3 = 0, 1, 2, 3, 4
DTYPES = {"f16": DTYPE_F16, "bf16": DTYPE_BF16, "f32": DTYPE_F32, "f16x2": DTYPE_F16X2, "bf16x3": DTYPE_BF16X3}
FP32_ACT_DTYPES = ("f32", "f16x2", "bf16x3")      # engines that keep fp32 activations in the workspace
OP_STEM, OP_CONV, OP_MASK, OP_HEAD, OP_MAXPOOL, OP_DENSE = 1, 2, 3, 4, 5, 6
PROFILE_SLOTS = 8
CONV_FAMILY_KERNELS = ("conv3x3_patch_kernel", "conv_igemm_wide_kernel", "conv_igemm_kernel", "conv3x3_pw_kernel", "conv1x1_stream_kernel",
                       "conv3x3_s2_kernel", "conv_split_kernel", "conv1x1_seam_kernel")
ABI_VERSION = 600             # BMI_VERSION of include/bayesnn_fpga_amd.h this binding was written against
CONV_FAMILIES = len(CONV_FAMILY_KERNELS)     # BMI_CONV_FAMILIES
PROFILE_NAMES = {OP_STEM: "stem", OP_CONV: "conv_igemm", OP_MASK: "mask", OP_HEAD: "head", OP_MAXPOOL: "maxpool",
                 OP_DENSE: "dense"}


class Site(C.Structure):
    _fields_ = [("kind", C.c_int32), ("site_id", C.c_int32), ("p", C.c_float), ("num_masks", C.c_int32),
                ("masks", C.c_void_p)]


class TensorDesc(C.Structure):
    _fields_ = [("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32)]


class OpDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("in_", C.c_int32), ("out", C.c_int32), ("residual", C.c_int32), ("in2", C.c_int32),
                ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("relu", C.c_int32),
                ("weight", C.c_void_p), ("weight2", C.c_void_p), ("scale", C.c_void_p), ("bias", C.c_void_p), ("site", Site),
                ("bias_post", C.c_void_p), ("site_pos", C.c_int32)]


class ModelDesc(C.Structure):
    _fields_ = [("n_tensors", C.c_int32), ("tensors", C.POINTER(TensorDesc)), ("n_ops", C.c_int32),
                ("ops", C.POINTER(OpDesc)), ("n_exits", C.c_int32), ("out_dim", C.c_int32), ("dtype", C.c_int32)]


class BmiError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__(f"{what} failed: {error_string(code)} ({code})")


_lib = None

_PROTOS = {
    "bmi_version": (C.c_int, []),
    "bmi_error_string": (C.c_char_p, [C.c_int]),
    "bmi_set_option": (C.c_int, [C.c_char_p, C.c_int32]),
    "bmi_engine_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "bmi_create": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]),
    "bmi_destroy": (C.c_int, [C.c_void_p]),
    "bmi_plan": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "bmi_query": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                            C.POINTER(C.c_int32)]),
    "bmi_tensor_info": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)] + [C.POINTER(C.c_int32)] * 5),
    "bmi_image_offset_ok": (C.c_int, [C.c_void_p, C.c_int32]),
    "bmi_forward_mcd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bmi_forward_mcd_images": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bmi_forward_mcd_samples": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_int32, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "bmi_forward_mcd_exit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_double, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_size_t,
                                       C.c_void_p]),
    "bmi_finalize": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p]),
    "bmi_finalize_checked": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "bmi_profile_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "bmi_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "bmi_philox_mask": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "bmi_stem_conv_fwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 9 + [C.c_void_p]),
    "bmi_mask_bits": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Site), C.c_int32, C.c_int32, C.c_uint64,
                                C.c_void_p]),
    "bmi_conv_igemm_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float] + [C.c_void_p] * 5 + [C.c_int32] * 11 + [C.POINTER(Site), C.c_int32, C.c_int32,
                                                                         C.c_uint64, C.c_int32, C.c_void_p]),
    "bmi_conv3x3_shortcut_fwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 7 + [C.c_void_p]),
    "bmi_profile_launches": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)] + [C.POINTER(C.c_int32)] * 4 + [C.POINTER(C.c_double)] * 3),
    "bmi_profile_conv_families": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                            C.POINTER(C.c_double)]),
    "bmi_conv1x1_seam_fwd": (C.c_int, [C.c_void_p] * 10 + [C.c_int32] * 7 + [C.c_void_p]),
    "bmi_conv_pair_fwd": (C.c_int, [C.c_void_p] * 9 + [C.c_int32] * 11 + [C.c_void_p]),
    "bmi_mask_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Site),
                                 C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_void_p]),
    "bmi_maxpool2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "bmi_head_fused": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                 C.POINTER(Site), C.POINTER(Site), C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.c_int32,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bmi_dense_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 5 +
                      [C.POINTER(Site), C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_void_p]),
}

EXPORTS = tuple(_PROTOS.keys())


def lib():
    """Loads the library (once).  Raises if it is absent — there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP library has not been built "
                "(run `python -m bayesnn_fpga_amd._build`).  bayesnn_fpga_amd has no CPU fallback.")
        # torch first: PyTorch-ROCm ships its own libamdhip64, and the process must end up with ONE HIP runtime.  Loaded
        # before torch, this library binds the system ROCm runtime instead; the streams and device pointers torch
        # hands over then belong to a different runtime and every launch fails (BMI_ERR_HIP).
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.bmi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} reports C-ABI version {l.bmi_version()}, this binding expects {ABI_VERSION}: "
                               "rebuild the library (python -m bayesnn_fpga_amd._build)")
        _lib = l
        # BMI_OPTIONS="name=value,name=value": bmi_set_option pairs applied once at load (profiling arms: rocprofv3 wraps bench.py,
        # which has no option flag of its own)
        for kv in filter(None, os.environ.get("BMI_OPTIONS", "").split(",")):
            nm, _, val = kv.partition("=")
            if l.bmi_set_option(nm.strip().encode(), int(val)) != BMI_OK:
                raise RuntimeError(f"BMI_OPTIONS: bmi_set_option({nm.strip()}, {val}) failed")
    return _lib


def error_string(code):
    return lib().bmi_error_string(int(code)).decode()


def check(code, what):
    if code != BMI_OK:
        raise BmiError(code, what)


def set_option(name, value):
    """Process DEFAULT of a kernel-selection switch (bmi_set_option): e.g. set_option("mfma_shape_patch", 16).  Read by the single-kernel entry
    points and COPIED into every engine created afterwards; a live engine keeps its copy (``MCDEngine.set_option`` edits one engine's)."""
    check(lib().bmi_set_option(name.encode(), int(value)), f"bmi_set_option({name})")


def make_site(kind=SITE_NONE, site_id=0, p=0.0, num_masks=0, masks_ptr=None):
    return Site(int(kind), int(site_id), float(p), int(num_masks), masks_ptr)
