"""ctypes binding of libnsr.so (the C ABI declared in include/nsr.h).

This is exactly the stub a reference maintainer would add: plain
pointers and sizes, torch only supplies device memory (``tensor.data_ptr()``) and
the current HIP stream.  There is no fallback: a missing library or a missing
symbol raises at import/first use, and every non-zero status becomes a RuntimeError.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NSR_LIB_PATH", os.path.join(_HERE, "libnsr.so"))   # override: ablation builds (development)

NSR_FP32, NSR_BF16, NSR_F16X3, NSR_F16 = 0, 1, 2, 3
NSR_ERR_RANGE = -5
# numerics status word of a packed network (include/nsr.h)
FLAGS = {1: "WEIGHT_RANGE", 2: "INPUT_RANGE", 4: "ACTIVATION_RANGE", 8: "OUTPUT_NONFINITE"}
NSR_FLAG_OUTPUT_NONFINITE = 8
NSR_F16X3_GEMM = 18   # include/nsr_train.h: training entry points only
NSR_OPT_GAMMA, NSR_OPT_COLOR_NONE = 1, 2     # include/nsr.h: colour-head option word (nsr_weights_set_options)
NSR_WHITE_BKGD, NSR_SIGMA_SOFTPLUS = 1, 2    # include/nsr.h: renderer option word (the `white_bkgd` argument)
NSR_TRAIN_GAMMA_CORRECT, NSR_TRAIN_COLOR_NONE, NSR_TRAIN_STOP_GRAD = 4, 8, 16    # include/nsr_train.h: the training entry points' own bits of that word
PRECISIONS = {"fp32": NSR_FP32, "bf16": NSR_BF16, "f16x3": NSR_F16X3, "f16": NSR_F16}
NSR_F16X3_BWD3, NSR_F16X3_BWD2, NSR_F16X3_BWD1, NSR_F16X3_BWDM = 19, 20, 21, 22   # chain path, MFMAs per product of the backward chain named explicitly
TRAIN_PRECISIONS = {"fp32": NSR_FP32, "f16x3": NSR_F16X3, "f16x3_gemm": NSR_F16X3_GEMM,
                    "f16x3_bwd3": NSR_F16X3_BWD3, "f16x3_bwd2": NSR_F16X3_BWD2, "f16x3_bwd1": NSR_F16X3_BWD1, "f16x3_bwdm": NSR_F16X3_BWDM}

# symbol -> (restype, argtypes); must list every function of include/*.h
SIGNATURES = {
    "nsr_version": (c_int, []),
    "nsr_status_string": (c_char_p, [c_int]),
    "nsr_packed_weights_bytes": (c_size_t, [c_int]),
    "nsr_pack_weights": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_void_p]),
    "nsr_pack_weights_async": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_void_p]),
    "nsr_weights_status": (c_int, [c_void_p, c_int, c_int, POINTER(c_uint), c_void_p]),
    "nsr_weights_set_gamma": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "nsr_weights_set_options": (c_int, [c_void_p, c_int, ctypes.c_uint, c_void_p]),
    "nsr_gen_rays": (c_int, [POINTER(c_float), c_int, c_int, c_double, c_int, c_int, c_float, c_float, c_void_p, c_void_p]),
    "nsr_gen_rays_range": (c_int, [POINTER(c_float), c_int, c_int, c_double, c_int, c_int, c_float, c_float, c_int64, c_int64,
                                   c_void_p, c_void_p]),
    "nsr_posenc": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "nsr_sample_along_rays": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsr_mlp_forward": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "nsr_render_rays": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "nsr_render_rays_composited": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsr_composite": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_int,
                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsr_resample_along_rays": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p,
                                        c_void_p, c_void_p, c_void_p]),
    "nsr_forward_rays_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "nsr_forward_rays_workspace_bytes_for": (c_size_t, [c_int, c_int64, c_int, c_int]),
    "nsr_forward_rays": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int,
                                 POINTER(c_void_p), c_void_p, c_size_t, c_void_p]),
    "nsr_forward_rays_profiled": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int,
                                          POINTER(c_void_p), c_void_p, c_size_t, c_void_p, POINTER(c_void_p)]),
    "nsr_event_create": (c_int, [POINTER(c_void_p)]),
    "nsr_event_destroy": (c_int, [c_void_p]),
    "nsr_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "nsr_sr_mean": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "nsr_unflatten": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    # ---- include/nsr_train.h
    "nsr_train_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "nsr_train_workspace_bytes_for": (c_size_t, [c_int, c_int64, c_int, c_int]),
    "nsr_train_loss_and_grads": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                         c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_int, c_int64,
                                         POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nsr_train_loss_and_grads_var": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                             c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_int, c_int64,
                                             POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p,
                                             c_void_p, c_void_p]),
    "nsr_train_status_reset": (c_int, [c_void_p, c_void_p]),
    "nsr_train_status": (c_int, [c_void_p, c_int, POINTER(c_uint), c_void_p]),
    "nsr_adam_step": (c_int, [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_int,
                              c_float, c_float, c_float, c_float, c_void_p]),
    # ---- include/nsr_warp.h
    "nsr_depth_warp": (c_int, [c_void_p, c_int, c_int, c_double, POINTER(c_float), POINTER(c_double), c_int, c_void_p,
                               c_void_p, c_void_p, c_void_p]),
    # ---- include/nsr_refine.h
    "nsr_refine_packed_bytes": (c_size_t, [c_int]),
    "nsr_refine_pack_weights": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_void_p]),
    "nsr_refine_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "nsr_refine_workspace_bytes_for": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "nsr_refine_forward": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                   c_size_t, c_void_p]),
    "nsr_refine_packed_bytes_noref": (c_size_t, [c_int]),
    "nsr_refine_pack_weights_noref": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_void_p]),
    "nsr_refine_forward_noref": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "nsr_refine_tile": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "nsr_refine_gather": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                  c_void_p, c_void_p]),
    "nsr_refine_stitch": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    # ---- include/nsr_image.h
    "nsr_lanczos_ksize": (c_int, [c_int, c_int]),
    "nsr_lanczos_coeffs": (c_int, [c_int, c_int, c_void_p, c_void_p]),
    "nsr_resample_pass_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "nsr_image_to_targets": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "nsr_rgba_premultiply_u8": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "nsr_image_to_targets_rgba": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "nsr_linear": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64,
                           c_int64, c_int, c_int, c_void_p]),
    "nsr_split_weights": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "nsr_linear_f16x3": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64, c_int64, c_int, c_int,
                                 c_void_p]),
}

_lib = None


class NsrError(RuntimeError):
    """A libnsr entry point returned a non-zero nsr_status."""


class NsrNumericsError(NsrError, FloatingPointError):
    """The numerics status word of a packed network is non-zero: non-finite inputs / outputs, or values outside the
    operand range of the precision -- what the reference traps with ``isnan(out_rgbs).any()`` ->
    ``pdb.set_trace()`` (models/nerf_downX_model.py:273-274)."""

    def __init__(self, msg: str, flags: int = 0):
        super().__init__(msg)
        self.flags = flags


def flag_names(flags: int):
    return [name for bit, name in FLAGS.items() if flags & bit]


def load() -> ctypes.CDLL:
    """dlopen libnsr.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch FIRST: libnsr.so needs libamdhip64.so.7 and so does torch, which ships its own copy; the dynamic linker keeps
    # whichever instance of that SONAME is loaded first for the whole process.  With libnsr first the process ends up on the
    # system runtime underneath torch's other bundled ROCm libraries and the first kernel launch fails (seen on the GPU box
    # when build() -- which binds the library without touching torch -- ran in front of smoke() in one process).
    try:
        import torch  # noqa: F401
    except ImportError:      # a torch-less host can still bind the C ABI (host-only entry points, symbol checks)
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m nerf_sr_amd.build` "
            "(needs hipcc). nerf_sr_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().nsr_status_string(status).decode()
        raise NsrError(f"{what} failed: nsr_status {status} ({msg})")
