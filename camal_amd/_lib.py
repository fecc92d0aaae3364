"""ctypes binding of libcamal_hip.so (C ABI: include/camal_hip.h).

There is NO fallback: if the library is missing or a call fails, this raises.  The product
path never routes through ``oracle/`` or any CPU restatement.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CAMAL_HIP_LIB") or os.path.join(_HERE, "libcamal_hip.so")  # override: A/B builds

ARCH_CAMAL, ARCH_RESNET = 0, 1
FP32, BF16 = 0, 1
PROF_CLASSES = ("stem", "conv", "conv_cam_head", "post", "other")


class CamalHipError(RuntimeError):
    pass


_lib = None

_vp, _i, _d, _sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t

# name -> (restype, argtypes); every symbol include/camal_hip.h declares
SIGNATURES = {
    "camal_version": (C.c_char_p, []),
    "camal_last_error": (C.c_char_p, []),
    "camal_member_blob_floats": (_sz, [_i]),
    "camal_create": (_i, [_i, _i, C.POINTER(_i), _i, C.POINTER(_vp)]),
    "camal_create_arch": (_i, [_i, _i, C.POINTER(_i), _i, _i, C.POINTER(_vp)]),
    "camal_member_blob_floats_arch": (_sz, [_i, _i]),
    "camal_create_widths": (_i, [_i, _i, C.POINTER(_i), _i, _i, _i, C.POINTER(_vp)]),
    "camal_member_blob_floats_widths": (_sz, [_i, _i, _i]),
    "camal_destroy": (_i, [_vp]),
    "camal_set_member_weights": (_i, [_vp, _i, C.POINTER(C.c_float), _sz]),
    "camal_set_chunk_windows": (_i, [_vp, _i]),
    "camal_set_fused": (_i, [_vp, _i]),
    "camal_reserve": (_i, [_vp, _i, _i]),
    "camal_forward_cam": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "camal_forward_cam_tap": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "camal_localize": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "camal_localize_i8": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _d, _d, _vp, _vp, _vp, _vp]),
    "camal_postprocess": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _d, _d, _vp, _vp, _vp, _vp, _vp]),
    "camal_attention": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "camal_windows_count": (_i, [C.c_longlong, _i, _i]),
    "camal_windows_workspace_bytes": (_sz, [_i]),
    "camal_build_windows": (_i, [_vp, _i, _i, C.c_longlong, _i, _i, _i, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "camal_windows_static_scratch_bytes": (_sz, []),
    "camal_build_windows_static": (_i, [_vp, _i, _i, C.c_longlong, _i, _i, _i, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "camal_nilm_workspace_bytes": (_sz, [_sz]),
    "camal_nilm_sums": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _sz, _vp]),
    "camal_trainer_param_count": (_sz, [_i]),
    "camal_trainer_running_count": (_sz, []),
    "camal_trainer_create": (_i, [_i, _i, _i, _i, C.POINTER(_vp)]),
    "camal_trainer_create_arch": (_i, [_i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "camal_trainer_param_count_arch": (_sz, [_i, _i]),
    "camal_trainer_create_widths": (_i, [_i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "camal_trainer_param_count_widths": (_sz, [_i, _i, _i]),
    "camal_trainer_running_count_widths": (_sz, [_i]),
    "camal_trainer_destroy": (_i, [_vp]),
    "camal_trainer_set_grad_scale": (_i, [_vp, C.c_float]),
    "camal_train_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, C.c_float, _vp, _vp, _vp, _vp, _vp]),
    "camal_trainer_peek": (_i, [_vp, _i, _i, _i, _i, _vp, _sz, C.POINTER(_i), _vp]),
    "camal_train_bwd_dlogits": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "camal_ce_eval": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "camal_adamw_step": (_i, [_vp, _vp, _vp, _vp, _sz, _i, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _vp]),
    "camal_set_profiling": (_i, [_vp, _i]),
    "camal_get_profile": (_i, [_vp, C.POINTER(_i), C.POINTER(_d)]),
}


def load():
    """Load the library once; raises CamalHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CamalHipError(
            f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C camal_amd/csrc).  camal_amd has no CPU fallback.")
    # torch bundles its own libamdhip64 (same SONAME as the one hipcc links against).  Import it
    # FIRST so that the loader binds this library to the runtime torch already initialised: two
    # HIP runtimes in one process cannot share device pointers or streams.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code):
    if code != 0:
        msg = load().camal_last_error().decode("utf-8", "replace")
        raise CamalHipError(f"libcamal_hip error {code}: {msg}")


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())
