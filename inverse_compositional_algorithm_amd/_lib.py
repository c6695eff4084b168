"""ctypes binding of libica_hip.so (include/ica_hip.h).  There is no fallback: if the HIP library is
missing the import of the compute path fails loudly."""
from __future__ import annotations

import ctypes as C
import os

ICA_ABI_VERSION = 4   # include/ica_hip.h
ICA_MAX_SCALES = 16
ICA_TRACE_STRIDE = 10
ICA_FLAG_EXACT_UPDATE = 1
ICA_STORAGE_F32, ICA_STORAGE_U8 = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ICA_LIB", os.path.join(_HERE, "libica_hip.so"))  # ICA_LIB: experiment builds only


class IcaError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libica_hip status {status}: {message}")
        self.status = status


class Problem(C.Structure):
    _fields_ = [("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("channels", C.c_int32),
                ("transform_type", C.c_int32), ("robust_type", C.c_int32), ("nscales", C.c_int32),
                ("max_iter", C.c_int32), ("nanifoutside", C.c_int32), ("delta", C.c_int32), ("flags", C.c_int32),
                ("storage", C.c_int32), ("tol", C.c_double), ("lambda_", C.c_double), ("nu", C.c_double),
                ("lambda0", C.c_double), ("lambdaN", C.c_double), ("lambda_ratio", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("pair_iterations", C.c_int64 * ICA_MAX_SCALES), ("pair_iterations_total", C.c_int64),
                ("iter_launches", C.c_int64 * ICA_MAX_SCALES), ("iter_kernel_ms", C.c_double * ICA_MAX_SCALES),
                ("iter_kernel_bytes", C.c_double * ICA_MAX_SCALES), ("level_width", C.c_int32 * ICA_MAX_SCALES),
                ("level_height", C.c_int32 * ICA_MAX_SCALES),
                ("first_launches", C.c_int64 * ICA_MAX_SCALES), ("first_kernel_ms", C.c_double * ICA_MAX_SCALES),
                ("first_kernel_bytes", C.c_double * ICA_MAX_SCALES), ("shader_clock_ghz", C.c_double)]


_P = C.c_void_p
_SIGNATURES = {
    "ica_version": (C.c_int, []),
    "ica_device_count": (C.c_int, []),
    "ica_strerror": (C.c_char_p, [C.c_int]),
    "ica_last_error": (C.c_char_p, [_P]),
    "ica_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "ica_destroy": (C.c_int, [_P]),
    "ica_set_stream": (C.c_int, [_P, _P]),
    "ica_synchronize": (C.c_int, [_P]),
    "ica_set_profiling": (C.c_int, [_P, C.c_int]),
    "ica_op_shader_clock_dev": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double)]),
    "ica_nparams": (C.c_int, [C.c_int]),
    "ica_zoom_size": (C.c_int, [C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ica_prepare": (C.c_int, [_P, C.POINTER(Problem)]),
    "ica_solve_batch_dev": (C.c_int, [_P, C.POINTER(Problem), _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(Stats)]),
    "ica_solve_batch": (C.c_int, [_P, C.POINTER(Problem), _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(Stats)]),
    "ica_op_warp_dev": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P]),
    "ica_op_minmax_dev": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "ica_op_normal_equations_dev": (C.c_int, [_P, C.POINTER(Problem), _P, _P, _P, _P, _P, _P]),
    "ica_op_solve_update_dev": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "ica_op_pyramid_down_dev": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _P, _P]),
    "ica_op_zoom_in_parameters_dev": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ica_op_transform_image_dev": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "ica_op_ref_jacobian_dev": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "ica_op_ref_steepest_descent_dev": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "ica_op_ref_normal_dev": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "ica_op_ref_rhop_dev": (C.c_int, [_P, C.c_size_t, _P, C.c_double, C.c_int, _P]),
    "ica_op_ref_robust_error_dev": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_double, C.c_int, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load() -> C.CDLL:
    """dlopen libica_hip.so and declare every entry point of include/ica_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels carry their own libamdhip64.so.7 / libhsa-runtime64.so.1 (same SONAMEs as /opt/rocm).
    # A process must end up with ONE HIP runtime, shared by torch (device memory, streams) and this library,
    # so torch's copy has to be mapped before libica_hip.so asks the loader for that SONAME.
    import torch  # noqa: F401
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built (python -m "
            "inverse_compositional_algorithm_amd.build, or __graft_entry__.build()). There is no CPU fallback.")
    if "ICA_LIB" not in os.environ:
        # the in-tree library must be the one build() wrote from the present sources (experiment builds carry no stamp)
        from . import build
        if os.path.isdir(build.CSRC) and build.is_stale():
            raise RuntimeError(f"{LIB_PATH} does not match its sources or its build stamp: rebuild it "
                               "(python -m inverse_compositional_algorithm_amd.build)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.ica_version() != ICA_ABI_VERSION:
        raise RuntimeError(f"libica_hip.so has ABI version {lib.ica_version()}, this package binds version "
                           f"{ICA_ABI_VERSION}: rebuild it (python -m inverse_compositional_algorithm_amd.build)")
    _lib = lib
    return lib


def check(status: int, ctx=None) -> None:
    if status == 0:
        return
    lib = load()
    msg = lib.ica_strerror(status).decode()
    if ctx is not None:
        extra = lib.ica_last_error(ctx).decode()
        if extra:
            msg += f" ({extra})"
    raise IcaError(status, msg)
