"""ctypes binding of the C-ABI library `libodil_hip.so` (include/odil_hip.h).

The library is the product's only compute path.  There is no CPU fallback: if the
shared object is missing, or a call is made with tensors that are not on a HIP
device, the call raises.
"""

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ODIL_HIP_LIB") or os.path.join(_HERE, "libodil_hip.so")

_P = c_void_p
_I64P = ctypes.POINTER(c_int64)

# name -> argument types with R standing for the real type (c_double / c_float).
_R = "real"
_SIGNATURES = {
    "interp_add": [_P, _P, _P, _I64P, c_int, c_char_p, _R, _R, _P],
    "interp_adj": [_P, _P, _P, _I64P, c_int, c_char_p, _R, _P],
    "interp_add_ld": [_P, c_int64, _P, _P, _I64P, c_int, c_char_p, _R, _R, _P],
    "interp_adj_ld": [_P, _P, c_int64, _I64P, c_int, c_char_p, _P],
    "interp_adj_cut": [_P, _P, _P, _I64P, c_int, c_char_p, _R, c_int, c_int, _P],
    "interp_adj_cut_adam": [_P, _P, _I64P, c_int, c_char_p, c_int, c_int, _P, _P, _P, _R, _R, _R, _R, _P, _P],
    "restrict": [_P, _P, _I64P, c_int, c_char_p, _P],
    "restrict_adj": [_P, _P, _I64P, c_int, c_char_p, _P],
    "conv_valid": [_P, _P, _P, _I64P, _I64P, _I64P, _I64P, c_int, c_int, _P],
    "mg_synth": [_P, _P, _P, _P, _I64P, c_int, c_int, c_char_p, _P],
    "mg_synth_adj": [_P, _P, _P, _P, _I64P, c_int, c_int, c_char_p, _P],
    "mg_synth_adj_adam": [_P, _P, _P, _P, _I64P, c_int, c_int, c_char_p, _P, _P, _P, _R, _R, _R, _R, _P, _P],
    "field_gather": [_P, _P, _I64P, c_int, c_char_p, c_char_p, _I64P, _P],
    "field_scatter": [_P, _P, _I64P, c_int, c_char_p, c_char_p, _I64P, c_int, _P],
    "mean_reduce": [_P, c_int64, c_int, _P, _P, _P],
    "poisson_residual": [_P, _P, _P, _I64P, c_int, _P, _P, _P, _P],
    "poisson_small_epochs": [_P, _P, _P, _P, _P, _P, _P, _I64P, c_int, c_int, _P, _P, c_int, _R, _R, _R, _P, _P, _P, _P],
    "poisson_jacobi": [_P, _P, _P, _I64P, c_int, _P, _R, _P],
    "poisson_jacobi2": [_P, _P, _P, _I64P, c_int, _P, _R, _R, c_int, _P],
    "poisson_jacobi2_synth": [_P, _P, _P, _P, _I64P, _P, _R, _R, c_int, _P],
    "poisson_residual_restrict": [_P, _P, _P, _I64P, c_int, _P, _R, _P, _P, _P],
    "poisson_residual_restrict_slab": [_P, _P, _P, _I64P, c_int, _P, _R, c_int64, c_int64, c_double, _P, _P, _P],
    "poisson_residual_synth": [_P, _P, _P, _P, _I64P, _P, c_int64, c_int64, c_double, _P, _P, _P],
    "poisson_jacobi_synth": [_P, _P, _P, _P, _I64P, _P, _R, _P],
    "poisson_residual_slab": [_P, _P, _P, _I64P, c_int, _P, c_int64, c_int64, c_double, _P, _P, _P],
    "poisson_adjoint": [_P, _P, _I64P, c_int, _P, _R, _P],
    "poisson_adjoint_adam": [_P, _P, _P, _P, _P, _I64P, c_int, _P, _R, _R, _R, _R, _R, _P, _P],
    "poisson_adjoint_transpose_adam": [_P, _P, _P, _I64P, _P, _R, _P, _P, _P, _P, _P, _P, _R, _R, _R, _R, _P, c_int,
                                       c_int, _P],
    "poisson_jac_coeffs": [_P, _I64P, c_int, _P, _P],
    "poisson_jac_match": [_P, _I64P, c_int, _P, _P, _P, _P],
    "adam_step": [_P, _P, _P, _P, c_int64, _R, _R, _R, _R, _P, _P],
    "adam_step_pieces": [_P, _P, _P, _P, c_int64, c_int64, c_int64, c_int64, _R, _R, _R, _R, _P, _P],
    "planes_copy": [_P, _P, _P, c_int, c_int64, c_int, c_int, _P],
    "axpy": [_P, _P, c_int64, _R, _P],
    "scale": [_P, _P, c_int64, _R, _P, _P],
    "addcmul": [_P, _P, _P, c_int64, c_int, _P],
    "dots": [_P, c_int64, c_int, _P, c_int64, _P, _P, _P],
    "dots3": [_P, c_int64, c_int, _P, _P, _P, c_int64, _P, _P, _P],
    "lbfgs_probe": [_P, _P, c_int64, _P, _P, _P],
    "lincomb": [_P, _R, _P, c_int64, c_int, _P, c_int64, _P],
    "stencil_apply": [_P, _I64P, c_int, _P, _P, _I64P, c_int, c_int, _P],
    "stencil_march": [_P, _I64P, c_int, c_int, _P, _P, _I64P, c_int, c_int, c_int, _P],
    "stencil_var_smooth": [_P, _P, _P, _P, _I64P, c_int, _R, c_int, _P],
    "stencil_var_smooth2": [_P, _P, _P, _P, _I64P, c_int, _R, _R, c_int, _P],
    "stencil_var_residual_restrict": [_P, _P, _P, _P, _I64P, c_int, _R, _P, _P, _P],
    "stencil_var_residual_restrict_slab": [_P, _P, _P, _P, _I64P, c_int, _R, c_int64, c_int64, c_double, _P, _P, _P],
    "stencil_var_coarsen": [_P, _P, _I64P, c_int, _P],
    "stencil_var_coarsen_axes": [_P, _P, _I64P, c_int, _P, _P],
    "stencil_vcycle_tail": [_P, _I64P, _P, c_int, c_int, _P, _P, _P, _P, c_int64, _P, c_int, _P, c_int, _P, c_int, c_int, _P],
    "max_abs_diff": [_P, _P, c_int64, _P, _P, _P],
    "max_abs_rows": [_P, c_int, c_int64, _P, _P, _P],
    "csr_assemble": [_P, _I64P, c_int, _I64P, c_int, c_int64, _P, _P, _P, _P],
    "dense_block_xty": [_P, _P, c_int64, c_int, c_int, c_int64, c_int64, _P, _P, _P],
    "dense_block_gram": [_P, c_int64, c_int, c_int64, _P, _P, _P],
}

EXPORTED = [
    "odil_last_error", "odil_version", "odil_device_count", "odil_reduce_workspace_bytes", "odil_dots_workspace_bytes",
    "odil_dense_block_workspace_bytes", "odil_narrow_scale", "odil_widen_axpy", "odil_poisson_small_epochs_resident",
] + [
    "odil_{}_{}".format(name, suffix) for name in _SIGNATURES for suffix in ("f64", "f32")
]

_lib = None


class OdilHipError(RuntimeError):
    pass


def load():
    """Loads the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OdilHipError(
            "HIP extension not built: {} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C odil_amd/csrc`. There is no CPU fallback.".format(LIB_PATH)
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.odil_last_error.restype = c_char_p
    lib.odil_last_error.argtypes = []
    lib.odil_version.restype = c_int
    lib.odil_device_count.restype = c_int
    lib.odil_reduce_workspace_bytes.restype = c_size_t
    lib.odil_dots_workspace_bytes.restype = c_size_t
    lib.odil_dots_workspace_bytes.argtypes = [c_int]
    lib.odil_dense_block_workspace_bytes.restype = c_size_t
    lib.odil_poisson_small_epochs_resident.restype = c_int
    lib.odil_poisson_small_epochs_resident.argtypes = [_I64P, c_int, c_int, c_int]
    for name in ("odil_narrow_scale", "odil_widen_axpy"):  # (mixed precision: no type suffix)
        fn = getattr(lib, name)
        fn.restype = c_int
        fn.argtypes = [_P, _P, c_int64, c_double, _P, _P]
    for name, sig in _SIGNATURES.items():
        for suffix, real in (("f64", c_double), ("f32", c_float)):
            fn = getattr(lib, "odil_{}_{}".format(name, suffix))
            fn.restype = c_int
            fn.argtypes = [real if a is _R else a for a in sig]
    _lib = lib
    return lib


def suffix_of(dtype):
    if dtype == torch.float64:
        return "f64"
    if dtype == torch.float32:
        return "f32"
    raise TypeError("unsupported dtype {} (float32 / float64 only)".format(dtype))


def call(name, dtype, *args, unserved_ok=False):
    """Calls odil_<name>_<f32|f64>(*args); raises OdilHipError on a non-zero status.  unserved_ok: the entry point may
    return 1 = "layout not served, nothing launched" (odil_interp_add_ld / _adj_ld); returns whether it served."""
    fn = _entry.get((name, dtype))
    if fn is None:
        fn = _entry[(name, dtype)] = getattr(load(), "odil_{}_{}".format(name, suffix_of(dtype)))
    status = fn(*args)
    if status == 1 and unserved_ok:
        return False
    if status != 0:
        raise OdilHipError("odil_{}: {} (status {})".format(name, load().odil_last_error().decode(), status))
    return True


_entry = dict()  # (name, dtype) -> the bound entry point (a launch is a few microseconds of host time: no lookups per call)


def ptr(t):
    """Device pointer of a tensor (or None)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError("expected a torch.Tensor, got {}".format(type(t).__name__))
    if not t.is_cuda:
        raise OdilHipError("tensor is on '{}': the HIP kernels need device memory (no CPU fallback)".format(t.device))
    if not t.is_contiguous():
        raise OdilHipError("tensor must be contiguous (C order)")
    return c_void_p(t.data_ptr())


def i64(values):
    values = [int(v) for v in values]
    return (c_int64 * len(values))(*values)


def ptr_array(tensors):
    """HOST array of device pointers."""
    arr = (c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
        if t is not None and (not t.is_cuda or not t.is_contiguous()):
            raise OdilHipError("level arrays must be contiguous device tensors")
    return arr


def host_reals(values, dtype):
    np_dtype = np.float64 if dtype == torch.float64 else np.float32
    arr = np.ascontiguousarray(np.asarray(values, dtype=np_dtype))
    return arr, arr.ctypes.data_as(c_void_p)


def stream_ptr():
    """The current HIP stream of the current device as a pointer argument.  (torch.cuda.current_stream() builds a Stream
    object through several Python layers, ~8 us per call -- more than the launch it precedes costs; the raw handle is one
    C call.)"""
    return c_void_p(_raw_stream(_current_device()))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_current_device = getattr(torch._C, "_cuda_getDevice", None)
if _raw_stream is None or _current_device is None:  # (another torch build: the documented route)
    def stream_ptr():  # noqa: F811
        return c_void_p(torch.cuda.current_stream().cuda_stream)


def reduce_workspace_elems():
    return load().odil_reduce_workspace_bytes() // 8
