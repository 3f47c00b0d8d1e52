"""Runtime selection: the environment contract of the reference
(reference src/odil/runtime.py:1-82) for the ROCm backend.

  ODIL_BACKEND  ''/'rocm' (anything else is rejected: there is one backend here)
  ODIL_DTYPE    float32 (default) | float64      -> `dtype`
  ODIL_JIT      kept for compatibility; kernels are precompiled, graphs are captured
                by the optimizers where it pays                     -> `enable_jit`
  ODIL_MT       host threading knob of the reference; no effect on the device path
  ODIL_FUSE     1 (default): recognise affine stencil operators and route them to the
                fused HIP kernels (core.Problem); 0: always use the generic path

TRACE    1 (default): trace every other operator once and run it as one generated HIP
                kernel (stencil_jit.py); 0: generic autograd path

`tf` and `jax` are None (reference examples test them to pick code paths).
"""

import os
import sys

import numpy

from .backend import ModRocm

backend_name = os.environ.get("ODIL_BACKEND", "") or "rocm"
if backend_name not in ("rocm", "hip"):
    sys.stderr.write(f"Unknown ODIL_BACKEND='{backend_name}', options are: rocm\n")
    raise RuntimeError("unsupported ODIL_BACKEND")

enable_jit = bool(int(os.environ.get("ODIL_JIT", 0)))
enable_fuse = bool(int(os.environ.get("ODIL_FUSE", 1)))
enable_trace = bool(int(os.environ.get("ODIL_TRACE", 1)))
tf = None
jax = None

dtype_name = os.environ.get("ODIL_DTYPE", "float32")
if dtype_name not in ("float32", "float64"):
    raise RuntimeError(f"Expected ODIL_DTYPE=float32 or float64, got '{dtype_name}'")
dtype = numpy.dtype(dtype_name)

_mod = None


def get_mod():
    """The process-wide `mod` on this rank's HIP device (LOCAL_RANK aware)."""
    global _mod
    if _mod is None:
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("odil_amd.runtime: no HIP device visible; there is no CPU fallback")
        local_rank = int(os.environ.get("LOCAL_RANK", 0))
        if local_rank < torch.cuda.device_count():
            torch.cuda.set_device(local_rank)
        _mod = ModRocm()
    return _mod


def __getattr__(name):
    if name == "mod":
        return get_mod()
    raise AttributeError(name)
