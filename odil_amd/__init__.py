"""odil_amd: MI355X-native (gfx950) implementation of the ODIL hot path.

Mirrors the public names of the reference package `odil`
(reference src/odil/__init__.py:3-61) for the path SURVEY.md section 8 scopes:
stencil residual + adjoint, multigrid-decomposition transfers, Adam / L-BFGS
updates and the Newton sparse-Jacobian assemble / normal-equations solve.
All arithmetic runs in hand-written HIP kernels behind the C-ABI of
include/odil_hip.h; there is no CPU fallback.
"""

from . import _lib  # noqa: F401

__version__ = "0.1.0"
