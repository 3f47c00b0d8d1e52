"""odil_amd: MI355X-native (gfx950) implementation of the ODIL hot path.

Mirrors the public names of the reference package `odil`
(reference src/odil/__init__.py:3-61) for the path SURVEY.md section 8 scopes:
stencil residual + adjoint, multigrid-decomposition transfers, Adam / L-BFGS
updates and the Newton sparse-Jacobian assemble / normal-equations solve.
All arithmetic the framework owns runs in hand-written HIP kernels behind the C-ABI of
include/odil_hip.h; there is no CPU fallback.

    import odil_amd as odil        # user operators written for `odil` run unchanged
"""

from . import _lib, backend, core, history, io, linsolver, optimizer, util  # noqa: F401
from .backend import ModBase, ModNumpy, ModRocm  # noqa: F401
from .core import (  # noqa: F401
    Array,
    Context,
    Domain,
    Field,
    MultigridField,
    NeuralNet,
    Problem,
    State,
    interp_to_finer,
    restrict_to_coarser,
)
from .history import History  # noqa: F401
from .io import parse_raw_xmf, read_raw, read_raw_with_xmf, write_raw_slab, write_raw_with_xmf, write_raw_xmf  # noqa: F401
from .optimizer import EarlyStopError  # noqa: F401
from .util import make_callback, optimize, printlog, set_log_file, setup_outdir  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    # `odil.runtime` is loaded lazily like in the reference (src/odil/__init__.py:46-61):
    # importing it selects the device.
    if name == "runtime":
        import importlib

        return importlib.import_module(".runtime", __name__)
    raise AttributeError(name)
