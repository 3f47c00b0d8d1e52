"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/odil_hip.h declares (no compute without a GPU), argument validation
works on the host, and the product refuses CPU tensors instead of falling back."""

import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "odil_hip.h")).read()
    return sorted(set(re.findall(r"\b(odil_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from odil_amd import _lib

    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_lib.EXPORTED) == names
    assert lib.odil_version() >= 100
    assert lib.odil_reduce_workspace_bytes() >= 4096 * 8


def test_host_side_validation_reports_errors():
    from ctypes import c_int, c_void_p

    from odil_amd import _lib

    lib = _lib.load()
    # invalid loc string -> ODIL_E_INVAL before anything touches the device
    status = lib.odil_interp_add_f64(
        c_void_p(16), None, c_void_p(16), _lib.i64([4, 4]), c_int(2), b"cx", 1.0, 1.0, None
    )
    assert status == -1
    assert b"loc" in lib.odil_last_error()
    status = lib.odil_poisson_adjoint_f64(c_void_p(16), c_void_p(16), _lib.i64([4] * 5), c_int(5), None, 1.0, None)
    assert status == -1


def test_no_cpu_fallback():
    from odil_amd import _lib, ops

    with pytest.raises(_lib.OdilHipError):
        ops.interp_add(torch.zeros(4, 4, dtype=torch.float64), "cc")
    with pytest.raises(_lib.OdilHipError):
        ops.adam_step(*[torch.zeros(8) for _ in range(4)], 0.1, 0.1, 0.001, 1e-7)
