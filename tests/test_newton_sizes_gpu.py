"""Newton steps (reference util.py:152-187 -> core.py:1113-1217 -> linsolver.py:4-87) on grids that are NOT powers of two and
in the size range between the dense factorisation and the large multigrid runs -- found by running the examples at N = 75,
100, 125, 300, 501, 1000, 99999, 100000: extents that stop halving early, odd finest levels, 49152 < unknowns <= 2e5
(`direct` used to take 45000 - 50000 CG iterations on the normal equations there, 24 - 29 s, some without converging), a
time-explicit operator beyond its stability limit (substitution returned finite garbage), and operators Newton cannot
linearise (now a ValueError that says why, instead of an IndexError from the dense scatter)."""

import importlib
import os
import sys

import pytest
import torch
from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_newton(modname, argv, epochs=1, double="1"):
    import odil_amd as odil

    for sub in os.listdir(os.path.join(ROOT, "examples")):
        p = os.path.join(ROOT, "examples", sub)
        if p not in sys.path:
            sys.path.insert(0, p)
    ex = importlib.import_module(modname)
    odil.util.set_log_file(open(os.devnull, "w"))
    args = ex.parse_args(argv + ["--optimizer", "newton", "--multigrid", "0", "--double", double])
    problem, state = ex.make_problem(args)
    args.epoch_start, args.epochs = 0, epochs
    before = float(problem.eval_loss_grad(state)[0])
    seen = []
    odil.util.optimize(args, "newton", problem, state, lambda s, e, p: seen.append(p.get("linsolver") if hasattr(p, "get") else None))
    after = float(problem.eval_loss_grad(state)[0])
    return before, after, [s for s in seen if s]


@pytest.mark.parametrize("argv,shortcut,drop", [
    (["--ndim", "3", "--N", "75"], "1", 1e-14), (["--ndim", "3", "--N", "75"], "0", 1e-14),   # odd finest level: GCR around padded cycles
    (["--ndim", "2", "--N", "501"], "1", 1e-14),
    (["--ndim", "2", "--N", "300"], "0", 1e-14),    # 90000 unknowns, 75^2 two levels down
    (["--ndim", "2", "--N", "256"], "1", 1e-14),    # 65536 unknowns: beyond the dense factorisation, below the old multigrid threshold
    (["--ndim", "2", "--N", "1000"], "1", 1e-14),   # 125^2 three levels down
    # 1-D, |A| = 4e10: u + delta rounds at 1e-16, which the operator turns into residuals of 1e-6 .. 1e-3 -- the loss cannot
    # go lower in float64 (it went from 135 to 135 in 50000 CG iterations before)
    (["--ndim", "1", "--N", "100000"], "0", 1e-10), (["--ndim", "1", "--N", "99999"], "0", 1e-6),
])
def test_poisson_newton_step_on_awkward_sizes(argv, shortcut, drop, monkeypatch):
    monkeypatch.setenv("ODIL_NEWTON_SHORTCUT", shortcut)
    before, after, status = run_newton("poisson", argv)
    assert after <= drop * before, (argv, before, after, status)
    assert status and "niter" in status[-1] and status[-1]["niter"] <= 40, status  # (not the 50000-iteration CG)


@pytest.mark.parametrize("argv", [["--N", "50", "--kind", "jump"], ["--N", "45", "--kind", "jump"], ["--N", "75", "--kind", "smooth"]])
def test_variable_coefficient_newton_step_on_awkward_sizes(argv):
    before, after, status = run_newton("diffusion", argv)
    assert after <= 1e-14 * before, (argv, before, after, status)
    assert status[-1]["niter"] <= 40, status


def test_substitution_is_only_accepted_when_it_meets_the_equations():
    """wave with dt > dx: the explicit recurrence amplifies rounding by its growth factor at every level; the substituted
    'solution' was finite and wrong by 1e146.  The residual is checked and the normal-equation routes take over."""
    before, after, status = run_newton("wave", ["--Nt", "40", "--Nx", "120"])
    assert after <= 1e-10 * before, (before, after, status)
    assert "substitution" not in status[-1].get("method", ""), status
    before, after, status = run_newton("wave", ["--Nt", "120", "--Nx", "40"])  # stable: substitution stays
    assert after <= 1e-10 * before and "substitution" in status[-1]["method"], (before, after, status)


def test_operators_that_shift_arrays_themselves_are_refused_with_a_reason():
    with pytest.raises(ValueError, match="not pointwise in ctx.field"):
        run_newton("infer_constant", ["--Nt", "20", "--Nx", "30"])


def test_lincomb_with_vectors_of_one_element():
    from odil_amd import ops

    dev = torch.device("cuda:0")
    a = torch.arange(1.0, 8.0, dtype=torch.float64, device=dev).reshape(1, 7).t()  # (7, 1) with strides (1, 7)
    y = torch.full((1,), 2.0, dtype=torch.float64, device=dev)
    ops.lincomb(y, 1.0, a.contiguous(), torch.ones(7, dtype=torch.float64, device=dev))
    assert float(y) == 2.0 + 28.0


@pytest.mark.parametrize("modname,argv,epochs,drop", [
    ("poisson", ["--ndim", "2", "--N", "48"], 1, 1e-10),          # dense route: was 3.7e2 -> 2.5e-1 in float32 arithmetic
    ("poisson", ["--ndim", "3", "--N", "64"], 1, 1e-8),           # float32 V-cycles
    ("diffusion", ["--N", "128", "--kind", "smooth"], 2, 1e-10),  # cycles stop at the float32 floor: accepted (was: 1000 CG iterations)
    ("wave", ["--Nt", "64", "--Nx", "32"], 1, 1e-10),             # substitution on the float64 copy (float32: rejected, dense LU)
    ("heat_tmax", ["--linsolver_damp", "1e-4"], 3, 2e-3),
    ("heat", ["--Nt", "64", "--Nx", "64", "--infer_k", "1", "--imposed", "stripe", "--kwreg", "1"], 3, 1e-6),  # reference run case 2n
])
def test_float32_problems_take_their_exact_solves_in_float64(modname, argv, epochs, drop):
    """`--double 0` (the default of the reference's heat example, whose documented Newton run is float32): the exact routes
    of `linsolver.solve` -- substitution, block cyclic reduction, Schur complement, dense factorisation of M^T M -- work on
    a float64 copy of the operator, the iterate is rounded back; multigrid cycles that stop at the float32 rounding floor
    well below the right-hand side are accepted instead of handed to CG on the normal equations."""
    import numpy as np

    import odil_amd as odil

    np.random.seed(1)
    odil.runtime.get_mod().random.set_seed(1)
    before, after, status = run_newton(modname, argv, epochs=epochs, double="0")
    assert after <= drop * before, (modname, argv, before, after, status)
    assert all(s.get("niter", 0) <= 40 for s in status), status
