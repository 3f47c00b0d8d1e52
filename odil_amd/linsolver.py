"""Normal-equations solve of the Newton step (reference src/odil/linsolver.py:4-87), on
the device and matrix-free.

The reference forms A = M^T M (+ damp^2 I + dampdiag^2 diag(A)) and b = M^T rhs with
scipy.sparse and factorises A with SuperLU (`direct`, linsolver.py:17-26) or hands it to an
iterative routine.  Here M stays a `core.LinearizedOperator` (per-shift coefficient
arrays + dense blocks), A is applied as M^T (M x) with the HIP stencil kernels and the
system is solved by Jacobi-preconditioned conjugate gradients with deterministic dot
products (odil_dots).  `direct` therefore means "CG to round-off" (tol 1e-14 relative,
bounded by `--linsolver_maxiter` if given, else 20 n), which reproduces the reference's
Newton iterate to solver tolerance; `cg` / `bicgstab` / `multigrid` use `--linsolver_tol`.
cupy / sparseqr / pyamg variants of the reference are optional third-party paths and are
not provided.
"""

import numpy as np
import torch

from . import ops


def _dot(a, b):
    return ops.dots(a[None], b)[0]


def cg_normal(op, rhs, damp=0.0, dampdiag=0.0, tol=1e-14, maxiter=None, status=None, x0=None):
    """Solves (M^T M + damp^2 I + dampdiag^2 diag(M^T M)) x = M^T rhs by preconditioned CG."""
    n = op.shape[1]
    dtype, device = op.dtype, op.device
    b = op.rmatvec(rhs)
    diag = op.normal_diagonal()
    shift = None
    if damp or dampdiag:
        shift = torch.full_like(diag, float(damp) ** 2)
        if dampdiag:
            ops.axpy(shift, diag, float(dampdiag) ** 2)
        ops.axpy(diag, shift, 1.0)

    def apply_a(v):
        av = op.rmatvec(op.matvec(v))
        if shift is not None:
            ops.addcmul(av, shift, v)
        return av

    # Jacobi preconditioner (unit where a column is empty)
    minv = torch.where(diag > 0, 1.0 / diag, torch.ones_like(diag))
    x = torch.zeros(n, dtype=dtype, device=device) if x0 is None else x0.clone()
    r = b.clone()
    if x0 is not None:
        ops.axpy(r, apply_a(x), -1.0)
    z = torch.empty_like(r)
    ops.addcmul(z, minv, r, accumulate=False)
    p = z.clone()
    rz = float(_dot(r, z))
    bnorm = float(_dot(b, b)) ** 0.5
    maxiter = maxiter or 20 * n
    niter = 0
    res = float(_dot(r, r)) ** 0.5
    while niter < maxiter and res > tol * max(bnorm, 1e-300):
        ap = apply_a(p)
        pap = float(_dot(p, ap))
        if pap <= 0:
            break
        alpha = rz / pap
        ops.axpy(x, p, alpha)
        ops.axpy(r, ap, -alpha)
        ops.addcmul(z, minv, r, accumulate=False)
        rz_new = float(_dot(r, z))
        beta = rz_new / rz
        rz = rz_new
        ops.scale(p, beta, out=p)
        ops.axpy(p, z, 1.0)
        niter += 1
        res = float(_dot(r, r)) ** 0.5
    if status is not None:
        status["residual"] = res
        status["niter"] = niter
    return x


def solve(matr, rhs, args, status=None, linsolver="direct"):
    """Reference signature (linsolver.py:4).  `matr` is a `core.LinearizedOperator`; returns the
    solution as a device vector."""
    from .core import LinearizedOperator

    if status is None:
        status = dict()
    if not isinstance(matr, LinearizedOperator):
        raise TypeError(
            "odil_amd.linsolver.solve expects the device operator returned by Problem.linearize_device(); "
            "got {} (host sparse matrices are not solved here: there is no CPU path)".format(type(matr).__name__)
        )
    maxiter = getattr(args, "linsolver_maxiter", None)
    damp = getattr(args, "linsolver_damp", 0) or 0
    dampdiag = getattr(args, "linsolver_dampdiag", 0) or 0
    tol = getattr(args, "linsolver_tol", 1e-10)
    if not torch.is_tensor(rhs):
        rhs = torch.as_tensor(np.asarray(rhs), dtype=matr.dtype, device=matr.device)
    if linsolver not in ("direct", "directsq", "cg", "bicgstab", "multigrid", "lsqr"):
        raise ValueError("Unknown linsolver=" + linsolver)
    # Square Poisson stencil without damping: M d = rhs has the solution of the normal equations
    # and is solved by geometric multigrid V-cycles (gmg.py) -- the only option that scales to
    # 512^3.  `multigrid` always takes it when it applies, `direct` above 2e5 unknowns.
    if not damp and not dampdiag and (linsolver == "multigrid" or (linsolver == "direct" and matr.ncols > 200000)):
        from . import gmg

        rec = gmg.recognise_poisson(matr)
        if rec is not None:
            shape, h2 = rec
            solver = gmg.PoissonGMG(shape, h2, matr.dtype, matr.device)
            gtol = 1e-12 if linsolver == "direct" else tol
            x = solver.solve(rhs.reshape(shape).contiguous(), tol=gtol, maxiter=maxiter or 60, status=status)
            return x.reshape(-1)
    if linsolver in ("direct", "directsq"):
        return cg_normal(matr, rhs, damp, dampdiag, tol=1e-14, maxiter=maxiter, status=status)
    else:
        return cg_normal(matr, rhs, damp, dampdiag, tol=tol, maxiter=maxiter or 1000, status=status)


def add_arguments(parser):
    """Same flag names as the reference (linsolver.py:90-131)."""
    parser.add_argument("--linsolver", type=str, default="direct",
                        choices=("direct", "directsq", "cg", "bicgstab", "multigrid", "lsqr"), help="Linear solver")
    parser.add_argument("--linsolver_tol", type=float, default=1e-10, help="Convergence tolerance of iterative solvers")
    parser.add_argument("--linsolver_maxiter", type=int, default=None, help="Maximum number of iterations")
    parser.add_argument("--linsolver_damp", type=float, default=0, help="Damping: adds damp^2 * I to the normal matrix")
    parser.add_argument("--linsolver_dampdiag", type=float, default=0, help="Adds dampdiag^2 * diag to the normal matrix")
    parser.add_argument("--linsolver_verbose", type=int, default=0, help="Print the status of the linear solver")
    parser.add_argument("--linsolver_history", type=int, default=0, help="Write the solver status to the history")
    parser.add_argument("--lr", type=float, default=1e-3, help="Learning rate")
    parser.add_argument("--nlvl", type=int, default=None, help="Number of multigrid levels")
    # accepted for command-line compatibility with the reference's host multigrid solver options
    parser.add_argument("--smooth_pre", type=int, default=None)
    parser.add_argument("--smooth_post", type=int, default=None)
    parser.add_argument("--omega", type=float, default=None)
    parser.add_argument("--ndirect", type=int, default=None)
    parser.add_argument("--restriction", type=str, default=None)
