"""Normal-equations solve of the Newton step (reference src/odil/linsolver.py:4-87), on
the device and matrix-free.

The reference forms A = M^T M (+ damp^2 I + dampdiag^2 diag(A)) and b = M^T rhs with
scipy.sparse and factorises A with SuperLU (`direct`, linsolver.py:17-26) or hands it to an
iterative routine.  Here M stays a `core.LinearizedOperator` (per-shift coefficient
arrays + dense blocks), A is applied as M^T (M x) with the HIP stencil kernels and the
system is solved by Jacobi-preconditioned conjugate gradients with deterministic dot
products (odil_dots) and no host synchronisation inside the iteration.  `direct` is a dense
Cholesky of A = M^T M up to 49152 unknowns, memory permitting (one f64 GEMM + rocSOLVER, `dense_normal`), geometric
multigrid for the recognised stencils beyond the dense factorisation's reach (gmg.py), and otherwise "CG to
round-off" (tol 1e-14 relative, bounded by `--linsolver_maxiter` if given, else 20 n), which
reproduces the reference's Newton iterate to solver tolerance; `cg` / `bicgstab` / `multigrid`
use `--linsolver_tol`.
cupy / sparseqr / pyamg variants of the reference are optional third-party paths and are
not provided.
"""

import numpy as np
import torch

from . import ops


def _dot(a, b):
    return ops.dots(a[None], b)[0]


def cg_normal(op, rhs, damp=0.0, dampdiag=0.0, tol=1e-14, maxiter=None, status=None, x0=None, check_every=25, b=None):
    """Solves (M^T M + damp^2 I + dampdiag^2 diag(M^T M)) x = M^T rhs (or = b when `b` is given) by
    Jacobi-preconditioned CG.

    The iteration runs without host synchronisation: the scalars <r, z>, <p, A p>, alpha and beta
    stay 0-d device tensors (deterministic odil_dots; updates through odil_lincomb with device
    coefficients) and the residual norm is read back only every `check_every` iterations."""
    n = op.shape[1]
    dtype, device = op.dtype, op.device
    b = op.rmatvec(rhs) if b is None else b
    diag = op.normal_diagonal()
    shift = None
    if damp or dampdiag:
        # reference linsolver.py:19-23: damp^2 I is added FIRST, dampdiag^2 times the diagonal of the already damped
        # matrix second: A_ii -> (A_ii + damp^2) (1 + dampdiag^2)
        shift = torch.full_like(diag, float(damp) ** 2 * (1.0 + float(dampdiag) ** 2))
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
    # rows (p, z, p'): the next direction p' = beta p + z is formed from two adjacent rows into the
    # third, and the roles of rows 0 and 2 alternate (no aliasing inside odil_lincomb)
    buf = torch.empty((3, n), dtype=dtype, device=device)
    p, z, flip = buf[0], buf[1], False
    ops.addcmul(z, minv, r, accumulate=False)
    p.copy_(z)
    rz = _dot(r, z)
    bnorm = float(_dot(b, b)) ** 0.5
    # (no --linsolver_maxiter: 20 n as SciPy's CG would, but bounded -- a singular system never reaches the tolerance,
    # and 20 n iterations at a few hundred thousand unknowns are hours)
    maxiter = maxiter or min(20 * n, 50000)
    niter = 0
    res = float(_dot(r, r)) ** 0.5
    coef = torch.ones(3, dtype=dtype, device=device)  # (beta, 1) on rows 0:2, or (1, beta) on rows 1:3
    ok = res > tol * max(bnorm, 1e-300)
    while ok and niter < maxiter:
        for _ in range(min(check_every, maxiter - niter)):
            ap = apply_a(p)
            pap = _dot(p, ap)
            alpha = torch.where(pap > 0, rz / pap, torch.zeros_like(rz)).reshape(1)  # converged exactly: stay put
            ops.lincomb(x, 1.0, p[None], alpha)
            ops.lincomb(r, 1.0, ap[None], -alpha)
            ops.addcmul(z, minv, r, accumulate=False)
            rz_new = _dot(r, z)
            beta = torch.where(rz > 0, rz_new / rz, torch.zeros_like(rz))
            rz = rz_new
            if not flip:
                coef[0:1], coef[1:2] = beta, 1.0
                ops.lincomb(buf[2], 0.0, buf[0:2], coef[0:2])
                p = buf[2]
            else:
                coef[1:2], coef[2:3] = 1.0, beta
                ops.lincomb(buf[0], 0.0, buf[1:3], coef[1:3])
                p = buf[0]
            flip = not flip
            niter += 1
        res = float(_dot(r, r)) ** 0.5
        ok = res > tol * max(bnorm, 1e-300) and float(pap) > 0 and res == res
    if status is not None:
        status["residual"] = res
        status["niter"] = niter
    if not bool(torch.isfinite(x).all()):
        raise FloatingPointError("cg_normal: the iterate is not finite (residual {})".format(res))
    return x


def _solve_small_spd(a, b, info=None, rcond=None, floor=0.0):
    """x with a x = b for the p x p (p <= 63) Schur complement of the dense columns.  Network weights often leave it
    SINGULAR (redundant directions: the reference's SuperLU then returns some member of the solution set): when the
    plain solve is not finite or misses the equations, the minimum-norm solution through the eigen-decomposition."""
    if not (bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())):
        return None  # (an inner solve broke down: the caller takes another route)
    if rcond is not None:
        # coefficients that were rounded to float32: a direction the exact complement annihilates survives at 1e-7 of its
        # scale, the plain solve follows it out to 1e7 (heat with the network: loss 45 -> 2e9 in one step) -- cut at the
        # data's precision instead of the arithmetic's
        # (`floor`: the complement is a DIFFERENCE G - C^T Z of nearly equal matrices -- what is left below the rounding of
        # G itself is noise whatever its size relative to the complement's own largest eigenvalue)
        sym = 0.5 * (a + a.t())
        try:
            w, v = torch.linalg.eigh(sym)
        except RuntimeError:
            try:
                # (singular values are |eigenvalues|; a negative eigenvalue shows as u = -v: the sign comes back from
                # diag(u^T v), so the direction is applied with the eigenvalue's own sign)
                u_, sv, vt = torch.linalg.svd(sym)
                v = vt.t()
                w = sv * torch.sign((u_ * v).sum(dim=0))
            except RuntimeError:
                return None
        cut = max(rcond * float(w.abs().max()), float(floor))
        keep = w.abs() > cut
        inv = torch.where(keep, 1.0 / torch.where(keep, w, torch.ones_like(w)), torch.zeros_like(w))
        x = v @ (inv * (v.t() @ b))
        if info is not None:
            info["schur_complement"] = "minimum-norm solution, {} of {} directions kept (cut at {:.1e})".format(int(keep.sum()), w.numel(), cut)
        return x
    try:
        x = torch.linalg.solve(a, b)
        ok = bool(torch.isfinite(x).all()) and float((a @ x - b).norm()) <= 1e-8 * max(float(b.norm()), 1e-300)
    except RuntimeError:
        x, ok = None, False
    if not ok:
        try:
            x = torch.linalg.pinv(a, hermitian=True) @ b
        except RuntimeError:
            # the device eigensolver gives up on some ill-conditioned complements (error 42 of syevd): the SVD route of the
            # same library; when that fails too the caller takes another solver
            try:
                x = torch.linalg.pinv(0.5 * (a + a.t())) @ b
            except RuntimeError:
                return None
        if info is not None:
            info["schur_complement"] = "singular: minimum-norm solution"
    return x


def schur_normal(op, rhs, damp=0.0, dampdiag=0.0, maxiter=None, status=None, inner=None):
    """`direct` for systems with DENSE columns (`Array` / `NeuralNet` unknowns, reference core.py:1189-1203).

    With M = [S | D] (S: the stencil blocks, matrix-free; D: rows x p dense, p <= 63) the normal equations
    (reference linsolver.py:17-23) are solved through the Schur complement of the stencil part:
        G = D^T D,  g = D^T r                      one pass of the MFMA kernel over [D | r]  (ops.dense_xty)
        C = S^T D,  c = S^T r                      p + 1 transposed stencil applications
        (S^T S) [Z | z] = [C | c]                  p + 1 matrix-free CG solves (cg_normal on the stencil part)
        (G - C^T Z) y = g - C^T z                  C^T [Z | z] again on the matrix cores; a p x p solve
        x = z - Z y
    Neither M nor S is ever densified.  Returns the full solution vector, or None when the system has no dense
    columns / too many of them.  inner: callable [k, n_s] -> [k, n_s] that applies (S^T S + damping)^{-1} to all
    right-hand sides at once (the block-tridiagonal direct solver, blocktri.py) instead of the CG solves."""
    dense_keys = []
    for row0, nrows, kind, key, payload in op.blocks:
        if kind == "dense" and key not in dense_keys:
            dense_keys.append(key)
    dense_keys.sort(key=lambda k: op.key_to_offset[k])
    p = sum(op.key_to_size[k] for k in dense_keys)
    if not dense_keys or p > 63:
        return None
    dtype, device = op.dtype, op.device
    col0, pos = dict(), 0
    for k in dense_keys:
        col0[k] = pos
        pos += op.key_to_size[k]
    # [D | r]: row-major, rows x (p + 1)
    daug = torch.zeros((op.nrows, p + 1), dtype=dtype, device=device)
    stencil_blocks = []
    for blk in op.blocks:
        row0, nrows, kind, key, payload = blk
        if kind == "dense":
            daug[row0:row0 + nrows, col0[key]:col0[key] + payload.shape[1]] += payload
        else:
            stencil_blocks.append(blk)
    daug[:, p] = rhs
    gg = ops.dense_xty(daug[:, :p], daug)  # p x (p + 1) = [D^T D | D^T r]
    G, g = gg[:, :p].clone(), gg[:, p].clone()
    if damp or dampdiag:
        G.diagonal().add_(float(damp) ** 2 + float(dampdiag) ** 2 * (G.diagonal().clone() + float(damp) ** 2))  # (as cg_normal)
    x = torch.zeros(op.ncols, dtype=dtype, device=device)
    dcols = torch.cat([torch.arange(op.key_to_offset[k], op.key_to_offset[k] + op.key_to_size[k], device=device)
                       for k in dense_keys])
    info = dict(method="schur-mfma", dense_columns=p)
    data_rcond = 1e-5 if getattr(op, "source_dtype", None) == torch.float32 else None
    if not stencil_blocks:
        y = _solve_small_spd(G, g, info, rcond=data_rcond)
        if y is None:
            return None
        niter = 0
    else:
        import copy

        op_s = copy.copy(op)
        op_s.blocks = stencil_blocks
        dt = daug.t().contiguous()  # (p + 1) x rows: the columns of [D | r] as contiguous vectors
        cz = torch.stack([op_s.rmatvec(dt[j]) for j in range(p + 1)])  # rows: C_j = S^T D_j, last: S^T r
        zs, niter, worst = [], 0, 0.0
        if inner is not None:
            # the stencil unknowns may be a sub-range of the vector: the inner solver sees its own field only
            zs = torch.zeros_like(cz)
            off, size = inner.offset, inner.size
            zs[:, off:off + size] = inner.solve(cz[:, off:off + size].contiguous())
            niter = 1
            info["method"] = "schur-mfma + block-tridiagonal direct"
        else:
            for j in range(p + 1):
                sub = dict()
                # (p + 1 solves: without --linsolver_maxiter each is bounded well below the single-solve default)
                zs.append(cg_normal(op_s, None, damp, dampdiag, tol=1e-14, maxiter=maxiter or min(20 * op_s.ncols, 5000),
                                    status=sub, b=cz[j]))
                niter = max(niter, sub.get("niter", 0))
                worst = max(worst, sub.get("residual", 0.0))
            zs = torch.stack(zs)
        info.update(inner_solves=p + 1, inner_residual_max=worst)  # (an inner solve cut short by maxiter shows here)
        ct = cz[:p].t().contiguous()  # unknowns x p
        zt = zs.t().contiguous()      # unknowns x (p + 1)
        czz = ops.dense_xty(ct, zt)   # C^T [Z | z]
        y = _solve_small_spd(G - czz[:, :p], g - czz[:, p], info, rcond=data_rcond,
                             floor=0.0 if data_rcond is None else 1e-6 * float(G.diagonal().abs().max()))
        if y is None:
            return None
        x.copy_(zs[p])
        ops.lincomb(x, 1.0, zs[:p].contiguous(), (-y).contiguous())
    x[dcols] = y
    if status is not None:
        r = op.rmatvec(op.matvec(x) - rhs)
        status.update(info)
        status["residual"] = float(_dot(r, r)) ** 0.5
        status["niter"] = niter
    return x


def blocktri_normal(op, rhs, damp=0.0, dampdiag=0.0, status=None):
    """`direct` for operators whose stencil part acts on ONE field and couples neighbouring levels of one axis only
    (implicit time stepping: reference examples/heat/heat.py:36-137): the normal equations (reference
    linsolver.py:17-26) by block cyclic reduction (blocktri.py), dense `NeuralNet` / `Array` columns through the
    Schur complement.  None when the structure is not there (the caller goes on to the general routes)."""
    from . import blocktri
    from .core import Field

    keys = {key for _, _, kind, key, _ in op.blocks if kind == "stencil"}
    if len(keys) != 1:
        return None
    (key,) = keys
    if not isinstance(op.key_to_field[key], Field):
        return None
    dense_keys = {k for _, _, kind, k, _ in op.blocks if kind == "dense"}
    if set(op.key_to_field) - {key} - dense_keys and any(op.key_to_size[k] for k in set(op.key_to_field) - {key} - dense_keys):
        return None  # unknowns no block refers to: the general routes regularise or report them
    if key in dense_keys:
        return None
    if not blocktri.plausible(op, key):
        return None  # (before any grid-sized product is formed)
    try:
        inner = blocktri.BlockTridiagonalNormal(op, key, damp, dampdiag)
    except (RuntimeError, MemoryError) as e:
        # ONLY out of memory while forming S^T S falls through to the matrix-free routes (they need far less); a launch
        # error, a shape bug or an assertion is a defect and must not hide behind a slower solver
        if not (isinstance(e, (MemoryError, torch.cuda.OutOfMemoryError)) or "out of memory" in str(e).lower()):
            raise
        from .util import printlog

        printlog("odil_amd: block cyclic reduction ran out of memory ({}); using the matrix-free routes".format(str(e).splitlines()[0]))
        return None
    if not inner.ok:
        return None
    inner.offset, inner.size = op.key_to_offset[key], op.key_to_size[key]
    try:
        if dense_keys:
            if sum(op.key_to_size[k] for k in dense_keys) > 63:
                return None
            return schur_normal(op, rhs, damp, dampdiag, status=status, inner=inner)
        b = op.rmatvec(rhs)
        x = torch.zeros(op.ncols, dtype=op.dtype, device=op.device)
        x[inner.offset:inner.offset + inner.size] = inner.solve(b[inner.offset:inner.offset + inner.size][None].contiguous())[0]
    except blocktri.NotPositiveDefinite:
        return None  # a singular normal matrix: the general routes (damping, CG) deal with it
    if status is not None:
        r = op.rmatvec(op.matvec(x) - rhs)
        status["residual"] = float(_dot(r, r)) ** 0.5
        status["niter"] = 1
        status["method"] = "block-tridiagonal direct (axis {}, {} levels of {} points)".format(
            inner.axis, inner.shape[inner.axis], inner.size // inner.shape[inner.axis])
    return x


def recognise_marching(op):
    """(coeffs [nshift, size], shifts, diagonal slot, field shape, axis, direction) when M is square, acts on ONE
    field and is triangular along one axis with only the unknown itself on the diagonal block -- the Jacobian of
    an operator that is explicit in time (examples/wave: shifts (0, 0), (-1, 0), (-2, 0), (-1, +-1)) -- and no
    coefficient wraps around the ends of that axis; None otherwise."""
    from .core import Field

    if len(op.key_to_field) != 1 or op.nrows != op.ncols:
        return None
    (key, field), = op.key_to_field.items()
    if not isinstance(field, Field):
        return None
    shape = tuple(field.array.shape)
    ndim = len(shape)
    shifts, coeffs = [], []
    for row0, nrows, kind, k, payload in op.blocks:
        if kind != "stencil" or row0 != 0 or nrows != op.ncols:
            return None
        coeff, shift, loc, vshape = payload
        if loc != field.loc or tuple(vshape) != shape:
            return None
        norm = tuple(((s + n // 2) % n) - n // 2 for s, n in zip(shift, shape))  # periodic roll: |shift| <= n / 2
        if norm in shifts:
            coeffs[shifts.index(norm)] = coeffs[shifts.index(norm)] + coeff.reshape(shape)
        else:
            shifts.append(norm)
            coeffs.append(coeff.reshape(shape))
    zero = (0,) * ndim
    if zero not in shifts or len(shifts) > 32:
        return None
    diag = shifts.index(zero)
    for axis in range(ndim):
        for direction in (1, -1):
            if not all(s == zero or direction * s[axis] < 0 for s in shifts):
                continue
            # rows whose neighbour would lie across the end of the axis must not use it
            ok = float(coeffs[diag].abs().min()) > 0
            for s, c in zip(shifts, coeffs):
                k = abs(s[axis])
                if ok and s != zero:
                    edge = c.narrow(axis, 0, k) if direction > 0 else c.narrow(axis, shape[axis] - k, k)
                    ok = float(edge.abs().max()) == 0
            if ok:
                return torch.stack(coeffs).contiguous(), shifts, diag, shape, axis, direction
    return None


def march_solve(rec, rhs, status=None):
    coeffs, shifts, diag, shape, axis, direction = rec
    x = ops.stencil_march(coeffs, shifts, diag, rhs.reshape(shape).contiguous(), axis, direction)
    if status is not None:
        r = ops.stencil_apply(coeffs, shifts, x) - rhs.reshape(shape)
        status["residual"] = float(_dot(r.reshape(-1), r.reshape(-1))) ** 0.5
        status["niter"] = 1
        status["method"] = "substitution along axis {}".format(axis)
    return x.reshape(-1)


SCHUR_MIN_UNKNOWNS = 16384  # systems with dense columns: Schur complement above this size, dense Cholesky below
DENSE_MAX_UNKNOWNS = 49152  # `direct` factorises the dense normal matrix up to here, memory permitting


def _dense_fits(op):
    """M (rows x unknowns), A = M^T M and the Cholesky factor as dense matrices must fit comfortably:
    16384 unknowns take 6 GB in f64, 49152 take 58 GB of the 288 GB of an MI355X."""
    if op.ncols > DENSE_MAX_UNKNOWNS:
        return False
    esize = 8 if op.dtype == torch.float64 else 4
    need = (op.nrows * op.ncols + 2 * op.ncols * op.ncols) * esize
    free = torch.cuda.mem_get_info(op.device)[0] if op.device.type == "cuda" else 0
    return need <= 0.5 * free


def dense_normal(op, rhs, damp=0.0, dampdiag=0.0, status=None):
    """`direct` for small systems, as the reference's SuperLU solve of A = M^T M (linsolver.py:17-26):
    M is scattered into a dense device matrix, A = M^T M is one f64 GEMM (rocBLAS, the only
    GEMM-shaped work of this path) and A x = M^T rhs is solved by Cholesky (rocSOLVER), LU when A
    is not numerically positive definite.  Returns None when the factorisation fails."""
    m = op.to_dense()
    a = m.t() @ m
    b = m.t() @ rhs
    if damp or dampdiag:
        d = a.diagonal().clone()
        a.diagonal().add_(float(damp) ** 2 + float(dampdiag) ** 2 * (d + float(damp) ** 2))  # (reference linsolver.py:19-23)
    chol, info = torch.linalg.cholesky_ex(a)
    if int(info) == 0:
        x = torch.cholesky_solve(b[:, None], chol)[:, 0]
        method = "dense-cholesky"
    else:
        lu, piv, info = torch.linalg.lu_factor_ex(a)
        if int(info) != 0:
            return None
        x = torch.linalg.lu_solve(lu, piv, b[:, None])[:, 0]
        method = "dense-lu"
    if status is not None:
        r = a @ x - b
        status["residual"] = float(_dot(r.contiguous(), r.contiguous())) ** 0.5
        status["niter"] = 1
        status["method"] = method
    return x


def solve(matr, rhs, args, status=None, linsolver="direct", consume=False):
    """Reference signature (linsolver.py:4).  `matr` is a `core.LinearizedOperator`; returns the
    solution as a device vector.  consume=True (the Newton driver): the result may be a work buffer of the solver, valid
    until the next solve -- no copy of it is made."""
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
    # 512^3.  `multigrid` always takes it when it applies, `direct` beyond the reach of the dense factorisation (49152
    # unknowns; up to round 5 only above 2e5, and 256^2 or N = 100000 in 1-D went to 45000 - 50000 CG iterations).
    if not damp and not dampdiag and (linsolver == "multigrid" or (linsolver == "direct" and matr.ncols > DENSE_MAX_UNKNOWNS)):
        import os

        from . import gmg

        # ODIL_GMG = auto (default) | poisson | stencil: `stencil` sends even the constant-coefficient Laplacian through
        # the variable-coefficient cycle (measurement, tests), `poisson` switches that cycle off
        mode = os.environ.get("ODIL_GMG", "auto")
        gtol = 1e-12 if linsolver == "direct" else tol
        # ODIL_GMG_MIXED=1: float32 V-cycles inside a float64 residual loop (gmg.solve_mixed; float64 problems only)
        mixed = matr.dtype == torch.float64 and bool(int(os.environ.get("ODIL_GMG_MIXED", 0)))
        rec = gmg.recognise_poisson(matr) if mode != "stencil" else None
        if rec is not None:
            shape, h2 = rec
            sub = dict()
            if mixed and all(n % 2 == 0 for n in shape):  # (an odd finest level is solved by GCR in one precision)
                x = gmg.solve_mixed(gmg.PoissonGMG(shape, h2, matr.dtype, matr.device, lite=True),
                                    gmg.PoissonGMG(shape, h2, torch.float32, matr.device),
                                    rhs.reshape(shape).contiguous(), tol=gtol, maxiter=maxiter or 60, status=sub)
            else:
                # (the constant-coefficient solver depends on shape, spacing and dtype only: kept with the domain, so that the
                # next Newton step finds its level buffers and coarsest-grid inverse)
                cache = matr.domain.__dict__.setdefault("_poisson_gmg", dict())
                key = (tuple(shape), tuple(float(v) for v in h2), matr.dtype, str(matr.device))
                solver = cache.get(key)
                if solver is None:
                    cache.clear()
                    solver = cache[key] = gmg.PoissonGMG(shape, h2, matr.dtype, matr.device)
                x = solver.solve(rhs.reshape(shape).contiguous(), tol=gtol, maxiter=maxiter or 60, status=sub, copy=not consume)
            # cells far from cubes (point smoothing with full coarsening loses its rate) can leave the cycles short of the
            # tolerance: the iterate is then handed to the normal-equation CG below as its starting point, not returned
            bnorm = sub["bnorm"] if "bnorm" in sub else float(_dot(rhs, rhs)) ** 0.5
            if sub.get("converged", True) or sub.get("residual", 0.0) <= 1e-6 * bnorm or (
                    sub.get("stagnated") and sub.get("residual", 0.0) <= 1e-3 * bnorm):
                status.update(sub)
                return x.reshape(-1)
            from .util import printlog

            printlog("odil_amd: Poisson multigrid stopped at relative residual {:.1e}; finishing with CG on the normal equations".format(
                sub.get("residual", float("nan")) / max(float(_dot(rhs, rhs)) ** 0.5, 1e-300)))
            return cg_normal(matr, rhs, tol=min(gtol, 1e-10), maxiter=maxiter, status=status, x0=x.reshape(-1))
        # Any other square (2 d + 1)-point operator on one cell-centred field (variable-coefficient diffusion, reaction,
        # convection, other wall closures): V-cycles on its own coefficient arrays.  M d = rhs is solved, which for a
        # nonsingular square M is the solution of the normal equations; cycles that do not contract hand over to the
        # normal-equation routes below.
        coeffs = gmg.recognise_stencil(matr) if mode != "poisson" else None
        if coeffs is not None:
            sub = dict()
            mixed = mixed and all(n % 2 == 0 for n in coeffs.shape[1:])
            if mixed:
                solver = gmg.StencilGMG(coeffs, store=torch.float32)
                x = gmg.solve_mixed(gmg.StencilGMG(coeffs, lite=True), solver, rhs.reshape(tuple(coeffs.shape[1:])).contiguous(),
                                    tol=gtol, maxiter=maxiter or 60, status=sub)
            else:
                solver = gmg.StencilGMG(coeffs)
                x = solver.solve(rhs.reshape(tuple(coeffs.shape[1:])).contiguous(), tol=gtol, maxiter=maxiter or 60, status=sub,
                                 copy=not consume)
            # (a residual below the tolerance, or cycles that stopped at the rounding floor of the working precision
            # well below the right-hand side: the iterate is finite)
            if sub.get("converged") or (sub.get("stagnated") and sub.get("residual", 0.0) <= 1e-3 * (
                    sub["bnorm"] if "bnorm" in sub else float(_dot(rhs, rhs)) ** 0.5)):
                sub["method"] = "gmg-vcycle (variable coefficients, {} levels{})".format(
                    solver.nlvl, "; float32 cycles, float64 residual" if mixed else "")
                status.update(sub)
                return x.reshape(-1)
            from .util import printlog

            printlog("odil_amd: variable-coefficient multigrid did not converge (relative residual {:.1e} after {} cycles); "
                     "using the normal-equation routes".format(sub.get("residual", float("nan")), sub.get("niter", 0)))
            del solver, coeffs, x
    # float32 problems: the EXACT routes below work on a float64 copy of the operator -- the normal matrix squares the
    # condition number, which at 1e3 - 1e4 for M already exceeds what float32 resolves (heat with the network, 64 x 64:
    # loss 45 -> 8e3 in one float32 step, 45 -> 0.5 with the copy; the reference's float32 SuperLU solve sits in between).
    # The iterate is rounded back to the problem's precision; the matrix-free routes (multigrid above, CG below) stay
    # in float32.
    if matr.dtype == torch.float32 and linsolver in ("direct", "directsq", "multigrid"):
        wide = matr.promoted()
        x = _exact_routes(wide, rhs.double(), damp, dampdiag, maxiter, status, linsolver)
        if x is not None:
            return x.to(torch.float32)
    else:
        x = _exact_routes(matr, rhs, damp, dampdiag, maxiter, status, linsolver)
        if x is not None:
            return x
    if linsolver in ("direct", "directsq"):
        return cg_normal(matr, rhs, damp, dampdiag, tol=1e-14, maxiter=maxiter, status=status)
    return cg_normal(matr, rhs, damp, dampdiag, tol=tol, maxiter=maxiter or 1000, status=status)


def _exact_routes(matr, rhs, damp, dampdiag, maxiter, status, linsolver):
    """Substitution, block cyclic reduction, Schur complement, dense factorisation -- in this order, whichever applies;
    None when none does (the caller iterates on the normal equations)."""
    # Square and triangular along one axis (time-explicit operators): M d = rhs by substitution is exact and has the
    # solution of the normal equations
    if not damp and not dampdiag and linsolver in ("direct", "directsq", "multigrid"):
        rec = recognise_marching(matr)
        if rec is not None:
            sub = dict()
            x = march_solve(rec, rhs, sub)
            # exact in exact arithmetic; a time-explicit scheme run beyond its stability limit (wave with dt > dx) amplifies
            # rounding by the growth factor of every level: the iterate is then finite garbage.  Accept what meets the
            # equations, leave the rest to the normal-equation routes (which the reference takes for everything)
            if bool(torch.isfinite(x).all()) and sub["residual"] <= 1e-6 * max(float(_dot(rhs, rhs)) ** 0.5, 1e-300):
                status.update(sub)
                return x
            # (or a singular diagonal block met on the way: the general solvers below regularise or report it)
    if linsolver in ("direct", "directsq"):
        x = blocktri_normal(matr, rhs, damp, dampdiag, status)
        if x is not None and bool(torch.isfinite(x).all()):
            return x
        has_dense = any(kind == "dense" for _, _, kind, _, _ in matr.blocks)
        # small systems: ONE dense Cholesky of the normal matrix beats the p + 1 inner CG solves of the Schur route
        small = _dense_fits(matr) and matr.ncols <= SCHUR_MIN_UNKNOWNS
        if has_dense and not small:
            try:
                x = schur_normal(matr, rhs, damp, dampdiag, maxiter=maxiter, status=status)
            except FloatingPointError:  # an inner CG solve broke down: the dense / CG routes below still apply
                x = None
            if x is not None and bool(torch.isfinite(x).all()):
                return x
        if _dense_fits(matr):
            x = dense_normal(matr, rhs, damp, dampdiag, status=status)
            if x is not None and bool(torch.isfinite(x).all()):
                return x
    return None


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
