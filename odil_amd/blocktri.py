"""Direct solve of the Newton step's normal equations for operators that couple NEIGHBOURING levels of one axis
only -- the implicit-in-time discretisations (reference examples/heat/heat.py:36-137: two time levels, face gradients
averaged in time), whose Jacobian no substitution can invert (reference src/odil/linsolver.py:17-26 hands
A = M^T M to SuperLU).

With S the stencil part of M acting on one field, A = S^T S is again a stencil operator; its coefficient arrays are
sums of products of S's coefficient arrays (formed on the device, a few elementwise passes over the grid).  When the
shifts of A along one axis stay within -1 .. +1 and nothing wraps around the ends of that axis, A is BLOCK
TRIDIAGONAL along it: n levels, blocks of nb = (points of one level) squared.  Block cyclic reduction solves such a
system in log2(n) rounds of batched dense Cholesky factorisations and GEMMs (rocSOLVER / rocBLAS through torch: the f64
matrix cores), every round halving the number of levels -- no sequential sweep over the levels, no iteration, the
answer to round-off.  256 x 512 (heat): ~4e11 flop, tens of milliseconds; the Jacobi-preconditioned CG on the same
normal equations it replaces took 0.7 - 2.3 s.  Dense columns (`NeuralNet` / `Array` unknowns) are eliminated by the
Schur complement of linsolver.schur_normal with this solver for the inner systems (all right-hand sides at once).
"""

import torch

BLOCK_MAX = 2048  # points of one level: dense blocks of nb x nb


def _centred(s, n):
    return ((s + n // 2) % n) - n // 2


def normal_stencil(op, key, damp=0.0, dampdiag=0.0):
    """{shift: coefficient array} of A = S^T S (+ damp^2 I + dampdiag^2 diag) for the stencil blocks of `op` on field
    `key`, or None when a block is not a plain stencil on the field's own grid.
    A[j, j + d] = sum over outputs and pairs (s, s') with s' - s = d of (c_s c_s')(j - s)."""
    field = op.key_to_field[key]
    shape = tuple(field.array.shape)
    by_out = dict()
    for row0, nrows, kind, k, payload in op.blocks:
        if kind != "stencil" or k != key:
            continue
        coeff, shift, loc, vshape = payload
        if loc != field.loc or tuple(vshape) != shape:
            return None
        s = tuple(_centred(a, n) for a, n in zip(shift, shape))
        terms = by_out.setdefault(row0, dict())
        terms[s] = terms[s] + coeff.reshape(shape) if s in terms else coeff.reshape(shape)
    if not by_out:
        return None
    dims = tuple(range(len(shape)))
    normal = dict()
    for terms in by_out.values():
        items = list(terms.items())
        for s, cs in items:
            for s2, cs2 in items:
                d = tuple(_centred(b - a, n) for a, b, n in zip(s, s2, shape))
                contrib = torch.roll(cs * cs2, shifts=s, dims=dims)
                normal[d] = normal[d] + contrib if d in normal else contrib
    zero = (0,) * len(shape)
    if zero not in normal:
        return None
    if damp or dampdiag:
        normal[zero] = (normal[zero] + float(damp) ** 2) * (1.0 + float(dampdiag) ** 2)  # reference linsolver.py:19-23
    return normal


def plausible(op, key):
    """Cheap necessary condition for `recognise` to succeed, from the shifts alone (no grid-sized array is touched):
    some axis with at least 4 levels, blocks of at most BLOCK_MAX points, along which the shifts of every output's
    stencil entries differ by at most one level (S^T S then couples neighbouring levels only)."""
    field = op.key_to_field[key]
    shape = tuple(field.array.shape)
    by_out = dict()
    for row0, nrows, kind, k, payload in op.blocks:
        if kind == "stencil" and k == key:
            by_out.setdefault(row0, []).append(tuple(_centred(a, n) for a, n in zip(payload[1], shape)))
    if not by_out:
        return False
    for axis, n in enumerate(shape):
        nb = 1
        for d, m in enumerate(shape):
            if d != axis:
                nb *= m
        if n >= 4 and nb <= BLOCK_MAX and all(max(s[axis] for s in ss) - min(s[axis] for s in ss) <= 1 for ss in by_out.values()):
            return True
    return False


def recognise(normal, shape):
    """The axis along which the normal operator is block tridiagonal (shifts within -1 .. 1, no coupling across the
    ends) with the smallest blocks, or None."""
    best = None
    for axis, n in enumerate(shape):
        nb = 1
        for d, m in enumerate(shape):
            if d != axis:
                nb *= m
        if n < 4 or nb > BLOCK_MAX or any(abs(d[axis]) > 1 for d in normal):
            continue
        ok = True
        for d, c in normal.items():
            if d[axis] == -1:
                ok = ok and float(c.narrow(axis, 0, 1).abs().max()) == 0.0
            elif d[axis] == 1:
                ok = ok and float(c.narrow(axis, n - 1, 1).abs().max()) == 0.0
        if ok and (best is None or nb < best[1]):
            best = (axis, nb)
    return None if best is None else best[0]


def dense_blocks(normal, shape, axis):
    """(L, D, U): dense blocks [n, nb, nb] of the block rows -- L[i] couples level i to i - 1, U[i] to i + 1."""
    n = shape[axis]
    other = [m for d, m in enumerate(shape) if d != axis]
    nb = 1
    for m in other:
        nb *= m
    any_c = next(iter(normal.values()))
    dev, dt = any_c.device, any_c.dtype
    idx = torch.arange(nb, device=dev).reshape(other if other else [1])
    rows = torch.arange(nb, device=dev).reshape(1, nb) * nb
    out = [torch.zeros((n, nb * nb), dtype=dt, device=dev) for _ in range(3)]
    odims = tuple(range(len(other)))
    for d, c in normal.items():
        do = tuple(v for k, v in enumerate(d) if k != axis)
        cols = torch.roll(idx, shifts=tuple(-v for v in do), dims=odims).reshape(1, nb) if other else idx.reshape(1, 1)
        vals = c.movedim(axis, 0).reshape(n, nb)
        out[d[axis] + 1].scatter_add_(1, (rows + cols).expand(n, nb), vals)
    return tuple(t.view(n, nb, nb) for t in out)


class NotPositiveDefinite(ArithmeticError):
    pass


def _cholesky(A):
    c, info = torch.linalg.cholesky_ex(A)
    if bool(info.any()):
        raise NotPositiveDefinite("a diagonal block of the normal equations is not positive definite")
    return c


def solve_block_tridiagonal(L, D, U, B):
    """X with L[i] X[i-1] + D[i] X[i] + U[i] X[i+1] = B[i] (L[0], U[n-1] ignored) by block cyclic reduction, for a
    symmetric positive definite system.  L, D, U: [n, nb, nb]; B: [n, nb, k].  Every round eliminates the odd levels
    with ONE batched Cholesky factorisation.  Raises NotPositiveDefinite for a singular / indefinite system."""
    n = D.shape[0]
    if n == 1:
        return torch.cholesky_solve(B, _cholesky(D))
    nb, k = D.shape[1], B.shape[2]
    ne, m = (n + 1) // 2, n // 2  # even / odd levels
    # the diagonal blocks of an SPD block-tridiagonal matrix and of its Schur complements are SPD: batched Cholesky
    # (128 x 512 x 512 with 1025 right-hand sides: 8 ms; the batched LU route of this stack takes 250 ms at that size
    # when its triangular solves do not fail outright -- tools/probe_batched_solve.py)
    sol = torch.cholesky_solve(torch.cat([L[1::2], U[1::2], B[1::2]], dim=2), _cholesky(D[1::2]))
    iL, iU, iB = sol[:, :, :nb], sol[:, :, nb:2 * nb], sol[:, :, 2 * nb:]
    Le, De, Ue, Be = L[0::2], D[0::2].clone(), U[0::2], B[0::2].clone()
    Ln, Un = torch.zeros_like(De), torch.zeros_like(De)
    # right odd neighbour of even level 2k: odd index k (exists for k < m)
    De[:m] -= Ue[:m] @ iL
    Un[:m] = -(Ue[:m] @ iU)
    Be[:m] -= Ue[:m] @ iB
    # left odd neighbour of even level 2k, k >= 1: odd index k - 1
    De[1:] -= Le[1:] @ iU[: ne - 1]
    Ln[1:] = -(Le[1:] @ iL[: ne - 1])
    Be[1:] -= Le[1:] @ iB[: ne - 1]
    Xe = solve_block_tridiagonal(Ln, De, Un, Be)
    Xo = iB - iL @ Xe[:m]
    if ne - 1 > 0:
        Xo[: ne - 1] -= iU[: ne - 1] @ Xe[1:]
    X = torch.empty((n, nb, k), dtype=B.dtype, device=B.device)
    X[0::2], X[1::2] = Xe, Xo
    return X


class BlockTridiagonalNormal:
    """(S^T S + damping)^{-1} for the stencil part S of a linearised operator on field `key`."""

    def __init__(self, op, key, damp=0.0, dampdiag=0.0):
        self.ok = False
        field = op.key_to_field[key]
        self.shape = tuple(field.array.shape)
        normal = normal_stencil(op, key, damp, dampdiag)
        if normal is None:
            return
        axis = recognise(normal, self.shape)
        if axis is None:
            return
        self.axis = axis
        self.L, self.D, self.U = dense_blocks(normal, self.shape, axis)
        self.ok = True

    def solve(self, b):
        """b: [k, size] right-hand sides (rows) -> [k, size] solutions."""
        k = b.shape[0]
        n = self.shape[self.axis]
        rhs = b.reshape((k,) + self.shape).movedim(1 + self.axis, 1)  # [k, n, other...]
        rhs = rhs.reshape(k, n, -1).permute(1, 2, 0).contiguous()       # [n, nb, k]
        x = solve_block_tridiagonal(self.L, self.D, self.U, rhs)
        other = tuple(m for d, m in enumerate(self.shape) if d != self.axis)
        x = x.permute(2, 0, 1).reshape((k, n) + other).movedim(1, 1 + self.axis)
        return x.reshape(k, -1).contiguous()
