"""NumPy restatement (TEST INFRASTRUCTURE) of the variable-coefficient multigrid of odil_amd/gmg.py: StencilGMG and
odil_amd/csrc/stencil_mg.hip -- the operator application, a Jacobi sweep, the restricted residual, the coarse-operator
construction (aggregates of 2^d cells: half of the piecewise-constant Galerkin product of the second-order part + the
antisymmetric part + the row sums, cell-Peclet limiter) and a V(2, 2) cycle.  The HIP kernels are held to it
(tests/test_stencil_gmg_gpu.py); tests/test_stencil_gmg_host.py records the contraction numbers of the algorithm itself.
Coefficient order: (0, -e_0, +e_0, -e_1, +e_1, ...), neighbours wrap periodically."""

import numpy as np

from oracle import odil_np as onp


def shifts_of(ndim):
    out = [(0,) * ndim]
    for a in range(ndim):
        out += [tuple(-1 if j == a else 0 for j in range(ndim)), tuple(1 if j == a else 0 for j in range(ndim))]
    return out


def apply(coeffs, x):
    nd = x.ndim
    y = coeffs[0] * x
    for s, c in zip(shifts_of(nd)[1:], coeffs[1:]):
        y = y + c * np.roll(x, [-v for v in s], axis=tuple(range(nd)))
    return y


def jacobi(coeffs, x, b, omega):
    return x - omega * (apply(coeffs, x) - b) / coeffs[0]


def restrict_mean(r):
    nd = r.ndim
    blocks = r.reshape([v for n in r.shape for v in (n // 2, 2)])
    return blocks.mean(axis=tuple(range(1, 2 * nd, 2)))


def _galerkin_p0(coeffs, halve=None):
    """R A P0 with R = mean of the children, P0 = piecewise constant: again 2 d + 1 arrays.  halve: the axes along which
    two cells are merged (default: all -- 2^d children); the other axes keep their cells."""
    nd = coeffs[0].ndim
    sh = coeffs[0].shape
    halve = [True] * nd if halve is None else list(halve)
    blocks = lambda a: a.reshape([v for n, on in zip(sh, halve) for v in ((n // 2, 2) if on else (n, 1))])  # noqa: E731
    pair_axes = tuple(range(1, 2 * nd, 2))
    c0 = blocks(coeffs[0]).sum(axis=pair_axes)
    out = []
    for a in range(nd):
        cm, cp = blocks(coeffs[1 + 2 * a]), blocks(coeffs[2 + 2 * a])
        other = tuple(ax for ax in pair_axes if ax != 2 * a + 1)

        def red(arr, child):
            idx = [slice(None)] * (2 * nd)
            idx[2 * a + 1] = child
            sub = arr[tuple(idx)]
            return sub.sum(axis=tuple(ax if ax < 2 * a + 1 else ax - 1 for ax in other))

        if halve[a]:
            c0 = c0 + red(cm, 1) + red(cp, 0)  # couplings between the two children along a are internal
            out += [red(cm, 0), red(cp, 1)]
        else:
            out += [red(cm, 0), red(cp, 0)]    # every cell keeps both neighbours along a
    w = 1.0 / 2 ** sum(halve)
    return [c0 * w] + [c * w for c in out]


def coarsen_axes(coeffs, halve):
    """The coarse operator when only the axes `halve` are merged (semi-coarsening: the strongly coupled axes of an
    anisotropic operator).  Same split as `coarsen`; the factor 1/2 of the second-order part belongs to the axes whose
    spacing doubles, so that part is taken apart axis by axis: A2 = sum_a A2_a (the couplings of axis a with their share of
    the diagonal, zero row sums) + a diagonal remainder on incomplete rows (wall closures), which goes with the incomplete
    axis -- 1/2 when that one is merged, 1 otherwise."""
    nd = coeffs[0].ndim
    halve = list(halve)
    complete, sym, anti = [], [None] * (1 + 2 * nd), [None] * (1 + 2 * nd)
    for a in range(nd):
        cm, cp = coeffs[1 + 2 * a], coeffs[2 + 2 * a]
        complete.append((cm != 0) & (cp != 0))
        cm_next, cp_prev = np.roll(cm, -1, axis=a), np.roll(cp, 1, axis=a)
        pair_p, pair_m = (cp != 0) & (cm_next != 0), (cm != 0) & (cp_prev != 0)
        sym[2 + 2 * a] = np.where(pair_p, 0.5 * (cp + cm_next), cp)
        anti[2 + 2 * a] = np.where(pair_p, 0.5 * (cp - cm_next), 0.0)
        sym[1 + 2 * a] = np.where(pair_m, 0.5 * (cm + cp_prev), cm)
        anti[1 + 2 * a] = np.where(pair_m, 0.5 * (cm - cp_prev), 0.0)
    sym[0], anti[0] = coeffs[0].copy(), np.zeros_like(coeffs[0])
    all_complete = np.logical_and.reduce(complete)
    z = np.where(all_complete, sum(sym), 0.0)
    zeros = lambda: [np.zeros_like(z) for _ in range(1 + 2 * nd)]  # noqa: E731
    a0 = zeros()
    a0[0] = z
    g = _galerkin_p0(a0, halve)
    for k, c in enumerate(_galerkin_p0(anti, halve)):
        g[k] = g[k] + c
    g1 = [c.copy() for c in _galerkin_p0(anti, halve)]
    g2 = zeros()
    g2 = [np.zeros_like(g[0]) for _ in range(1 + 2 * nd)]
    rem = sym[0] - z
    for a in range(nd):
        part = zeros()
        part[1 + 2 * a], part[2 + 2 * a] = sym[1 + 2 * a], sym[2 + 2 * a]
        part[0] = -(sym[1 + 2 * a] + sym[2 + 2 * a])
        rem = rem - part[0]
        f = 0.5 if halve[a] else 1.0
        for k, c in enumerate(_galerkin_p0(part, halve)):
            g2[k] = g2[k] + f * c
    # the remainder: zero on complete rows; on a row next to a wall the closure of the incomplete axis
    half_rows = np.zeros(z.shape, dtype=bool)
    for a in range(nd):
        if halve[a]:
            half_rows |= ~complete[a]
    for rows, f in ((half_rows, 0.5), (~half_rows, 1.0)):
        part = zeros()
        part[0] = np.where(rows, rem, 0.0)
        g2[0] = g2[0] + f * _galerkin_p0(part, halve)[0]
    for k in range(1, 1 + 2 * nd):
        s2, n1 = g2[k], g1[k]
        need = np.abs(n1) > np.abs(s2)
        sgn = np.where(s2 != 0, np.sign(s2), np.where(g2[0] > 0, -1.0, 1.0))
        s_new = np.where(need, sgn * np.abs(n1), s2)
        g2[0] = g2[0] - (s_new - s2)
        g2[k] = s_new
    return [a + b for a, b in zip(g2, g)]


def coarsen(coeffs):
    nd = coeffs[0].ndim
    complete, sym, anti = [], [None] * (1 + 2 * nd), [None] * (1 + 2 * nd)
    for a in range(nd):
        cm, cp = coeffs[1 + 2 * a], coeffs[2 + 2 * a]
        complete.append((cm != 0) & (cp != 0))
        cm_next, cp_prev = np.roll(cm, -1, axis=a), np.roll(cp, 1, axis=a)  # the reverse couplings
        pair_p, pair_m = (cp != 0) & (cm_next != 0), (cm != 0) & (cp_prev != 0)
        sym[2 + 2 * a] = np.where(pair_p, 0.5 * (cp + cm_next), cp)
        anti[2 + 2 * a] = np.where(pair_p, 0.5 * (cp - cm_next), 0.0)
        sym[1 + 2 * a] = np.where(pair_m, 0.5 * (cm + cp_prev), cm)
        anti[1 + 2 * a] = np.where(pair_m, 0.5 * (cm - cp_prev), 0.0)
    sym[0], anti[0] = coeffs[0].copy(), np.zeros_like(coeffs[0])
    z = np.where(np.logical_and.reduce(complete), sum(sym), 0.0)
    a0 = [z] + [np.zeros_like(z) for _ in range(2 * nd)]
    a2 = [c - r for c, r in zip(sym, a0)]
    g2 = [0.5 * c for c in _galerkin_p0(a2)]
    g1, g0 = _galerkin_p0(anti), _galerkin_p0(a0)
    for k in range(1, 1 + 2 * nd):
        s2, n1 = g2[k], g1[k]
        need = np.abs(n1) > np.abs(s2)
        sgn = np.where(s2 != 0, np.sign(s2), np.where(g2[0] > 0, -1.0, 1.0))
        s_new = np.where(need, sgn * np.abs(n1), s2)
        g2[0] = g2[0] - (s_new - s2)
        g2[k] = s_new
    return [a + b + c for a, b, c in zip(g2, g1, g0)]


def chebyshev_weights(n, nd):
    lo, hi = 1.0 / nd, 2.0
    mid, half = 0.5 * (hi + lo), 0.5 * (hi - lo)
    return [1.0 / (mid - half * np.cos(np.pi * (2 * k + 1) / (2 * n))) for k in range(n)]


def hierarchy(coeffs, min_size=2):
    levels = [coeffs]
    while all(s % 2 == 0 and s // 2 >= min_size for s in levels[-1][0].shape):
        levels.append(coarsen(levels[-1]))
    return levels


def vcycle(levels, lvl, x, b, nu=2, top2=False):
    """top2: two cycles on the first coarse level (what StencilGMG does in 3-D)."""
    coeffs = levels[lvl]
    nd = x.ndim
    if lvl == len(levels) - 1:
        n = x.size
        amat = np.zeros((n, n))
        for j in range(n):
            e = np.zeros(n)
            e[j] = 1
            amat[:, j] = apply(coeffs, e.reshape(x.shape)).ravel()
        return (np.linalg.pinv(amat, rcond=1e-12) @ b.ravel()).reshape(x.shape)
    w = chebyshev_weights(nu, nd)
    for k in range(nu):
        x = jacobi(coeffs, x, b, w[k])
    rc = restrict_mean(b - apply(coeffs, x))
    ec = vcycle(levels, lvl + 1, np.zeros_like(rc), rc, nu)
    if top2 and lvl == 0 and len(levels) > 2:
        ec = vcycle(levels, lvl + 1, ec, rc, nu)
    x = x + onp.interp_to_finer(ec, "c" * nd)
    for k in range(nu):
        x = jacobi(coeffs, x, b, w[k])
    return x


def poisson_coeffs(shape):
    c = onp.poisson_jac_coeffs(shape, onp.step(shape))
    return [c[s] for s in shifts_of(len(shape))]


def diffusion_coeffs(shape, kfun, sigma=0.0):
    """div(k grad u) - sigma u with the Poisson example's quadratic wall ghosts (examples/diffusion/diffusion.py)."""
    nd = len(shape)
    h = [1.0 / n for n in shape]
    xs = np.meshgrid(*[(np.arange(n) + 0.5) / n for n in shape], indexing="ij")
    c0, out = np.zeros(shape) - sigma, []
    for a in range(nd):
        xp = [x + (0.5 * h[a] if j == a else 0.0) for j, x in enumerate(xs)]
        xm = [x - (0.5 * h[a] if j == a else 0.0) for j, x in enumerate(xs)]
        kp, km = kfun(*xp) / h[a] ** 2, kfun(*xm) / h[a] ** 2
        idx = np.arange(shape[a]).reshape([-1 if j == a else 1 for j in range(nd)])
        lo, hi = idx == 0, idx == shape[a] - 1
        # ghost below: u_g = (u_+ - 6 u) / 3  ->  k_- (u_g - u) = k_- (u_+ / 3 - 3 u)
        cm = np.where(lo, 0, km) + np.where(hi, kp / 3, 0)
        cp = np.where(hi, 0, kp) + np.where(lo, km / 3, 0)
        c0 = c0 - np.where(lo, 3 * km, km) - np.where(hi, 3 * kp, kp)
        out += [cm * np.ones(shape), cp * np.ones(shape)]
    return [c0] + out


def add_upwind_convection(coeffs, v, axis=0):
    """- v du/dx_axis by first-order upwinding (v > 0) on the rows away from the walls of that axis."""
    shape = coeffs[0].shape
    n = shape[axis]
    idx = np.arange(n).reshape([-1 if j == axis else 1 for j in range(len(shape))])
    interior = (idx > 0) & (idx < n - 1)
    out = [c.copy() for c in coeffs]
    out[0] = out[0] - np.where(interior, v * n, 0)
    out[1 + 2 * axis] = out[1 + 2 * axis] + np.where(interior, v * n, 0)
    return out


def pick_axes(coeffs, min_size=2):
    """The axes StencilGMG merges on this level: largest coupling (the smaller of the two directions) within a factor 2 of
    the largest of all axes; None when one of them cannot be halved."""
    nd = coeffs[0].ndim
    strength = [min(np.abs(coeffs[1 + 2 * a]).max(), np.abs(coeffs[2 + 2 * a]).max()) for a in range(nd)]
    halve = [s >= 0.5 * max(strength) for s in strength]
    ok = all(n % 2 == 0 and n // 2 >= min_size for n, on in zip(coeffs[0].shape, halve) if on)
    return halve if ok else None


def hierarchy_axes(coeffs):
    levels, halves = [coeffs], []
    while True:
        h = pick_axes(levels[-1])
        if h is None:
            return levels, halves
        levels.append(coarsen_axes(levels[-1], h))
        halves.append(h)


def vcycle_axes(levels, halves, lvl, x, b, nu=2):
    """V(nu, nu) with semi-coarsened transitions: mean over the merged children, linear interpolation along merged axes."""
    coeffs, nd = levels[lvl], x.ndim
    if lvl == len(levels) - 1:
        n = x.size
        amat = np.zeros((n, n))
        for j in range(n):
            e = np.zeros(n)
            e[j] = 1
            amat[:, j] = apply(coeffs, e.reshape(x.shape)).ravel()
        return (np.linalg.pinv(amat, rcond=1e-12) @ b.ravel()).reshape(x.shape)
    w = chebyshev_weights(nu, nd)
    for k in range(nu):
        x = jacobi(coeffs, x, b, w[k])
    r = b - apply(coeffs, x)
    h = halves[lvl]
    rc = r.reshape([v for n, on in zip(r.shape, h) for v in ((n // 2, 2) if on else (n, 1))]).mean(axis=tuple(range(1, 2 * nd, 2)))
    ec = vcycle_axes(levels, halves, lvl + 1, np.zeros_like(rc), rc, nu)
    x = x + onp.interp_to_finer(ec, "".join("c" if on else "." for on in h))
    for k in range(nu):
        x = jacobi(coeffs, x, b, w[k])
    return x
