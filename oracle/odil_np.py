"""CPU oracle: a NumPy restatement of the ODIL hot path (TEST INFRASTRUCTURE ONLY).

Every function restates one piece of the reference algorithm (cselab/odil v0.1.8,
paths relative to the reference root) and cites the lines it follows.  The
restatement is *pinned* against golden vectors produced by running the
reference's own code in the build container (tests/golden/make_golden.py,
fixtures tests/golden/*.npz) -- see tests/test_oracle_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this module.  The product (odil_amd/) never does: it calls the HIP
kernels through the C-ABI and fails loudly when they are missing.

Adjoints are written out by hand (scatter/fold form), deliberately in a
different formulation from the gather form the HIP kernels use, so the two
implementations cross-check each other.
"""

import itertools

import numpy as np

# --------------------------------------------------------------------------
# Grid geometry (reference src/odil/core.py:61-77, 99-215)
# --------------------------------------------------------------------------


def field_shape(cshape, loc):
    """core.py:165-177: 'n' axes carry one more point than cells."""
    return tuple(int(s) + 1 if c == "n" else int(s) for s, c in zip(cshape, loc))


def mg_cshapes(cshape, mg_axes=None, mg_nlvl=None):
    """core.py:65-73: level shapes fine->coarse, nlvl = min_axes round(log2 n)."""
    ndim = len(cshape)
    mg_axes = mg_axes or [True] * ndim
    nlvl_max = min(int(round(np.log2(n))) if ax else max(cshape) for n, ax in zip(cshape, mg_axes))
    nlvl = nlvl_max if mg_nlvl is None else min(mg_nlvl, nlvl_max)
    return [tuple(n >> lvl if ax else n for n, ax in zip(cshape, mg_axes)) for lvl in range(nlvl)]


def points_1d(lower, upper, n, loc, dtype):
    """core.py:99-107."""
    if loc == "c":
        x = np.linspace(lower, upper, n, endpoint=False, dtype=dtype)
        if len(x) > 1:
            x += (x[1] - x[0]) * 0.5
        return x
    return np.linspace(lower, upper, n + 1, dtype=dtype)


def points(cshape, loc=None, lower=0.0, upper=1.0, dtype=np.float64):
    """core.py:125-136 (all dims, no '.' axes)."""
    ndim = len(cshape)
    loc = loc or "c" * ndim
    lower = (np.ones(ndim, dtype=dtype) * lower).astype(dtype)
    upper = (np.ones(ndim, dtype=dtype) * upper).astype(dtype)
    xx = [points_1d(lower[d], upper[d], cshape[d], loc[d], dtype) for d in range(ndim)]
    return np.meshgrid(*xx, indexing="ij")


def step(cshape, lower=0.0, upper=1.0, dtype=np.float64):
    """core.py:199-200: (upper - lower) / cshape per axis, in the domain dtype."""
    ndim = len(cshape)
    lower = (np.ones(ndim, dtype=dtype) * lower).astype(dtype)
    upper = (np.ones(ndim, dtype=dtype) * upper).astype(dtype)
    return tuple((upper[i] - lower[i]) / cshape[i] for i in range(ndim))


# --------------------------------------------------------------------------
# Prolongation P and its transpose (core.py:606-700; method 'stack')
# --------------------------------------------------------------------------


def _interp_tables(loc):
    """Per output parity d (one bit per non-'.' axis): list of (r, w) and sum(w).

    core.py:675-687: offsets dd in meshgrid 'ij' order; weight(r, d) =
    3**(#c-axes with r == d) if r <= d on all node axes else 0.
    """
    bits = [[0] if l == "." else [0, 1] for l in loc]
    dd = list(itertools.product(*bits))
    sc = [i for i, l in enumerate(loc) if l == "c"]
    sn = [i for i, l in enumerate(loc) if l == "n"]

    def weight(r, d):
        if all(r[i] - d[i] <= 0 for i in sn):
            return 3 ** sum(1 - abs(r[i] - d[i]) for i in sc)
        return 0

    return {d: ([(r, weight(r, d)) for r in dd], sum(weight(r, d) for r in dd)) for d in dd}


def _upad(u, loc):
    """core.py:640-643: ghost = 2*symmetric - reflect, jointly over all 'c' axes."""
    pad_width = [(1, 1) if l == "c" else (0, 0) for l in loc]
    return 2 * np.pad(u, pad_width, mode="symmetric") - np.pad(u, pad_width, mode="reflect")


def fine_shape(shape, loc):
    return tuple({"c": 2 * s, "n": 2 * s - 1, ".": s}[l] for s, l in zip(shape, loc))


def _parity_slices(shape, loc, s):
    """For fine parity s: (fine slice, source start offset in (padded) coarse, count) per axis.

    'c' axis: fine k = 2 i + s reads padded index (i + s) + r (see core.py:689-696:
    fine = res[1:-3] and res[2 ip + d] with d = 1 - s).
    'n' axis: fine k = 2 i + s reads coarse index i + r, i < n - s.
    """
    fsl, start, count, d = [], [], [], []
    for n, l, sa in zip(shape, loc, s):
        if l == "c":
            fsl.append(slice(sa, 2 * n, 2))
            start.append(sa)
            count.append(n)
            d.append(1 - sa)
        elif l == "n":
            fsl.append(slice(sa, 2 * n - 1, 2))
            start.append(0)
            count.append(n - sa)
            d.append(sa)
        else:
            fsl.append(slice(None))
            start.append(0)
            count.append(n)
            d.append(0)
    return tuple(fsl), start, count, tuple(d)


def interp_to_finer(u, loc, depth=1):
    """Linear prolongation, one level (core.py:606-700, 'stack' summation order)."""
    if depth == 0:
        return u
    u = np.asarray(u)
    assert len(loc) == u.ndim
    tables = _interp_tables(loc)
    upad = _upad(u, loc)
    out = np.empty(fine_shape(u.shape, loc), dtype=u.dtype)
    bits = [[0] if l == "." else [0, 1] for l in loc]
    for s in itertools.product(*bits):
        fsl, start, count, d = _parity_slices(u.shape, loc, s)
        terms, sumw = tables[d]
        acc = 0
        for r, w in terms:
            if not w:
                continue
            src = tuple(slice(st + ra, st + ra + c) for st, ra, c in zip(start, r, count))
            acc = acc + w * upad[src]
        out[fsl] = acc / sumw
    return interp_to_finer(out, loc, depth - 1)


def _fold_pad(gpad, shape, loc, mode, cut=(False, False)):
    """Transpose of np.pad(mode) on the 'c' axes: scatter-add ghosts back.  `cut`: axis 0 is
    an interior slab interface at its low / high end -- the padded entry there is not a
    ghost of this array and is dropped (slab decomposition)."""
    g = gpad
    for ax, (n, l) in enumerate(zip(shape, loc)):
        if l != "c":
            continue
        idx = np.pad(np.arange(n), (1, 1), mode=mode)
        out_shape = list(g.shape)
        out_shape[ax] = n
        out = np.zeros(out_shape, dtype=g.dtype)
        lo = 1 if (ax == 0 and cut[0]) else 0
        hi = n + 2 - (1 if (ax == 0 and cut[1]) else 0)
        sel = tuple(slice(lo, hi) if a == ax else slice(None) for a in range(g.ndim))
        np.add.at(out, tuple(idx[lo:hi] if a == ax else slice(None) for a in range(g.ndim)), g[sel])
        g = out
    return g


def interp_to_finer_adj(gfine, loc, coarse_shape, cut=(False, False)):
    """P^T: cotangent of interp_to_finer (what autodiff yields, core.py:1100 / :1062)."""
    gfine = np.asarray(gfine)
    tables = _interp_tables(loc)
    pshape = tuple(n + 2 if l == "c" else n for n, l in zip(coarse_shape, loc))
    gpad = np.zeros(pshape, dtype=gfine.dtype)
    bits = [[0] if l == "." else [0, 1] for l in loc]
    for s in itertools.product(*bits):
        fsl, start, count, d = _parity_slices(coarse_shape, loc, s)
        terms, sumw = tables[d]
        gs = gfine[fsl] / sumw
        for r, w in terms:
            if not w:
                continue
            src = tuple(slice(st + ra, st + ra + c) for st, ra, c in zip(start, r, count))
            gpad[src] += w * gs
    return 2 * _fold_pad(gpad, coarse_shape, loc, "symmetric", cut) - _fold_pad(gpad, coarse_shape, loc, "reflect", cut)


# --------------------------------------------------------------------------
# Restriction (core.py:703-755, method 'conv'; backend.py:112-126)
# --------------------------------------------------------------------------


def restrict_to_coarser(u, loc, depth=1):
    """Full weighting: 'c' [1,1]/2, 'n' [1,2,1]/4 with linearly extrapolated ghosts,
    stride 2 VALID.  The integer stride is applied to every axis (backend.py:118-119),
    so '.' axes are subsampled too."""
    if depth == 0:
        return u
    u = np.asarray(u)
    pad_width = [(1, 1) if l == "n" else (0, 0) for l in loc]
    upad = 2 * np.pad(u, pad_width, mode="symmetric") - np.pad(u, pad_width, mode="reflect")
    wloc = {"n": np.array([1, 2, 1]) * 0.25, "c": np.array([1, 1]) * 0.5, ".": np.array([1.0])}
    res = upad
    for ax, l in enumerate(loc):
        w = wloc[l].astype(u.dtype)
        k = len(w)
        nout = (res.shape[ax] - k) // 2 + 1
        acc = 0
        for j in range(k):
            sl = [slice(None)] * res.ndim
            sl[ax] = slice(j, j + 2 * (nout - 1) + 1, 2)
            acc = acc + w[j] * res[tuple(sl)]
        res = acc
    return restrict_to_coarser(res, loc, depth - 1)


def restrict_to_coarser_adj(gcoarse, loc, fine_shape_):
    """R^T: cotangent of restrict_to_coarser (needed by `poisson --mgloss`, poisson.py:116-122)."""
    gcoarse = np.asarray(gcoarse)
    pshape = tuple(n + 2 if l == "n" else n for n, l in zip(fine_shape_, loc))
    wloc = {"n": np.array([1, 2, 1]) * 0.25, "c": np.array([1, 1]) * 0.5, ".": np.array([1.0])}
    # transpose of the separable strided correlation, axis by axis (last applied first)
    g = gcoarse
    for ax in reversed(range(len(loc))):
        w = wloc[loc[ax]].astype(gcoarse.dtype)
        nout = g.shape[ax]
        shape = list(g.shape)
        shape[ax] = pshape[ax]
        up = np.zeros(shape, dtype=gcoarse.dtype)
        for j in range(len(w)):
            sl = [slice(None)] * g.ndim
            sl[ax] = slice(j, j + 2 * (nout - 1) + 1, 2)
            up[tuple(sl)] += w[j] * g
        g = up
    # transpose of upad = 2 * symmetric - reflect on the 'n' axes
    nloc = "".join("c" if l == "n" else "." for l in loc)  # _fold_pad folds the axes marked 'c'
    return 2 * _fold_pad(g, fine_shape_, nloc, "symmetric") - _fold_pad(g, fine_shape_, nloc, "reflect")


# --------------------------------------------------------------------------
# Multigrid synthesis u = sum_l P^l (f_l w_l) and its adjoint (core.py:245-263)
# --------------------------------------------------------------------------


def mg_loc(loc, axes=None):
    """core.py:259: axes without decomposition interpolate as '.'."""
    axes = axes or [True] * len(loc)
    return "".join(l if ax else "." for l, ax in zip(loc, axes))


def multigrid_to_regular(terms, loc, factors=None, axes=None):
    factors = factors or [1] * len(terms)
    arrays = [t * f for t, f in zip(terms, factors)]
    iloc = mg_loc(loc, axes)
    res = arrays[-1]
    for array in reversed(arrays[:-1]):
        res = array + interp_to_finer(res, iloc)
    return res


def multigrid_to_regular_adj(gfine, shapes, loc, factors=None, axes=None):
    """Returns [f_l * (P^T)^l gfine] for the level array shapes `shapes`."""
    factors = factors or [1] * len(shapes)
    iloc = mg_loc(loc, axes)
    grads = []
    g = np.asarray(gfine)
    for lvl, shape in enumerate(shapes):
        if lvl > 0:
            g = interp_to_finer_adj(g, iloc, shape)
        grads.append(g * factors[lvl])
    return grads


# --------------------------------------------------------------------------
# Stencil access (core.py:910-975) and its adjoint
# --------------------------------------------------------------------------


def field_access(array, field_loc, shift=None, loc=None):
    """ctx.field: 'c'->'n' zero-pad low (core.py:956-960), periodic roll by -shift
    (:962-963), 'n'->'c' drop last (:965-969)."""
    ndim = array.ndim
    shift = tuple(shift) if shift else (0,) * ndim
    loc = loc or field_loc
    pad_flag = [lf == "c" and l == "n" for lf, l in zip(field_loc, loc)]
    if any(pad_flag):
        array = np.pad(array, [(1, 0) if f else (0, 0) for f in pad_flag], mode="constant")
    if any(shift):
        array = np.roll(array, np.negative(shift), range(ndim))
    trim_flag = [lf == "n" and l == "c" for lf, l in zip(field_loc, loc)]
    if any(trim_flag):
        array = array[tuple(slice(0, -1 if f else None) for f in trim_flag)]
    return array


def field_access_adj(g, src_shape, field_loc, shift=None, loc=None):
    """Transpose of field_access: un-trim (zero), roll back, un-pad (drop)."""
    ndim = len(src_shape)
    shift = tuple(shift) if shift else (0,) * ndim
    loc = loc or field_loc
    trim_flag = [lf == "n" and l == "c" for lf, l in zip(field_loc, loc)]
    if any(trim_flag):
        g = np.pad(g, [(0, 1) if f else (0, 0) for f in trim_flag], mode="constant")
    if any(shift):
        g = np.roll(g, shift, range(ndim))
    pad_flag = [lf == "c" and l == "n" for lf, l in zip(field_loc, loc)]
    if any(pad_flag):
        g = g[tuple(slice(1, None) if f else slice(None) for f in pad_flag)]
    assert g.shape == tuple(src_shape)
    return g


# --------------------------------------------------------------------------
# Poisson workload (examples/poisson/poisson.py:57-123, core.py:1439-1445)
# --------------------------------------------------------------------------


def extrap_quadh(u0, u1, u1p):
    """core.py:1439-1445."""
    return (u0 - 6 * u1 + 8 * u1p) / 3


def poisson_ref_u(cshape, dtype=np.float64):
    """poisson.py:18-24 ('hat')."""
    xw = points(cshape, dtype=dtype)
    p = 5
    u = np.prod([(1 - x) * x * 5 for x in xw], axis=0)
    return (u**p / (1 + u**p)) ** (1 / p)


def _poisson_stencil(u, dw):
    """Laplacian with zero-Dirichlet ghosts, poisson.py:57-68 and :104-112 op order."""
    ndim = u.ndim
    iw = np.meshgrid(*[np.arange(n) for n in u.shape], indexing="ij")
    nw = u.shape
    zero = u.dtype.type(0)
    u_ww = []
    for i in range(ndim):
        qwm = np.roll(u, 1, i)
        qwp = np.roll(u, -1, i)
        qm = np.where(iw[i] == 0, extrap_quadh(qwp, u, zero), qwm)
        qp = np.where(iw[i] == nw[i] - 1, extrap_quadh(qwm, u, zero), qwp)
        u_ww.append((qp - 2 * u + qm) / dw[i] ** 2)
    return sum(u_ww)


def poisson_discrete_rhs(ref_u, dw):
    """poisson.py:71-86."""
    return _poisson_stencil(ref_u, dw)


def poisson_residual(u, rhs, dw):
    """poisson.py:89-113: fu = sum_i u_ww - rhs."""
    return _poisson_stencil(u, dw) - rhs


def poisson_jac_coeffs(shape, dw, dtype=np.float64):
    """Per-shift coefficient arrays d(sum fu)/d u_shift, as `eval_operator_grad` with
    `distinct_shift=True` returns them (core.py:1313-1361).
    Keys: shift tuples (0..), (-1 on axis i), (+1 on axis i)."""
    ndim = len(shape)
    iw = np.meshgrid(*[np.arange(n) for n in shape], indexing="ij")
    one = np.ones(shape, dtype=dtype)
    coeffs = {(0,) * ndim: np.zeros(shape, dtype=dtype)}
    for i in range(ndim):
        h2 = dw[i] ** 2
        lo = iw[i] == 0
        hi = iw[i] == shape[i] - 1
        sm = tuple(-1 if j == i else 0 for j in range(ndim))
        sp = tuple(1 if j == i else 0 for j in range(ndim))
        # qm = where(lo, (qwp - 6 q)/3, qwm); qp = where(hi, (qwm - 6 q)/3, qwp)
        c_m = np.where(lo, 0, one) + np.where(hi, one / 3, 0)
        c_p = np.where(hi, 0, one) + np.where(lo, one / 3, 0)
        c_0 = -2 * one + np.where(lo, -2 * one, 0) + np.where(hi, -2 * one, 0)
        coeffs[sm] = c_m / h2
        coeffs[sp] = c_p / h2
        coeffs[(0,) * ndim] = coeffs[(0,) * ndim] + c_0 / h2
    return coeffs


def poisson_adjoint(fbar, dw):
    """J^T fbar for the Poisson operator: sum_s roll(c_s * fbar, +s)."""
    coeffs = poisson_jac_coeffs(fbar.shape, dw, dtype=fbar.dtype)
    ndim = fbar.ndim
    g = np.zeros_like(fbar)
    for shift, c in coeffs.items():
        g = g + np.roll(c * fbar, shift, range(ndim))
    return g


def poisson_loss_grad(terms, rhs, dw, loc=None):
    """eval_loss_grad for the Poisson problem with a multigrid unknown
    (core.py:1082-1104): loss = mean(fu**2); grads wrt every level array."""
    loc = loc or "c" * rhs.ndim
    u = multigrid_to_regular(terms, loc) if len(terms) > 1 else terms[0]
    fu = poisson_residual(u, rhs, dw)
    loss = np.mean(np.square(fu))
    gu = poisson_adjoint(2 * fu / fu.size, dw)
    grads = multigrid_to_regular_adj(gu, [t.shape for t in terms], loc)
    return loss, grads, fu


# --------------------------------------------------------------------------
# Optimizers (optimizer.py:256-341)
# --------------------------------------------------------------------------


def adam_step(x, m, v, grads, local_epoch, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-7, dtype=np.float64):
    """optimizer.py:311-319 (Keras convention: epsilon outside the sqrt)."""
    dtype = np.dtype(dtype).type
    lr, beta_1, beta_2 = dtype(lr), dtype(beta_1), dtype(beta_2)
    local_epoch = dtype(local_epoch)
    beta_1_power = beta_1**local_epoch
    beta_2_power = beta_2**local_epoch
    alpha = lr * np.sqrt(1 - beta_2_power) / (1 - beta_1_power)
    m = [m + (g - m) * (1 - beta_1) for m, g in zip(m, grads)]
    v = [v + (np.square(g) - v) * (1 - beta_2) for v, g in zip(v, grads)]
    x = [x - (m * alpha) / (np.sqrt(v) + epsilon) for x, m, v in zip(x, m, v)]
    return x, m, v


def adam_run(x0, loss_grad, epochs, lr, callback=None, dtype=np.float64, **kw):
    """optimizer.py:328-336."""
    x = [np.array(e, dtype=dtype) for e in x0]
    m = [np.zeros_like(e) for e in x]
    v = [np.zeros_like(e) for e in x]
    losses = []
    for epoch in range(1, epochs + 1):
        loss, grads = loss_grad(x)
        losses.append(loss)
        x, m, v = adam_step(x, m, v, grads, epoch, lr, dtype=dtype, **kw)
        if callback is not None:
            callback(x, epoch, loss)
    return x, losses


def gd_run(x0, loss_grad, epochs, lr):
    """optimizer.py:262-277."""
    x = [np.array(e) for e in x0]
    losses = []
    for _ in range(epochs):
        loss, grads = loss_grad(x)
        losses.append(loss)
        for i in range(len(x)):
            x[i] = x[i] - grads[i] * lr
    return x, losses


def lbfgsb_run(x0, loss_grad, epochs, m=50, maxls=50, pgtol=1e-16, factr=0):
    """optimizer.py:54-117: SciPy fmin_l_bfgs_b on the flat float64 vector.
    Third-party arithmetic (scipy pinned 1.16.2 in uv.lock:1499; the container and
    the GPU image ship 1.15.3): the oracle calls the same SciPy routine the
    reference calls."""
    from scipy import optimize

    shapes = [a.shape for a in x0]
    sizes = [int(np.prod(s)) for s in shapes]
    losses = []
    iter_losses = []

    def unflat(x):
        return [s.reshape(shp) for s, shp in zip(np.split(x, np.cumsum(sizes)[:-1]), shapes)]

    def func(x):
        loss, grads = loss_grad(unflat(np.array(x, dtype=np.float64)))
        losses.append(float(loss))
        return np.float64(loss), np.concatenate([np.asarray(g, dtype=np.float64).ravel() for g in grads])

    def cb(x):
        iter_losses.append(losses[-1])

    x0f = np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in x0])
    x, f, info = optimize.fmin_l_bfgs_b(
        func=func, x0=x0f, maxiter=epochs, pgtol=pgtol, m=m, maxls=maxls, factr=factr, maxfun=np.inf, callback=cb
    )
    return unflat(x), losses, iter_losses, info


# --------------------------------------------------------------------------
# Newton: linearize (core.py:1113-1217) + normal equations (linsolver.py:17-26)
# --------------------------------------------------------------------------


def field_to_matrix(coeff, shift, field_shape_, field_loc, loc, offset, size_all):
    """core.py:1144-1171: rows = arange(n_out); cols = offset + roll(pad/trim(arange))."""
    import scipy.sparse as sp

    size = int(np.prod(field_shape_))
    cols = offset + np.arange(size).reshape(field_shape_)
    cols = field_access(cols, field_loc, shift, loc)
    rows = np.arange(coeff.size)
    return sp.csr_array((coeff.ravel(), (rows, cols.ravel())), shape=(coeff.size, size_all))


def poisson_linearize(u, rhs, dw):
    """vector, matrix of `Problem.linearize` for the Poisson operator on a plain Field."""
    import scipy.sparse as sp

    coeffs = poisson_jac_coeffs(u.shape, dw, dtype=u.dtype)
    loc = "c" * u.ndim
    matrix = sp.csr_array((u.size, u.size), dtype=u.dtype)
    for shift, c in coeffs.items():
        matrix = matrix + field_to_matrix(c, shift, u.shape, loc, loc, 0, u.size)
    return poisson_residual(u, rhs, dw).ravel(), matrix


def solve_normal_direct(matrix, rhs, damp=0.0, dampdiag=0.0):
    """linsolver.py:17-26: (M^T M + damp^2 I + dampdiag^2 diag) x = M^T rhs, SuperLU MMD_ATA."""
    import scipy.sparse
    import scipy.sparse.linalg

    a = matrix.T.dot(matrix).tocsr()
    if damp:
        a = a + damp**2 * scipy.sparse.eye(matrix.shape[1], format="csr")
    if dampdiag:
        a = a + dampdiag**2 * scipy.sparse.diags(a.diagonal())
    return scipy.sparse.linalg.spsolve(a.tocsc(), matrix.T.dot(rhs), permc_spec="MMD_ATA")
