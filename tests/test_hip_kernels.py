"""GPU parity tests: every HIP kernel, called through the C-ABI, against the oracle and
the golden fixtures generated from the reference.  f64 tolerances are a few ulp-scale
(1e-13 relative to the array's max); the north_star loss-trajectory tolerance is 1e-6."""

import numpy as np
import pytest
import torch
from conftest import load_golden

from oracle import odil_np as onp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def to(x, dev):
    return torch.tensor(np.ascontiguousarray(x), device=dev)


def rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b))) / max(1.0, float(np.max(np.abs(b)))) if b.size else 0.0


INTERP_CASES = [str(c) for c in load_golden("interp")["cases"]]


@pytest.mark.parametrize("loc", INTERP_CASES)
def test_interp_and_adjoint_vs_golden(dev, loc):
    from odil_amd import ops

    g = load_golden("interp")
    u = g[f"{loc}/u"]
    fine = ops.interp_add(to(u, dev), loc)
    # same summation order as the reference 'stack' path and no FMA contraction: bit-exact
    assert np.array_equal(fine.cpu().numpy(), g[f"{loc}/fine"])
    if f"{loc}/fine2" in g:
        assert np.array_equal(ops.interp_to_finer(to(u, dev), loc, depth=2).cpu().numpy(), g[f"{loc}/fine2"])
    gu = ops.interp_adj(to(g[f"{loc}/gfine"], dev), loc, u.shape)
    assert rel(gu, g[f"{loc}/gu"]) < 1e-14


@pytest.mark.parametrize("loc", [str(c) for c in load_golden("interp_conv")["cases"]])
def test_interp_conv_reference_values(dev, loc):
    """`interp_to_finer(method="conv")` AS THE REFERENCE COMPUTES IT (core.py:645-667; fixtures from its own function):
    the public function with method="conv" -- the tracer workload's default -- and the cotangent kernel."""
    import odil_amd as odil
    from odil_amd import ops

    g = load_golden("interp_conv")
    u = g[f"{loc}/u"]
    fine = odil.core.interp_to_finer(to(u, dev), loc=loc, method="conv", mod=odil.runtime.get_mod())
    assert rel(fine, g[f"{loc}/fine"]) < 1e-15
    if f"{loc}/fine2" in g:
        fine2 = odil.core.interp_to_finer(to(u, dev), loc=loc, method="conv", mod=odil.runtime.get_mod(), depth=2)
        assert rel(fine2, g[f"{loc}/fine2"]) < 2e-15
    assert rel(ops.interp_adj(to(g[f"{loc}/gfine"], dev), loc, u.shape), g[f"{loc}/gu"]) < 1e-14


@pytest.mark.parametrize("loc", [str(c) for c in load_golden("restrict")["cases"]])
def test_restrict_reference_values(dev, loc):
    """`restrict_to_coarser` AS THE REFERENCE COMPUTES IT (core.py:703-755 + backend.py:112-126; fixtures from its own
    function, '.' axes subsampled by the integer stride): values, depth 2, cotangent."""
    import odil_amd as odil
    from odil_amd import ops

    g = load_golden("restrict")
    u = g[f"{loc}/u"]
    mod = odil.runtime.get_mod()
    assert rel(odil.core.restrict_to_coarser(to(u, dev), loc=loc, mod=mod), g[f"{loc}/coarse"]) < 1e-15
    assert rel(odil.core.restrict_to_coarser(to(u, dev), loc=loc, mod=mod, depth=2), g[f"{loc}/coarse2"]) < 2e-15
    assert rel(ops.restrict_adj(to(g[f"{loc}/gcoarse"], dev), loc, u.shape), g[f"{loc}/gu"]) < 1e-15


@pytest.mark.parametrize(
    "loc,shape", [("ccc", (5, 6, 7)), ("ncc", (5, 4, 6)), ("cc", (33, 130)), ("c", (700,)), ("cn", (9, 300)),
                  ("ncc", (9, 16, 20)), ("ncc", (67, 8, 12)), ("ncc", (3, 6, 4)), ("ncc", (4, 2, 2)),
                  ("nccc", (3, 4, 6, 8)), ("nccc", (6, 9, 4, 6)), (".ccc", (3, 5, 4, 6)), ("nccc", (2, 4, 2, 2))]
)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_interp_random_vs_oracle(dev, loc, shape, dtype):
    from odil_amd import ops

    rng = np.random.default_rng(1)
    u = rng.standard_normal(shape).astype(dtype)
    add = rng.standard_normal(onp.fine_shape(shape, loc)).astype(dtype)
    out = ops.interp_add(to(u, dev), loc, add=to(add, dev), coarse_scale=0.5, add_scale=2.0)
    ref = dtype(2.0) * add + onp.interp_to_finer(dtype(0.5) * u, loc)
    assert np.array_equal(out.cpu().numpy(), ref)
    gf = rng.standard_normal(onp.fine_shape(shape, loc)).astype(dtype)
    gu, gs = ops.interp_adj(to(gf, dev), loc, shape, scale=3.0)
    refg = onp.interp_to_finer_adj(gf, loc, shape)
    tol = 1e-13 if dtype == np.float64 else 2e-6
    assert rel(gu, refg) < tol
    assert rel(gs, dtype(3.0) * refg) < tol


def test_interp_exact_on_linear_functions(dev):
    """reference tests/test_mg_interp.py:11-32 through the HIP kernel."""
    from odil_amd import ops

    for ndim in [1, 2, 3, 4]:
        for loc in {s[:ndim] for s in ["cccc", "nnnn", "cnnn", "nccc"]}:
            cshapeh = tuple(3 + np.arange(ndim))
            cshape = tuple(2 * np.array(cshapeh))

            def func(xx):
                return sum(x * np.sqrt(i + 1) for i, x in enumerate(xx))

            u = func(onp.points(cshape, loc))
            uh = func(onp.points(cshapeh, loc))
            ui = ops.interp_add(to(uh, dev), loc).cpu().numpy()
            assert np.max(np.abs(ui - u)) <= 100 * np.finfo(np.float64).eps


def test_restrict_exact_on_linear_functions_with_jumps(dev):
    """reference tests/test_mg_restrict.py:11-41 through the HIP kernel, + oracle parity."""
    from odil_amd import ops

    for ndim in [1, 2, 3, 4]:
        for loc in {s[:ndim] for s in ["cccc", "nnnn", "cnnn", "nccc"]}:
            cshapeh = tuple(3 + np.arange(ndim))
            cshape = tuple(2 * np.array(cshapeh))

            def func(xx):
                res = np.zeros_like(xx[0])
                for i in range(len(xx)):
                    res += xx[i] * (i + 1)
                    res += np.where(xx[i] == 0, 10.0, 0.0)
                    res += np.where(xx[i] == 1, 10.0, 0.0)
                return res

            u = func(onp.points(cshape, loc))
            uh = func(onp.points(cshapeh, loc))
            uhr = ops.restrict_to_coarser(to(u, dev), loc).cpu().numpy()
            assert np.max(np.abs(uhr - uh)) <= 100 * np.finfo(np.float64).eps, (ndim, loc)
    rng = np.random.default_rng(5)
    for loc, shape in [("cc", (8, 12)), ("nn", (9, 13)), ("c.n", (6, 5, 7)), ("ccc", (8, 4, 6))]:
        u = rng.standard_normal(shape)
        assert rel(ops.restrict_to_coarser(to(u, dev), loc), onp.restrict_to_coarser(u, loc)) < 1e-14


MG_CASES = [str(c) for c in load_golden("mg")["cases"]]


@pytest.mark.parametrize("name", MG_CASES)
def test_mg_synth_and_adjoint_vs_golden(dev, name):
    from odil_amd import ops

    g = load_golden("mg")
    nlvl = int(g[f"{name}/nlvl"])
    loc = str(g[f"{name}/loc"])
    axes = [bool(a) for a in g[f"{name}/axes"]]
    factors = [float(f) for f in g[f"{name}/factors"]]
    iloc = onp.mg_loc(loc, axes)
    terms = [to(g[f"{name}/w{i}"], dev) for i in range(nlvl)]
    u = ops.mg_synth(terms, iloc, factors=factors)
    assert np.array_equal(u.cpu().numpy(), g[f"{name}/u"])
    grads = ops.mg_synth_adj(to(g[f"{name}/gu"], dev), [t.shape for t in terms], iloc, factors=factors)
    for i in range(nlvl):
        assert rel(grads[i], g[f"{name}/g{i}"]) < 1e-14


def test_field_access_vs_golden(dev):
    from odil_amd import ops

    g = load_golden("field_access")
    for name in g["cases"]:
        floc, loc = str(g[f"{name}/field_loc"]), str(g[f"{name}/loc"])
        shift = tuple(int(s) for s in g[f"{name}/shift"])
        a = g[f"{name}/a"]
        out = ops.field_gather(to(a, dev), floc, shift, loc)
        assert np.array_equal(out.cpu().numpy(), g[f"{name}/out"])
        ga = ops.field_scatter(to(g[f"{name}/g"], dev), a.shape, floc, shift, loc)
        assert np.array_equal(ga.cpu().numpy(), g[f"{name}/ga"])


@pytest.mark.parametrize(
    "name", ["poisson_1d_N256", "poisson_2d_N32", "poisson_3d_N16", "poisson_2d_N8", "poisson_3d_N8"]
)
def test_poisson_loss_grad_vs_golden(dev, name):
    from odil_amd import ops

    g = load_golden(name)
    ndim, N, nlvl = int(g["ndim"]), int(g["N"]), int(g["nlvl"])
    cshape = (N,) * ndim
    dw = onp.step(cshape)
    h2 = [d**2 for d in dw]
    loc = "c" * ndim
    terms = [to(g[f"rand/w{i}"], dev) for i in range(nlvl)]
    rhs = to(g["rhs"], dev)
    u = ops.mg_synth(terms, loc)
    fu, loss = ops.poisson_residual(u, rhs, h2)
    assert rel(fu, g["rand/fu"]) < 1e-14
    assert abs(float(loss) - float(g["rand/loss"])) <= 1e-13 * float(g["rand/loss"])
    gu = ops.poisson_adjoint(fu, h2, 2.0 / fu.numel())
    grads = ops.mg_synth_adj(gu, [t.shape for t in terms], loc)
    for i in range(nlvl):
        assert rel(grads[i], g[f"rand/g{i}"]) < 1e-13
    # loss-only call
    _, loss2 = ops.poisson_residual(u, rhs, h2, want_fu=False)
    assert float(loss2) == float(loss)


@pytest.mark.parametrize("shape", [(7,), (5, 9), (3, 4, 5), (2, 2, 2), (6, 1030)])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_poisson_ragged_shapes_vs_oracle(dev, shape, dtype):
    from odil_amd import ops

    rng = np.random.default_rng(7)
    dw = onp.step(shape, dtype=dtype)
    h2 = [d**2 for d in dw]
    u = rng.standard_normal(shape).astype(dtype)
    rhs = rng.standard_normal(shape).astype(dtype)
    fu, loss = ops.poisson_residual(to(u, dev), to(rhs, dev), h2)
    fref = onp.poisson_residual(u, rhs, dw)
    tol = 1e-14 if dtype == np.float64 else 1e-6
    assert rel(fu, fref) < tol
    # same operation order, no FMA contraction, exact x/3: the residual is BIT-IDENTICAL to NumPy's
    assert np.array_equal(fu.cpu().numpy(), fref), "residual differs from the oracle in the last bits"
    lref = np.mean(np.square(fref.astype(np.float64)))
    assert abs(float(loss) - lref) <= (1e-13 if dtype == np.float64 else 1e-6) * lref
    fb = rng.standard_normal(shape).astype(dtype)
    gu = ops.poisson_adjoint(to(fb, dev), h2, 1.0)
    assert rel(gu, onp.poisson_adjoint(fb, dw)) < (1e-13 if dtype == np.float64 else 1e-5)
    coeffs = ops.poisson_jac_coeffs(shape, h2, to(u, dev).dtype, dev).cpu().numpy()
    ref = onp.poisson_jac_coeffs(shape, dw, dtype=dtype)
    ndim = len(shape)
    order = [(0,) * ndim]
    for i in range(ndim):
        order += [tuple(-1 if j == i else 0 for j in range(ndim)), tuple(1 if j == i else 0 for j in range(ndim))]
    for k, s in enumerate(order):
        assert rel(coeffs[k], ref[s]) < tol


def test_mean_reduce(dev):
    from odil_amd import ops

    rng = np.random.default_rng(3)
    for n in [1, 63, 1000, 100003]:
        x = rng.standard_normal(n)
        assert abs(float(ops.mean_reduce(to(x, dev))) - np.mean(x**2)) <= 1e-14 * np.mean(x**2)
        assert abs(float(ops.mean_reduce(to(x, dev), square=False)) - np.mean(x)) <= 1e-14
    x32 = rng.standard_normal(5000).astype(np.float32)
    assert abs(float(ops.mean_reduce(to(x32, dev))) - np.mean(x32.astype(np.float64) ** 2)) < 1e-6
    # determinism
    x = to(rng.standard_normal(1 << 20), dev)
    assert float(ops.mean_reduce(x)) == float(ops.mean_reduce(x))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_adam_step_vs_oracle(dev, dtype):
    from odil_amd import ops

    rng = np.random.default_rng(11)
    n = 10007
    x, g = rng.standard_normal(n).astype(dtype), rng.standard_normal(n).astype(dtype)
    m, v = (rng.standard_normal(n) * 0.1).astype(dtype), (rng.random(n) * 0.1).astype(dtype)
    t = dtype
    lr, b1, b2, eps = t(0.005), t(0.9), t(0.999), 1e-7
    epoch = t(3)
    alpha = lr * np.sqrt(1 - b2**epoch) / (1 - b1**epoch)
    xr, mr, vr = onp.adam_step([x], [m], [v], [g], 3, 0.005, dtype=dtype)
    tx, tm, tv, tg = to(x, dev), to(m, dev), to(v, dev), to(g, dev)
    ops.adam_step(tx, tm, tv, tg, alpha, 1 - b1, 1 - b2, eps)
    if dtype == np.float64:
        assert np.array_equal(tm.cpu().numpy(), mr[0]) and np.array_equal(tv.cpu().numpy(), vr[0])
        assert rel(tx, xr[0]) < 1e-15
    else:
        assert rel(tm, mr[0]) < 1e-6 and rel(tv, vr[0]) < 1e-6 and rel(tx, xr[0]) < 1e-6
    # unaligned views take the scalar path
    ops.adam_step(tx[1:], tm[1:], tv[1:], tg[1:], alpha, 1 - b1, 1 - b2, eps)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("geom", [(5, 3 * 64, 0, 64), (5, 3 * 64, 2 * 64, 64), (7, 100, 3, 41), (1, 4096, 1024, 2048)])
def test_adam_step_pieces_equals_the_update_of_the_slices(dev, dtype, geom):
    """odil_adam_step_pieces (planes next to a slab interface of an array whose sharded axis is not the leading one):
    the listed pieces get exactly odil_adam_step's update, everything else is untouched."""
    from odil_amd import ops

    pieces, stride, offset, count = geom
    n = pieces * stride + 17
    rng = np.random.default_rng(5)
    mk = lambda scale=1.0: to((rng.standard_normal(n) * scale).astype(dtype), dev)
    x, m, g = mk(), mk(0.1), mk()
    v = to((rng.random(n) * 0.1).astype(dtype), dev)
    want = [t.clone() for t in (x, m, v)]
    hyper = (1e-3, 0.1, 0.001, 1e-7)
    for o in range(pieces):
        sl = slice(o * stride + offset, o * stride + offset + count)
        xs, ms, vs = (t[sl].clone() for t in want)
        ops.adam_step(xs, ms, vs, g[sl].clone(), *hyper)
        for t, u in zip(want, (xs, ms, vs)):
            t[sl] = u
    ops.adam_step_pieces(x, m, v, g, pieces, stride, offset, count, *hyper)
    for got, ref in zip((x, m, v), want):
        assert torch.equal(got, ref)


def test_vector_ops(dev):
    from odil_amd import ops

    rng = np.random.default_rng(13)
    n, k = 5003, 7
    a, b = rng.standard_normal((k, n)), rng.standard_normal(n)
    d = ops.dots(to(a, dev), to(b, dev)).cpu().numpy()
    assert np.max(np.abs(d - a @ b)) < 1e-11
    b2, b3 = rng.standard_normal(n), rng.standard_normal(n)
    d3 = ops.dots3(to(a, dev), [to(b, dev), to(b2, dev), to(b3, dev)]).cpu().numpy()
    assert np.max(np.abs(d3 - np.stack([a @ b, a @ b2, a @ b3]))) < 1e-11
    d2 = ops.dots3(to(a, dev), [to(b, dev), to(b2, dev)]).cpu().numpy()
    assert np.array_equal(d2[:2], d3[:2])
    # long vectors, L-BFGS sized history: the variant that reads the right-hand vectors once
    for dtype, tol in [(np.float64, 1e-10), (np.float32, 1e-3)]:
        n2, k2 = 70004, 50
        big = rng.standard_normal((k2, n2)).astype(dtype)
        rhs = [rng.standard_normal(n2).astype(dtype) for _ in range(3)]
        got = ops.dots3(to(big, dev), [to(v, dev) for v in rhs]).cpu().numpy()
        want = np.stack([big.astype(np.float64) @ v.astype(np.float64) for v in rhs])
        assert np.max(np.abs(got - want)) < tol * np.sqrt(n2)
        got2 = ops.dots3(to(big, dev), [to(rhs[0], dev), to(rhs[1], dev)]).cpu().numpy()
        assert np.array_equal(got2[:2], got[:2]) and np.all(got2[2] == 0)
        again = ops.dots3(to(big, dev), [to(v, dev) for v in rhs]).cpu().numpy()
        assert np.array_equal(again, got)  # deterministic
        # interleaved S / Y history of L-BFGS (m = 50): 100 strided rows in one launch
        wide = rng.standard_normal((100, n2)).astype(dtype)
        tw = to(wide, dev)
        got = ops.dots3(tw, [to(v, dev) for v in rhs]).cpu().numpy()
        want = np.stack([wide.astype(np.float64) @ v.astype(np.float64) for v in rhs])
        assert np.max(np.abs(got - want)) < tol * np.sqrt(n2)
        odd = ops.dots3(tw[1::2], [to(v, dev) for v in rhs]).cpu().numpy()  # row stride 2 n
        assert np.max(np.abs(odd - want[:, 1::2])) < tol * np.sqrt(n2)
        # what the line search reads after an evaluation
        for nn in [1, 777, n2]:
            gv, dv = rng.standard_normal(nn).astype(dtype), rng.standard_normal(nn).astype(dtype)
            out = torch.zeros(3, dtype=to(gv, dev).dtype, device=dev)
            ops.lbfgs_probe(to(gv, dev), to(dv, dev), out)
            g64, d64 = gv.astype(np.float64), dv.astype(np.float64)
            want = np.array([g64 @ d64, g64 @ g64, np.max(np.abs(g64))])
            assert np.max(np.abs(out.cpu().numpy() - want)) < tol * np.sqrt(nn) and float(out[2]) == want[2]
    y = rng.standard_normal(n)
    coef = rng.standard_normal(k)
    ty = to(y, dev)
    ops.lincomb(ty, 0.5, to(a, dev), to(coef, dev))
    assert rel(ty, 0.5 * y + coef @ a) < 1e-13
    ty = to(y, dev)
    ops.axpy(ty, to(b, dev), -0.25)
    assert rel(ty, y - 0.25 * b) < 1e-15


def test_adam_trajectory_poisson_vs_golden(dev):
    """Loss trajectory of the reference AdamNativeOptimizer on Poisson MG, tolerance 1e-6 rel."""
    from odil_amd import ops

    for name in ["poisson_1d_N256", "poisson_2d_N32", "poisson_3d_N16"]:
        g = load_golden(name)
        ndim, N, nlvl = int(g["ndim"]), int(g["N"]), int(g["nlvl"])
        cshape = (N,) * ndim
        loc = "c" * ndim
        dw = onp.step(cshape)
        h2 = [d**2 for d in dw]
        shapes = onp.mg_cshapes(cshape)
        sizes = [int(np.prod(s)) for s in shapes]
        flat = torch.zeros(sum(sizes), dtype=torch.float64, device=dev)
        m, v, gflat = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros_like(flat)
        views = [t.view(s) for t, s in zip(flat.split(sizes), shapes)]
        gviews = [t.view(s) for t, s in zip(gflat.split(sizes), shapes)]
        rhs = to(g["rhs"], dev)
        losses = []
        ref = g["adam/losses"]
        lr, b1, b2 = np.float64(0.005), np.float64(0.9), np.float64(0.999)
        for epoch in range(1, len(ref) + 1):
            u = ops.mg_synth(views, loc)
            fu, loss = ops.poisson_residual(u, rhs, h2)
            ops.poisson_adjoint(fu, h2, 2.0 / fu.numel(), out=gviews[0])
            ops.mg_synth_adj(gviews[0], shapes, loc, grads=gviews)
            losses.append(float(loss))
            e = np.float64(epoch)
            alpha = lr * np.sqrt(1 - b2**e) / (1 - b1**e)
            ops.adam_step(flat, m, v, gflat, alpha, 1 - b1, 1 - b2, 1e-7)
        assert np.max(np.abs(np.array(losses) - ref) / ref) < 1e-6, name
        for i in range(nlvl):
            assert rel(views[i], g[f"adam/w{i}"]) < 1e-7


@pytest.mark.parametrize("name", ["newton_poisson_1d_N8", "newton_poisson_2d_N6", "newton_poisson_3d_N4"])
def test_newton_blocks_vs_golden(dev, name):
    import scipy.sparse as sp

    from odil_amd import ops

    g = load_golden(name)
    u0 = g["u0"]
    shape = u0.shape
    ndim = len(shape)
    dw = onp.step(shape)
    h2 = [d**2 for d in dw]
    coeffs = ops.poisson_jac_coeffs(shape, h2, torch.float64, dev)
    shifts = [(0,) * ndim]
    for i in range(ndim):
        shifts += [tuple(-1 if j == i else 0 for j in range(ndim)), tuple(1 if j == i else 0 for j in range(ndim))]
    for k, s in enumerate(shifts):
        assert rel(coeffs[k], g["coeff/" + ",".join(str(v) for v in s)]) < 1e-14
    indptr, indices, data = [t.cpu().numpy() for t in ops.csr_assemble(coeffs, shifts, shape)]
    M = sp.csr_array((data, indices, indptr), shape=(u0.size, u0.size)).toarray()
    assert rel(M, g["matrix"]) < 1e-14
    x = np.random.default_rng(0).standard_normal(shape)
    assert rel(ops.stencil_apply(coeffs, shifts, to(x, dev)), (g["matrix"] @ x.ravel()).reshape(shape)) < 1e-13
    assert (
        rel(ops.stencil_apply(coeffs, shifts, to(x, dev), transpose=True), (g["matrix"].T @ x.ravel()).reshape(shape))
        < 1e-13
    )


def test_large_roundtrip_properties(dev):
    """Size-independent properties at a large size (no oracle): <P x, y> == <x, P^T y>,
    linearity of the residual and <J u, f> == <u, J^T f>."""
    from odil_amd import ops

    torch.manual_seed(0)
    cs = (64, 64, 64)
    x = torch.randn(cs, dtype=torch.float64, device=dev)
    y = torch.randn(tuple(2 * s for s in cs), dtype=torch.float64, device=dev)
    lhs = float((ops.interp_add(x, "ccc") * y).sum())
    rhs = float((x * ops.interp_adj(y, "ccc", cs)).sum())
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), 1.0)
    shape = (128, 128, 128)
    h2 = [(1.0 / s) ** 2 for s in shape]
    u1, u2 = [torch.randn(shape, dtype=torch.float64, device=dev) for _ in range(2)]
    zero = torch.zeros_like(u1)
    f1, _ = ops.poisson_residual(u1, zero, h2)
    f2, _ = ops.poisson_residual(u2, zero, h2)
    f12, _ = ops.poisson_residual(u1 + u2, zero, h2)
    assert float((f12 - f1 - f2).abs().max()) <= 1e-9 * float(f12.abs().max())
    g2 = ops.poisson_adjoint(f2, h2, 1.0)
    a, b = float((f1 * f2).sum()), float((u1 * g2).sum())
    assert abs(a - b) <= 1e-10 * abs(a)


@pytest.mark.parametrize("fuse", [0, 1])
def test_device_resident_adam_driver_vs_golden(dev, fuse, monkeypatch):
    """The bench driver (packed state, optional Adam-in-adjoint fusion) follows the reference's
    AdamNativeOptimizer trajectory (1e-6 rel) and both variants agree bit for bit."""
    from odil_amd.poisson_path import PoissonMultigridAdam

    monkeypatch.setenv("ODIL_FUSE_ADAM0", str(fuse))
    finals = []
    for name in ["poisson_2d_N32", "poisson_3d_N16"]:
        g = load_golden(name)
        ref = g["adam/losses"]
        run = PoissonMultigridAdam(int(g["ndim"]), int(g["N"]), device=dev, rhs=to(g["rhs"], dev))
        assert run.fuse_adam0 == bool(fuse)
        losses = []
        for _ in range(len(ref)):
            run.epoch()
            losses.append(run.last_loss())
        assert np.max(np.abs(np.array(losses) - ref) / ref) < 1e-6, name
        for i in range(int(g["nlvl"])):
            assert rel(run.w[i], g[f"adam/w{i}"]) < 1e-7
        finals.append(run.x.clone())
    test_device_resident_adam_driver_vs_golden.results = getattr(
        test_device_resident_adam_driver_vs_golden, "results", {})
    test_device_resident_adam_driver_vs_golden.results[fuse] = finals
    res = test_device_resident_adam_driver_vs_golden.results
    if 0 in res and 1 in res:
        for a, b in zip(res[0], res[1]):
            assert torch.equal(a, b)


def test_restrict_adjoint_vs_oracle(dev):
    from odil_amd import ops

    rng = np.random.default_rng(21)
    for loc, shape in [("c", (8,)), ("n", (9,)), ("cc", (6, 8)), ("nn", (7, 9)), ("cn", (6, 9)), ("c.n", (4, 5, 7)),
                       ("ccc", (4, 6, 8)), ("nnn", (5, 7, 9)), ("nccc", (5, 4, 6, 4))]:
        cshape = ops.coarse_shape(shape, loc)
        y = rng.standard_normal(cshape)
        got = ops.restrict_adj(to(y, dev), loc, shape)
        assert rel(got, onp.restrict_to_coarser_adj(y, loc, shape)) < 1e-14, loc


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("cshape", [(4, 6, 10), (2, 2, 2), (9, 5, 7), (32, 40, 64), (5, 130, 3)])
def test_poisson_residual_with_fused_prolongation_is_bit_identical(dev, dtype, cshape):
    """odil_poisson_residual_synth: u = w0 + P coarse formed in registers; fu must equal, bit for bit,
    odil_interp_add followed by odil_poisson_residual (same arithmetic, same order), the loss to round-off."""
    from odil_amd import ops

    rng = np.random.default_rng(12)
    fshape = tuple(2 * s for s in cshape)
    coarse = to(rng.standard_normal(cshape).astype(dtype), dev)
    w0 = to(rng.standard_normal(fshape).astype(dtype), dev)
    rhs = to(rng.standard_normal(fshape).astype(dtype), dev)
    for h2 in ([0.25**2, 0.125**2, 0.5**2], [0.1**2, 0.3**2, 0.07**2]):  # power-of-two steps multiply, others divide
        h2 = [dtype(v) for v in h2]
        u = ops.interp_add(coarse, "ccc", add=w0)
        fu_ref, loss_ref = ops.poisson_residual(u, rhs, h2)
        fu, loss = ops.poisson_residual_synth(coarse, w0, rhs, h2)
        assert torch.equal(fu, fu_ref)
        assert abs(float(loss) - float(loss_ref)) <= (1e-13 if dtype == np.float64 else 1e-5) * float(loss_ref)
        want = onp.poisson_residual(u.cpu().numpy().astype(np.float64), rhs.cpu().numpy().astype(np.float64),
                                    [float(np.sqrt(v)) for v in h2])
        assert rel(fu, want) < (1e-12 if dtype == np.float64 else 1e-4)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("cshape", [(4, 6, 10), (2, 2, 2), (9, 5, 7), (32, 40, 64), (5, 130, 3)])
def test_jacobi_sweep_with_fused_prolongation_is_bit_identical(dev, dtype, cshape):
    """odil_poisson_jacobi_synth: the sweep of x + P coarse with the prolongation in registers equals, bit for bit,
    odil_interp_add followed by odil_poisson_jacobi (walls on every side included: the diagonal changes there)."""
    from odil_amd import ops

    rng = np.random.default_rng(14)
    fshape = tuple(2 * s for s in cshape)
    coarse = to(rng.standard_normal(cshape).astype(dtype), dev)
    x = to(rng.standard_normal(fshape).astype(dtype), dev)
    b = to(rng.standard_normal(fshape).astype(dtype), dev)
    assert ops.jacobi_synth_supported(fshape, x.dtype)
    for h2 in ([0.25**2, 0.125**2, 0.5**2], [0.1**2, 0.3**2, 0.07**2]):
        h2 = [dtype(v) for v in h2]
        u = ops.interp_add(coarse, "ccc", add=x)
        want = ops.poisson_jacobi(u, b, h2, 0.8, torch.empty_like(u))
        got = ops.poisson_jacobi_synth(coarse, x, b, h2, 0.8, torch.empty_like(x))
        assert torch.equal(got, want)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(12,), (6, 10), (8, 6, 10), (5, 7, 9)])
def test_poisson_jacobi_sweep(dev, dtype, shape):
    """odil_poisson_jacobi == x - omega (A x - b) / diag(A) with A, diag from the residual / Jacobian kernels."""
    from odil_amd import ops

    rng = np.random.default_rng(31)
    h2 = [dtype(v) for v in [0.25**2, 0.1**2, 0.3**2][: len(shape)]]
    x = to(rng.standard_normal(shape).astype(dtype), dev)
    b = to(rng.standard_normal(shape).astype(dtype), dev)
    omega = 0.8
    out = ops.poisson_jacobi(x, b, h2, omega, torch.empty_like(x))
    r, _ = ops.poisson_residual(x, b, h2)
    diag = ops.poisson_jac_coeffs(shape, h2, x.dtype, dev)[0]
    want = x - omega * r / diag
    assert rel(out, want.cpu().numpy()) < (1e-13 if dtype == np.float64 else 1e-5)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(16,), (1024,), (6, 12), (40, 1000), (8, 6, 12), (5, 7, 8), (2, 2, 4), (33, 31, 136),
                                   (20, 45, 520), (70, 16, 256), (3, 100, 4)])
def test_two_jacobi_sweeps_in_one_pass_are_bit_identical(dev, dtype, shape):
    """odil_poisson_jacobi2 (the intermediate iterate kept on the CU) == two calls of odil_poisson_jacobi, bit for
    bit: whole-row and tiled x-windows (halo packs), ragged y-tiles, 1-D / 2-D, chunks of 1 .. all planes, walls on
    every side, steps that multiply (powers of two) and steps that divide."""
    from odil_amd import ops

    rng = np.random.default_rng(41)
    x = to(rng.standard_normal(shape).astype(dtype), dev)
    b = to(rng.standard_normal(shape).astype(dtype), dev)
    assert ops.jacobi2_supported(shape, x.dtype)
    for h2 in ([0.25**2, 0.125**2, 0.5**2], [0.1**2, 0.3**2, 0.07**2]):
        h2 = [dtype(v) for v in h2[: len(shape)]]
        y1 = ops.poisson_jacobi(x, b, h2, 0.9, torch.empty_like(x))
        want = ops.poisson_jacobi(y1, b, h2, 0.6, torch.empty_like(x))
        for zc in (0, 1, 3, 64):
            got = ops.poisson_jacobi2(x, b, h2, 0.9, 0.6, torch.full_like(x, float("nan")), zc_hint=zc)
            assert torch.equal(got, want), (shape, zc)
        # from the ZERO vector (u = NULL: nothing read for the iterate): the bits of the same calls on an array of zeros
        zero = torch.zeros_like(x)
        y1 = ops.poisson_jacobi(zero, b, h2, 0.9, torch.empty_like(x))
        assert torch.equal(ops.poisson_jacobi(None, b, h2, 0.9, torch.full_like(x, float("nan"))), y1), shape
        want = ops.poisson_jacobi(y1, b, h2, 0.6, torch.empty_like(x))
        for zc in (0, 3):
            got = ops.poisson_jacobi2(None, b, h2, 0.9, 0.6, torch.full_like(x, float("nan")), zc_hint=zc)
            assert torch.equal(got, want), (shape, zc, "zero start")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("cshape", [(4, 6, 10), (2, 2, 2), (9, 5, 7), (16, 20, 64), (6, 33, 70), (35, 7, 130), (3, 50, 2)])
def test_correction_and_two_sweeps_in_one_pass_are_bit_identical(dev, dtype, cshape):
    """odil_poisson_jacobi2_synth (x + P coarse formed on the fly, the first sweep kept on the CU) == odil_interp_add
    followed by two calls of odil_poisson_jacobi, bit for bit: walls of the joint ghost rule on every side, whole-row and
    tiled x-windows (two halo packs), ragged y-tiles, chunks of 2 .. all planes."""
    from odil_amd import ops

    rng = np.random.default_rng(43)
    fshape = tuple(2 * s for s in cshape)
    coarse = to(rng.standard_normal(cshape).astype(dtype), dev)
    x = to(rng.standard_normal(fshape).astype(dtype), dev)
    b = to(rng.standard_normal(fshape).astype(dtype), dev)
    for h2 in ([0.25**2, 0.125**2, 0.5**2], [0.1**2, 0.3**2, 0.07**2]):
        h2 = [dtype(v) for v in h2]
        u = ops.interp_add(coarse, "ccc", add=x)
        y1 = ops.poisson_jacobi(u, b, h2, 0.9, torch.empty_like(x))
        want = ops.poisson_jacobi(y1, b, h2, 0.6, torch.empty_like(x))
        for zc in (0, 2, 6, 64):
            got = ops.poisson_jacobi2_synth(coarse, x, b, h2, 0.9, 0.6, torch.full_like(x, float("nan")), zc_hint=zc)
            assert torch.equal(got, want), (cshape, zc, int((got != want).sum()))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(4, 6, 8), (8, 12, 16), (64, 32, 520), (2, 2, 4)])
def test_poisson_residual_restrict(dev, dtype, shape):
    """odil_poisson_residual_restrict == scale * 2^d * restrict(residual) and the same loss, from the
    oracle-checked residual and restriction kernels."""
    from odil_amd import ops

    rng = np.random.default_rng(37)
    h2 = [dtype(v) for v in [0.25**2, 0.1**2, 0.3**2]]
    x = to(rng.standard_normal(shape).astype(dtype), dev)
    b = to(rng.standard_normal(shape).astype(dtype), dev)
    assert ops.residual_restrict_supported(shape, x.dtype)
    out = torch.full(tuple(n // 2 for n in shape), 7.0, dtype=x.dtype, device=dev)
    loss = torch.zeros((), dtype=x.dtype, device=dev)
    ops.poisson_residual_restrict(x, b, h2, -0.125, out, loss)
    r, want_loss = ops.poisson_residual(x, b, h2)
    want = -ops.restrict_to_coarser(r, "ccc")
    tol = 1e-13 if dtype == np.float64 else 1e-5
    assert rel(out, want.cpu().numpy()) < tol
    assert abs(float(loss) - float(want_loss)) < tol * float(want_loss)
    assert not ops.residual_restrict_supported((8, 8), x.dtype) and not ops.residual_restrict_supported((8, 6, 7), x.dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("fine", [(8, 32, 128), (16, 40, 136), (24, 72, 264)])
def test_tiled_transpose_is_bit_identical(dev, dtype, fine, monkeypatch):
    """The LDS-staged P^T of the large 'ccc' levels (k_interp_adj_tile) == the register-window kernel
    bit for bit (whole and partial tiles, both walls in every axis), == the oracle's transpose to rounding,
    with and without the Adam update of the coarse level."""
    from odil_amd import ops

    rng = np.random.default_rng(41)
    shapes = [fine, tuple(n // 2 for n in fine)]
    g = to(rng.standard_normal(fine).astype(dtype), dev)
    monkeypatch.setenv("ODIL_ADJ_TILE", "1")
    tiled = [t.clone() for t in ops.mg_synth_adj(g, shapes, "ccc")]
    monkeypatch.setenv("ODIL_ADJ_TILE", "0")
    plain = [t.clone() for t in ops.mg_synth_adj(g, shapes, "ccc")]
    assert torch.equal(tiled[1], plain[1])
    want = onp.interp_to_finer_adj(g.cpu().numpy().astype(np.float64), "ccc", shapes[1])
    assert rel(tiled[1], want) < (1e-13 if dtype == np.float64 else 2e-6)
    # with Adam of the coarse level inside the launch
    res = {}
    for mode in ["1", "0"]:
        monkeypatch.setenv("ODIL_ADJ_TILE", mode)
        x = [None, to(rng.standard_normal(shapes[1]).astype(dtype) * 0 + 0.5, dev)]
        m = [None, torch.zeros_like(x[1])]
        v = [None, torch.zeros_like(x[1])]
        grads = [g, torch.empty_like(x[1])]
        ops.mg_synth_adj_adam(g, shapes, "ccc", grads, x, m, v, 0.01, 0.1, 0.001, 1e-7)
        res[mode] = (x[1].clone(), m[1].clone(), v[1].clone(), grads[1].clone())
    for a, b in zip(res["1"], res["0"]):
        assert torch.equal(a, b)
    # a leading batch axis: the space part of the 4-D space-time transposes, one volume per grid row
    g4 = to(rng.standard_normal((3,) + tuple(fine)).astype(dtype), dev)
    out = {}
    for mode in ["1", "0"]:
        monkeypatch.setenv("ODIL_ADJ_TILE", mode)
        out[mode] = ops.interp_adj(g4, ".ccc", (3,) + tuple(shapes[1])).clone()
    assert torch.equal(out["1"], out["0"])
    for b in range(3):
        monkeypatch.setenv("ODIL_ADJ_TILE", "1")
        assert torch.equal(out["1"][b], ops.interp_adj(g4[b].contiguous(), "ccc", shapes[1]))
    # slab interfaces along the marched axis (multi-GPU path)
    for cut in [(True, False), (False, True), (True, True)]:
        out = {}
        for mode in ["1", "0"]:
            monkeypatch.setenv("ODIL_ADJ_TILE", mode)
            out[mode] = ops.interp_adj(g, "ccc", shapes[1], cut=cut).clone()
        assert torch.equal(out["1"], out["0"])
        assert rel(out["1"], onp.interp_to_finer_adj(g.cpu().numpy().astype(np.float64), "ccc", shapes[1], cut=cut)) < (
            1e-13 if dtype == np.float64 else 2e-6)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("fine", [(8, 32, 128), (16, 40, 136), (24, 72, 264)])
def test_adjoint_and_transpose_in_one_launch(dev, dtype, fine, monkeypatch):
    """odil_poisson_adjoint_transpose_adam == odil_poisson_adjoint(_adam) followed by the first P^T (+Adam of
    that level), bit for bit: the gradient of the finest level, the coarse gradient, and both updated states.
    (The separate P^T through the kernels that sum in the same order x, y, z as the fused launch: the row-marching
    float kernel sums x, z, y -- test_row_marching_float_transpose.)"""
    from odil_amd import ops

    monkeypatch.setenv("ODIL_ADJ_ROWS", "0")

    rng = np.random.default_rng(43)
    coarse = tuple(n // 2 for n in fine)
    shapes = [fine, coarse]
    h2 = [dtype(v) for v in [0.25**2, 0.1**2, 0.3**2]]
    scale = dtype(2.0 / np.prod(fine))
    fu = to(rng.standard_normal(fine).astype(dtype), dev)
    assert ops.adjoint_transpose_supported(fine)
    # gradients only
    g0 = torch.full(fine, 7.0, dtype=fu.dtype, device=dev)
    g1 = torch.full(coarse, 7.0, dtype=fu.dtype, device=dev)
    ops.poisson_adjoint_transpose(fu, h2, scale, g1, g0=g0)
    g0_ref = ops.poisson_adjoint(fu, h2, scale)
    g1_ref = ops.mg_synth_adj(g0_ref, shapes, "ccc")[1]
    assert torch.equal(g0, g0_ref) and torch.equal(g1, g1_ref)
    # with the Adam update of both levels (and without storing g0)
    def fresh():
        r = np.random.default_rng(47)
        mk = lambda s, pos=False: to((np.abs(r.standard_normal(s)) if pos else r.standard_normal(s)).astype(dtype), dev)
        return [mk(fine), mk(coarse)], [mk(fine), mk(coarse)], [mk(fine, True), mk(coarse, True)]

    x, m, v = fresh()
    g1 = torch.empty(coarse, dtype=fu.dtype, device=dev)
    ops.poisson_adjoint_transpose(fu, h2, scale, g1, g0=None, adam0=(x[0], m[0], v[0]), adam1=(x[1], m[1], v[1]),
                                  alpha=0.01, one_minus_b1=0.1, one_minus_b2=0.001, eps=1e-7)
    xr, mr, vr = fresh()
    gr = [torch.empty(fine, dtype=fu.dtype, device=dev), torch.empty(coarse, dtype=fu.dtype, device=dev)]
    ops.poisson_adjoint_adam(fu, h2, scale, gr[0], xr[0], mr[0], vr[0], 0.01, 0.1, 0.001, 1e-7)
    ops.mg_synth_adj_adam(gr[0], shapes, "ccc", gr, xr, mr, vr, 0.01, 0.1, 0.001, 1e-7)
    assert torch.equal(g1, gr[1])
    for a, b in zip(x + m + v, xr + mr + vr):
        assert torch.equal(a, b)


@pytest.mark.parametrize("loc,cshape", [("ccc", (4, 4, 32)), ("ccc", (5, 7, 32)), ("ccc", (18, 16, 64)), ("ccc", (9, 33, 128)),
                                        ("ccc", (16, 40, 128)), (".ccc", (3, 6, 12, 32)), (".ccc", (5, 18, 20, 128)),
                                        (".ccc", (2, 4, 5, 64)), ("ccc", (64, 64, 128)), ("nccc", (5, 4, 32, 64))])
def test_row_marching_float_transpose(dev, loc, cshape, monkeypatch):
    """k_interp_adj_rows (float volumes with rows of 64 / 128 / 256 fine cells: a lane owns one 16 B pack of a fine row,
    x neighbours by wave shifts, four or six fine planes per lane, rows marched) against the float64 oracle, against
    the kernels it replaces (same terms, other order of summation: a few ulp), with the scaled copy and with the Adam
    update of the coarse level inside the launch; bit-reproducible."""
    from odil_amd import ops

    rng = np.random.default_rng(sum(cshape))
    fshape = ops.fine_shape(cshape, loc)
    g64 = rng.standard_normal(fshape)
    g = to(g64.astype(np.float32), dev)
    ref = onp.interp_to_finer_adj(g64.astype(np.float32).astype(np.float64), loc, cshape)
    monkeypatch.setenv("ODIL_ADJ_ROWS", "0")
    old = ops.interp_adj(g, loc, cshape)
    monkeypatch.setenv("ODIL_ADJ_ROWS", "1")
    new = ops.interp_adj(g, loc, cshape)
    assert rel(new, ref) < 1e-6 and rel(new, old.cpu().numpy()) < 1e-6
    assert torch.equal(new, ops.interp_adj(g, loc, cshape))
    out, scaled = ops.interp_adj(g, loc, cshape, scale=0.5)
    assert torch.equal(out, new) and torch.equal(scaled, 0.5 * new)
    mk = lambda pos=False: to((np.abs(rng.standard_normal(cshape)) if pos else rng.standard_normal(cshape)).astype(np.float32), dev)
    x, m, v = mk(), mk(), mk(True)
    state = [t.clone() for t in (x, m, v)]
    gout = torch.empty(cshape, dtype=torch.float32, device=dev)
    ops.interp_adj_adam(g, loc, cshape, gout, x, m, v, 0.01, 0.1, 0.001, 1e-7)
    assert torch.equal(gout, new)
    xr, mr, vr = state
    ops.adam_step(xr, mr, vr, new, 0.01, 0.1, 0.001, 1e-7)
    for a, b in ((x, xr), (m, mr), (v, vr)):
        assert rel(a, b.cpu().numpy()) < 1e-6


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("loc,full,lo,n", [("nccc", (5, 8, 8, 16), 1, 6), ("nccc", (9, 7, 128, 128), 2, 4), (".ccc", (3, 10, 4, 8), 1, 8),
                                          ("nccc", (3, 5, 4, 4), 1, 3), ("nccc", (33, 6, 16, 32), 0, 4)])
def test_transfers_on_views_with_a_leading_stride(dev, dtype, loc, full, lo, n):
    """odil_interp_add_ld / odil_interp_adj_ld: the coarse operand of P (the result of P^T) as a VIEW of a larger array
    along axis 1 -- a ghost-extended level array of the slab paths without its outer planes.  The marching kernels of
    the 4-D layouts read / write it in place (bit-identical to the contiguous call); layouts they do not serve (third
    case: 3 coarse planes) take the copy inside ops.interp_add / interp_adj.  Nothing outside the view is touched."""
    from odil_amd import ops

    rng = np.random.default_rng(61)
    big = to(rng.standard_normal(full).astype(dtype), dev)
    view = big.narrow(1, lo, n)
    assert not view.is_contiguous()
    cshape = tuple(view.shape)
    add = to(rng.standard_normal(ops.fine_shape(cshape, loc)).astype(dtype), dev)
    want = ops.interp_add(view.contiguous(), loc, add=add, coarse_scale=0.5, add_scale=2.0)
    got = ops.interp_add(view, loc, add=add, coarse_scale=0.5, add_scale=2.0)
    assert torch.equal(got, want)
    g = to(rng.standard_normal(ops.fine_shape(cshape, loc)).astype(dtype), dev)
    for route in (ops.interp_adj, ops.interp_adj_best):
        want = route(g, loc, cshape)
        dst_big = torch.full(full, 7.0, dtype=big.dtype, device=dev)
        route(g, loc, cshape, out=dst_big.narrow(1, lo, n))
        assert torch.equal(dst_big.narrow(1, lo, n), want)
        rest = torch.ones(full, dtype=torch.bool)
        rest[:, lo:lo + n] = False
        assert bool((dst_big.cpu()[rest] == 7.0).all())


def test_float_transpose_into_a_view_with_an_odd_leading_stride(dev):
    """P^T of float '.ccc' volumes with fine rows of 64 cells is served by the row-marching kernel, which stores 8-byte
    packs at `lead * stride + ...`: a result view whose leading stride is ODD (here 4 * 4 * 32 + 1 elements) must take
    another route -- every odd leading index would be a misaligned vector store -- and still give the contiguous result."""
    from odil_amd import ops

    rng = np.random.default_rng(67)
    loc, cshape = ".ccc", (3, 4, 4, 32)
    g = to(rng.standard_normal(ops.fine_shape(cshape, loc)).astype(np.float32), dev)
    want = ops.interp_adj(g, loc, cshape)
    ld = 4 * 4 * 32 + 1
    base = torch.full((3 * ld + 8,), 7.0, dtype=torch.float32, device=dev)
    view = base.as_strided(cshape, (ld, 4 * 32, 32, 1))
    for route in (ops.interp_adj, ops.interp_adj_best):
        base.fill_(7.0)
        route(g, loc, cshape, out=view)
        # (the kernel that takes over sums in another order than the row-marching one: float results differ in the last bits)
        assert float((view - want).abs().max()) <= 4e-6 * float(want.abs().max())
        touched = torch.zeros_like(base, dtype=torch.bool)
        touched.as_strided(cshape, (ld, 4 * 32, 32, 1)).fill_(True)
        assert bool((base[~touched] == 7.0).all())


def test_adjoint_and_transpose_in_one_launch_is_race_free(dev):
    """The one-launch kernel hands g0 from the lanes that form it to the lanes that consume it through LDS, one
    plane behind: a missing barrier shows as run-to-run differences on a grid with many resident workgroups.
    Twelve runs on (64, 128, 256) must all equal the separate kernels bit for bit."""
    from odil_amd import ops

    rng = np.random.default_rng(53)
    fine = (64, 128, 256)
    coarse = tuple(n // 2 for n in fine)
    h2 = [np.float64(v) for v in [0.25**2, 0.1**2, 0.3**2]]
    scale = np.float64(2.0 / np.prod(fine))
    fu = to(rng.standard_normal(fine), dev)
    state0 = [to(rng.standard_normal(s), dev) for s in (fine, coarse)] + [to(rng.standard_normal(s), dev) for s in (fine, coarse)] \
        + [to(np.abs(rng.standard_normal(s)), dev) for s in (fine, coarse)]

    def clone():
        x, m, v = [t.clone() for t in state0[0:2]], [t.clone() for t in state0[2:4]], [t.clone() for t in state0[4:6]]
        return x, m, v

    xr, mr, vr = clone()
    gr = [torch.empty(fine, dtype=fu.dtype, device=dev), torch.empty(coarse, dtype=fu.dtype, device=dev)]
    ops.poisson_adjoint_adam(fu, h2, scale, gr[0], xr[0], mr[0], vr[0], 0.01, 0.1, 0.001, 1e-7)
    ops.mg_synth_adj_adam(gr[0], [fine, coarse], "ccc", gr, xr, mr, vr, 0.01, 0.1, 0.001, 1e-7)
    for rep in range(12):
        x, m, v = clone()
        g1 = torch.empty(coarse, dtype=fu.dtype, device=dev)
        ops.poisson_adjoint_transpose(fu, h2, scale, g1, g0=None, adam0=(x[0], m[0], v[0]), adam1=(x[1], m[1], v[1]),
                                      alpha=0.01, one_minus_b1=0.1, one_minus_b2=0.001, eps=1e-7)
        assert torch.equal(g1, gr[1]), rep
        for a, b in zip(x + m + v, xr + mr + vr):
            assert torch.equal(a, b), rep


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_two_step_transpose_of_space_time_layout(dev, dtype, monkeypatch):
    """Large 'nccc' arrays take P^T = (P^T over the node axis) o (P^T over the three cell axes)
    (ops.mg_synth_adj); it must equal the one-kernel chain."""
    from odil_amd import ops

    rng = np.random.default_rng(17)
    shapes = [(9, 128, 128, 64), (5, 64, 64, 32), (3, 32, 32, 16)]
    if dtype == np.float32:
        shapes = [(17, 128, 128, 64), (9, 64, 64, 32), (5, 32, 32, 16)]
    g = to(rng.standard_normal(shapes[0]).astype(dtype), dev)
    assert ops._two_step_adjoint(g, shapes, "nccc", False)
    two = ops.mg_synth_adj(g, shapes, "nccc")
    monkeypatch.setattr(ops, "_two_step_adjoint", lambda *a: False)
    one = ops.mg_synth_adj(g, shapes, "nccc")
    tol = 1e-13 if dtype == np.float64 else 2e-6
    for a, b in zip(two, one):
        assert a.shape == b.shape and rel(a, b.cpu().numpy()) < tol
    # node-axis-only transpose against the oracle
    y = rng.standard_normal((9, 6, 5, 4))
    got = ops.interp_adj(to(y, dev), "n...", (5, 6, 5, 4))
    assert rel(got, onp.interp_to_finer_adj(y, "n...", (5, 6, 5, 4))) < 1e-14


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_dense_block_xty_on_the_matrix_cores(dev, dtype):
    """odil_dense_block_xty (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 tiles, fixed-order reduction) against
    NumPy: X^T Y for tall skinny matrices -- the D^T D, D^T r and C^T Z products of the Newton normal equations
    with dense `Array` / `NeuralNet` columns (reference core.py:1189-1203, linsolver.py:17-23) -- at row counts that
    are not multiples of anything, every column-tile count, strided (column-sliced) operands; bit-reproducible."""
    from odil_amd import ops

    rng = np.random.default_rng(5)
    tol = 1e-13 if dtype == np.float64 else 2e-5
    for n, px, py in [(1, 1, 1), (5, 3, 17), (1000, 16, 16), (4099, 46, 47), (100003, 64, 64), (70001, 33, 1), (257, 64, 5)]:
        x, y = rng.standard_normal((n, px)).astype(dtype), rng.standard_normal((n, py)).astype(dtype)
        got = ops.dense_xty(to(x, dev), to(y, dev))
        want = x.astype(np.float64).T @ y.astype(np.float64)
        assert got.shape == (px, py)
        assert np.max(np.abs(got.cpu().numpy() - want)) <= tol * max(1.0, np.max(np.abs(want))) * np.sqrt(n), (n, px, py)
        assert torch.equal(got, ops.dense_xty(to(x, dev), to(y, dev)))
    # column slices of one wider matrix (row stride > columns): [D | r] as the solver passes it
    d = rng.standard_normal((3001, 36)).astype(dtype)
    td = to(d, dev)
    got = ops.dense_xty(td[:, :35], td)
    want = d[:, :35].astype(np.float64).T @ d.astype(np.float64)
    assert np.max(np.abs(got.cpu().numpy() - want)) <= tol * np.max(np.abs(want)) * 60


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_planes_copy_vs_index_arithmetic(dtype):
    """odil_planes_copy (pack / unpack / unpack-add of the slab exchanges' planes, one launch) against the index
    arithmetic it replaces (tests/slab_oracle_ops.PlaneList): planes of arrays cut along an inner and along the
    leading axis, runs that are / are not multiples of four elements, a single-element plane; bit-exact."""
    from slab_oracle_ops import PlaneList as RefList

    from odil_amd import ops

    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(11)
    for planes in (
        [(5, 7, 6 * 24, 24), (7 * 6 * 24 + 3 * 24 + 5, 7, 6 * 24, 24), (3000, 1, 40, 40)],   # inner-axis cuts, runs of 24 / 40
        [(2, 3, 50, 10), (400, 9, 11, 1), (600, 1, 1, 1)],                                      # runs of 10, 1: scalar path
    ):
        n = 4000
        arr = torch.randn(n, generator=gen, dtype=dtype)
        ref, got = RefList(planes, "cpu"), ops.PlaneList(planes, dev)
        assert got.count == ref.count
        d_arr = arr.to(dev)
        msg = got.pack(d_arr)
        assert torch.equal(msg.cpu(), ref.pack(arr))
        buf = torch.randn(ref.count, generator=gen, dtype=dtype)
        a1, a2 = arr.clone(), arr.clone()
        ref.unpack(a1, buf)
        ref.unpack_add(a2, buf)
        d1, d2 = arr.to(dev), arr.to(dev)
        got.unpack(d1, buf.to(dev))
        got.unpack_add(d2, buf.to(dev))
        assert torch.equal(d1.cpu(), a1) and torch.equal(d2.cpu(), a2)


@pytest.mark.parametrize("cshape,lo,n", [((5, 4, 32, 32), 0, 4), ((3, 7, 8, 64), 2, 3), ((3, 6, 6, 128), 1, 4),
                                          ((2, 4, 4, 160), 0, 2), ((4, 2, 2, 32), 0, 2), ((9, 20, 16, 64), 1, 18)])
@pytest.mark.parametrize("with_add", [True, False])
def test_tiled_float_prolongation_of_a_node_leading_axis_equals_the_marching_kernel(dev, cshape, lo, n, with_add, monkeypatch):
    """k_interp_add_lead_tile ('nccc' float: the fine volumes 2k, 2k + 1 of a tile of coarse cells staged through LDS)
    against the register-window marching kernel it replaces (ODIL_LEAD_TILE=0): the same bits -- the 32 x 8 x 2 tile (the other shapes measured in round 4 were removed in round 5), tiles that hang over the array on z, y and x, walls inside every tile, a coarse operand that is
    a view along axis 1 (slab paths), with and without the fine addend -- and the float64 oracle to rounding."""
    from odil_amd import ops

    rng = np.random.default_rng(77)
    big = to(rng.standard_normal(cshape).astype(np.float32), dev)
    view = big.narrow(1, lo, n)
    loc = "nccc"
    fshape = ops.fine_shape(tuple(view.shape), loc)
    add = to(rng.standard_normal(fshape).astype(np.float32), dev) if with_add else None
    monkeypatch.setenv("ODIL_LEAD_TILE", "1")
    got = ops.interp_add(view, loc, add=add, coarse_scale=0.5, add_scale=2.0)
    monkeypatch.setenv("ODIL_LEAD_TILE", "0")
    want = ops.interp_add(view, loc, add=add, coarse_scale=0.5, add_scale=2.0)
    assert torch.equal(got, want)
    ref = onp.interp_to_finer(view.cpu().numpy().astype(np.float64) * 0.5, loc)
    if with_add:
        ref = ref + 2.0 * add.cpu().numpy().astype(np.float64)
    assert rel(got, ref) < 2e-6


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("loc,cshape", [("ccc", (6, 8, 32)), ("ccc", (3, 5, 64)), ("ccc", (4, 3, 160)), ("ccc", (16, 16, 128)),
                                         (".ccc", (3, 4, 6, 32)), ("nccc", (4, 3, 8, 64))])
def test_tiled_prolongation_of_volumes_and_batches_equals_the_marching_kernels(dev, dtype, loc, cshape, monkeypatch):
    """The LDS-tiled prolongation with ONE fine volume per leading index (3-D 'ccc', batches '.ccc') and, in float64, the
    pairs of a node-centred leading axis: the same bits as the marching kernels (ODIL_LEAD_TILE=0), the oracle to rounding."""
    from odil_amd import ops

    rng = np.random.default_rng(5)
    c = to(rng.standard_normal(cshape).astype(dtype), dev)
    add = to(rng.standard_normal(ops.fine_shape(cshape, loc)).astype(dtype), dev)
    res = []
    for flag in ("1", "0"):
        monkeypatch.setenv("ODIL_LEAD_TILE", flag)
        res.append((ops.interp_add(c, loc, add=add, coarse_scale=0.5, add_scale=2.0), ops.interp_add(c, loc)))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    ref = onp.interp_to_finer(c.cpu().numpy().astype(np.float64), loc)
    assert rel(res[0][1], ref) < (2e-6 if dtype == np.float32 else 1e-14)
