"""Pins the NumPy oracle (oracle/odil_np.py) against the golden vectors produced by
the reference's own code (tests/golden/make_golden.py).  CPU only."""

import numpy as np
import pytest
from conftest import load_golden

from oracle import odil_np as onp

EPS = np.finfo(np.float64).eps


def close(a, b, tol=1e-13):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(np.max(np.abs(b)))) if b.size else 1.0
    err = float(np.max(np.abs(a - b))) / scale if b.size else 0.0
    assert err <= tol, err


INTERP_CASES = [str(c) for c in load_golden("interp")["cases"]]


@pytest.mark.parametrize("loc", INTERP_CASES)
def test_interp_forward_and_adjoint(loc):
    g = load_golden("interp")
    u = g[f"{loc}/u"]
    fine = onp.interp_to_finer(u, loc)
    # Same summation order as the reference 'stack' path: bit-exact.
    assert np.array_equal(fine, g[f"{loc}/fine"])
    if f"{loc}/fine2" in g:
        assert np.array_equal(onp.interp_to_finer(u, loc, depth=2), g[f"{loc}/fine2"])
    gu = onp.interp_to_finer_adj(g[f"{loc}/gfine"], loc, u.shape)
    close(gu, g[f"{loc}/gu"], 1e-14)


CONV_CASES = [str(c) for c in load_golden("interp_conv")["cases"]]


@pytest.mark.parametrize("loc", CONV_CASES)
def test_interp_conv_reference_values(loc):
    """`interp_to_finer(method="conv")` of the reference itself (core.py:645-667 through the conv_transpose of
    tests/golden/ref_shim.py; the tracer workload's default, veltracer.py:150): the oracle's one prolongation (the
    'stack' summation order) gives the same values to rounding, and the same cotangent."""
    g = load_golden("interp_conv")
    u = g[f"{loc}/u"]
    assert float(g[f"{loc}/stack_minus_conv"]) <= 16 * EPS  # the reference's two methods agree with each other
    close(onp.interp_to_finer(u, loc), g[f"{loc}/fine"], 1e-15)
    if f"{loc}/fine2" in g:
        close(onp.interp_to_finer(u, loc, depth=2), g[f"{loc}/fine2"], 2e-15)
    close(onp.interp_to_finer_adj(g[f"{loc}/gfine"], loc, u.shape), g[f"{loc}/gu"], 1e-14)


@pytest.mark.parametrize("loc", [str(c) for c in load_golden("restrict")["cases"]])
def test_restrict_reference_values(loc):
    """`restrict_to_coarser` of the reference itself (core.py:703-755, the strided convolution of backend.py:112-126
    -- integer stride on '.' axes included), values at depth 1 and 2 and the cotangent."""
    g = load_golden("restrict")
    u = g[f"{loc}/u"]
    close(onp.restrict_to_coarser(u, loc), g[f"{loc}/coarse"], 1e-15)
    close(onp.restrict_to_coarser(u, loc, depth=2), g[f"{loc}/coarse2"], 2e-15)
    close(onp.restrict_to_coarser_adj(g[f"{loc}/gcoarse"], loc, u.shape), g[f"{loc}/gu"], 1e-15)


def test_interp_exact_on_linear_functions():
    """reference tests/test_mg_interp.py:11-32 restated on the oracle."""
    for ndim in [1, 2, 3, 4]:
        for loc in {s[:ndim] for s in ["cccc", "nnnn", "cnnn", "nccc"]}:
            cshapeh = tuple(3 + np.arange(ndim))
            cshape = tuple(2 * np.array(cshapeh))

            def func(xx):
                return sum(x * np.sqrt(i + 1) for i, x in enumerate(xx))

            u = func(onp.points(cshape, loc))
            uh = func(onp.points(cshapeh, loc))
            assert np.max(np.abs(onp.interp_to_finer(uh, loc) - u)) <= 100 * EPS


def test_restrict_exact_on_linear_functions_with_jumps():
    """reference tests/test_mg_restrict.py:11-41 restated on the oracle."""
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
            uhr = onp.restrict_to_coarser(u, loc)
            assert np.max(np.abs(uhr - uh)) <= 100 * EPS, (ndim, loc)


MG_CASES = [str(c) for c in load_golden("mg")["cases"]]


@pytest.mark.parametrize("name", MG_CASES)
def test_multigrid_synthesis_and_adjoint(name):
    g = load_golden("mg")
    nlvl = int(g[f"{name}/nlvl"])
    loc = str(g[f"{name}/loc"])
    axes = [bool(a) for a in g[f"{name}/axes"]]
    factors = [float(f) for f in g[f"{name}/factors"]]
    cshape = tuple(int(c) for c in g[f"{name}/cshape"])
    terms = [g[f"{name}/w{i}"] for i in range(nlvl)]
    assert [onp.field_shape(cs, loc) for cs in onp.mg_cshapes(cshape, axes)] == [t.shape for t in terms]
    u = onp.multigrid_to_regular(terms, loc, factors, axes)
    assert np.array_equal(u, g[f"{name}/u"])
    grads = onp.multigrid_to_regular_adj(g[f"{name}/gu"], [t.shape for t in terms], loc, factors, axes)
    for i in range(nlvl):
        close(grads[i], g[f"{name}/g{i}"], 1e-14)


def test_field_access():
    g = load_golden("field_access")
    for name in g["cases"]:
        floc, loc = str(g[f"{name}/field_loc"]), str(g[f"{name}/loc"])
        shift = tuple(int(s) for s in g[f"{name}/shift"])
        a = g[f"{name}/a"]
        assert np.array_equal(onp.field_access(a, floc, shift, loc), g[f"{name}/out"])
        ga = onp.field_access_adj(g[f"{name}/g"], a.shape, floc, shift, loc)
        assert np.array_equal(ga, g[f"{name}/ga"])


POISSON = ["poisson_1d_N256", "poisson_2d_N32", "poisson_3d_N16", "poisson_2d_N8", "poisson_3d_N8"]


@pytest.mark.parametrize("name", POISSON)
def test_poisson_loss_grad_and_adam(name):
    g = load_golden(name)
    ndim, N, nlvl = int(g["ndim"]), int(g["N"]), int(g["nlvl"])
    cshape = (N,) * ndim
    dw = onp.step(cshape)
    ref_u = onp.poisson_ref_u(cshape)
    close(ref_u, g["ref_u"], 1e-15)
    rhs = onp.poisson_discrete_rhs(ref_u, dw)
    close(rhs, g["rhs"], 1e-14)
    rhs = g["rhs"]
    assert nlvl == len(onp.mg_cshapes(cshape))
    terms = [g[f"rand/w{i}"] for i in range(nlvl)]
    loss, grads, fu = onp.poisson_loss_grad(terms, rhs, dw)
    close(fu, g["rand/fu"], 1e-14)
    assert abs(loss - float(g["rand/loss"])) <= 1e-14 * abs(loss)
    for i in range(nlvl):
        close(grads[i], g[f"rand/g{i}"], 1e-13)

    # Adam trajectory from zero (reference AdamNativeOptimizer, lr=0.005).
    def loss_grad(x):
        loss, grads, _ = onp.poisson_loss_grad(x, rhs, dw)
        return loss, grads

    losses_ref = g["adam/losses"]
    x0 = [np.zeros(s) for s in [onp.field_shape(cs, "c" * ndim) for cs in onp.mg_cshapes(cshape)]]
    x, losses = onp.adam_run(x0, loss_grad, len(losses_ref), float(g["lr"]))
    # north_star tolerance: loss trajectory within 1e-6 relative.
    assert np.max(np.abs(np.array(losses) - losses_ref) / losses_ref) < 1e-6
    for i in range(nlvl):
        close(x[i], g[f"adam/w{i}"], 1e-7)


def test_poisson_known_answers_from_survey():
    """SURVEY.md 8(c) anchors (reference code + reference AdamNativeOptimizer, f64)."""
    g = load_golden("poisson_1d_N256")
    assert abs(np.sum(np.abs(g["rhs"])) - 2560.0000052663336) < 1e-9
    ll = g["adam/losses"]
    for i, v in [(0, 135.42885058760373), (1, 1368107.267716066), (2, 280292.88918997056)]:
        assert abs(ll[i] - v) / v < 1e-12
    for i, v in [(5, 742302.57750732475), (9, 462251.28067223774)]:
        assert abs(ll[i] - v) / v < 1e-6
    g = load_golden("poisson_2d_N32")
    for i, v in [(0, 371.60309995622487), (1, 1833.3563979244263), (2, 512.77536456229643)]:
        assert abs(g["adam/losses"][i] - v) / v < 1e-12
    g = load_golden("poisson_3d_N16")
    for i, v in [(0, 633.38922172230809), (1, 647.18695614438343), (2, 564.55327518546585)]:
        assert abs(g["adam/losses"][i] - v) / v < 1e-12


def test_poisson_f32_no_multigrid():
    g = load_golden("poisson_2d_N16_f32_nomg")
    dw = onp.step((16, 16), dtype=np.float32)
    w = g["rand/w0"]
    assert w.dtype == np.float32
    loss, grads, fu = onp.poisson_loss_grad([w], g["rhs"], dw)
    assert fu.dtype == np.float32
    close(fu, g["rand/fu"], 2e-6)
    assert abs(loss - float(g["rand/loss"])) <= 1e-5 * abs(loss)
    close(grads[0], g["rand/g0"], 1e-5)


@pytest.mark.parametrize("name", ["newton_poisson_1d_N8", "newton_poisson_2d_N6", "newton_poisson_3d_N4"])
def test_newton_poisson(name):
    g = load_golden(name)
    u0, rhs = g["u0"], g["rhs"]
    dw = onp.step(u0.shape)
    coeffs = onp.poisson_jac_coeffs(u0.shape, dw)
    assert len(coeffs) == len(g["shifts"])
    for sname in g["shifts"]:
        shift = tuple(int(s) for s in str(sname).split(","))
        close(coeffs[shift], g[f"coeff/{sname}"], 1e-14)
    vector, matrix = onp.poisson_linearize(u0, rhs, dw)
    close(vector, g["vector"], 1e-14)
    close(matrix.toarray(), g["matrix"], 1e-14)
    delta = onp.solve_normal_direct(matrix, -vector)
    close(delta, g["delta"], 1e-9)
    close(u0 + delta.reshape(u0.shape), g["ref_u"], 1e-9)  # linear problem: one Newton step solves it


def test_lbfgsb_trajectory():
    import scipy

    g = load_golden("lbfgsb_poisson_2d_N32")
    rhs = g["rhs"]
    cshape = rhs.shape
    dw = onp.step(cshape)

    def loss_grad(x):
        loss, grads, _ = onp.poisson_loss_grad(x, rhs, dw)
        return loss, grads

    x0 = [np.zeros(cs) for cs in onp.mg_cshapes(cshape)]
    x, losses, iter_losses, info = onp.lbfgsb_run(x0, loss_grad, int(g["epochs"]), m=int(g["m"]), maxls=int(g["maxls"]))
    ref = g["iter_losses"]
    n = min(len(ref), len(iter_losses))
    assert n >= 10
    rel = np.abs(np.array(iter_losses[:n]) - ref[:n]) / ref[:n]
    if str(g["scipy_version"]) == scipy.__version__:
        assert rel[:10].max() < 1e-6, rel


def test_restrict_adjoint_is_the_transpose():
    """<R x, y> == <x, R^T y> for the oracle pair (R itself is pinned by the exactness test above)."""
    rng = np.random.default_rng(9)
    for loc, shape in [("c", (8,)), ("n", (9,)), ("cc", (6, 8)), ("nn", (7, 9)), ("cn", (6, 9)), ("c.n", (4, 5, 7)),
                       ("ccc", (4, 6, 8)), ("nnn", (5, 7, 9))]:
        x = rng.standard_normal(shape)
        rx = onp.restrict_to_coarser(x, loc)
        y = rng.standard_normal(rx.shape)
        lhs = np.sum(rx * y)
        rhs = np.sum(x * onp.restrict_to_coarser_adj(y, loc, shape))
        assert abs(lhs - rhs) <= 1e-13 * max(1.0, abs(lhs)), (loc, lhs, rhs)
