"""GPU tests of the operator API: user callbacks written for the reference run through
Domain / Context / Problem on the HIP kernels and reproduce the reference's numbers
(golden fixtures) -- reference tests test_optimize.py / test_newton.py restated."""

import argparse
import os
import sys

import numpy as np
import pytest
import torch
from conftest import ROOT, load_golden

import odil_amd as odil
from oracle import odil_np as onp

sys.path.insert(0, os.path.join(ROOT, "examples", "poisson"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mod():
    assert torch.cuda.is_available()
    return odil.runtime.get_mod()


def npy(t):
    return t.detach().cpu().numpy()


def rel(a, b):
    a = npy(a) if isinstance(a, torch.Tensor) else np.asarray(a)
    return float(np.max(np.abs(a - b))) / max(1.0, float(np.max(np.abs(b))))


def poisson_args(ndim, N, multigrid=1, **kw):
    import poisson

    args = poisson.parse_args([])
    args.ndim, args.N, args.multigrid = ndim, N, multigrid
    for k, v in kw.items():
        setattr(args, k, v)
    args.epoch_start = 0
    return poisson, args


@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("name", ["poisson_1d_N256", "poisson_2d_N32", "poisson_3d_N16"])
def test_poisson_problem_loss_grad_vs_golden(mod, name, fuse, monkeypatch):
    g = load_golden(name)
    ndim, N, nlvl = int(g["ndim"]), int(g["N"]), int(g["nlvl"])
    monkeypatch.setattr(odil.runtime, "enable_fuse", fuse)
    poisson, args = poisson_args(ndim, N)
    problem, state = poisson.make_problem(args)
    domain = problem.domain
    # device pow() and NumPy's differ in the last bits of ref_u; 1/h^2 amplifies that to ~1e-12
    assert rel(problem.extra.rhs, g["rhs"]) < 1e-11
    problem.extra.rhs = mod.array(g["rhs"])  # parity is checked on identical inputs
    arrays = [mod.array(g[f"rand/w{i}"]) for i in range(nlvl)]
    domain.arrays_to_state(arrays, state)
    loss, grads, terms, names, norms = problem.eval_loss_grad(state)
    assert (problem._fused is not None) == fuse
    assert isinstance(loss, np.ndarray) and loss.shape == ()
    assert abs(float(loss) - float(g["rand/loss"])) <= 1e-12 * float(g["rand/loss"])
    assert abs(float(norms[0]) - np.sqrt(float(g["rand/loss"]))) <= 1e-12 * float(norms[0])
    for i in range(nlvl):
        assert rel(grads[i], g[f"rand/g{i}"]) < 1e-12
    values, names2 = problem.eval_operator(state)
    assert rel(values[0], g["rand/fu"]) < 1e-13


@pytest.mark.parametrize("fuse", [False, True])
def test_adam_trajectory_through_optimize_grad(mod, fuse, monkeypatch):
    """reference AdamNativeOptimizer loss trajectory, tolerance 1e-6 relative (north_star)."""
    monkeypatch.setattr(odil.runtime, "enable_fuse", fuse)
    for name in ["poisson_1d_N256", "poisson_2d_N32", "poisson_3d_N16"]:
        g = load_golden(name)
        ref = g["adam/losses"]
        poisson, args = poisson_args(int(g["ndim"]), int(g["N"]), epochs=len(ref), lr=0.005)
        problem, state = poisson.make_problem(args)
        problem.extra.rhs = mod.array(g["rhs"])
        losses = []

        def callback(state, epoch, pinfo):
            losses.append(float(pinfo["loss"]))

        odil.util.optimize_grad(args, "adam", problem, state, callback)
        # callback sees the loss evaluated BEFORE each update, plus the initial evaluation (epoch 0)
        got = np.array(losses[1:])
        assert got.shape == ref.shape
        assert np.max(np.abs(got - ref) / ref) < 1e-6, name
        arrays = problem.domain.arrays_from_state(state)
        for i in range(int(g["nlvl"])):
            assert rel(arrays[i], g[f"adam/w{i}"]) < 1e-7


def test_lbfgsb_trajectory_vs_golden(mod):
    g = load_golden("lbfgsb_poisson_2d_N32")
    ref = g["iter_losses"]
    poisson, args = poisson_args(2, 32, epochs=int(g["epochs"]))
    problem, state = poisson.make_problem(args)
    problem.extra.rhs = mod.array(g["rhs"])
    losses = []

    def callback(state, epoch, pinfo):
        losses.append(float(pinfo["loss"]))

    odil.util.optimize_grad(args, "lbfgsb", problem, state, callback)
    got = np.array(losses[1:])
    n = min(len(got), len(ref))
    assert n >= 20
    relerr = np.abs(got[:n] - ref[:n]) / ref[:n]
    assert relerr[:12].max() < 1e-6, relerr


def test_poisson_f32_no_multigrid_generic_path(mod, monkeypatch):
    g = load_golden("poisson_2d_N16_f32_nomg")
    monkeypatch.setattr(odil.runtime, "enable_fuse", False)
    poisson, args = poisson_args(2, 16, multigrid=0, double=0)
    problem, state = poisson.make_problem(args)
    problem.extra.rhs = mod.array(g["rhs"])
    state.fields["u"].array = mod.array(g["rand/w0"])
    loss, grads, *_ = problem.eval_loss_grad(state)
    assert grads[0].dtype == torch.float32
    assert abs(float(loss) - float(g["rand/loss"])) <= 1e-5 * float(g["rand/loss"])
    assert rel(grads[0], g["rand/g0"]) < 1e-5


# ---------------------------------------------------------------------------------------
# reference tests/test_newton.py:11-44 (operator) and :115-148 (check), same data as the golden
# ---------------------------------------------------------------------------------------
def newton_operator(ctx):
    mod = ctx.mod
    extra = ctx.extra
    res = []
    u_xm = ctx.field("ufx", 0, 0, loc="cc")  # face (i-1/2, j) seen from cell (i, j)
    u_xp = ctx.field("ufx", 1, 0, loc="cc")  # face (i+1/2, j)
    hx = ctx.step("x")
    res += [(u_xp - u_xm) / hx - extra.ref["dudx"]]
    ufx = ctx.field("ufx")
    ixfx = ctx.indices("x", loc="nc")
    mask = mod.where(ixfx == 0, ctx.cast(1), ctx.cast(0))
    res += [(ufx - extra.ref["ufx"]) * mask]
    uc = ctx.field("uc")
    res += [(u_xp + u_xm) * 0.5 - uc]
    a = ctx.field("a")
    res += [a - extra.ref["a"]]
    net_out = ctx.neural_net("net")(*extra.ref["net_in"])
    for i in range(extra.Nnet):
        res += [(f"net{i}", net_out[i] - extra.ref["net_out"][i])]
    return res


def damped_want(g, y):
    a = g["matrix"].T @ g["matrix"]
    a = a + 0.09 * np.eye(len(a))  # reference linsolver.py:19-23: damp first, dampdiag on the damped diagonal
    return np.linalg.solve(a + 0.04 * np.diag(np.diag(a)), g["matrix"].T @ y)


def make_newton_problem(mod, g):
    Nx, Ny, Na, Nnet = 3, 2, 5, 5
    domain = odil.Domain(cshape=(Nx, Ny), dimnames=["x", "y"], lower=(0, 0), upper=(Nx, Ny), dtype=np.float64,
                         multigrid=0)
    net = odil.NeuralNet([g["x0/3"]], [g["x0/4"]], activation="none")
    state = odil.State(fields={
        "uc": odil.Field(g["x0/0"], loc="cc"),
        "ufx": odil.Field(g["x0/1"], loc="nc"),
        "a": odil.Array(g["x0/2"]),
        "net": net,
    })
    state = domain.init_state(state)
    extra = argparse.Namespace(Nnet=Nnet)
    extra.ref = {k: mod.array(g[f"ref/{k}"]) for k in ["uc", "ufx", "dudx", "a", "net_in", "net_out"]}
    return odil.Problem(newton_operator, domain, extra), state


def test_newton_linearize_and_step_vs_golden(mod):
    g = load_golden("test_newton")
    problem, state = make_newton_problem(mod, g)
    domain = problem.domain
    vector, matrix = problem.linearize(state)
    assert rel(vector, g["vector"]) < 1e-14
    assert rel(matrix.toarray(), g["matrix"]) < 1e-13
    # device operator: M x and M^T y agree with the assembled matrix
    _, op = problem.linearize_device(state)
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(op.shape[1]), rng.standard_normal(op.shape[0])
    assert rel(op.matvec(mod.array(x)), g["matrix"] @ x) < 1e-13
    assert rel(op.rmatvec(mod.array(y)), g["matrix"].T @ y) < 1e-13
    assert rel(op.normal_diagonal(), np.sum(g["matrix"] ** 2, axis=0)) < 1e-13
    # the three ways the normal equations are solved agree: dense Cholesky (`direct`, small systems),
    # the dense matrix is the assembled one, CG without host synchronisation
    assert rel(op.to_dense(), g["matrix"]) < 1e-13
    rhs = mod.array(y)
    status = dict()
    x_dense = odil.linsolver.dense_normal(op, rhs, status=status)
    assert status["method"] == "dense-cholesky" and status["residual"] < 1e-10
    status = dict()
    x_cg = odil.linsolver.cg_normal(op, rhs, tol=1e-14, status=status, check_every=7)
    assert status["niter"] % 7 == 0 and status["niter"] > 0
    want = np.linalg.solve(g["matrix"].T @ g["matrix"], g["matrix"].T @ y)
    assert rel(x_dense, want) < 1e-9 and rel(x_cg, want) < 1e-8
    # ... and the route `direct` takes for systems with dense columns: D^T D and C^T Z on the matrix cores,
    # Schur complement against the matrix-free stencil part (no densification of M)
    status = dict()
    x_schur = odil.linsolver.schur_normal(op, rhs, status=status)
    assert status["method"] == "schur-mfma" and status["dense_columns"] == 5 + 25 + 5 and status["residual"] < 1e-9
    assert rel(x_schur, want) < 1e-9
    assert rel(odil.linsolver.schur_normal(op, rhs, damp=0.3, dampdiag=0.2), damped_want(g, y)) < 1e-9
    damped = odil.linsolver.dense_normal(op, rhs, damp=0.3, dampdiag=0.2)
    a = g["matrix"].T @ g["matrix"]
    # reference linsolver.py:19-23: damp^2 I first, then dampdiag^2 times the diagonal of the DAMPED matrix
    a = a + 0.09 * np.eye(len(a))
    assert rel(damped, np.linalg.solve(a + 0.04 * np.diag(np.diag(a)), g["matrix"].T @ y)) < 1e-9
    assert rel(odil.linsolver.cg_normal(op, rhs, damp=0.3, dampdiag=0.2, tol=1e-14), damped.cpu().numpy()) < 1e-8
    # one Newton step through optimize_newton
    args = argparse.Namespace(epoch_start=0, epochs=1, linsolver="direct", linsolver_maxiter=None, linsolver_damp=0,
                              linsolver_dampdiag=0, linsolver_tol=1e-10, linsolver_verbose=0)
    odil.util.optimize_newton(args, problem, state)
    arrays = domain.arrays_from_state(state)
    for k, a in enumerate(arrays):
        assert rel(a, g[f"x1/{k}"]) < 1e-9, k
    # the reference test's own pass criterion: rms error < 1e-6 per field (test_newton.py:131-146)
    ref = problem.extra.ref
    for key in ["ufx", "uc", "a"]:
        err = domain.field(state, key) - ref[key]
        assert float(torch.sqrt(torch.mean(err**2))) < 1e-6
    out = torch.stack(domain.neural_net(state, "net")(*ref["net_in"]))
    assert float(torch.sqrt(torch.mean((out - ref["net_out"]) ** 2))) < 1e-6


@pytest.mark.parametrize("name", ["newton_poisson_1d_N8", "newton_poisson_2d_N6", "newton_poisson_3d_N4"])
def test_newton_poisson_vs_golden(mod, name):
    g = load_golden(name)
    ndim, N = g["u0"].ndim, g["u0"].shape[0]
    poisson, args = poisson_args(ndim, N, multigrid=0, epochs=1, linsolver="direct", linsolver_maxiter=None)
    problem, state = poisson.make_problem(args)
    problem.extra.rhs = mod.array(g["rhs"])
    state.fields["u"].array = mod.array(g["u0"])
    vector, matrix = problem.linearize(state)
    assert rel(vector, g["vector"]) < 1e-13
    assert rel(matrix.toarray(), g["matrix"]) < 1e-13
    odil.util.optimize_newton(args, problem, state)
    assert rel(state.fields["u"].array, g["u1"]) < 1e-8
    assert rel(state.fields["u"].array, g["ref_u"]) < 1e-8  # linear problem: one step solves it


# ---------------------------------------------------------------------------------------
# reference tests/test_optimize.py:10-113: fields at cc/nn/nc/cn + Array + NeuralNet, MG on
# ---------------------------------------------------------------------------------------
def optimize_operator(ctx):
    extra = ctx.extra
    res = []
    for key in ["uc", "un", "ufx", "ufy"]:
        res += [(key, ctx.field(key) - extra.ref[key])]
    res += [("a", ctx.field("a") - extra.ref["a"])]
    net_a = ctx.neural_net("net")(ctx.field("a"))[0]
    res += [("net_a", net_a - extra.ref["net_a"])]
    return res


@pytest.mark.parametrize("opt", ["lbfgsb", "adamn"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_optimize_mixed_state(mod, opt, dtype):
    mod.random.set_seed(1)
    domain = odil.Domain(cshape=(8, 4), dimnames=["x", "y"], lower=(0, 0), upper=(2, 1), multigrid=1,
                         mg_interp="stack", mg_axes=[True, True], dtype=dtype)
    state = odil.State(fields={
        "uc": odil.Field(np.zeros(domain.size(loc="cc")), loc="cc"),
        "un": odil.Field(np.zeros(domain.size(loc="nn")), loc="nn"),
        "ufx": odil.Field(np.zeros(domain.size(loc="nc")), loc="nc"),
        "ufy": odil.Field(np.zeros(domain.size(loc="cn")), loc="cn"),
        "a": odil.Array(np.zeros(5)),
        "net": domain.make_neural_net([1, 7, 1]),
    })
    state = domain.init_state(state)
    func = lambda x, y: x * 0.25 + y * 0.5
    extra = argparse.Namespace()
    extra.ref = {k: func(*domain.points(loc=l)) for k, l in [("uc", "cc"), ("un", "nn"), ("ufx", "nc"), ("ufy", "cn")]}
    extra.ref["a"] = mod.array(np.arange(5, dtype=dtype))
    extra.ref["net_a"] = extra.ref["a"] * 0.5
    problem = odil.Problem(optimize_operator, domain, extra)
    args = argparse.Namespace(epoch_start=0, epochs=1000, lr=0.1, bfgs_m=50, bfgs_maxls=50, bfgs_pgtol=None,
                              adam_epsilon=None, adam_beta_1=None, adam_beta_2=None, callback_update_state=0)
    try:
        odil.util.optimize_grad(args, opt, problem, state, None)
    except odil.EarlyStopError:
        pass
    error = [domain.field(state, key) - extra.ref[key] for key in ["uc", "un", "ufx", "ufy", "a"]]
    error.append(domain.neural_net(state, "net")(domain.field(state, "a"))[0] - extra.ref["net_a"])
    total = float(torch.sqrt(sum(torch.mean(torch.square(e)) for e in error)))
    assert total < 1e-2, total


@pytest.mark.parametrize("ndim,N", [(2, 64), (3, 32)])
def test_newton_multigrid_solver(mod, ndim, N):
    """`--linsolver multigrid`: V-cycles on the recognised Poisson stencil reach the Newton iterate
    (= the exact discrete solution for this linear problem) in a handful of cycles."""
    poisson, args = poisson_args(ndim, N, multigrid=0, epochs=1, linsolver="multigrid", linsolver_maxiter=None,
                                 linsolver_tol=1e-11)
    problem, state = poisson.make_problem(args)
    statuses = []

    def callback(state, epoch, pinfo):
        if "linsolver" in pinfo:
            statuses.append(pinfo["linsolver"])

    odil.util.optimize_newton(args, problem, state, callback)
    assert statuses and statuses[0]["method"] == "gmg-vcycle" and statuses[0]["niter"] <= 25, statuses
    err = state.fields["u"].array - problem.extra.ref_u
    assert float(err.abs().max()) < 1e-8
    loss = problem.eval_loss_grad(state)[0]
    assert float(loss) < 1e-10


def test_newton_multigrid_matches_direct_small(mod):
    g = load_golden("newton_poisson_3d_N4")
    poisson, args = poisson_args(3, 4, multigrid=0, epochs=1, linsolver="multigrid", linsolver_maxiter=None,
                                 linsolver_tol=1e-13)
    problem, state = poisson.make_problem(args)
    problem.extra.rhs = mod.array(g["rhs"])
    state.fields["u"].array = mod.array(g["u0"])
    odil.util.optimize_newton(args, problem, state)
    assert rel(state.fields["u"].array, g["u1"]) < 1e-9


def test_poisson_mgloss_gradient(mod, monkeypatch):
    """`--mgloss 2` (reference poisson.py:116-122): extra loss terms on restricted residuals; their
    gradient flows through R^T.  Checked against the oracle composition."""
    poisson, args = poisson_args(2, 16, mgloss=2)
    problem, state = poisson.make_problem(args)
    rng = np.random.default_rng(4)
    arrays = [mod.array(rng.standard_normal(tuple(a.shape)) * 0.1) for a in problem.domain.arrays_from_state(state)]
    problem.domain.arrays_to_state(arrays, state)
    loss, grads, terms, names, norms = problem.eval_loss_grad(state)
    assert len(terms) == 3 and problem._fused is None
    # oracle: same composition in NumPy
    cshape = (16, 16)
    dw = onp.step(cshape)
    w = [a.cpu().numpy() for a in arrays]
    rhs = problem.extra.rhs.cpu().numpy()
    u = onp.multigrid_to_regular(w, "cc")
    f0 = onp.poisson_residual(u, rhs, dw)
    f1 = onp.restrict_to_coarser(f0, "cc")
    f2 = onp.restrict_to_coarser(f1, "cc")
    want = sum(np.mean(f**2) for f in (f0, f1, f2))
    assert abs(float(loss) - want) <= 1e-12 * want
    g2 = 2 * f2 / f2.size
    g1 = 2 * f1 / f1.size + onp.restrict_to_coarser_adj(g2, "cc", f1.shape)
    g0 = 2 * f0 / f0.size + onp.restrict_to_coarser_adj(g1, "cc", f0.shape)
    gw = onp.multigrid_to_regular_adj(onp.poisson_adjoint(g0, dw), [a.shape for a in w], "cc")
    for a, b in zip(grads, gw):
        assert rel(a, b) < 1e-11


@pytest.mark.parametrize("ndim,N", [(3, 32), (2, 64)])
def test_newton_poisson_through_geometric_multigrid(mod, ndim, N):
    """`--linsolver multigrid` on the recognised Poisson operator: the Newton step comes from the fused
    residual and V-cycles (Chebyshev-weighted Jacobi sweeps, odil_poisson_jacobi) and equals the step
    of the general route (coefficient arrays -> normal equations) -- one step solves the linear problem."""
    poisson, args = poisson_args(ndim, N, multigrid=0, epochs=1, linsolver="multigrid", linsolver_maxiter=None)
    args.linsolver_tol = 1e-12
    problem, state = poisson.make_problem(args)
    odil.util.optimize_newton(args, problem, state)
    assert problem._fused is not None and "_gmg" in problem._fused.__dict__  # the fast route was taken
    u_fast = state.fields["u"].array.clone()
    err = u_fast - problem.extra.ref_u
    assert float(err.abs().max()) < 1e-8
    # general route on the same problem
    problem2, state2 = poisson.make_problem(args)
    vector, matrix = problem2.linearize_device(state2)
    status = dict()
    delta = odil.linsolver.solve(matrix, -vector, args, status, "multigrid")
    assert status["method"] == "gmg-vcycle" and status["niter"] < 25
    assert float((state2.fields["u"].array.reshape(-1) + delta - u_fast.reshape(-1)).abs().max()) < 1e-9


def test_dense_jacobian_through_a_hip_transfer(mod):
    """Dense Jacobian columns of network parameters (`tape.jacobian`, reference core.py:1347-1349) when the network's
    output passes through one of this package's HIP-backed autograd functions (a prolongation): the one-pass-per-parameter
    route needs a second derivative those functions do not have -- it must fall back to row-by-row passes, not drop the
    block.  Checked against central differences of the operator in the weights."""
    domain = odil.Domain(cshape=(6,), dimnames=["x"], multigrid=0, dtype=np.float64)
    mod.random.set_seed(7)
    state = odil.State(fields={"u": odil.Field(np.zeros(12), loc="c", cshape=(12,)),
                               "a": odil.Array(np.linspace(-1, 1, 6)), "net": domain.make_neural_net([1, 4, 1])})
    state = domain.init_state(state)

    def operator(ctx):
        coarse = ctx.neural_net("net")(ctx.field("a"))[0]
        return [("f", odil.core.interp_to_finer(coarse, loc="c", mod=ctx.mod) * 3.0 - 0.5)]

    problem = odil.Problem(operator, domain)
    vector, matrix = problem.linearize(state)
    dense = matrix.toarray()
    arrays = domain.arrays_from_state(state)
    sizes = [int(a.numel()) for a in arrays]
    nfield = sizes[0]
    assert dense.shape == (12, sum(sizes)) and np.all(dense[:, :nfield] == 0)  # u does not enter
    h, col = 1e-6, nfield
    for k in range(1, len(arrays)):
        for e in range(sizes[k]):
            vals = []
            for sign in (1.0, -1.0):
                pert = [a.clone() for a in arrays]
                pert[k].view(-1)[e] += sign * h
                domain.arrays_to_state(pert, state)
                vals.append(npy(problem.eval_operator(state)[0][0]).reshape(-1))
            domain.arrays_to_state(arrays, state)
            fd = (vals[0] - vals[1]) / (2 * h)
            assert np.max(np.abs(dense[:, col] - fd)) <= 1e-7 * max(1.0, np.max(np.abs(fd))), (k, e)
            col += 1
    assert np.max(np.abs(dense[:, nfield + sizes[1]:])) > 0.1  # the network block is there
