"""The quasi-Newton / Newton drivers of the slab path (odil_amd/slab_solvers.py) on CPU: ranks as threads of one process
and as gloo processes, kernels replaced by their NumPy-oracle doubles.  L-BFGS-B on the decomposed domain must follow
the undivided run of the same `lbfgsb_minimize` iterate for iterate (the reductions differ in summation order only); the
matrix-free CG Newton step must reproduce the undivided solve."""

import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import odil_np as onp  # noqa: E402


def make_rhs(N, world):
    return np.random.default_rng(7).standard_normal((N * world, N, N))


def undivided_lbfgs(N, world, rhs, maxiter, m):
    from test_lbfgs_host_logic import NumpyVectors

    from odil_amd.optimizer import lbfgsb_minimize

    cshape = (N * world, N, N)
    shapes = onp.mg_cshapes(cshape)
    sizes = [int(np.prod(s)) for s in shapes]
    dw = (1.0 / N,) * 3

    def fun(x):
        terms = [t.reshape(s) for t, s in zip(np.split(x, np.cumsum(sizes)[:-1]), shapes)]
        loss, grads, _ = onp.poisson_loss_grad(terms, rhs, dw)
        return float(loss), np.concatenate([g.ravel() for g in grads])

    x = np.zeros(sum(sizes))
    res = lbfgsb_minimize(x, fun, NumpyVectors(x.size, m), maxiter, m=m)
    return res, [t.reshape(s) for t, s in zip(np.split(x, np.cumsum(sizes)[:-1]), shapes)]


def lbfgs_rank(rank, world, comm, N, rhs, maxiter, m):
    import slab_oracle_ops
    from slab_oracle_vectors import TorchCpuSlabVectors

    from odil_amd import slab
    from odil_amd.slab_solvers import SlabPoissonLbfgs

    slab.hip_ops = slab_oracle_ops
    run = SlabPoissonLbfgs(N, rank, world, dtype=torch.float64, device=torch.device("cpu"), rhs_global=torch.from_numpy(rhs))
    res = run.minimize(comm, maxiter, m=m, vectors=TorchCpuSlabVectors(run.n_unknowns_local, m, comm))
    return res, [w.clone().numpy() for w in run.owned_levels()]


def check_lbfgs(results, ref, world):
    res_ref, x_ref = ref
    for r, (res, owned) in enumerate(results):
        assert res["nit"] == res_ref["nit"] and res["funcalls"] == res_ref["funcalls"]
        assert abs(res["f"] - res_ref["f"]) <= 1e-9 * abs(res_ref["f"])
        for lvl, want in enumerate(x_ref):
            nz = want.shape[0] // world
            np.testing.assert_allclose(owned[lvl], want[r * nz:(r + 1) * nz], rtol=0, atol=1e-8 * np.abs(want).max())


@pytest.mark.parametrize("world", [2, 3])
def test_slab_lbfgs_thread_ranks_follow_the_undivided_run(world):
    from odil_amd.slab_solvers import run_threads

    N, maxiter, m = 8, 8, 5
    rhs = make_rhs(N, world)
    results = run_threads(world, lambda rank, comm: lbfgs_rank(rank, world, comm, N, rhs, maxiter, m))
    check_lbfgs(results, undivided_lbfgs(N, world, rhs, maxiter, m), world)


def gloo_worker(rank, world, N, maxiter, m, port, out):
    from odil_amd.slab import TorchDistComm

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = lbfgs_rank(rank, world, TorchDistComm(rank, world), N, make_rhs(N, world), maxiter, m)
        torch.save(res, os.path.join(out, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_slab_lbfgs_two_gloo_ranks_follow_the_undivided_run(tmp_path):
    world, N, maxiter, m = 2, 8, 6, 4
    port = 29500 + (os.getpid() + 77) % 2000
    mp.spawn(gloo_worker, args=(world, N, maxiter, m, port, str(tmp_path)), nprocs=world, join=True)
    results = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=False) for r in range(world)]
    check_lbfgs(results, undivided_lbfgs(N, world, make_rhs(N, world), maxiter, m), world)


def newton_rank(rank, world, comm, N, rhs, maxiter, nz=None):
    import slab_oracle_ops

    from odil_amd import slab_solvers

    slab_solvers.hip_ops = slab_oracle_ops
    run = slab_solvers.SlabPoissonNewtonCG(N, rank, world, dtype=torch.float64, device=torch.device("cpu"),
                                           rhs_global=torch.from_numpy(rhs), nz=nz)
    loss0, loss1 = run.step(comm, maxiter=maxiter, tol=1e-13)
    return loss0, loss1, dict(run.status), run.owned(run.u).clone().numpy()


@pytest.mark.parametrize("world", [2, 4])
def test_slab_newton_cg_step_solves_the_undivided_problem(world):
    from odil_amd.slab import LocalComm
    from odil_amd.slab_solvers import run_threads

    N = 4
    cshape = (N * world, N, N)
    dw = (1.0 / N,) * 3
    ref_u = np.random.default_rng(3).standard_normal(cshape)
    rhs = onp.poisson_discrete_rhs(ref_u, dw)  # f(u) = Lap(u) - rhs is linear: one exact step lands on ref_u
    results = run_threads(world, lambda rank, comm: newton_rank(rank, world, comm, N, rhs, 400))
    for r, (loss0, loss1, status, u) in enumerate(results):
        assert abs(loss0 - np.mean(rhs**2)) <= 1e-12 * np.mean(rhs**2)
        assert loss1 < 1e-16 * loss0 and status["niter"] < 400
        np.testing.assert_allclose(u, ref_u[r * N:(r + 1) * N], rtol=0, atol=1e-8)
    # the undivided box through the same class (one rank holding every plane): the same iteration count and solution
    loss0, loss1, status, u = newton_rank(0, 1, LocalComm(), N, rhs, 400, nz=N * world)
    assert status["niter"] == results[0][2]["niter"]
    np.testing.assert_allclose(u, np.concatenate([res[3] for res in results]), rtol=0, atol=1e-10)


def test_thread_ranks_report_the_first_error_and_do_not_hang():
    from odil_amd.slab_solvers import run_threads

    def body(rank, comm):
        comm.exchange("gather", torch.zeros(1), None)
        if rank == 1:
            raise ValueError("rank 1 failed")
        comm.exchange("gather", torch.zeros(1), None)  # the others wait here: the broken barrier releases them
        return rank

    with pytest.raises(ValueError, match="rank 1 failed"):
        run_threads(3, body)


def undivided_traced_lbfgs(which, world, nx_rank, maxiter, m):
    """L-BFGS-B (the same `lbfgsb_minimize`, NumPy vectors) on the UNDIVIDED problem evaluated by the generic oracle."""
    from test_lbfgs_host_logic import NumpyVectors
    from test_slab_traced_cpu import make_problem

    from odil_amd.optimizer import lbfgsb_minimize
    from oracle import odil_generic as og

    problem, state = make_problem(which, world, nx_rank)
    geom = og.Geometry.of(problem.domain)
    fields = og.fields_of_state(problem.domain, state)

    def get():
        out = []
        for f in fields.values():
            out += list(f["terms"]) if f["kind"] == "mg" else (
                [f["array"]] if f["kind"] in ("field", "array") else list(f["weights"]) + list(f["biases"]))
        return out

    def put(x):
        k = 0
        for f in fields.values():
            if f["kind"] == "mg":
                f["terms"] = x[k:k + len(f["terms"])]
                k += len(f["terms"])
            elif f["kind"] in ("field", "array"):
                f["array"] = x[k]
                k += 1
            else:
                nw, nb = len(f["weights"]), len(f["biases"])
                f["weights"], f["biases"] = x[k:k + nw], x[k + nw:k + nw + nb]
                k += nw + nb

    x0 = [np.asarray(a, dtype=np.float64) for a in get()]
    shapes, sizes = [a.shape for a in x0], [a.size for a in x0]
    split = lambda v: [p.reshape(s) for p, s in zip(np.split(v, np.cumsum(sizes)[:-1]), shapes)]

    def fun(v):
        put(split(v))
        loss, grads = og.eval_loss_grad(problem.operator, geom, fields, problem.extra, tracers=problem.tracers)[:2]
        return float(loss), np.concatenate([np.asarray(g).ravel() for g in grads])

    x = np.concatenate([a.ravel() for a in x0])
    res = lbfgsb_minimize(x, fun, NumpyVectors(x.size, m), maxiter, m=m)
    return res, split(x)


def traced_lbfgs_rank(rank, world, comm, which, nx_rank, maxiter, m):
    import slab_oracle_ops
    import slab_traced_double
    from slab_oracle_vectors import TorchCpuTailVectors
    from test_slab_traced_cpu import local_extra, make_problem

    from odil_amd import slab_traced
    from odil_amd.slab_solvers import SlabTracedLbfgs

    slab_traced.hip_ops = slab_oracle_ops
    problem, state = make_problem(which, world, nx_rank)
    run = slab_traced.SlabTracedAdam(problem, state, rank, world, axis=1, lr=0.01, device=torch.device("cpu"),
                                     kernels=slab_traced_double.make_kernels(local_extra))
    drv = SlabTracedLbfgs(run)
    res = drv.minimize(comm, maxiter, m=m, vectors=TorchCpuTailVectors(drv.n, m, comm, drv.n_own, world))
    return res, [a.clone().numpy() for a in run.owned_arrays()], drv.n_own, drv.n


@pytest.mark.parametrize("which,world,nx_rank", [("veltracer", 2, 8), ("veltracer", 4, 2), ("heat2d", 2, 8),
                                                  ("veltracer-factors", 2, 8)])
def test_slab_traced_lbfgs_thread_ranks_follow_the_undivided_run(which, world, nx_rank):
    """Any traced operator under L-BFGS-B on the slabs: (4, 2) leaves 2, 1, 0.5 cells of x per rank on the three levels --
    the coarser two are held whole by every rank and count once in the reductions; heat2d: network parameters likewise."""
    from odil_amd.slab_solvers import run_threads

    maxiter, m = 6, 4
    results = run_threads(world, lambda rank, comm: traced_lbfgs_rank(rank, world, comm, which, nx_rank, maxiter, m))
    res_ref, x_ref = undivided_traced_lbfgs(which, world, nx_rank, maxiter, m)
    for r, (res, owned, n_own, n) in enumerate(results):
        assert res["nit"] == res_ref["nit"] and res["funcalls"] == res_ref["funcalls"]
        assert abs(res["f"] - res_ref["f"]) <= 1e-9 * abs(res_ref["f"])
        if (world, nx_rank) == (4, 2) or which == "heat2d":
            assert n > n_own  # a replicated tail exists
        for i, ref in enumerate(x_ref):
            got = owned[i]
            if got.shape == ref.shape:
                want = ref
            else:
                k = ref.shape[1] // world
                want = ref[:, r * k:(r + 1) * k]
            assert got.shape == want.shape
            assert np.max(np.abs(got - want)) <= 1e-8 * max(1.0, np.max(np.abs(want))), (i, r)


def gmg_rank(rank, world, comm, N, rhs, nz=None, tol=1e-13, agg_cells=0):
    import slab_oracle_ops

    from odil_amd import slab_solvers

    slab_solvers.hip_ops = slab_oracle_ops
    run = slab_solvers.SlabPoissonNewtonGMG(N, rank, world, dtype=torch.float64, device=torch.device("cpu"),
                                            rhs_global=torch.from_numpy(rhs), nz=nz, agg_cells=agg_cells)
    loss0, loss1 = run.step(comm, maxiter=40, tol=tol)
    return loss0, loss1, dict(run.status), run.owned(run.u).clone().numpy()


@pytest.mark.parametrize("world,N,nz", [(2, 8, 8), (4, 8, 4), (3, 8, 8), (2, 16, 16)])
def test_slab_newton_multigrid_step_solves_the_undivided_problem(world, N, nz):
    """The slab form of the geometric multigrid (two to three slab levels, the rest agglomerated) lands on the exact
    discrete solution of the UNDIVIDED box -- the Newton iterate of the reference's direct solve -- in a dozen cycles,
    with the same cycle count as one rank holding every plane."""
    from odil_amd.slab import LocalComm
    from odil_amd.slab_solvers import run_threads

    cshape = (nz * world, N, N)
    dw = (1.0 / N,) * 3
    ref_u = np.random.default_rng(3).standard_normal(cshape)
    rhs = onp.poisson_discrete_rhs(ref_u, dw)
    results = run_threads(world, lambda rank, comm: gmg_rank(rank, world, comm, N, rhs, nz=nz))
    scale = np.abs(ref_u).max()
    for r, (loss0, loss1, status, u) in enumerate(results):
        assert abs(loss0 - np.mean(rhs**2)) <= 1e-12 * np.mean(rhs**2)
        assert status["converged"] and status["niter"] <= 20, status
        assert loss1 < 1e-18 * loss0
        np.testing.assert_allclose(u, ref_u[r * nz:(r + 1) * nz], rtol=0, atol=1e-9 * scale)
    # every rank took the same decisions
    assert len({res[2]["niter"] for res in results}) == 1
    # one rank holding every plane (no exchange at all): the same cycle count, the same solution to round-off
    loss0, loss1, status, u = gmg_rank(0, 1, LocalComm(), N, rhs, nz=nz * world)
    # (its hierarchy is deeper -- a rank with few planes agglomerates earlier --, so the counts may differ by a few cycles)
    assert abs(status["niter"] - results[0][2]["niter"]) <= 3
    np.testing.assert_allclose(u, np.concatenate([res[3] for res in results]), rtol=0, atol=1e-9 * scale)


@pytest.mark.parametrize("tol", [1e-4, 1e-6])
def test_slab_newton_multigrid_reports_the_finest_level_residual(tol):
    """With a LOOSE tolerance the step must stop on the FINEST level's residual (the cycle's recursion once overwrote it
    with the deepest slab level's: `converged` after one cycle at a true relative residual of 0.08): the loss after the
    step, sqrt(loss1 / loss0) = |A delta + f| / |f|, is then of the order of the tolerance, and the reported residual
    bounds it (it belongs to the pre-smoothed iterate: slightly pessimistic)."""
    from odil_amd.slab_solvers import run_threads

    world, N, nz = 2, 32, 32
    ref_u = np.random.default_rng(5).standard_normal((nz * world, N, N))
    rhs = onp.poisson_discrete_rhs(ref_u, (1.0 / N,) * 3)
    results = run_threads(world, lambda rank, comm: gmg_rank(rank, world, comm, N, rhs, nz=nz, tol=tol))
    for loss0, loss1, status, u in results:
        true_rel = float(np.sqrt(loss1 / loss0))
        assert status["converged"] and status["residual"] <= tol
        assert true_rel <= 1.5 * tol, (true_rel, status)
        assert true_rel <= 1.5 * status["residual"], (true_rel, status)
        assert status["niter"] >= 3  # (one cycle contracts by ~0.1: 1e-4 cannot be reached in one)


def gmg_gloo_worker(rank, world, N, nz, port, out):
    from odil_amd.slab import TorchDistComm

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cshape = (nz * world, N, N)
        ref_u = np.random.default_rng(3).standard_normal(cshape)
        rhs = onp.poisson_discrete_rhs(ref_u, (1.0 / N,) * 3)
        res = gmg_rank(rank, world, TorchDistComm(rank, world), N, rhs, nz=nz)
        torch.save(res, os.path.join(out, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_slab_newton_multigrid_two_gloo_ranks(tmp_path):
    world, N, nz = 2, 8, 8
    port = 29500 + (os.getpid() + 311) % 2000
    mp.spawn(gmg_gloo_worker, args=(world, N, nz, port, str(tmp_path)), nprocs=world, join=True)
    ref_u = np.random.default_rng(3).standard_normal((nz * world, N, N))
    for r in range(world):
        loss0, loss1, status, u = torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=False)
        assert status["converged"] and loss1 < 1e-18 * loss0
        np.testing.assert_allclose(u, ref_u[r * nz:(r + 1) * nz], rtol=0, atol=1e-9 * np.abs(ref_u).max())


# ---- the variable-coefficient multigrid on the slabs (slab_solvers.SlabStencilGMG) ------------------------------------------
def _diffusion_global(shape, seed=3, reaction=1.0):
    """-div(k grad u) + r u on the global box: smooth k (1 : 20), zero-Dirichlet walls half a cell off on every side."""
    rng = np.random.default_rng(seed)
    grids = np.meshgrid(*[(np.arange(n) + 0.5) / n for n in shape], indexing="ij")
    k = 1.0 + 19.0 * np.prod([np.sin(np.pi * g * rng.integers(1, 3)) ** 2 for g in grids], axis=0)
    off, diag = [], reaction * rng.uniform(0.5, 1.0, shape)
    for a in range(3):
        h2 = (1.0 / shape[a]) ** 2
        km, kp = 0.5 * (k + np.roll(k, 1, axis=a)) / h2, 0.5 * (k + np.roll(k, -1, axis=a)) / h2
        idx = np.arange(shape[a]).reshape([-1 if j == a else 1 for j in range(3)])
        diag = diag + np.where(idx == 0, 2.0 * k / h2, km) + np.where(idx == shape[a] - 1, 2.0 * k / h2, kp)
        off += [-np.where(idx == 0, 0.0, km), -np.where(idx == shape[a] - 1, 0.0, kp)]
    return [diag] + off


def stencil_gmg_rank(rank, world, comm, coeffs, b, nz, tol=1e-10, agg_cells=0):
    import slab_oracle_ops

    from odil_amd import slab_solvers

    own = slice(rank * nz, (rank + 1) * nz)
    c = torch.from_numpy(np.ascontiguousarray(np.stack([a[own] for a in coeffs])))
    run = slab_solvers.SlabStencilGMG(c, rank, world, ops=slab_oracle_ops, agg_cells=agg_cells)
    x = run.solve(comm, torch.from_numpy(np.ascontiguousarray(b[own])), tol=tol, maxiter=60)
    return x.numpy(), dict(run.status), len(run.mlv)


@pytest.mark.parametrize("world,shape", [(2, (16, 8, 8)), (4, (16, 8, 8)), (2, (32, 16, 16)), (3, (24, 8, 16))])
def test_slab_variable_coefficient_multigrid_solves_the_undivided_system(world, shape):
    """The slab form of the variable-coefficient cycle (ghost-extended coefficient arrays, coarse operators formed on the
    extended arrays, the rest agglomerated) lands on the solution of the UNDIVIDED system -- dense solve of the assembled
    operator -- and needs the cycles of the undivided NumPy hierarchy (+- 3: a rank with few planes agglomerates earlier)."""
    import stencil_gmg_np as sg

    from odil_amd.slab_solvers import run_threads

    coeffs = _diffusion_global(shape)
    rng = np.random.default_rng(9)
    b = rng.standard_normal(shape)
    n = int(np.prod(shape))
    if n <= 4096:
        amat = np.stack([sg.apply(coeffs, e.reshape(shape)).ravel() for e in np.eye(n)], axis=1)
        want = np.linalg.solve(amat, b.ravel()).reshape(shape)
    else:  # (the undivided NumPy cycles, converged)
        levels = sg.hierarchy(coeffs)
        want = np.zeros(shape)
        for _ in range(80):
            want = sg.vcycle(levels, 0, want, b, top2=True)
    nz = shape[0] // world
    results = run_threads(world, lambda rank, comm: stencil_gmg_rank(rank, world, comm, coeffs, b, nz))
    scale = np.abs(want).max()
    for r, (x, status, nlev) in enumerate(results):
        assert status["converged"] and status["niter"] <= 45, status
        np.testing.assert_allclose(x, want[r * nz:(r + 1) * nz], rtol=0, atol=1e-7 * scale)
    assert len({res[1]["niter"] for res in results}) == 1
    # the undivided hierarchy's cycle count (two cycles on its first coarse level where the slab hierarchy has them)
    levels = sg.hierarchy(coeffs)
    x, it = np.zeros(shape), 0
    while np.linalg.norm(b - sg.apply(coeffs, x)) > 1e-10 * np.linalg.norm(b) and it < 60:
        x = sg.vcycle(levels, 0, x, b, top2=results[0][2] > 2)
        it += 1
    assert abs(it - results[0][1]["niter"]) <= 4, (it, results[0][1])


def stencil_gmg_gloo_worker(rank, world, shape, port, out):
    from odil_amd.slab import TorchDistComm

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        coeffs = _diffusion_global(shape)
        b = np.random.default_rng(9).standard_normal(shape)
        res = stencil_gmg_rank(rank, world, TorchDistComm(rank, world), coeffs, b, shape[0] // world)
        torch.save(res, os.path.join(out, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_slab_variable_coefficient_multigrid_two_gloo_ranks(tmp_path):
    import stencil_gmg_np as sg

    world, shape = 2, (16, 8, 8)
    port = 29500 + (os.getpid() + 577) % 2000
    mp.spawn(stencil_gmg_gloo_worker, args=(world, shape, port, str(tmp_path)), nprocs=world, join=True)
    coeffs = _diffusion_global(shape)
    b = np.random.default_rng(9).standard_normal(shape)
    n = int(np.prod(shape))
    amat = np.stack([sg.apply(coeffs, e.reshape(shape)).ravel() for e in np.eye(n)], axis=1)
    want = np.linalg.solve(amat, b.ravel()).reshape(shape)
    nz = shape[0] // world
    for r in range(world):
        x, status, _ = torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=False)
        assert status["converged"]
        np.testing.assert_allclose(x, want[r * nz:(r + 1) * nz], rtol=0, atol=1e-8 * np.abs(want).max())


def test_slab_multigrids_agglomerate_small_levels_by_default():
    """With the default `agg_cells` a level of <= 32^3 cells per rank is not a slab level (four exchanges of ~65 us per level
    and cycle) but part of the agglomerated box: (64, 32, 32) on two ranks keeps ONE slab level, and both cycles still land
    on the undivided solution."""
    import stencil_gmg_np as sg

    from odil_amd.slab_solvers import run_threads

    world, shape = 2, (64, 32, 32)
    nz = shape[0] // world
    coeffs = _diffusion_global(shape)
    b = np.random.default_rng(9).standard_normal(shape)
    results = run_threads(world, lambda rank, comm: stencil_gmg_rank(rank, world, comm, coeffs, b, nz, agg_cells=32**3))
    levels = sg.hierarchy(coeffs)
    want = np.zeros(shape)
    for _ in range(80):
        want = sg.vcycle(levels, 0, want, b, top2=True)
    for r, (x, status, nlev) in enumerate(results):
        assert nlev == 1 and status["converged"], (nlev, status)
        np.testing.assert_allclose(x, want[r * nz:(r + 1) * nz], rtol=0, atol=1e-7 * np.abs(want).max())
    N = 32
    ref_u = np.random.default_rng(3).standard_normal(shape)
    rhs = onp.poisson_discrete_rhs(ref_u, (1.0 / N,) * 3)
    results = run_threads(world, lambda rank, comm: gmg_rank(rank, world, comm, N, rhs, nz=nz, agg_cells=32**3))
    for r, (loss0, loss1, status, u) in enumerate(results):
        assert status["converged"] and "1 slab levels" in status["method"], status
        np.testing.assert_allclose(u, ref_u[r * nz:(r + 1) * nz], rtol=0, atol=1e-8 * np.abs(ref_u).max())
