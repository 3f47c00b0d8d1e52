"""The quasi-Newton / Newton drivers of the slab path (odil_amd/slab_solvers.py) on the HIP kernels, ranks emulated as
threads on one GPU: L-BFGS-B on the decomposed domain follows the undivided oracle run iterate for iterate; the matrix-free
CG Newton step solves the undivided problem."""

import os

import numpy as np
import pytest
import torch

from oracle import odil_np as onp

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("world", [2, 4])
def test_slab_lbfgs_emulated_ranks_follow_the_undivided_oracle_run(world):
    from odil_amd.slab_solvers import SlabPoissonLbfgs, run_threads

    N, maxiter, m = 16, 10, 6
    dev = torch.device("cuda:0")
    rhs = make_rhs(N, world)

    def body(rank, comm):
        torch.cuda.set_device(dev)
        run = SlabPoissonLbfgs(N, rank, world, dtype=torch.float64, device=dev, rhs_global=torch.from_numpy(rhs))
        res = run.minimize(comm, maxiter, m=m)
        return res, [w.cpu().numpy() for w in run.owned_levels()]

    results = run_threads(world, body)
    res_ref, x_ref = undivided_lbfgs(N, world, rhs, maxiter, m)
    for r, (res, owned) in enumerate(results):
        assert res["nit"] == res_ref["nit"] and res["funcalls"] == res_ref["funcalls"]
        assert abs(res["f"] - res_ref["f"]) <= 1e-9 * abs(res_ref["f"])
        for lvl, want in enumerate(x_ref):
            nz = want.shape[0] // world
            np.testing.assert_allclose(owned[lvl], want[r * nz:(r + 1) * nz], rtol=0, atol=1e-8 * np.abs(want).max())


def test_slab_lbfgs_one_rank_equals_the_single_gpu_optimizer():
    """World 1 through the slab driver == `LbfgsVectors` + the undivided kernels (the reductions are the same launches)."""
    from odil_amd import ops
    from odil_amd.optimizer import LbfgsVectors, lbfgsb_minimize
    from odil_amd.slab import LocalComm
    from odil_amd.slab_solvers import SlabPoissonLbfgs

    N, maxiter, m = 16, 8, 5
    dev = torch.device("cuda:0")
    rhs = make_rhs(N, 1)
    run = SlabPoissonLbfgs(N, 0, 1, dtype=torch.float64, device=dev, rhs_global=torch.from_numpy(rhs))
    res = run.minimize(LocalComm(), maxiter, m=m)
    shapes = onp.mg_cshapes((N, N, N))
    sizes = [int(np.prod(s)) for s in shapes]
    h2 = [(1.0 / N) ** 2] * 3
    trhs = torch.from_numpy(rhs).to(dev)
    x = torch.zeros(sum(sizes), dtype=torch.float64, device=dev)

    def fg(xflat):
        terms = [t.view(s) for t, s in zip(xflat.split(sizes), shapes)]
        u = ops.mg_synth(terms, "ccc")
        fu, loss = ops.poisson_residual(u, trhs, h2)
        gu = ops.poisson_adjoint(fu, h2, 2.0 / fu.numel())
        return loss, torch.cat([g.reshape(-1) for g in ops.mg_synth_adj(gu, shapes, "ccc")])

    ref = lbfgsb_minimize(x, fg, LbfgsVectors(x.numel(), m, dev), maxiter, m=m)
    assert res["nit"] == ref["nit"] and res["funcalls"] == ref["funcalls"]
    assert abs(res["f"] - ref["f"]) <= 1e-10 * abs(ref["f"])
    got = torch.cat([w.reshape(-1) for w in run.owned_levels()])
    assert float((got - x).abs().max()) <= 1e-9 * float(x.abs().max())


@pytest.mark.parametrize("world", [2, 4])
def test_slab_newton_cg_emulated_ranks_solve_the_undivided_problem(world):
    from odil_amd.slab_solvers import SlabPoissonNewtonCG, run_threads

    N = 8
    dev = torch.device("cuda:0")
    cshape = (N * world, N, N)
    dw = (1.0 / N,) * 3
    ref_u = np.random.default_rng(3).standard_normal(cshape)
    rhs = onp.poisson_discrete_rhs(ref_u, dw)

    def body(rank, comm):
        torch.cuda.set_device(dev)
        run = SlabPoissonNewtonCG(N, rank, world, dtype=torch.float64, device=dev, rhs_global=torch.from_numpy(rhs))
        loss0, loss1 = run.step(comm, maxiter=3000, tol=1e-13)
        return loss0, loss1, dict(run.status), run.owned(run.u).cpu().numpy()

    for r, (loss0, loss1, status, u) in enumerate(run_threads(world, body)):
        assert abs(loss0 - np.mean(rhs**2)) <= 1e-12 * np.mean(rhs**2)
        assert loss1 < 1e-14 * loss0 and status["niter"] < 3000
        np.testing.assert_allclose(u, ref_u[r * N:(r + 1) * N], rtol=0, atol=1e-7)


@pytest.mark.parametrize("world,N,nz", [(2, 16, 16), (4, 32, 8), (3, 64, 64)])
def test_slab_newton_multigrid_emulated_ranks_solve_the_undivided_problem(world, N, nz):
    """`SlabPoissonNewtonGMG` on the HIP kernels, ranks as threads on one GPU: the Newton step by slab-decomposed V-cycles
    lands on the exact discrete solution of the undivided box (the iterate of the reference's direct solve,
    linsolver.py:17-26) in a dozen cycles -- where the unpreconditioned CG of the test above needs hundreds -- and on the
    iterate of the single-GPU `gmg.PoissonGMG` for the same box."""
    from odil_amd import gmg
    from odil_amd.slab_solvers import SlabPoissonNewtonGMG, run_threads

    dev = torch.device("cuda:0")
    cshape = (nz * world, N, N)
    dw = (1.0 / N,) * 3
    gen = torch.Generator().manual_seed(3)
    ref_u = torch.randn(cshape, generator=gen, dtype=torch.float64)
    from odil_amd import ops

    h2 = [np.float64(1.0 / N) ** 2] * 3
    rhs, _ = ops.poisson_residual(ref_u.to(dev), torch.zeros(cshape, dtype=torch.float64, device=dev), h2)
    rhs_host = rhs.cpu()

    def body(rank, comm):
        torch.cuda.set_device(dev)
        run = SlabPoissonNewtonGMG(N, rank, world, dtype=torch.float64, device=dev, rhs_global=rhs_host, nz=nz, agg_cells=0)
        loss0, loss1 = run.step(comm, maxiter=40, tol=1e-13)
        return loss0, loss1, dict(run.status), run.owned(run.u).cpu()

    results = run_threads(world, body)
    scale = float(ref_u.abs().max())
    for r, (loss0, loss1, status, u) in enumerate(results):
        assert status["converged"] and status["niter"] <= 20, status
        assert loss1 < 1e-18 * loss0
        assert float((u - ref_u[r * nz:(r + 1) * nz]).abs().max()) <= 1e-8 * scale  # (cond(A) ~ 1e4 at N = 64: the residual floor)
    if all(s % 2 == 0 for s in cshape):
        solver = gmg.PoissonGMG(cshape, h2, torch.float64, dev)
        x = solver.solve(rhs, tol=1e-13, maxiter=40)  # A x = rhs: x = ref_u
        whole = torch.cat([res[3] for res in results])
        assert float((x.cpu() - whole).abs().max()) <= 1e-8 * scale


@pytest.mark.parametrize("world,shape", [(2, (32, 16, 16)), (4, (64, 32, 32)), (3, (48, 64, 32)), (2, (256, 128, 128))])
def test_slab_variable_coefficient_multigrid_emulated_ranks(world, shape):
    """`SlabStencilGMG` on the HIP kernels, ranks as threads on one GPU: variable-coefficient diffusion with a reaction
    term on the box cut along axis 0 -- the slab-decomposed cycles (coefficient ghosts exchanged at set-up, coarse operators
    on the extended arrays, sweeps in pairs on the large levels, the bottom agglomerated) reach the iterate of the single-GPU
    `gmg.StencilGMG` on the undivided box."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_slab_solvers_cpu import _diffusion_global

    from odil_amd import gmg
    from odil_amd.slab_solvers import SlabStencilGMG, run_threads

    dev = torch.device("cuda:0")
    coeffs = torch.as_tensor(np.stack(_diffusion_global(shape))).to(dev)
    b = torch.as_tensor(np.random.default_rng(9).standard_normal(shape)).to(dev)
    nz = shape[0] // world

    def body(rank, comm):
        torch.cuda.set_device(dev)
        run = SlabStencilGMG(coeffs[:, rank * nz:(rank + 1) * nz].contiguous(), rank, world, pair_min_cells=32**3,
                             agg_cells=0 if shape[0] < 256 else 32**3)
        x = run.solve(comm, b[rank * nz:(rank + 1) * nz].contiguous(), tol=1e-10, maxiter=60)
        return x.cpu(), dict(run.status)

    results = run_threads(world, body)
    solver = gmg.StencilGMG(coeffs)
    st = dict()
    want = solver.solve(b, tol=1e-11, maxiter=60, status=st).cpu()
    assert st["converged"]
    scale = float(want.abs().max())
    for r, (x, status) in enumerate(results):
        assert status["converged"], status
        # (the single-GPU solver starts by nested iteration and may hand over to GCR; the slab cycles are plain V-cycles
        # from zero: up to twice the passes on the hardest of these boxes)
        assert status["niter"] <= 2 * st["niter"] + 6, (status, st)
        assert float((x - want[r * nz:(r + 1) * nz]).abs().max()) <= 1e-7 * scale, (r, status)
    assert len({res[1]["niter"] for res in results}) == 1


@pytest.mark.parametrize("which,world,nx_rank", [("veltracer", 2, 16), ("veltracer3d", 4, 4), ("heat2d", 2, 16)])
def test_slab_traced_lbfgs_emulated_ranks_follow_the_single_gpu_optimizer(which, world, nx_rank):
    """L-BFGS-B of a traced operator on the slabs (generated kernels in slab mode, ranks as threads on one GPU; (4, 4):
    the coarsest level is held whole by every rank; heat2d: network parameters) against `LbfgsbOptimizer` on the
    undivided traced path: the same iteration / evaluation counts, the unknowns to 1e-8."""
    import argparse

    import odil_amd as odil
    from odil_amd.slab_solvers import SlabTracedLbfgs, run_threads
    from odil_amd.slab_traced import SlabTracedAdam
    from test_slab_gpu import _traced_problem

    maxiter, m = 8, 5
    problem, state = _traced_problem(which, world, 1, nx_rank)
    dev = torch.device("cuda:0")

    def body(rank, comm):
        torch.cuda.set_device(dev)
        run = SlabTracedAdam(problem, state, rank, world, axis=1, lr=0.01)
        drv = SlabTracedLbfgs(run)
        res = drv.minimize(comm, maxiter, m=m)
        return res, [a.clone() for a in run.owned_arrays()], drv.n - drv.n_own

    results = run_threads(world, body)
    a = argparse.Namespace(epoch_start=0, epochs=maxiter, lr=0.01, bfgs_m=m, bfgs_pgtol=None, bfgs_maxls=None,
                           adam_epsilon=None, adam_beta_1=None, adam_beta_2=None, callback_update_state=0)
    odil.util.set_log_file(open(os.devnull, "w"))
    _, optinfo = None, None
    odil.util.optimize_grad(a, "lbfgsb", problem, state, None)
    assert problem._traced is not None
    want = problem.domain.arrays_from_state(state)
    want_loss = float(problem.eval_loss_grad(state)[0])
    for r, (res, owned, ntail) in enumerate(results):
        assert res["nit"] == maxiter
        assert abs(res["f"] - want_loss) <= 1e-9 * abs(want_loss), (res["f"], want_loss)
        if which == "heat2d" or (world, nx_rank) == (4, 4):
            assert ntail > 0
        for i, (got, ref) in enumerate(zip(owned, want)):
            if got.shape != ref.shape:
                n = ref.shape[1] // world
                ref = ref[:, r * n:(r + 1) * n]
            assert float((got - ref).abs().max()) <= 1e-8 * max(1.0, float(ref.abs().max())), (r, i)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_restricted_residuals_with_the_norm_of_a_plane_range(dtype):
    """`odil_poisson_residual_restrict_slab` / `odil_stencil_var_residual_restrict_slab`: the coarse right-hand side is the
    whole-array kernel's bit for bit; the norm counts the planes z0 <= z < z1 only (a rank's own planes of a ghost-extended
    array) and equals the plain residual's sum over them."""
    from test_slab_solvers_cpu import _diffusion_global

    from odil_amd import ops

    dev = torch.device("cuda:0")
    shape, z0, z1 = (20, 16, 24), 2, 18
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(shape, generator=gen, dtype=torch.float64).to(dev, dtype)
    b = torch.randn(shape, generator=gen, dtype=torch.float64).to(dev, dtype)
    h2 = [1.0 / 16**2] * 3
    cshape = tuple(n // 2 for n in shape)
    tol = 1e-13 if dtype == torch.float64 else 1e-5
    whole, part = torch.empty(cshape, dtype=dtype, device=dev), torch.empty(cshape, dtype=dtype, device=dev)
    l0, l1 = torch.zeros((), dtype=dtype, device=dev), torch.zeros((), dtype=dtype, device=dev)
    ops.poisson_residual_restrict(x, b, h2, -0.125, whole, l0)
    ops.poisson_residual_restrict(x, b, h2, -0.125, part, l1, zrange=(z0, z1), denom=3.0)
    assert torch.equal(whole, part)
    fu, _ = ops.poisson_residual(x, b, h2)
    want = float((fu[z0:z1].double() ** 2).sum()) / 3.0
    assert abs(float(l1) - want) <= tol * want
    assert abs(float(l0) - float((fu.double() ** 2).mean())) <= tol * float(l0)
    coeffs = torch.as_tensor(np.stack(_diffusion_global(shape))).to(dev, dtype).contiguous()
    ops.stencil_var_residual_restrict(coeffs, x, b, 0.125, whole, l0)
    ops.stencil_var_residual_restrict(coeffs, x, b, 0.125, part, l1, zrange=(z0, z1), denom=3.0)
    assert torch.equal(whole, part)
    r = ops.stencil_var_residual(coeffs, x, b)
    want = float((r[z0:z1].double() ** 2).sum()) / 3.0
    assert abs(float(l1) - want) <= tol * want
    assert abs(float(l0) - float((r.double() ** 2).mean())) <= tol * float(l0)
