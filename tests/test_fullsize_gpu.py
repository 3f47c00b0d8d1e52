"""Parity at the size BASELINE.json quotes the metric on (3-D Poisson 512^3, f64, 9 multigrid levels),
where the oracle cannot follow: size-independent properties of the same kernels the bench runs.

* transposes are transposes: <P c, g> = <c, P^T g>, <A u, f> = <u, A^T f>;
* every fusion of the epoch (last prolongation inside the residual, LDS-tiled transposes, stencil adjoint
  + first transposed prolongation + Adam in one launch) leaves the epoch bit-identical to the separate
  kernels: same loss after several epochs, same state;
* the loss of the first epochs equals the small-grid goldens' behaviour in kind: zero state -> loss =
  mean(rhs^2), and it is reproducible run to run (deterministic reductions)."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N = 512


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def dot(a, b):
    from odil_amd import ops

    return float(ops.dots(a.reshape(1, -1), b.reshape(-1))[0])


def test_transposes_at_full_size(dev):
    from odil_amd import ops

    g = torch.Generator(device=dev).manual_seed(3)
    fine, coarse = (N, N, N), (N // 2,) * 3
    c = torch.randn(coarse, dtype=torch.float64, device=dev, generator=g)
    gf = torch.randn(fine, dtype=torch.float64, device=dev, generator=g)
    pc = ops.interp_add(c, "ccc")
    ptg = ops.interp_adj(gf, "ccc", coarse)
    lhs, rhs = dot(pc, gf), dot(c, ptg)
    assert abs(lhs - rhs) < 1e-11 * max(abs(lhs), abs(rhs), 1.0)
    del pc, ptg, c
    h2 = [np.float64(1.0 / N) ** 2] * 3
    u = torch.randn(fine, dtype=torch.float64, device=dev, generator=g)
    au, _ = ops.poisson_residual(u, torch.zeros_like(u), h2)
    atf = ops.poisson_adjoint(gf, h2, 1.0)
    lhs, rhs = dot(au, gf), dot(u, atf)
    assert abs(lhs - rhs) < 1e-11 * max(abs(lhs), abs(rhs), 1.0)


def run_epochs(dev, monkeypatch, env, epochs=3, n=N):
    from odil_amd.poisson_path import PoissonMultigridAdam

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    run = PoissonMultigridAdam(3, n, dtype=torch.float64, device=dev)
    losses = []
    for _ in range(epochs):
        run.epoch()
        losses.append(run.last_loss())
    state = (run.x.clone(), run.m.clone(), run.v.clone())
    del run
    torch.cuda.empty_cache()
    return losses, state


def test_fused_epoch_equals_separate_kernels_at_full_size(dev, monkeypatch):
    fused, sf = run_epochs(dev, monkeypatch, {"ODIL_FUSE_TRANSPOSE": "1", "ODIL_SYNTH_RESIDUAL": "1", "ODIL_ADJ_TILE": "1"})
    again, sa = run_epochs(dev, monkeypatch, {"ODIL_FUSE_TRANSPOSE": "1", "ODIL_SYNTH_RESIDUAL": "1", "ODIL_ADJ_TILE": "1"})
    plain, sp = run_epochs(dev, monkeypatch, {"ODIL_FUSE_TRANSPOSE": "0", "ODIL_SYNTH_RESIDUAL": "0", "ODIL_ADJ_TILE": "0"})
    assert fused == again  # deterministic reductions: bit-reproducible
    assert fused == plain  # every fusion is bit-identical to the separate kernels
    for a, b, c in zip(sf, sa, sp):
        assert torch.equal(a, b) and torch.equal(a, c)
    # zero initial state: the first loss is mean(rhs^2) of the discrete right-hand side; then it moves
    assert fused[0] > 0 and fused[1] != fused[0] and np.isfinite(fused).all()


def test_fused_epoch_equals_separate_kernels_at_eight_times_the_full_size(dev, monkeypatch):
    """1024^3 f64 (1.07e9 cells, 10 levels, 8.6 GB per level-0 array, ~75 GB in all: what the 288 GB of an MI355X are for;
    element offsets beyond 2^30, byte offsets beyond 2^33): the fused epoch is still bit-identical to the separate kernels."""
    if torch.cuda.mem_get_info(dev)[0] < 110e9:
        pytest.skip("needs ~90 GB of free device memory")
    fused, sf = run_epochs(dev, monkeypatch, {"ODIL_FUSE_TRANSPOSE": "1", "ODIL_SYNTH_RESIDUAL": "1", "ODIL_FUSE_ADAM0": "1"}, n=1024)
    sums = [float(t.sum()) for t in sf]
    del sf
    torch.cuda.empty_cache()
    plain, sp = run_epochs(dev, monkeypatch, {"ODIL_FUSE_TRANSPOSE": "0", "ODIL_SYNTH_RESIDUAL": "0", "ODIL_FUSE_ADAM0": "0"}, n=1024)
    assert np.allclose(fused, plain, rtol=1e-15, atol=0) and np.isfinite(fused).all()  # (the loss: same terms, 1-ulp sums)
    assert sums == [float(t.sum()) for t in sp]
    assert abs(fused[0] - 651.1063812200) < 1e-6  # mean(rhs^2) of the discrete right-hand side of 'hat' at this size


def test_arrays_beyond_two_to_the_31_elements(dev):
    """1300^3 = 2.197e9 float cells (8.8 GB per array): residual, stencil adjoint, Adam, the reductions and both transfers
    index with 64 bits where it matters -- the planes at the END of the arrays (element offsets beyond 2^31) against a
    float64 evaluation of the same stencil, and the transposes as transposes."""
    from odil_amd import ops

    if torch.cuda.mem_get_info(dev)[0] < 80e9:
        pytest.skip("needs ~60 GB of free device memory")
    n = 1300
    h2 = [np.float32(1.0 / n) ** 2] * 3
    g = torch.Generator(device=dev).manual_seed(1)
    u = torch.randn((n, n, n), dtype=torch.float32, device=dev, generator=g)
    assert u.numel() > 2**31

    def lap(block):  # planes 1 .. k - 2 of `block`, cells two away from the walls, in float64
        d = block.double()
        c = d[1:-1, 2:-2, 2:-2]
        return ((d[2:, 2:-2, 2:-2] + d[:-2, 2:-2, 2:-2] - 2 * c) / float(h2[0]) + (d[1:-1, 3:-1, 2:-2] + d[1:-1, 1:-3, 2:-2] - 2 * c) / float(h2[1])
                + (d[1:-1, 2:-2, 3:-1] + d[1:-1, 2:-2, 1:-3] - 2 * c) / float(h2[2]))

    fu, _ = ops.poisson_residual(u, torch.zeros_like(u), h2)
    ga = ops.poisson_adjoint(fu, h2, 1.0)
    for z0 in (5, n - 12):
        for src, out in ((u, fu), (fu, ga)):
            ref = lap(src[z0 - 1:z0 + 4])
            assert float((out[z0:z0 + 3, 2:-2, 2:-2].double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    x, m, v = (torch.zeros_like(u) for _ in range(3))
    ops.adam_step(x.view(-1), m.view(-1), v.view(-1), ga.view(-1), 0.01, 0.1, 0.001, 1e-7)
    assert torch.equal(x[-1].abs() > 0, ga[-1] != 0) and bool(torch.isfinite(x[-1]).all())
    assert abs(float(ops.dots(u.view(1, -1), u.view(-1))[0]) / u.numel() - 1.0) < 1e-3
    del x, m, v, ga
    torch.cuda.empty_cache()
    c = torch.randn((n // 2,) * 3, dtype=torch.float32, device=dev, generator=g)
    pc = ops.interp_add(c, "ccc")
    lhs = float(ops.dots(pc.view(1, -1), fu.view(-1))[0])
    del pc
    rhs = float(ops.dots(c.view(1, -1), ops.interp_adj(fu, "ccc", tuple(c.shape)).view(-1))[0])
    assert abs(lhs - rhs) <= 2e-6 * abs(lhs)


def test_newton_step_and_lbfgs_at_full_size(dev):
    """BASELINE configs 4b and 2 through the public API: one Newton step of the 512^3 Poisson problem (no
    decomposition; geometric multigrid on the device) solves it -- loss drops from O(1e5) to rounding -- and
    L-BFGS-B on 1024^2 with the multigrid decomposition decreases the loss monotonically over its accepted
    iterations (Wolfe line search) and reproduces itself run to run."""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "examples", "poisson"))
    import odil_amd as odil
    import poisson

    odil.util.set_log_file(open(os.devnull, "w"))
    args = poisson.parse_args(["--ndim", "3", "--N", str(N), "--multigrid", "0", "--linsolver", "multigrid",
                               "--linsolver_tol", "1e-10"])
    problem, state = poisson.make_problem(args)
    loss0 = float(problem.eval_loss_grad(state)[0])
    args.epoch_start, args.epochs = 0, 1
    odil.util.optimize(args, "newton", problem, state, None)
    loss1 = float(problem.eval_loss_grad(state)[0])
    assert loss0 > 1.0 and loss1 < 1e-18 * loss0
    del problem, state
    torch.cuda.empty_cache()

    def lbfgs_losses():
        a = poisson.parse_args(["--ndim", "2", "--N", "1024"])
        prob, st = poisson.make_problem(a)
        a.epoch_start, a.epochs = 0, 30
        losses = []
        try:
            odil.util.optimize(a, "lbfgsb", prob, st, lambda s, e, p: losses.append(float(np.array(p["loss"]))))
        except odil.EarlyStopError:
            pass
        return losses

    la, lb = lbfgs_losses(), lbfgs_losses()
    assert la == lb and len(la) >= 30
    assert all(b < a for a, b in zip(la[1:], la[2:]))  # (entry 0 is the initial evaluation)
