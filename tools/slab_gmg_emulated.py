#!/usr/bin/env python3
"""The slab multigrid solvers with ranks emulated as threads on one GPU: time per cycle and rank, exchanges per cycle, against
the single-GPU solver on ONE rank's box (weak scaling: what a rank would cost if its exchanges were free is the single-GPU
cycle; what the emulation adds is the slab bookkeeping -- ghost planes, plane copies, thread hand-over).
    python3 tools/slab_gmg_emulated.py [world] [N] [poisson|varcoef]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from odil_amd import gmg  # noqa: E402
from odil_amd.slab import LocalComm  # noqa: E402
from odil_amd.slab_solvers import SlabPoissonNewtonGMG, SlabStencilGMG, run_threads  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
which = sys.argv[3] if len(sys.argv) > 3 else "poisson"
dev = torch.device("cuda:0")
torch.manual_seed(5)


class Counting:
    """Wraps a ThreadComm: counts point-to-point exchanges and collectives of one rank."""

    def __init__(self, comm):
        self.comm, self.halo, self.gather = comm, 0, 0

    def exchange(self, kind, *a):
        if kind == "gather":
            self.gather += 1
        else:
            self.halo += 1
        return self.comm.exchange(kind, *a)

    def __getattr__(self, name):
        return getattr(self.comm, name)


def coefficient_box(shape):
    """Seven arrays of div(k grad u) - c u with a smooth k on the unit-spaced box (walls: the missing neighbour's coupling
    stays on the diagonal -- a Dirichlet closure)."""
    z, y, x = [torch.linspace(0, 1, n, device=dev, dtype=torch.float64) for n in shape]
    k = 1.0 + 0.5 * torch.sin(6 * z)[:, None, None] * torch.cos(5 * y)[None, :, None] * torch.sin(4 * x)[None, None, :]
    c = torch.zeros((7,) + tuple(shape), dtype=torch.float64, device=dev)
    diag = torch.full(tuple(shape), 0.1, dtype=torch.float64, device=dev)
    for d in range(3):
        kk = k.movedim(d, 0)
        face = 0.5 * (kk[1:] + kk[:-1])
        lo = torch.zeros_like(kk)
        hi = torch.zeros_like(kk)
        lo[1:] = face
        hi[:-1] = face
        lo[0] = kk[0] * 2
        hi[-1] = kk[-1] * 2
        diag += (lo + hi).movedim(0, d)
        lo[0] = 0
        hi[-1] = 0
        c[1 + 2 * d] = -lo.movedim(0, d)
        c[2 + 2 * d] = -hi.movedim(0, d)
    c[0] = diag
    return c


if which == "poisson":
    rhs = torch.randn((world * N, N, N), dtype=torch.float64)

    def body(rank, comm):
        torch.cuda.set_device(dev)
        cc = Counting(comm)
        run = SlabPoissonNewtonGMG(N, rank, world, dtype=torch.float64, device=dev, rhs_global=rhs, nz=N)
        run.step(cc, maxiter=2, tol=1e-30)
        torch.cuda.synchronize()
        run.u.zero_()
        cc.halo = cc.gather = 0
        t0 = time.perf_counter()
        run.step(cc, maxiter=10, tol=1e-30)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, run.status, cc.halo, cc.gather

    if world == 1:  # the slab driver's own bookkeeping, no hand-over between threads
        out = [body(0, LocalComm())]
    else:
        out = run_threads(world, body)
    t = max(o[0] for o in out)
    st = out[0][1]
    print("{} emulated ranks of {}^3, Poisson: {:.2f} ms per cycle for all ranks = {:.2f} per rank; {} cycles, {}; per cycle "
          "and rank {:.1f} plane exchanges, {:.1f} collectives".format(
              world, N, 1e3 * t / st["niter"], 1e3 * t / st["niter"] / world, st["niter"], st["method"],
              out[0][2] / st["niter"], out[0][3] / st["niter"]))
    solver = gmg.PoissonGMG((N, N, N), [1.0 / N**2] * 3, torch.float64, dev)
    b = rhs[:N].to(dev).contiguous()
    solver.solve(b, tol=1e-30, maxiter=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s1 = dict()
    solver.solve(b, tol=1e-30, maxiter=10, status=s1, fmg=False)
    torch.cuda.synchronize()
    print("single-GPU PoissonGMG on one rank's box {}^3: {:.2f} ms per cycle ({} cycles)".format(
        N, 1e3 * (time.perf_counter() - t0) / max(s1.get("niter", 10), 1), s1.get("niter")))
else:
    shape = (world * N, N, N)
    coeffs = coefficient_box(shape)
    b = torch.randn(shape, dtype=torch.float64, device=dev)

    def body(rank, comm):
        torch.cuda.set_device(dev)
        cc = Counting(comm)
        run = SlabStencilGMG(coeffs[:, rank * N:(rank + 1) * N].contiguous(), rank, world)
        run.solve(cc, b[rank * N:(rank + 1) * N].contiguous(), tol=1e-30, maxiter=2)
        torch.cuda.synchronize()
        cc.halo = cc.gather = 0
        t0 = time.perf_counter()
        run.solve(cc, b[rank * N:(rank + 1) * N].contiguous(), tol=1e-30, maxiter=10)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, dict(run.status), cc.halo, cc.gather

    if world == 1:
        out = [body(0, LocalComm())]
    else:
        out = run_threads(world, body)
    t = max(o[0] for o in out)
    st = out[0][1]
    print("{} emulated ranks of {}^3, variable coefficients: {:.2f} ms per cycle for all ranks = {:.2f} per rank; residual {:.2e} "
          "after {} cycles, {}; per cycle and rank {:.1f} plane exchanges, {:.1f} collectives".format(
              world, N, 1e3 * t / st["niter"], 1e3 * t / st["niter"] / world, st["residual"], st["niter"], st["method"],
              out[0][2] / st["niter"], out[0][3] / st["niter"]))
    solver = gmg.StencilGMG(coeffs[:, :N].contiguous())
    b1 = b[:N].contiguous()
    solver.solve(b1, tol=1e-30, maxiter=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s1 = dict()
    solver.solve(b1, tol=1e-30, maxiter=10, status=s1, fmg=False)
    torch.cuda.synchronize()
    print("single-GPU StencilGMG on one rank's box {}^3: {:.2f} ms per cycle ({} cycles, residual {:.2e})".format(
        N, 1e3 * (time.perf_counter() - t0) / max(s1.get("niter", 10), 1), s1.get("niter"), s1.get("residual", np.nan)))
