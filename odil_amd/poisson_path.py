"""Device-resident driver of the reference's hot loop for the Poisson workload.

One `epoch()` = what `AdamNativeOptimizer.run` does per iteration
(reference src/odil/optimizer.py:331-336) for `examples/poisson/poisson.py`:
  loss_grad:  u = multigrid_to_regular(w)         core.py:245-263   -> mg_synth
              fu = Lap(u) - rhs, loss = mean(fu^2) poisson.py:89-113 -> poisson_residual
              grads = d loss / d w_l              core.py:1100       -> poisson_adjoint + mg_synth_adj
  step:       Adam update of every level array    optimizer.py:311-319 -> adam_step
The multigrid unknowns of all levels live in ONE packed device buffer (the layout of
`Domain.pack_state`, core.py:436-443), so the optimizer update is a single launch and
the level arrays are views.  Nothing crosses to the host inside an epoch; the loss stays
a device scalar until asked for.
"""

import math

import numpy as np
import torch

from . import ops
from .fused import PoissonEvaluator


def mg_cshapes(cshape, mg_axes=None, mg_nlvl=None):
    """Level cell-shapes fine->coarse (reference core.py:65-73)."""
    ndim = len(cshape)
    mg_axes = mg_axes or [True] * ndim
    nlvl_max = min(int(round(math.log2(n))) if ax else max(cshape) for n, ax in zip(cshape, mg_axes))
    nlvl = nlvl_max if mg_nlvl is None else min(mg_nlvl, nlvl_max)
    shapes = [tuple(n >> lvl if ax else n for n, ax in zip(cshape, mg_axes)) for lvl in range(nlvl)]
    for fine, coarse in zip(shapes[:-1], shapes[1:]):  # same requirement as the reference (core.py:75-90)
        for f, c, ax in zip(fine, coarse, mg_axes):
            if ax and f != 2 * c:
                raise ValueError("Expected equal '{}' and '{}' with cshapes={}: every extent must be divisible by 2^{}"
                                 .format(f, 2 * c, shapes, nlvl - 1))
    return shapes


def hat_reference(cshape, dtype, device):
    """Reference solution 'hat' on cell centres (reference examples/poisson/poisson.py:21-24).
    Synthetic-input generation only (torch elementwise); not part of the timed path."""
    xs = []
    for n in cshape:
        x = torch.linspace(0.0, 1.0, n + 1, dtype=torch.float64, device=device)[:-1]
        x = x + (x[1] - x[0]) * 0.5 if n > 1 else x
        xs.append(x)
    grids = torch.meshgrid(*xs, indexing="ij")
    u = torch.ones(cshape, dtype=torch.float64, device=device)
    for x in grids:
        u = u * ((1 - x) * x * 5)
    p = 5
    u = (u**p / (1 + u**p)) ** (1 / p)
    return u.to(dtype)


class PoissonMultigridAdam:
    def __init__(self, ndim, N, dtype=torch.float64, device=None, lr=0.005, beta_1=0.9, beta_2=0.999,
                 epsilon=1e-7, multigrid=True, rhs=None, ref_u=None):
        """ref_u: the reference solution as a device tensor (default: 'hat' formed on the device); rhs: the right-hand
        side (default: the discrete Laplacian of ref_u by the residual kernel, reference poisson.py:71-86)."""
        self.ndim, self.N, self.dtype, self.device = ndim, N, dtype, device
        self.loc = "c" * ndim
        cshape = (N,) * ndim
        self.cshape = cshape
        self.shapes = mg_cshapes(cshape) if multigrid else [cshape]
        self.nlvl = len(self.shapes)
        self.sizes = [math.prod(s) for s in self.shapes]
        self.local_cells = math.prod(cshape)
        self.global_cells = self.local_cells
        self.n_unknowns_local = sum(self.sizes)
        npdt = np.float64 if dtype == torch.float64 else np.float32
        self.npdt = npdt
        step = [(npdt(1) - npdt(0)) / n for n in cshape]  # core.py:199-200 in the domain dtype
        self.h2 = [s**2 for s in step]
        n = self.n_unknowns_local
        self.x = torch.zeros(n, dtype=dtype, device=device)
        self.m = torch.zeros(n, dtype=dtype, device=device)
        self.v = torch.zeros(n, dtype=dtype, device=device)
        self.w = [t.view(s) for t, s in zip(self.x.split(self.sizes), self.shapes)]
        self.mw = [t.view(s) for t, s in zip(self.m.split(self.sizes), self.shapes)]
        self.vw = [t.view(s) for t, s in zip(self.v.split(self.sizes), self.shapes)]
        # rhs = discrete Laplacian of the reference solution (poisson.py:71-86): same kernel, rhs = 0
        if ref_u is None:
            ref_u = hat_reference(cshape, dtype, device)
        assert tuple(ref_u.shape) == tuple(cshape) and ref_u.dtype == dtype
        self.ref_u = ref_u
        if rhs is None:
            rhs, _ = ops.poisson_residual(ref_u, torch.zeros_like(ref_u), self.h2)
        self.ev = PoissonEvaluator(cshape, self.shapes, rhs, self.h2, name="", dtype=dtype, device=device)
        self.g = self.ev.g
        self.loss = self.ev.loss
        self.lr, self.b1, self.b2, self.eps = npdt(lr), npdt(beta_1), npdt(beta_2), epsilon
        self.t = 0
        import os

        self.fuse_adam0 = bool(int(os.environ.get("ODIL_FUSE_ADAM0", 1)))

    def loss_grad(self, timers=None):
        loss, _ = self.ev.loss_grad_arrays(self.w, timers)
        return loss

    def epoch(self, timers=None):
        self.t += 1
        t = self.npdt(self.t)
        alpha = self.lr * np.sqrt(1 - self.b2**t) / (1 - self.b1**t)  # optimizer.py:313-315
        omb1, omb2 = 1 - self.b1, 1 - self.b2
        fuse = self.fuse_adam0 and self.nlvl > 1
        if fuse:
            # every level is updated by the lane that forms its gradient (adjoint launch for the
            # finest level, P^T chain for the others): no optimizer launch at all
            self.ev.loss_grad_arrays(self.w, timers, adam=(self.mw, self.vw, alpha, omb1, omb2, self.eps))
            return
        self.loss_grad(timers)
        if timers is not None:
            a, b = timers.section("adam")
            a.record()
        ops.adam_step(self.x, self.m, self.v, self.g, alpha, omb1, omb2, self.eps)
        if timers is not None:
            b.record()

    def last_loss(self):
        return float(self.loss)
