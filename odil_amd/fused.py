"""Recognition of affine stencil operators and their fused evaluation.

A user `operator(ctx)` is arbitrary Python over `mod`, so only framework-owned arithmetic
can be hand-written generically.  But an operator that is AFFINE in a single cell-centred
field is completely determined by its per-shift Jacobian coefficient arrays (what
`Problem.eval_operator_grad` returns, reference core.py:1313-1361) and its value at one
state.  `detect()` evaluates the user's operator through the generic path at two probe
states; if (a) the coefficient arrays do not depend on the state, (b) they equal the
coefficients of the zero-Dirichlet Laplacian of the Poisson example (reference
examples/poisson/poisson.py:57-113) for this domain's steps, and (c) the operator did not
read `ctx.tracers`, then `fu = Lap(u) - rhs_eff` with `rhs_eff = Lap(u_A) - f(u_A)`, and
the problem is routed to the fused HIP kernels (residual + loss, adjoint, multigrid
synthesis / P^T chain).  Anything else keeps the generic path.  Like `jax.jit` in the
reference (core.py:1107), this bakes `extra` in at first evaluation.  ODIL_FUSE=0 disables it.
"""

import math

import numpy as np
import torch

from . import ops


class PoissonEvaluator:
    """Fused loss + gradient of `fu = Lap(u) - rhs` for a (multigrid) cell-centred unknown."""

    def __init__(self, cshape, shapes, rhs, h2, name="", dtype=torch.float64, device=None):
        self.cshape = tuple(cshape)
        self.ndim = len(cshape)
        self.loc = "c" * self.ndim
        self.shapes = [tuple(s) for s in shapes]
        self.nlvl = len(shapes)
        self.sizes = [math.prod(s) for s in self.shapes]
        self.rhs = rhs
        self.h2 = list(h2)
        self.names = [name]
        self.dtype, self.device = dtype, device
        self.npdt = np.float64 if dtype == torch.float64 else np.float32
        n = sum(self.sizes)
        self.g = torch.zeros(n, dtype=dtype, device=device)
        self.gw = [t.view(s) for t, s in zip(self.g.split(self.sizes), self.shapes)]
        self._u = None  # synthesised field, allocated on first use (the fused residual does not need it)
        self.fu = torch.empty(self.cshape, dtype=dtype, device=device)
        self.loss = torch.zeros((), dtype=dtype, device=device)
        self.work = ([None] + [torch.empty(s, dtype=dtype, device=device) for s in self.shapes[1:-1]] + [None])[
            : self.nlvl
        ]
        self.scale = self.npdt(2) / self.npdt(self.fu.numel())
        self.timers = None  # (measurement: event pairs around the launches of every evaluation, see loss_grad_arrays)
        import os

        # last prolongation fused into the residual (u never stored): 3-D, even extents, >= 2 levels
        self.synth_residual = (
            self.ndim == 3 and self.nlvl >= 2 and all(s % 2 == 0 and s >= 4 for s in self.cshape)
            and tuple(self.shapes[1]) == tuple(s // 2 for s in self.cshape)
            and bool(int(os.environ.get("ODIL_SYNTH_RESIDUAL", 1))))
        # stencil adjoint, first transposed prolongation and the Adam updates of both levels in one launch
        self.fuse_transpose = (
            self.ndim == 3 and self.nlvl >= 2 and ops.adjoint_transpose_supported(self.cshape)
            and tuple(self.shapes[1]) == tuple(s // 2 for s in self.cshape)
            and bool(int(os.environ.get("ODIL_FUSE_TRANSPOSE", 1))))

    def adjoint_and_transposes(self, arrays, adam, tic=lambda name: None, toc=lambda b: None):
        """grads (and, with `adam`, the update of every level) from the residual in self.fu."""
        if adam is not None:
            ml, vl, alpha, omb1, omb2, eps = adam
        if adam is not None and self.fuse_transpose:
            # stencil adjoint + first P^T + the updates of both levels in one launch: the finest-level gradient
            # never goes through memory and is not stored (gw[0] is left as it was).  Without the updates the
            # separate kernels are faster (the one-pass kernel is bound by its LDS / VALU phases at two
            # workgroups per CU: 1.6 ms against 0.54 + 0.45 ms at 512^3), so gradients alone keep them.
            b = tic("adjoint_transpose")
            ops.poisson_adjoint_transpose(self.fu, self.h2, self.scale, self.gw[1], g0=None,
                                          adam0=(arrays[0], ml[0], vl[0]), adam1=(arrays[1], ml[1], vl[1]),
                                          alpha=alpha, one_minus_b1=omb1, one_minus_b2=omb2, eps=eps)
            toc(b)
            if self.nlvl > 2:
                b = tic("mg_synth_adj")
                ops.mg_synth_adj_adam(self.gw[1], self.shapes[1:], self.loc, self.gw[1:], arrays[1:], ml[1:], vl[1:],
                                      alpha, omb1, omb2, eps)
                toc(b)
            return
        b = tic("adjoint")
        if adam is not None:
            ops.poisson_adjoint_adam(self.fu, self.h2, self.scale, self.gw[0], arrays[0], ml[0], vl[0], alpha, omb1,
                                     omb2, eps)
        else:
            ops.poisson_adjoint(self.fu, self.h2, self.scale, out=self.gw[0])
        toc(b)
        if self.nlvl > 1:
            b = tic("mg_synth_adj")
            if adam is not None:
                ops.mg_synth_adj_adam(self.gw[0], self.shapes, self.loc, self.gw, arrays, ml, vl, alpha, omb1, omb2, eps)
            else:
                ops.mg_synth_adj(self.gw[0], self.shapes, self.loc, grads=self.gw)
            toc(b)

    @property
    def u(self):
        if self._u is None and self.nlvl > 1:
            self._u = torch.empty(self.cshape, dtype=self.dtype, device=self.device)
        return self._u

    def loss_grad_arrays(self, arrays, timers=None, adam=None):
        """arrays: level arrays fine -> coarse.  Returns (loss 0-d tensor, grads views of one buffer).
        adam = (m_levels, v_levels, alpha, 1-b1, 1-b2, eps): also apply the Adam update of EVERY
        level array inside the launch that forms its gradient (adjoint for level 0, the P^T
        chain for the others); no separate optimizer launch is needed then."""

        timers = timers if timers is not None else self.timers

        def tic(name):
            if timers is None:
                return None
            a, b = timers.section(name)
            a.record()
            return b

        def toc(b):
            if b is not None:
                b.record()

        if self.synth_residual:
            b = tic("mg_synth")
            if self.nlvl > 2:
                coarse = ops.mg_synth(arrays[1:], self.loc, work=[None] + self.work[2:], out=self.work[1])
            else:
                coarse = arrays[1]
            toc(b)
            u = None
        elif self.nlvl > 1:
            b = tic("mg_synth")
            u = ops.mg_synth(arrays, self.loc, work=self.work, out=self.u)
            toc(b)
        else:
            u = arrays[0]
        b = tic("residual")
        if self.synth_residual:
            ops.poisson_residual_synth(coarse, arrays[0], self.rhs, self.h2, fu=self.fu, loss=self.loss)
        else:
            ops.poisson_residual(u, self.rhs, self.h2, fu=self.fu, loss=self.loss)
        toc(b)
        self.adjoint_and_transposes(arrays, adam, tic, toc)
        return self.loss, self.gw

    def eval_loss_grad(self, state):
        (field,) = state.fields.values()
        arrays = [t.array for t in field.terms] if hasattr(field, "terms") else [field.array]
        arrays = [a if a.is_contiguous() else a.contiguous() for a in arrays]
        loss, grads = self.loss_grad_arrays(arrays)
        return loss, list(grads), [loss], self.names, [torch.sqrt(loss)]


    def eval_loss_grad_adam(self, state, m, v, alpha, omb1, omb2, eps):
        """eval_loss_grad + the Adam step of every level inside the launches that form the
        gradients.  Returns (..., done=nlvl) or None when this configuration cannot fuse."""
        import os

        if self.nlvl < 2 or not int(os.environ.get("ODIL_FUSE_ADAM0", 1)):
            return None
        (field,) = state.fields.values()
        arrays = [t.array for t in field.terms]
        if not all(a.is_contiguous() for a in list(arrays) + list(m) + list(v)):
            return None
        loss, grads = self.loss_grad_arrays(arrays, adam=(m, v, alpha, omb1, omb2, eps))
        return loss, list(grads), [loss], self.names, [torch.sqrt(loss)], self.nlvl


    # ---- whole epochs in one launch (small 1-D / 2-D problems) --------------------------------------------------------------
    # ONE workgroup walks the epoch (odil_poisson_small_epochs) when the state fits its LDS (1-D N <= 1024 in float64, 2-D
    # 32^2; twice that in float32) -- 15 us per epoch at 1-D N = 256 against 58 replayed as a hipGraph -- and, from
    # global memory, for 1-D grids up to this many cells (N = 4096: 40 us against 82).  Beyond, the separate kernels win:
    # one workgroup is one CU's worth of bandwidth and a memory round trip per phase.
    small_max_cells = 4096
    small_force = False  # (tests: the global-memory form on any 1-D / 2-D problem)

    def small_plan(self, arrays, m, v):
        """The packed vectors (x, m, v, g) when `arrays` / m / v are the level slices of packed vectors in level order and
        the problem is small enough for one workgroup; None otherwise."""
        from ._lib import i64, load

        if self.ndim > 2 or self.nlvl > 12 or len(arrays) != self.nlvl or not self.small_max_cells:
            return None
        flat = [int(n) for shape in self.shapes for n in shape]
        resident = bool(load().odil_poisson_small_epochs_resident(i64(flat), self.nlvl, self.ndim, 8 if self.dtype == torch.float64 else 4))
        if not resident and not self.small_force and not (self.ndim == 1 and self.sizes[0] <= self.small_max_cells):
            return None
        if any(tuple(b) != tuple(n // 2 for n in a) for a, b in zip(self.shapes, self.shapes[1:])):
            return None

        def packed(levels):
            base, item, off = levels[0].data_ptr(), levels[0].element_size(), 0
            for t, size in zip(levels, self.sizes):
                if not t.is_contiguous() or t.numel() != size or t.dtype != self.dtype or t.data_ptr() != base + off * item:
                    return None
                off += size
            return levels[0]

        heads = [packed(list(levels)) for levels in (arrays, m, v, self.gw)]
        return None if any(hd is None for hd in heads) else heads

    def small_epochs(self, heads, alphas, losses, norms, omb1, omb2, eps):
        """alphas.numel() Adam epochs in ONE launch on the packed vectors of `small_plan`; the loss every epoch evaluated
        (and its square root) land in `losses` / `norms` (device tensors like `alphas`)."""
        if self.__dict__.get("_small_u") is None:
            self._small_u = torch.empty(sum(self.sizes), dtype=self.dtype, device=self.device)
        ops.poisson_small_epochs(heads[0], heads[1], heads[2], heads[3], self._small_u, self.fu, self.rhs, self.shapes,
                                 self.h2, alphas, omb1, omb2, eps, losses, norms)


def _close(a, b, rtol):
    scale = max(float(b.abs().max()), 1e-300)
    return float((a - b).abs().max()) <= rtol * scale


def detect(problem, state):
    """Returns a fused evaluator for `problem`, or None to keep the generic path."""
    from .core import Field, MultigridField, State

    domain = problem.domain
    if len(state.fields) != 1:
        return None
    (key, field), = state.fields.items()
    ndim = domain.ndim
    if ndim > 3 or field.loc != "c" * ndim:
        return None
    if isinstance(field, MultigridField):
        if field.factors is not None and any(float(f) != 1.0 for f in field.factors):
            return None
        axes = field.axes or domain.mg_axes
        if axes is not None and not all(axes):
            return None
        shapes = [tuple(t.array.shape) for t in field.terms]
    elif isinstance(field, Field):
        shapes = [tuple(field.array.shape)]
    else:
        return None
    cshape = tuple(domain.cshape)
    if shapes[0] != cshape or any(s < 2 for s in cshape):
        return None
    dtype = shapes and (field.terms[0].array if isinstance(field, MultigridField) else field.array).dtype
    device = domain.mod.device
    npdt = np.float64 if dtype == torch.float64 else np.float32
    h2 = [npdt(domain.step_by_dim(i)) ** 2 for i in range(ndim)]
    rtol = 1e-11 if dtype == torch.float64 else 1e-4

    gen = torch.Generator(device="cpu").manual_seed(12345)
    probes = []
    for k in range(2):
        # probe A is the zero state: Lap(0) == 0 exactly, so rhs_eff = -f(0) carries no rounding
        if k == 0:
            u = torch.zeros(cshape, dtype=dtype, device=device)
        else:
            u = torch.randn(cshape, generator=gen, dtype=torch.float64).to(dtype).to(device)
        pstate = State(fields={key: Field(u, loc=field.loc, cshape=cshape)}, initialized=True)
        try:
            accessed = [False]
            values, grads, names = _probe(problem, pstate, accessed)
        except Exception:
            return None
        if accessed[0] or len(values) != 1 or tuple(values[0].shape) != cshape:
            return None
        probes.append((u, values[0], grads[0], names))
    (ua, fa, ga, names), (ub, fb, gb, _) = probes
    want = [(0,) * ndim]
    for i in range(ndim):
        want += [tuple(-1 if j == i else 0 for j in range(ndim)), tuple(1 if j == i else 0 for j in range(ndim))]
    if any(k[0] != key or k[2] != field.loc for k in ga):
        return None
    if sorted(k[1] for k in ga) != sorted(want):
        return None
    coeffs = ops.poisson_jac_coeffs(cshape, h2, dtype, device)
    for slot, shift in enumerate(want):
        a, b = ga.get((key, shift, field.loc)), gb.get((key, shift, field.loc))
        if a is None or b is None:
            return None
        if not _close(a, b, rtol) or not _close(a, coeffs[slot], rtol):
            return None
    rhs_eff = -fa  # ua == 0
    fb_fused, _ = ops.poisson_residual(ub, rhs_eff, h2)
    if not _close(fb_fused, fb, 1e-9 if dtype == torch.float64 else 1e-3):
        return None
    from .util import printlog

    try:
        printlog("odil_amd: operator recognised as zero-Dirichlet Poisson stencil -> fused HIP kernels")
    except Exception:
        pass
    return PoissonEvaluator(cshape, shapes, rhs_eff.contiguous(), h2, name=names[0], dtype=dtype, device=device)


def _probe(problem, pstate, accessed):
    """eval_operator_grad on a probe state, recording whether the operator read ctx.tracers."""
    from . import core

    domain = problem.domain
    arrays = domain.arrays_from_state(pstate)
    leaves = [a.detach().requires_grad_(True) for a in arrays]
    shadow = problem._shadow_state(pstate, leaves)
    with torch.enable_grad():
        ctx = core.Context(domain, shadow, extra=problem.extra, tracers=problem.tracers, distinct_shift=True)
        names, values = problem._split_outputs(problem.operator(ctx))
        accessed[0] = ctx.tracers_accessed or bool(ctx.key_to_array_jac)
        if any(isinstance(v, core.Context.Raw) for v in values):
            accessed[0] = True
            return [], [], names
        grads = []
        for v in values:
            descs = list(ctx.desc_to_array.keys())
            symbols = [ctx.desc_to_array[d] for d in descs]
            gg = torch.autograd.grad(v.sum(), symbols, allow_unused=True, retain_graph=True)
            grads.append(dict(zip(descs, gg)))
    return [v.detach() for v in values], grads, names
