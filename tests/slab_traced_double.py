"""Test double of the generated slab kernels (odil_amd.slab_traced.HipSlabKernels) on CPU: evaluates the user
operator on ONE RANK's cells with torch-CPU autograd, through the generic oracle's Context
(oracle/odil_generic.py), so that the slab driver's exchange logic -- ghost planes, periodic wrap planes,
deferred halo-adds, partial loss sums -- runs over gloo without a GPU.  The HIP kernels themselves are checked
with emulated ranks on a GPU (tests/test_slab_gpu.py)."""

import numpy as np
import torch

from oracle import odil_generic as og


class _SlabContext(og.Context):
    """Context of one rank: fields are halo arrays (one cell beyond the owned ones on the sharded axis), index
    and point grids are the owned part of the GLOBAL ones."""

    def __init__(self, geom, halo_arrays, locs, params, extra, tracers, axis, off, n):
        super().__init__(geom, halo_arrays, locs, params, extra, tracers)
        self.axis, self.off, self.n = axis, off, n

    def _own(self, t):
        return t.narrow(self.axis, self.off, self.n)

    def indices(self, *dims, loc=None):
        return self.geom._pick([self._own(self.geom.grids(loc, "indices")[i]) for i in self.geom._dims(dims)], dims)

    def points(self, *dims, loc=None):
        return self.geom._pick([self._own(self.geom.grids(loc, "points")[i]) for i in self.geom._dims(dims)], dims)

    def field(self, key, *shift, loc=None, frozen=False):
        if key in self.params:
            return super().field(key, *shift, loc=loc, frozen=frozen)
        shift = list(int(s) for s in shift) or [0] * self.geom.ndim
        sa, shift[self.axis] = shift[self.axis], 0
        assert abs(sa) <= 1
        u = og.field_access(self.regular[key], self.locs[key], shift, loc or self.locs[key])
        u = u.narrow(self.axis, 1 + sa, self.n)
        return u.detach() if frozen else u


def make_kernels(local_extra):
    """Factory with the signature of HipSlabKernels; `local_extra(extra, off, n)` cuts the rank's part out of the
    operator's constant arrays (the generated kernel indexes the global arrays with global indices instead)."""

    class CpuSlabKernels:
        halo = 1

        def __init__(self, problem, state, axis, n, device):
            from odil_amd.core import Field, MultigridField

            self.problem, self.axis, self.n = problem, axis, n
            self.geom = og.Geometry.of(problem.domain)
            from odil_amd.core import Array, NeuralNet

            self.locs = {k: f.loc for k, f in state.fields.items() if isinstance(f, (Field, MultigridField))}
            self.src_keys = list(self.locs)
            self.gather_keys = list(self.locs)
            # replicated unknowns (networks, Arrays): their gradients are partial sums over the rank's cells
            self.param_fields = {k: f for k, f in state.fields.items() if isinstance(f, (NeuralNet, Array))}
            self.param_groups, ofs = dict(), 0
            for k, f in self.param_fields.items():
                lens = [int(a.numel()) for a in problem.domain.arrays_from_field(f)]
                self.param_groups[k] = (ofs, lens)
                ofs += sum(lens)
            self.pgrad = torch.zeros(max(1, ofs), dtype=torch.float64)
            self.world = problem.domain.cshape[axis] // n

        def set_geometry(self, off, lo, ea):
            self.off, self.lo, self.ea = off, lo, ea

        def set_params(self, fn):
            self.param_arrays = fn  # key -> this rank's (replicated) parameter arrays

        def forward(self, srcs, wlo, whi):
            a, n, lo = self.axis, self.n, self.lo
            self.leaves = dict()
            for key in self.src_keys:
                u = srcs[key]
                left = u.narrow(a, lo - 1, 1) if lo > 0 else wlo[key]
                right = u.narrow(a, lo + n, 1) if lo + n < u.shape[a] else whi[key]
                self.leaves[key] = torch.cat([left, u.narrow(a, lo, n), right], dim=a).detach().clone().requires_grad_(True)
            from odil_amd.core import NeuralNet

            extra = local_extra(self.problem.extra, self.off, n)
            params, pleaves = dict(), []
            for k, f in self.param_fields.items():
                arrs = [t.detach().clone().requires_grad_(True) for t in self.param_arrays(k)]
                pleaves += arrs
                if isinstance(f, NeuralNet):
                    nw = len(f.weights)
                    params[k] = ("net", arrs[:nw], arrs[nw:], f.activation, f.func_in, f.func_out)
                else:
                    params[k] = ("array", arrs[0])
            ctx = _SlabContext(self.geom, self.leaves, self.locs, params, extra, self.problem.tracers, a, self.off, n)
            names, values = og.split_outputs(self.problem.operator(ctx))
            # mean over the GLOBAL output: local sum / (local count * ranks) (windows of these operators are in t)
            self.terms = [(v * v).sum() / (v.numel() * self.world) for v in values]
            leaves = [self.leaves[k] for k in self.src_keys] + pleaves
            grads = torch.autograd.grad(sum(self.terms), leaves, allow_unused=True)
            grads = [torch.zeros_like(t) if g is None else g for g, t in zip(grads, leaves)]
            self.grads = dict(zip(self.src_keys, grads))
            if pleaves:
                self.pgrad = torch.cat([g.reshape(-1) for g in grads[len(self.src_keys):]])

        def gather(self, key, g, gwlo, gwhi):
            a, n, lo = self.axis, self.n, self.lo
            gh = self.grads[key]
            g.zero_()
            g.narrow(a, lo, n).copy_(gh.narrow(a, 1, n))
            (g.narrow(a, lo - 1, 1) if lo > 0 else gwlo).copy_(gh.narrow(a, 0, 1))
            (g.narrow(a, lo + n, 1) if lo + n < g.shape[a] else gwhi).copy_(gh.narrow(a, n + 1, 1))

        def partial_terms(self):
            return torch.stack([t.detach() for t in self.terms])

    return CpuSlabKernels
