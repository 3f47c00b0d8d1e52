"""Every combination of the multigrid options of `Domain` (reference core.py:62-78: `mg_axes`, `mg_nlvl`, `mg_factors`) with
cell- and node-centred fields on three grids: the synthesis `Domain.field` (reference core.py:245-263) against the NumPy
oracle's `multigrid_to_regular`, bit for bit, and -- for the cell-centred field of a heat-type operator -- loss and level
gradients of the generated kernels against autograd through the hand-written transfers."""

import os
import sys

import numpy as np
import pytest
import torch
from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import odil_np as onp  # noqa: E402

pytestmark = pytest.mark.gpu


def operator(ctx):
    mod = ctx.mod
    dt, dx = ctx.step("t", "x")
    it, ix = ctx.indices("t", "x")
    nx = ctx.size("x")
    u, um = ctx.field("u"), ctx.field("u", -1, 0)
    uxm, uxp = ctx.field("u", 0, -1), ctx.field("u", 0, 1)
    uxm = mod.where(ix == 0, -u, uxm)
    uxp = mod.where(ix == nx - 1, -u, uxp)
    return [mod.where(it == 0, u - mod.sin(np.pi * ctx.points("x")), (u - um) / dt - 0.1 * (uxm - 2 * u + uxp) / dx**2)]


@pytest.mark.parametrize("cshape", [(32, 64), (16, 16), (64, 8)])
@pytest.mark.parametrize("axes", [None, [True, True], [False, True], [True, False]])
def test_domain_multigrid_options(cshape, axes):
    import odil_amd as odil
    from odil_amd import runtime

    odil.util.set_log_file(open(os.devnull, "w"))
    saved = runtime.enable_trace
    try:
        for nlvl in (None, 2, 3):
            for geometric in (False, True):
                nl = len(odil.Domain(cshape=cshape, dimnames=("t", "x"), multigrid=True, mg_axes=axes, mg_nlvl=nlvl, dtype=np.float64).mg_cshapes)
                fac = [0.5**l for l in range(nl)] if geometric else None
                for loc in ("cc", "nc", "nn"):
                    res = {}
                    for trace in (True, False):
                        runtime.enable_trace = trace
                        domain = odil.Domain(cshape=cshape, dimnames=("t", "x"), multigrid=True, mg_axes=axes, mg_nlvl=nlvl,
                                             mg_factors=fac, dtype=np.float64)
                        state = domain.init_state(odil.State(fields={"u": odil.Field(None, loc=loc)}))
                        rng = np.random.default_rng(5)
                        terms = [rng.standard_normal(tuple(t.array.shape)) for t in state.fields["u"].terms]
                        for t, a in zip(state.fields["u"].terms, terms):
                            t.array.copy_(torch.as_tensor(a))
                        want = onp.multigrid_to_regular(terms, loc, factors=fac, axes=domain.mg_axes)
                        assert np.array_equal(domain.field(state, "u").cpu().numpy(), want), (cshape, axes, nlvl, fac, loc)
                        if loc != "cc":
                            break
                        problem = odil.Problem(operator, domain, None)
                        loss, grads = problem.eval_loss_grad(state)[:2]
                        assert (problem._traced is not None) == trace
                        res[trace] = (float(loss), [g.detach().cpu().numpy().copy() for g in grads])
                    if loc == "cc":
                        (l1, g1), (l0, g0) = res[True], res[False]
                        assert abs(l1 - l0) <= 1e-13 * abs(l0), (cshape, axes, nlvl, fac)
                        for a, b in zip(g1, g0):
                            assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), (cshape, axes, nlvl, fac)
    finally:
        runtime.enable_trace = saved


def make_operator(ndim):
    def operator(ctx):
        u = ctx.field("u")
        fu = u * u
        dw = ctx.step()
        for i in range(ndim):
            sm = [(-1 if j == i else 0) for j in range(ndim)]
            sp = [(1 if j == i else 0) for j in range(ndim)]
            fu = fu + (ctx.field("u", *sm) - 2 * u + ctx.field("u", *sp)) / dw[i] ** 2 * 1e-3
        return [fu]
    return operator


@pytest.mark.parametrize("cshape", [(8, 16, 32), (16, 8, 8), (4, 8, 16, 16), (8, 8, 8, 8), (2, 16, 16, 32)])
def test_domain_multigrid_options_in_three_and_four_dimensions(cshape):
    """The same in 3-D and 4-D (space-time fields of the flow-reconstruction workload): subsets of `mg_axes`, two levels or
    all, cell-centred / node-centred in time / node-centred everywhere."""
    import itertools

    import odil_amd as odil
    from odil_amd import runtime

    odil.util.set_log_file(open(os.devnull, "w"))
    nd = len(cshape)
    saved = runtime.enable_trace
    axsets = [None] + [list(a) for a in itertools.product([True, False], repeat=nd) if any(a) and not all(a)][:6]
    try:
        for axes in axsets:
            for nlvl in (None, 2):
                for loc in ("c" * nd, "n" + "c" * (nd - 1), "n" * nd):
                    res = {}
                    for trace in (True, False):
                        runtime.enable_trace = trace
                        domain = odil.Domain(cshape=cshape, dimnames=("t", "x", "y", "z")[:nd], multigrid=True, mg_axes=axes,
                                             mg_nlvl=nlvl, dtype=np.float64)
                        state = domain.init_state(odil.State(fields={"u": odil.Field(None, loc=loc)}))
                        rng = np.random.default_rng(5)
                        terms = [rng.standard_normal(tuple(t.array.shape)) for t in state.fields["u"].terms]
                        for t, a in zip(state.fields["u"].terms, terms):
                            t.array.copy_(torch.as_tensor(a))
                        want = onp.multigrid_to_regular(terms, loc, factors=None, axes=domain.mg_axes)
                        assert np.array_equal(domain.field(state, "u").cpu().numpy(), want), (cshape, axes, nlvl, loc)
                        if loc != "c" * nd:
                            break
                        problem = odil.Problem(make_operator(nd), domain, None)
                        loss, grads = problem.eval_loss_grad(state)[:2]
                        assert (problem._traced is not None) == trace
                        res[trace] = (float(loss), [g.detach().cpu().numpy().copy() for g in grads])
                    if res:
                        (l1, g1), (l0, g0) = res[True], res[False]
                        assert abs(l1 - l0) <= 1e-13 * abs(l0), (cshape, axes, nlvl)
                        for a, b in zip(g1, g0):
                            assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max(), (cshape, axes, nlvl)
    finally:
        runtime.enable_trace = saved
