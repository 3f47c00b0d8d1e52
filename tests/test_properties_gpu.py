"""Randomised structure tests of the transfer / access kernels through the C-ABI: random
locations ('c', 'n', '.'), ranks 1-4 and extents (even, odd, tiny), so that every dispatch
variant (generic, per-plane, z-marching 'ccc' / 'ncc' / 4-D, column pairs for float) meets the
same contracts: bit-exact prolongation against the oracle, transposes that are transposes,
restriction and its cotangent, the multigrid chain with factors, stencil access and its scatter."""

import numpy as np
import pytest
import torch

from oracle import odil_np as onp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def to(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a), device=dev)


def rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return float(np.max(np.abs(a - b))) / max(1.0, float(np.max(np.abs(b))))


def random_case(rng, allow_none=True):
    ndim = int(rng.integers(1, 5))
    kinds = "cn." if allow_none else "cn"
    while True:
        loc = "".join(rng.choice(list(kinds)) for _ in range(ndim))
        if any(c != "." for c in loc):
            break
    budget = 60000
    shape = []
    for c in loc:
        hi = max(3, int(round(budget ** (1.0 / ndim))))
        n = int(rng.integers(2, min(hi, 70) + 1))
        if rng.random() < 0.4:
            n = int(rng.choice([2, 3, 4, 8, 16, 32, 64]))
            n = min(n, hi + 8)
        shape.append(max(2, n))
    return loc, tuple(shape)


@pytest.mark.parametrize("seed", range(40))
def test_prolongation_and_transpose_random_layouts(dev, seed):
    from odil_amd import ops

    rng = np.random.default_rng(1000 + seed)
    loc, shape = random_case(rng)
    for dtype, tol in [(np.float64, 1e-13), (np.float32, 3e-6)]:
        u = rng.standard_normal(shape).astype(dtype)
        fshape = onp.fine_shape(shape, loc)
        add = rng.standard_normal(fshape).astype(dtype)
        out = ops.interp_add(to(u, dev), loc, add=to(add, dev))
        want = add + onp.interp_to_finer(u, loc)
        assert np.array_equal(out.cpu().numpy(), want), (loc, shape, dtype)  # reference summation order
        y = rng.standard_normal(fshape).astype(dtype)
        pty = ops.interp_adj(to(y, dev), loc, shape)
        assert rel(pty, onp.interp_to_finer_adj(y.astype(np.float64), loc, shape)) < tol * 10, (loc, shape)
        # <P x, y> == <x, P^T y>
        px = ops.interp_add(to(u, dev), loc).cpu().numpy().astype(np.float64)
        lhs, rhs = float(np.sum(px * y)), float(np.sum(u.astype(np.float64) * pty.cpu().numpy()))
        assert abs(lhs - rhs) <= tol * 100 * max(1.0, abs(lhs), float(np.sum(np.abs(px * y)))), (loc, shape)


@pytest.mark.parametrize("seed", range(20))
def test_restriction_and_cotangent_random_layouts(dev, seed):
    from odil_amd import ops

    rng = np.random.default_rng(2000 + seed)
    loc, cshape = random_case(rng)
    fshape = onp.fine_shape(cshape, loc)  # restriction of a fine array of this shape gives back cshape
    u = rng.standard_normal(fshape)
    r = ops.restrict_to_coarser(to(u, dev), loc)
    want = onp.restrict_to_coarser(u, loc)
    assert tuple(r.shape) == want.shape and rel(r, want) < 1e-14, (loc, fshape)
    y = rng.standard_normal(want.shape)
    rty = ops.restrict_adj(to(y, dev), loc, fshape)
    assert rel(rty, onp.restrict_to_coarser_adj(y, loc, fshape)) < 1e-13
    lhs, rhs = float(np.sum(want * y)), float(np.sum(u * rty.cpu().numpy()))
    assert abs(lhs - rhs) <= 1e-11 * max(1.0, float(np.sum(np.abs(want * y))))


@pytest.mark.parametrize("seed", range(16))
def test_multigrid_chain_with_factors_random_layouts(dev, seed):
    from odil_amd import ops

    rng = np.random.default_rng(3000 + seed)
    ndim = int(rng.integers(1, 5))
    loc = "".join(rng.choice(list("cn")) for _ in range(ndim))
    nlvl = int(rng.integers(2, 5))
    base = [int(rng.integers(1, 4)) for _ in range(ndim)]
    cshape = tuple(b * 2 ** (nlvl - 1) * (2 if ndim < 3 else 1) for b in base)
    shapes = [onp.field_shape(s, loc) for s in onp.mg_cshapes(cshape, mg_nlvl=nlvl)]
    factors = None if seed % 2 else [float(f) for f in rng.uniform(0.5, 2.0, size=len(shapes))]
    for dtype, tol in [(np.float64, 1e-13), (np.float32, 2e-6)]:
        terms = [rng.standard_normal(s).astype(dtype) for s in shapes]
        u = ops.mg_synth([to(t, dev) for t in terms], loc, factors=factors)
        want = onp.multigrid_to_regular([t.astype(np.float64) for t in terms], loc, factors=factors)
        assert rel(u, want) < tol * 10, (loc, shapes)
        if factors is None and dtype == np.float64:
            assert np.array_equal(u.cpu().numpy(), onp.multigrid_to_regular(terms, loc))
        g = rng.standard_normal(shapes[0]).astype(dtype)
        grads = ops.mg_synth_adj(to(g, dev), shapes, loc, factors=factors)
        wantg = onp.multigrid_to_regular_adj(g.astype(np.float64), shapes, loc, factors=factors)
        for a, b in zip(grads, wantg):
            assert rel(a, b) < tol * 20, (loc, shapes)


@pytest.mark.parametrize("seed", range(20))
def test_field_access_and_scatter_random_layouts(dev, seed):
    from odil_amd import ops

    rng = np.random.default_rng(4000 + seed)
    ndim = int(rng.integers(1, 5))
    floc = "".join(rng.choice(list("cn")) for _ in range(ndim))
    loc = "".join(rng.choice(list("cn")) for _ in range(ndim)) if seed % 2 else floc
    cshape = tuple(int(rng.integers(2, 9)) for _ in range(ndim))
    src = rng.standard_normal(onp.field_shape(cshape, floc))
    shift = tuple(int(s) for s in rng.integers(-9, 10, size=ndim))
    out = ops.field_gather(to(src, dev), floc, shift, loc)
    want = onp.field_access(src, floc, shift, loc)
    assert np.array_equal(out.cpu().numpy(), want), (floc, loc, shift)
    g = rng.standard_normal(want.shape)
    back = ops.field_scatter(to(g, dev), src.shape, floc, shift, loc)
    assert rel(back, onp.field_access_adj(g, src.shape, floc, shift, loc)) < 1e-14
    assert abs(float(np.sum(want * g)) - float(np.sum(src * back.cpu().numpy()))) < 1e-10
