"""The ALGORITHM of the variable-coefficient multigrid (odil_amd/gmg.py: StencilGMG, csrc/stencil_mg.hip), on its NumPy
restatement (tests/stencil_gmg_np.py), CPU only: what its V(2, 2) cycles contract by on the operator classes it is meant
for.  The HIP kernels are held to the same restatement on the GPU (tests/test_stencil_gmg_gpu.py); the Newton iterate it
must reach is the reference's `linsolver.solve("direct")` (reference src/odil/linsolver.py:17-26)."""

import numpy as np
import pytest
import stencil_gmg_np as sg


def cycles_to(coeffs, tol=1e-10, maxit=30):
    levels = sg.hierarchy(coeffs)
    rng = np.random.default_rng(0)
    b = sg.apply(coeffs, rng.standard_normal(coeffs[0].shape))
    x, r0, hist = np.zeros_like(b), np.linalg.norm(b), []
    for _ in range(maxit):
        x = sg.vcycle(levels, 0, x, b)
        hist.append(np.linalg.norm(b - sg.apply(coeffs, x)) / r0)
        if hist[-1] < tol:
            break
    return hist


smooth = lambda *x: 1 + 10 * np.prod([np.sin(np.pi * v) ** 2 for v in x], axis=0)  # noqa: E731
jump = lambda *x: np.where(np.abs(x[0] - 0.5) < 0.25, 1000.0, 1.0) * np.ones_like(x[0])  # noqa: E731


@pytest.mark.parametrize("name,make,limit", [
    ("poisson 2-D", lambda: sg.poisson_coeffs((32, 32)), 12),
    ("poisson 3-D", lambda: sg.poisson_coeffs((16, 16, 16)), 12),
    ("smooth k 2-D", lambda: sg.diffusion_coeffs((32, 32), smooth), 12),
    ("k jumps 1 : 1000, 2-D", lambda: sg.diffusion_coeffs((32, 32), jump), 13),
    ("k jumps 1 : 1000, 3-D", lambda: sg.diffusion_coeffs((16, 16, 16), jump), 13),
    ("reaction sigma ~ 1 / h^2", lambda: sg.diffusion_coeffs((32, 32), smooth, sigma=1000.0), 12),
    ("upwind convection, cell Peclet 0.6", lambda: sg.add_upwind_convection(sg.poisson_coeffs((32, 32)), 20.0), 12),
    ("upwind convection, cell Peclet 6", lambda: sg.add_upwind_convection(sg.poisson_coeffs((32, 32)), 200.0), 22),
    ("1-D, smooth k (aggregates of two cells: 0.47 per cycle)", lambda: sg.diffusion_coeffs((64,), smooth), 20),
    ("1-D, k jumps 1 : 1000 (linear interpolation across the jump: slower)", lambda: sg.diffusion_coeffs((64,), jump), 20),
])
def test_vcycles_contract(name, make, limit):
    hist = cycles_to(make())
    assert hist[-1] < 1e-10 and len(hist) <= limit, (name, len(hist), hist[-3:])


def test_coarse_operator_of_the_laplacian_is_the_rediscretised_one_in_the_interior():
    """away from the walls the construction reproduces the coefficients of the same stencil on the coarser grid exactly
    (1 / (2 h)^2), and every coarse row keeps the sign pattern / diagonal dominance of the fine one."""
    fine = sg.poisson_coeffs((16, 16))
    coarse = sg.coarsen(fine)
    inner = (slice(1, -1),) * 2
    assert np.allclose(coarse[0][inner], fine[0][4, 4] / 4, rtol=1e-14)
    for k in range(1, 5):
        assert np.allclose(coarse[k][inner], fine[k][4, 4] / 4, rtol=1e-14)
        assert (coarse[k] >= 0).all()
    assert (-coarse[0] >= sum(coarse[1:]) - 1e-9).all()


def test_two_cycles_on_the_first_coarse_level_pay_in_three_dimensions():
    """the smooth right-hand side of the examples (rhs = the operator on the 'hat' reference solution) from the zero start:
    V(2, 2) 17 cycles at 0.24 per cycle, with the doubled level-1 cycle 13 at 0.14 -- for 1 / 7 more work in 3-D."""
    from oracle import odil_np as onp

    shape = (32, 32, 32)
    for fine in (sg.poisson_coeffs(shape), sg.diffusion_coeffs(shape, jump)):
        levels = sg.hierarchy(fine)
        b = sg.apply(fine, onp.poisson_ref_u(shape))
        counts = []
        for top2 in (False, True):
            x, r0, n = np.zeros_like(b), np.linalg.norm(b), 0
            while np.linalg.norm(b - sg.apply(fine, x)) > 1e-10 * r0 and n < 40:
                x = sg.vcycle(levels, 0, x, b, top2=top2)
                n += 1
            counts.append(n)
        assert counts[1] <= 13 and counts[0] >= counts[1] + 3, counts


def test_semicoarsening_restatement():
    """`coarsen_axes` with every axis merged is `coarsen`; with the strongly coupled axes only, the cycle contracts on
    cells 1 : 4 and 1 : 8 (Poisson, a 1 : 1000 jump, a reaction term) where merging every axis DIVERGES."""
    jump = lambda *x: np.where(np.abs(x[0] - 0.5) < 0.25, 1000.0, 1.0) * np.ones_like(x[0])  # noqa: E731
    smooth = lambda *x: 1 + 10 * np.prod([np.sin(np.pi * v) ** 2 for v in x], axis=0)  # noqa: E731
    for c in (sg.poisson_coeffs((16, 16)), sg.diffusion_coeffs((16, 8, 8), jump, sigma=3.0), sg.add_upwind_convection(sg.poisson_coeffs((16, 16)), 20.0)):
        for a, b in zip(sg.coarsen(c), sg.coarsen_axes(c, [True] * c[0].ndim)):
            assert np.abs(a - b).max() <= 1e-14 * np.abs(a).max()
    rng = np.random.default_rng(0)
    for name, c, bound in (("poisson 64 x 16", sg.poisson_coeffs((64, 16)), 0.2), ("jump 64 x 16", sg.diffusion_coeffs((64, 16), jump), 0.3),
                           ("reaction 16 x 64", sg.diffusion_coeffs((16, 64), smooth, sigma=50.0), 0.2), ("poisson 16 x 16 x 4", sg.poisson_coeffs((16, 16, 4)), 0.2)):
        levels, halves = sg.hierarchy_axes(c)
        assert not all(halves[0]) and all(halves[-1]), (name, halves)
        xt = rng.standard_normal(c[0].shape)
        b = sg.apply(c, xt)
        x, hist = np.zeros_like(xt), []
        for _ in range(8):
            x = sg.vcycle_axes(levels, halves, 0, x, b)
            hist.append(np.linalg.norm(b - sg.apply(c, x)))
        assert (hist[-1] / hist[3]) ** 0.25 < bound, (name, hist)
        full = sg.hierarchy(c)
        y, grow = np.zeros_like(xt), []
        for _ in range(6):
            y = sg.vcycle(full, 0, y, b)
            grow.append(np.linalg.norm(b - sg.apply(c, y)))
        assert grow[-1] > 0.5 * grow[-2], (name, grow)  # (full coarsening: no useful contraction)
