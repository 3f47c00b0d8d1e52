#!/usr/bin/env python3
"""Variable-coefficient diffusion  div(k grad u) = f  in the unit cube with zero Dirichlet walls, solved by minimising the
discrete residual (Newton / Adam / L-BFGS-B) -- NOT one of the reference's examples: the smallest operator whose Jacobian
is a (2 d + 1)-point stencil with VARIABLE coefficients, i.e. the case `linsolver.solve` takes through the general
geometric multigrid (odil_amd/gmg.py: StencilGMG) rather than through the constant-coefficient Poisson cycle.  Written
against the reference's operator API exactly like examples/poisson/poisson.py (reference examples/poisson/poisson.py:57-113:
same stencil access, same quadratic wall ghosts `extrap_quadh`), with the conductivity k given on the cell faces.

    python examples/diffusion/diffusion.py --ndim 3 --N 64 --optimizer newton --linsolver multigrid --kind jump
"""

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import odil_amd as odil  # noqa: E402
from odil_amd import printlog  # noqa: E402


def conductivity(kind, xx):
    """k at the points xx (a list of coordinate arrays): 'smooth' 1 + 10 prod sin^2(pi x), 'jump' 1000 inside the slab
    |x_0 - 1/2| < 1/4 and 1 outside, 'one' the Laplacian."""
    if kind == "one":
        return xx[0] * 0 + 1
    if kind == "smooth":
        p = xx[0] * 0 + 1
        for x in xx:
            p = p * np.sin(np.pi * x) ** 2
        return 1 + 10 * p
    if kind == "jump":
        return np.where(np.abs(xx[0] - 0.5) < 0.25, 1000.0, 1.0) + xx[0] * 0
    raise ValueError("Unknown kind=" + kind)


def reference_solution(xx):
    u = xx[0] * 0 + 1
    for x in xx:
        u = u * np.sin(np.pi * x)
    return u


def flux_divergence(stencil, kfaces, dirs, iw, nw, dw, mod, sigma=0.0):
    """sum_i [k_i+ (u_i+ - u) - k_i- (u - u_i-)] / h_i^2 (- sigma u), wall ghosts by quadratic extrapolation through 0."""
    q = stencil[0]
    zero = mod.cast(0, q.dtype)
    total = None
    for i in dirs:
        qm, qp = stencil[2 * i + 1], stencil[2 * i + 2]
        gm = mod.where(iw[i] == 0, odil.core.extrap_quadh(qp, q, zero), qm)
        gp = mod.where(iw[i] == nw[i] - 1, odil.core.extrap_quadh(qm, q, zero), qp)
        km, kp = kfaces[i]
        term = (kp * (gp - q) - km * (q - gm)) / dw[i] ** 2
        total = term if total is None else total + term
    if sigma:
        total = total - sigma * q
    return total


def operator(ctx):
    mod, extra = ctx.mod, ctx.extra
    ndim = ctx.domain.ndim
    dirs = range(ndim)
    st = [ctx.field("u")]
    for i in dirs:
        st.append(ctx.field("u", *[-1 if j == i else 0 for j in dirs]))
        st.append(ctx.field("u", *[1 if j == i else 0 for j in dirs]))
    fu = flux_divergence(st, extra.kfaces, dirs, ctx.indices(), ctx.size(), ctx.step(), mod, extra.args.sigma)
    return [fu - extra.rhs]


def make_problem(args):
    dtype = np.float64 if args.double else np.float32
    ndim = args.ndim
    domain = odil.Domain(cshape=[args.N] * ndim, dimnames=["x", "y", "z"][:ndim], multigrid=args.multigrid, dtype=dtype)
    mod = domain.mod
    x1 = [np.asarray(mod.numpy(x), dtype=np.float64) for x in domain.points_1d()]
    step = [float(s) for s in domain.step()]
    kfaces = []
    for i in range(ndim):
        pair = []
        for sign in (-0.5, 0.5):
            coords = [x + (sign * step[i] if j == i else 0.0) for j, x in enumerate(x1)]
            xx = np.meshgrid(*coords, indexing="ij")
            pair.append(mod.cast(conductivity(args.kind, xx), dtype))
        kfaces.append(tuple(pair))
    xx = np.meshgrid(*x1, indexing="ij")
    ref_u = mod.cast(reference_solution(xx), dtype)
    # rhs = the discrete operator applied to the reference solution (as reference examples/poisson/poisson.py:71-86)
    st = [ref_u]
    for i in range(ndim):
        st += [mod.roll(ref_u, 1, i), mod.roll(ref_u, -1, i)]
    rhs = flux_divergence(st, kfaces, range(ndim), domain.indices(), domain.size(), domain.step(), mod, args.sigma)
    state = odil.State()
    state.fields["u"] = None
    state = domain.init_state(state)
    extra = argparse.Namespace(ref_u=ref_u, rhs=rhs, kfaces=kfaces, args=args)
    return odil.Problem(operator, domain, extra), state


def error_rms(domain, extra, state, key):
    du = domain.field(state, key) - extra.ref_u
    return float((du**2).mean() ** 0.5)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--ndim", type=int, choices=[1, 2, 3], default=2, help="Space dimension")
    parser.add_argument("--N", type=int, default=32, help="Grid size")
    parser.add_argument("--kind", type=str, default="smooth", choices=("one", "smooth", "jump"), help="Conductivity field")
    parser.add_argument("--sigma", type=float, default=0.0, help="Reaction coefficient: div(k grad u) - sigma u = f")
    odil.util.add_arguments(parser)
    odil.linsolver.add_arguments(parser)
    parser.set_defaults(frames=1, report_every=1, history_every=1, plot_every=1, history_full=50)
    parser.set_defaults(optimizer="newton", multigrid=0, lr=0.005, double=1, outdir="out_diffusion", linsolver="multigrid")
    return parser.parse_args(argv)


def main():
    args = parse_args()
    odil.setup_outdir(args)
    problem, state = make_problem(args)

    def report(problem, state, epoch, cbinfo):
        printlog("error: u:{:.5g}".format(error_rms(problem.domain, problem.extra, state, "u")))

    callback = odil.make_callback(problem, args, report_func=report)
    odil.util.optimize(args, args.optimizer, problem, state, callback)


if __name__ == "__main__":
    main()
