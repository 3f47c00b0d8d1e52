#!/usr/bin/env python3
"""Poisson equation in a d-dimensional unit cube with zero Dirichlet conditions, solved by
minimising the discrete residual -- the workload of reference examples/poisson/poisson.py,
written against the same operator API (`import odil_amd as odil`).

    python examples/poisson/poisson.py --ndim 3 --N 64 --optimizer adam --epochs 100
"""

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import odil_amd as odil  # noqa: E402
from odil_amd import printlog  # noqa: E402


def reference_solution(name, domain):
    xx = domain.points()
    if name == "hat":  # product of parabolas squashed into [0, 1)
        u = xx[0] * 0 + 1
        for x in xx:
            u = u * ((1 - x) * x * 5)
        p = 5
        return (u**p / (1 + u**p)) ** (1 / p)
    raise ValueError("Unknown name=" + name)


def laplacian(stencil, dirs, iw, nw, dw, mod):
    """Sum of second differences; ghosts at the walls by quadratic extrapolation through the
    wall value 0 (extrap_quadh), as in the reference operator."""
    q = stencil[0]
    zero = mod.cast(0, q.dtype)
    terms = []
    for i in dirs:
        qm, qp = stencil[2 * i + 1], stencil[2 * i + 2]
        gm = mod.where(iw[i] == 0, odil.core.extrap_quadh(qp, q, zero), qm)
        gp = mod.where(iw[i] == nw[i] - 1, odil.core.extrap_quadh(qm, q, zero), qp)
        terms.append((gp - 2 * q + gm) / dw[i] ** 2)
    return sum(terms)


def discrete_rhs(u, domain):
    mod = domain.mod
    dirs = range(domain.ndim)
    st = [u]
    for i in dirs:
        st += [mod.roll(u, 1, i), mod.roll(u, -1, i)]
    return laplacian(st, dirs, domain.indices(), domain.size(), domain.step(), mod)


def operator(ctx):
    mod = ctx.mod
    ndim = ctx.domain.ndim
    dirs = range(ndim)
    st = [ctx.field("u")]
    for i in dirs:
        st.append(ctx.field("u", *[-1 if j == i else 0 for j in dirs]))
        st.append(ctx.field("u", *[1 if j == i else 0 for j in dirs]))
    fu = laplacian(st, dirs, ctx.indices(), ctx.size(), ctx.step(), mod) - ctx.extra.rhs
    res = [fu]
    for _ in range(getattr(ctx.extra.args, "mgloss", 0) or 0):
        fu = odil.core.restrict_to_coarser(fu, loc="c" * ndim, mod=mod)
        res.append(fu)
    return res


def make_problem(args):
    dtype = np.float64 if args.double else np.float32
    ndim = args.ndim
    domain = odil.Domain(cshape=[args.N] * ndim, dimnames=["x", "y", "z", "sx", "sy", "sz"][:ndim],
                         multigrid=args.multigrid, dtype=dtype)
    if domain.multigrid:
        printlog("multigrid levels:", domain.mg_cshapes)
    ref_u = reference_solution(args.ref, domain)
    rhs = discrete_rhs(ref_u, domain)
    state = odil.State()
    state.fields["u"] = None
    state = domain.init_state(state)
    extra = argparse.Namespace(ref_u=ref_u, rhs=rhs, args=args)
    return odil.Problem(operator, domain, extra), state


def error_rms(domain, extra, state, key):
    du = domain.field(state, key) - extra.ref_u
    return float((du**2).mean() ** 0.5)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--ndim", type=int, choices=[1, 2, 3], default=2, help="Space dimension")
    parser.add_argument("--N", type=int, default=32, help="Grid size")
    parser.add_argument("--ref", type=str, default="hat", choices=("hat",), help="Reference solution")
    parser.add_argument("--mgloss", type=int, default=0, help="Extra restricted-residual terms in the loss")
    odil.util.add_arguments(parser)
    odil.linsolver.add_arguments(parser)
    parser.set_defaults(frames=4, report_every=100, history_every=10, plot_every=100, history_full=50)
    parser.set_defaults(optimizer="adam", multigrid=1, lr=0.005, double=1, outdir="out_poisson")
    return parser.parse_args(argv)


def main():
    args = parse_args()
    odil.setup_outdir(args)
    problem, state = make_problem(args)

    def report(problem, state, epoch, cbinfo):
        printlog("error: u:{:.5g}".format(error_rms(problem.domain, problem.extra, state, "u")))

    def history(problem, state, epoch, hist, cbinfo):
        hist.append("error_u", error_rms(problem.domain, problem.extra, state, "u"))

    callback = odil.make_callback(problem, args, report_func=report, history_func=history)
    odil.util.optimize(args, args.optimizer, problem, state, callback)


if __name__ == "__main__":
    main()
