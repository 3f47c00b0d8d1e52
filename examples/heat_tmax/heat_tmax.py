#!/usr/bin/env python3
"""Heat equation u_t = u_xx with an UNKNOWN final time: the time step is scaled by a scalar
unknown (`Array`), found so that the temperature at the domain centre reaches a given value
(same formulation as the reference's examples/heat_tmax/heat_tmax.py:23-75).  Exercises an
`Array` unknown inside the stencil, a row imposed exactly and a scalar output."""

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import odil_amd as odil  # noqa: E402
from odil_amd import printlog  # noqa: E402

OFFSETS = [(0, 0), (0, -1), (0, 1), (-1, 0), (-1, -1), (-1, 1)]  # (t, x) shifts of the stencil


def reference_u(t, x, tmax):
    return np.sin(np.asarray(x)) * np.exp(-np.asarray(t) * tmax)


def with_initial_row(u, extra, mod):
    """Row t = 0 of the unknown field is replaced by the initial condition."""
    return mod.concatenate([extra.u_init[None, :], u[1:]], axis=0)


def operator(ctx):
    extra, mod = ctx.extra, ctx.mod
    dt, dx = ctx.step("t", "x")
    it, ix = ctx.indices("t", "x", loc="nc")
    nx = ctx.size("x")
    coeff = ctx.field("coeff")

    def shifted(shift):
        # undo the shift, impose the initial row, shift again (keeps ctx.field for the Jacobian)
        back = lambda q, s: mod.roll(q, s, (0, 1))
        return back(with_initial_row(back(ctx.field("u", *shift), shift), extra, mod), np.negative(shift))

    u, uxm, uxp, um, umxm, umxp = [shifted(s) for s in OFFSETS]
    uxm = mod.where(ix == 0, -u, uxm)  # zero Dirichlet walls
    uxp = mod.where(ix == nx - 1, -u, uxp)
    umxm = mod.where(ix == 0, -um, umxm)
    umxp = mod.where(ix == nx - 1, -um, umxp)
    dt = dt * coeff[0]  # the unknown final time stretches the step
    lap = 0.5 * ((uxm - 2 * u + uxp) / dx**2 + (umxm - 2 * um + umxp) / dx**2)
    fu = mod.where(it == 0, ctx.cast(0), (u - um) / dt - lap)
    centre = nx // 2
    return [("eqn", fu), ("imp", extra.args.kimp * (u[-1, centre] - extra.u_final[centre]))]


def make_problem(args):
    dtype = np.float64 if args.double else np.float32
    domain = odil.Domain(cshape=(args.Nt, args.Nx), dimnames=("t", "x"), lower=(0, 0), upper=(1, np.pi), dtype=dtype,
                         multigrid=args.multigrid, mg_interp=args.mg_interp, mg_nlvl=args.nlvl)
    mod = domain.mod
    x1 = domain.points_1d("x", loc="c")
    extra = argparse.Namespace(args=args)
    extra.u_init = mod.cast(reference_u(0, x1, args.tmax_ref), dtype)
    extra.u_final = mod.cast(reference_u(1, x1, args.tmax_ref), dtype)
    state = odil.State(fields={
        "u": odil.Field(np.tile(reference_u(0, x1, args.tmax_ref), [args.Nt + 1, 1]), loc="nc"),
        "coeff": odil.Array([args.tmax_init]),
    })
    state = domain.init_state(state)
    return odil.Problem(operator, domain, extra), state


def tmax_of(problem, state):
    return float(problem.domain.mod.numpy(problem.domain.field(state, "coeff"))[0])


def parse_args(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--Nt", type=int, default=64)
    parser.add_argument("--Nx", type=int, default=64)
    parser.add_argument("--kimp", type=float, default=1)
    parser.add_argument("--tmax_ref", type=float, default=4.5)
    parser.add_argument("--tmax_init", type=float, default=1)
    odil.util.add_arguments(parser)
    odil.linsolver.add_arguments(parser)
    parser.set_defaults(frames=4, plot_every=1000, report_every=1000, history_every=200, optimizer="lbfgsb",
                        multigrid=1, double=1, outdir="out_heat_tmax")
    return parser.parse_args(argv)


def main():
    args = parse_args()
    odil.setup_outdir(args)
    problem, state = make_problem(args)
    callback = odil.make_callback(
        problem, args,
        report_func=lambda p, s, epoch, cbinfo: printlog("tmax={:.5g}".format(tmax_of(p, s))),
        history_func=lambda p, s, epoch, history, cbinfo: history.append("tmax", tmax_of(p, s)))
    odil.optimize(args, args.optimizer, problem, state, callback)


if __name__ == "__main__":
    main()
