#!/usr/bin/env python3
"""Infers three constants (diffusivity, source, velocity) of u_t + c u_x = nu u_xx + s from the
solution at the initial and the final time (same formulation as the reference's
examples/infer_constant/infer_constant.py:16-75): an `Array` of three unknowns multiplies the
stencil terms, first and last rows are imposed exactly, and the residual drops its first row."""

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import odil_amd as odil  # noqa: E402
from odil_amd import printlog  # noqa: E402

SHIFTS = [(0, 0), (0, -1), (0, 1), (-1, 0), (-1, -1), (-1, 1)]  # (t, x)


def reference_u(t, x, args):
    t, x = np.asarray(t, dtype=np.float64), np.asarray(x, dtype=np.float64)
    u = np.zeros(np.broadcast(t, x).shape)
    moved = x - t * args.c_vel
    for i in (1, 2, 3):
        k = 2 * i * np.pi
        u += np.cos(moved * k) * np.exp(-args.c_diff * k**2 * t)
    return u / 6 + args.c_src * t


def with_end_rows(u, extra, mod):
    return mod.concatenate([extra.u_init[None, :], u[1:-1], extra.u_final[None, :]], axis=0)


def operator(ctx):
    extra, mod = ctx.extra, ctx.mod
    dt, dx = ctx.step("t", "x")
    coeff = ctx.field("coeff")
    full = with_end_rows(ctx.field("u"), extra, mod)
    u, uxm, uxp, um, umxm, umxp = [mod.roll(full, shift=np.negative(s), axis=(0, 1)) for s in SHIFTS]
    lap = 0.5 * ((uxm - 2 * u + uxp) / dx**2 + (umxm - 2 * um + umxp) / dx**2)
    slope = 0.5 * ((u - uxm) / dx + (um - umxm) / dx)
    fu = (u - um) / dt - coeff[0] * lap - coeff[1] + coeff[2] * slope
    return [fu[1:]]


def make_problem(args):
    dtype = np.float64 if args.double else np.float32
    domain = odil.Domain(cshape=(args.Nt, args.Nx), dimnames=("t", "x"), lower=(0, -1), upper=(1, 1), dtype=dtype,
                         multigrid=args.multigrid, mg_interp=args.mg_interp, mg_nlvl=args.nlvl)
    mod = domain.mod
    x1 = domain.points_1d("x", loc="c")
    extra = argparse.Namespace(args=args)
    extra.u_init = mod.cast(reference_u(domain.lower[0], x1, args), dtype)
    extra.u_final = mod.cast(reference_u(domain.upper[0], x1, args), dtype)
    state = odil.State(fields={"coeff": odil.Array([0, 0, 0.001]), "u": odil.Field(None, loc="nc")})
    state = domain.init_state(state)
    return odil.Problem(operator, domain, extra), state


def constants(problem, state):
    return [float(v) for v in problem.domain.mod.numpy(problem.domain.field(state, "coeff"))]


def parse_args(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--Nt", type=int, default=64)
    parser.add_argument("--Nx", type=int, default=64)
    parser.add_argument("--c_diff", type=float, default=0.01, help="Diffusivity")
    parser.add_argument("--c_src", type=float, default=0.1, help="Uniform source")
    parser.add_argument("--c_vel", type=float, default=0.2, help="Advection velocity")
    odil.util.add_arguments(parser)
    odil.linsolver.add_arguments(parser)
    parser.set_defaults(frames=3, plot_every=50, report_every=50, history_every=10, optimizer="lbfgsb", multigrid=1,
                        double=1, outdir="out_infer_constant")
    return parser.parse_args(argv)


def main():
    args = parse_args()
    odil.setup_outdir(args)
    problem, state = make_problem(args)

    def history_func(p, s, epoch, history, cbinfo):
        for name, v in zip(("c_diff", "c_src", "c_vel"), constants(p, s)):
            history.append(name, v)

    callback = odil.make_callback(
        problem, args, history_func=history_func,
        report_func=lambda p, s, epoch, cbinfo: printlog("diff={:.5g}, src={:.5g}, vel={:.5g}".format(*constants(p, s))))
    odil.optimize(args, args.optimizer, problem, state, callback)


if __name__ == "__main__":
    main()
