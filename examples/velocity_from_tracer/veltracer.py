#!/usr/bin/env python3
"""Velocity reconstruction from two tracer snapshots on a (t, x, y) grid: the workload of
reference examples/velocity_from_tracer/veltracer.py (`operator_advection`), written against
the same operator API.  Unknowns u, vx, vy live at nodes in t and cell centres in x, y ('ncc').

    python examples/velocity_from_tracer/veltracer.py --Nt 32 --Nx 64 --Ny 64 --epochs 200
"""

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import odil_amd as odil  # noqa: E402
from odil_amd import printlog  # noqa: E402


def blob(x, y, t):
    """A blob carried by a uniform velocity and stretched in time."""
    dx = (x - 0.2 * t - 0.3) * (1 + t)
    dy = (y - 0.2 * t - 0.3) / (1 + t)
    return np.maximum(0, 1 - (dx**2 + dy**2) / 0.2**2) ** 0.2


def operator(ctx):
    mod, extra = ctx.mod, ctx.extra
    args = extra.args
    dt, dx, dy = ctx.step()
    it = ctx.indices(loc="ncc")[0]
    nt = ctx.size()[0]
    offsets = [(0, 0), (-1, 0), (1, 0), (0, -1), (0, 1)]  # centre, x-, x+, y-, y+

    def five(key, shift_t=0, frozen=False):
        return [ctx.field(key, shift_t, sx, sy, frozen=frozen) for sx, sy in offsets]

    def upwind(um, u, up, v):
        """First-order difference taken from the side the (frozen) velocity comes from."""
        return mod.where(v > 0, u - um, mod.where(v < 0, up - u, (up - um) * 0.5))

    vx_st, vy_st = five("vx"), five("vy")
    vx, vy = vx_st[0], vy_st[0]
    vx_frozen, vy_frozen = ctx.field("vx", frozen=True), ctx.field("vy", frozen=True)
    st = five("u", shift_t=-1)
    flux_x = vx * upwind(st[1], st[0], st[2], vx_frozen) / dx
    flux_y = vy * upwind(st[3], st[0], st[4], vy_frozen) / dy
    u = ctx.field("u")
    u_old = mod.where(it == 1, extra.u_init[None, :], st[0])
    fu = (u - u_old) / dt + flux_x + flux_y
    fu = mod.where(it == 0, (u - extra.u_init[None, :]) / dx, fu)
    zero = ctx.cast(0)
    fimp = mod.where(it == nt - 1, (u - extra.u_final[None, :]) / dx, zero)
    res = [fu, fimp * args.kimp]
    if args.kxreg:
        for q in (vx_st, vy_st):
            lap = (q[2] - 2 * q[0] + q[1]) / dx**2 + (q[4] - 2 * q[0] + q[3]) / dy**2
            res.append(lap * args.kxreg)
    if args.ktreg:
        k = args.ktreg / dt
        for key in ("vx", "vy"):
            d = (ctx.field(key) - ctx.field(key, -1, 0, 0)) * k
            res.append(mod.where(it == 0, zero, d))
    return res


def make_problem(args):
    dtype = np.float64 if args.double else np.float32
    domain = odil.Domain(cshape=(args.Nt, args.Nx, args.Ny), dimnames=("t", "x", "y"), lower=(0, 0, 0),
                         upper=(1, 1, 1), dtype=dtype, multigrid=args.multigrid, mg_interp=args.mg_interp,
                         mg_nlvl=args.nlvl)
    if domain.multigrid:
        printlog("multigrid levels:", domain.mg_cshapes)
    mod = domain.mod
    x, y = [mod.numpy(a) for a in domain.points("x", "y", loc=".cc")]
    extra = argparse.Namespace(args=args)
    extra.u_init = mod.cast(blob(x, y, 0), dtype)
    extra.u_final = mod.cast(blob(x, y, 1), dtype)
    state = odil.State()
    for key in ("u", "vx", "vy"):
        state.fields[key] = odil.Field(None, loc="ncc")
    state = domain.init_state(state)
    return odil.Problem(operator, domain, extra), state


def parse_args(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--Nt", type=int, default=None)
    parser.add_argument("--Nx", type=int, default=64)
    parser.add_argument("--Ny", type=int, default=None)
    parser.add_argument("--kxreg", type=float, default=0.01, help="Laplacian regularization weight")
    parser.add_argument("--ktreg", type=float, default=1, help="Time regularization weight")
    parser.add_argument("--kimp", type=float, default=10, help="Imposed values weight")
    odil.util.add_arguments(parser)
    odil.linsolver.add_arguments(parser)
    parser.set_defaults(outdir="out_veltracer", frames=5, plot_every=100, report_every=100, history_every=10,
                        optimizer="adam", lr=0.01, multigrid=1, mg_interp="conv")
    args = parser.parse_args(argv)
    args.Nt = args.Nt or args.Nx
    args.Ny = args.Ny or args.Nx
    return args


def main():
    args = parse_args()
    odil.setup_outdir(args)
    problem, state = make_problem(args)
    callback = odil.make_callback(problem, args)
    odil.util.optimize(args, args.optimizer, problem, state, callback)


if __name__ == "__main__":
    main()
