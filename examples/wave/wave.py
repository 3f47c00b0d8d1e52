#!/usr/bin/env python3
"""Wave equation u_tt = u_xx on (t, x) in [0, 1] x [-1, 1] as one discrete loss: three time
levels per row of the residual, Dirichlet walls through quadratic ghost values, initial
displacement and velocity imposed in rows 0 and 1 (same discretisation as the reference's
examples/wave/wave.py:29-75; written against `import odil_amd as odil`)."""

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import odil_amd as odil  # noqa: E402
from odil_amd import printlog  # noqa: E402

MODES = (1, 2, 3, 4, 5)


def exact(t, x):
    """Superposition of left- and right-running cosines and its time derivative."""
    t, x = np.asarray(t, dtype=np.float64), np.asarray(x, dtype=np.float64)
    u, ut = np.zeros(np.broadcast(t, x).shape), np.zeros(np.broadcast(t, x).shape)
    for i in MODES:
        k = i * np.pi
        u += np.cos((x - t + 0.5) * k) + np.cos((x + t - 0.5) * k)
        ut += k * np.sin((x - t + 0.5) * k) - k * np.sin((x + t - 0.5) * k)
    return u / (2 * len(MODES)), ut / (2 * len(MODES))


def operator(ctx):
    extra, mod = ctx.extra, ctx.mod
    dt, dx = ctx.step()
    it, ix = ctx.indices()
    nx = ctx.size()[1]
    now, old, older, old_xm, old_xp = [
        ctx.field("u", st, sx) for st, sx in [(0, 0), (-1, 0), (-2, 0), (-1, -1), (-1, 1)]]
    ghost = odil.core.extrap_quadh
    # wall values belong to the previous row of the stencil
    wall_lo = mod.roll(extra.left_u, 1, axis=0)[:, None]
    wall_hi = mod.roll(extra.right_u, 1, axis=0)[:, None]
    old_xm = mod.where(ix == 0, ghost(old_xp, old, wall_lo), old_xm)
    old_xp = mod.where(ix == nx - 1, ghost(old_xm, old, wall_hi), old_xp)
    rate = (now - old) / dt
    rate_old = mod.where(it == 1, extra.init_ut[None, :], (old - older) / dt)
    fu = (rate - rate_old) / dt - (old_xm - 2 * old + old_xp) / dx**2
    first_row = extra.init_u + 0.5 * dt * extra.init_ut
    fu = mod.where(it == 0, (now - first_row[None, :]) * extra.args.kimp, fu)
    return [("fu", fu)]


def make_problem(args):
    dtype = np.float64 if args.double else np.float32
    domain = odil.Domain(cshape=(args.Nt, args.Nx), dimnames=("t", "x"), lower=(0, -1), upper=(1, 1),
                         multigrid=args.multigrid, dtype=dtype)
    if domain.multigrid:
        printlog("multigrid levels:", domain.mg_cshapes)
    mod = domain.mod
    t1, x1 = domain.points_1d()
    tt, xx = np.meshgrid(t1, x1, indexing="ij")
    extra = argparse.Namespace(args=args)
    extra.ref_u, extra.ref_ut = [mod.cast(v, dtype) for v in exact(tt, xx)]
    extra.left_u = mod.cast(exact(t1, domain.lower[1])[0], dtype)
    extra.right_u = mod.cast(exact(t1, domain.upper[1])[0], dtype)
    extra.init_u, extra.init_ut = [mod.cast(v, dtype) for v in exact(domain.lower[0], x1)]
    state = odil.State()
    state.fields["u"] = np.zeros(domain.cshape)
    state = domain.init_state(state)
    return odil.Problem(operator, domain, extra), state


def rms_error(problem, state):
    u = problem.domain.field(state, "u")
    return float(((u - problem.extra.ref_u) ** 2).mean() ** 0.5)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--Nt", type=int, default=64)
    parser.add_argument("--Nx", type=int, default=64)
    parser.add_argument("--kimp", type=float, default=1, help="Weight of the imposed initial row")
    odil.util.add_arguments(parser)
    odil.linsolver.add_arguments(parser)
    parser.set_defaults(double=1, multigrid=1, outdir="out_wave", linsolver="direct", optimizer="lbfgsb", lr=0.001,
                        plot_every=100, report_every=10, history_full=5, history_every=10, frames=2)
    return parser.parse_args(argv)


def main():
    args = parse_args()
    odil.setup_outdir(args)
    problem, state = make_problem(args)
    callback = odil.make_callback(
        problem, args,
        report_func=lambda p, s, epoch, cbinfo: printlog("error: u:{:.5g}".format(rms_error(p, s))),
        history_func=lambda p, s, epoch, history, cbinfo: history.append("error_u", rms_error(p, s)))
    odil.util.optimize(args, args.optimizer, problem, state, callback)


if __name__ == "__main__":
    main()
