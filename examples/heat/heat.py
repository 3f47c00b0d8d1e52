#!/usr/bin/env python3
"""Inverse heat-conduction problem on a (t, x) grid: the workload of reference
examples/heat/heat.py (`operator_odil`), written against the same operator API.

  u_t = (k(u) u_x)_x, u = 0 on the walls, u(0, x) given; with --infer_k the conductivity is
  sigmoid(MLP(u)) * kmax, evaluated inside the stencil at the faces on FROZEN u, and the
  temperature is pinned at a set of imposed points.

    python examples/heat/heat.py --Nt 64 --Nx 64 --infer_k 1 --imposed stripe --epochs 200
"""

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import odil_amd as odil  # noqa: E402
from odil_amd import printlog  # noqa: E402


def initial_u(x, mod):
    bump = lambda z: mod.exp(-((z - 0.5) ** 2) * 50)
    return bump(x) - bump(-mod.cast(0.5, x.dtype))


def reference_k(u, mod):
    return 0.02 * mod.exp(-((u - 0.5) ** 2) * 20)


def anneal(epoch, period):
    return 0.5 ** (epoch / period) if period else 1


def operator(ctx):
    extra, mod = ctx.extra, ctx.mod
    args = extra.args
    dt, dx = ctx.step()
    it, ix = ctx.indices()
    nt, nx = ctx.size()
    epoch = ctx.tracers["epoch"]
    extrap_wall = odil.core.extrap_quadh
    extrap_init = odil.core.extrap_linear

    def two_levels(frozen):
        """[[u, u(x-), u(x+)] at t, the same at t-1] with initial and wall conditions as ghosts."""
        frozen = frozen and bool(args.keep_frozen)
        now = [ctx.field("u", 0, s, frozen=frozen) for s in (0, -1, 1)]
        old = [ctx.field("u", -1, s, frozen=frozen) for s in (0, -1, 1)]
        if args.keep_init:
            u0 = extra.init_u
            u0s = [u0, mod.roll(u0, 1, axis=0), mod.roll(u0, -1, axis=0)]
            for i in range(3):
                old[i] = mod.where(it == 0, extrap_init(now[i], u0s[i][None, :]), old[i])
        for q in (now, old):
            q[1] = mod.where(ix == 0, extrap_wall(q[2], q[0], 0), q[1])
            q[2] = mod.where(ix == nx - 1, extrap_wall(q[1], q[0], 0), q[2])
        return now, old

    q, qo = two_levels(frozen=False)
    u_t = (q[0] - qo[0]) / dt
    grad_m = ((q[0] + qo[0]) - (q[1] + qo[1])) / (2 * dx)  # face x - 1/2, mean of two time levels
    grad_p = ((q[2] + qo[2]) - (q[0] + qo[0])) / (2 * dx)
    f, fo = two_levels(frozen=True)
    face_m = ((f[0] + fo[0]) + (f[1] + fo[1])) * 0.25
    face_p = ((f[2] + fo[2]) + (f[0] + fo[0])) * 0.25
    if args.infer_k:
        net = ctx.neural_net("k_net")
        km = mod.sigmoid(net(face_m)[0]) * args.kmax
        kp = mod.sigmoid(net(face_p)[0]) * args.kmax
    else:
        km, kp = reference_k(face_m, mod), reference_k(face_p, mod)
    fu = u_t - (grad_p * kp - grad_m * km) / dx
    if not args.keep_init:
        fu = mod.where(it == 0, ctx.cast(0), fu)
    res = [("fu", fu)]
    if extra.imp_size:
        k = args.kimp * (np.prod(ctx.size()) / extra.imp_size) ** 0.5
        res.append(("imp", extra.imp_mask * (q[0] - extra.imp_u) * k))
    if args.kxreg:
        u_x = mod.where(ix == 0, ctx.cast(0), (q[0] - q[1]) / dx)
        res.append(("xreg", u_x * (args.kxreg * anneal(epoch, args.kxregdecay))))
    if args.ktreg:
        u_tt = mod.where(it == 0, ctx.cast(0), (q[0] - qo[0]) / dt)
        res.append(("treg", u_tt * (args.ktreg * anneal(epoch, args.ktregdecay))))
    if args.kwreg and args.infer_k:
        ww = ctx.domain.arrays_from_field(ctx.state.fields["k_net"])
        ww = mod.concatenate([mod.flatten(w) for w in ww], axis=0)
        res.append(("wreg", (mod.stop_gradient(ww) - ww) * (args.kwreg * anneal(epoch, args.kwregdecay))))
    return res


def imposed_mask(args, domain):
    size = int(np.prod(domain.cshape))
    rng = np.random.default_rng(args.seed)
    idx = np.arange(size)
    if args.imposed == "stripe":
        t = domain.mod.numpy(domain.points("t")).flatten()
        idx = idx[abs(t[idx] - 0.5) < 1 / 6]
    elif args.imposed == "none":
        idx = idx[:0]
    idx = np.unique(rng.permutation(idx)[: min(args.nimp, idx.size)])
    mask = np.zeros(size)
    mask[idx] = 1
    return mask.reshape(domain.cshape), len(idx)


def make_problem(args):
    dtype = np.float64 if args.double else np.float32
    domain = odil.Domain(cshape=(args.Nt, args.Nx), dimnames=("t", "x"), multigrid=args.multigrid, dtype=dtype)
    mod = domain.mod
    tt, xx = domain.points()
    x1 = mod.array(domain.points_1d()[1])
    extra = argparse.Namespace(args=args)
    extra.init_u = initial_u(x1, mod)
    extra.ref_u = initial_u(xx, mod)
    extra.imp_u = extra.ref_u
    mask, extra.imp_size = imposed_mask(args, domain)
    extra.imp_mask = mod.cast(mask, dtype)
    state = odil.State()
    state.fields["u"] = np.zeros(domain.cshape)
    if args.infer_k:
        state.fields["k_net"] = domain.make_neural_net([1] + list(args.arch_k) + [1])
    state = domain.init_state(state)
    return odil.Problem(operator, domain, extra), state


def parse_args(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--Nt", type=int, default=64)
    parser.add_argument("--Nx", type=int, default=64)
    parser.add_argument("--arch_k", type=int, nargs="*", default=[5, 5], help="Hidden layers of the conductivity net")
    parser.add_argument("--infer_k", type=int, default=0)
    for name in ["kxreg", "kxregdecay", "ktreg", "ktregdecay", "kwreg", "kwregdecay", "noise"]:
        parser.add_argument("--" + name, type=float, default=0)
    parser.add_argument("--kimp", type=float, default=2)
    parser.add_argument("--keep_frozen", type=int, default=1)
    parser.add_argument("--keep_init", type=int, default=1)
    parser.add_argument("--imposed", type=str, choices=["random", "stripe", "none"], default="none")
    parser.add_argument("--nimp", type=int, default=200)
    parser.add_argument("--kmax", type=float, default=0.1)
    odil.util.add_arguments(parser)
    odil.linsolver.add_arguments(parser)
    parser.set_defaults(outdir="out_heat", optimizer="adam", lr=0.001, double=0, multigrid=1, plot_every=2000,
                        report_every=500, history_full=10, history_every=100, frames=10)
    return parser.parse_args(argv)


def main():
    args = parse_args()
    odil.setup_outdir(args)
    problem, state = make_problem(args)
    callback = odil.make_callback(problem, args)
    odil.util.optimize(args, args.optimizer, problem, state, callback)


if __name__ == "__main__":
    main()
