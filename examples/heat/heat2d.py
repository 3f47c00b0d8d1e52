#!/usr/bin/env python3
"""Inverse heat conduction with TWO space dimensions, (t, x, y): the size BASELINE.json names for
the heat workload (512^2 x 256 t).  The reference example has one space dimension
(examples/heat/heat.py:36-137); this is its flux form applied per space axis -- same time
discretisation (two levels, face gradients averaged in time), conductivity sigmoid(MLP(u)) * kmax
evaluated at the faces on FROZEN u, zero Dirichlet walls through quadratic ghosts, initial row
through linear ghosts, imposed points, annealed smoothness terms.  Parity: this operator run by the
reference's own core.py gives the fixtures tests/golden/heat2d_*.npz.

    python examples/heat/heat2d.py --Nt 32 --Nx 64 --Ny 64 --infer_k 1 --imposed stripe --epochs 200
"""

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import odil_amd as odil  # noqa: E402
from odil_amd import printlog  # noqa: E402

SPACE = (1, 2)  # axes of (t, x, y) that carry diffusion


def unit(axis, s):
    shift = [0, 0, 0]
    shift[axis] = s
    return tuple(shift)


def initial_u(x, y, mod):
    bump = lambda z: mod.exp(-((z - 0.5) ** 2) * 50)
    edge = bump(-mod.cast(0.5, x.dtype))
    return (bump(x) - edge) * (bump(y) - edge)


def reference_k(u, mod):
    return 0.02 * mod.exp(-((u - 0.5) ** 2) * 20)


def anneal(epoch, period):
    return 0.5 ** (epoch / period) if period else 1


def operator(ctx):
    extra, mod = ctx.extra, ctx.mod
    args = extra.args
    steps = ctx.step()
    dt = steps[0]
    idx = ctx.indices()
    it = idx[0]
    sizes = ctx.size()
    epoch = ctx.tracers["epoch"]
    wall, start = odil.core.extrap_quadh, odil.core.extrap_linear
    u0 = extra.init_u[None]

    def read(shift, frozen):
        """u at `shift` and one time level below, with the initial row as a ghost."""
        st, sx, sy = shift
        now = ctx.field("u", st, sx, sy, frozen=frozen)
        old = ctx.field("u", st - 1, sx, sy, frozen=frozen)
        if args.keep_init:
            old = mod.where(it == 0, start(now, mod.roll(u0, (-sx, -sy), axis=(1, 2))), old)
        return now, old

    def stencil(frozen):
        """centre and, per space axis, (minus, plus) neighbours -- each as (now, old) -- walls as ghosts."""
        frozen = frozen and bool(args.keep_frozen)
        centre = read((0, 0, 0), frozen)
        sides = dict()
        for a in SPACE:
            minus, plus = read(unit(a, -1), frozen), read(unit(a, 1), frozen)
            lo, hi = idx[a] == 0, idx[a] == sizes[a] - 1
            minus = tuple(mod.where(lo, wall(p, c, 0), m) for m, p, c in zip(minus, plus, centre))
            plus = tuple(mod.where(hi, wall(m, c, 0), p) for m, p, c in zip(minus, plus, centre))
            sides[a] = (minus, plus)
        return centre, sides

    def conductivity(face):
        if args.infer_k:
            return mod.sigmoid(ctx.neural_net("k_net")(face)[0]) * args.kmax
        return reference_k(face, mod)

    (q, qo), sides = stencil(frozen=False)
    (f, fo), fsides = stencil(frozen=True)
    fu = (q - qo) / dt
    for a in SPACE:
        h = steps[a]
        (m, mo), (p, po) = sides[a]
        (fm, fmo), (fp, fpo) = fsides[a]
        grad_m = ((q + qo) - (m + mo)) / (2 * h)
        grad_p = ((p + po) - (q + qo)) / (2 * h)
        k_m = conductivity(((f + fo) + (fm + fmo)) * 0.25)
        k_p = conductivity(((fp + fpo) + (f + fo)) * 0.25)
        fu = fu - (grad_p * k_p - grad_m * k_m) / h
    if not args.keep_init:
        fu = mod.where(it == 0, ctx.cast(0), fu)
    res = [("fu", fu)]
    if extra.imp_size:
        k = args.kimp * (np.prod(sizes) / extra.imp_size) ** 0.5
        res.append(("imp", extra.imp_mask * (q - extra.imp_u) * k))
    if args.kxreg:
        w = args.kxreg * anneal(epoch, args.kxregdecay)
        for a, name in zip(SPACE, ("xreg", "yreg")):
            slope = mod.where(idx[a] == 0, ctx.cast(0), (q - sides[a][0][0]) / steps[a])
            res.append((name, slope * w))
    if args.ktreg:
        rate = mod.where(it == 0, ctx.cast(0), (q - qo) / dt)
        res.append(("treg", rate * (args.ktreg * anneal(epoch, args.ktregdecay))))
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
    domain = odil.Domain(cshape=(args.Nt, args.Nx, args.Ny), dimnames=("t", "x", "y"), multigrid=args.multigrid,
                         dtype=dtype)
    if domain.multigrid:
        printlog("multigrid levels:", domain.mg_cshapes)
    mod = domain.mod
    _, x1, y1 = [mod.array(p) for p in domain.points_1d()]
    tt, xx, yy = domain.points()
    extra = argparse.Namespace(args=args)
    extra.init_u = initial_u(x1[:, None], y1[None, :], mod)
    extra.ref_u = initial_u(xx, yy, mod)
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
    parser.add_argument("--Nt", type=int, default=32)
    parser.add_argument("--Nx", type=int, default=32)
    parser.add_argument("--Ny", type=int, default=None)
    parser.add_argument("--arch_k", type=int, nargs="*", default=[5, 5], help="Hidden layers of the conductivity net")
    parser.add_argument("--infer_k", type=int, default=0)
    for name in ["kxreg", "kxregdecay", "ktreg", "ktregdecay"]:
        parser.add_argument("--" + name, type=float, default=0)
    parser.add_argument("--kimp", type=float, default=2)
    parser.add_argument("--keep_frozen", type=int, default=1)
    parser.add_argument("--keep_init", type=int, default=1)
    parser.add_argument("--imposed", type=str, choices=["random", "stripe", "none"], default="none")
    parser.add_argument("--nimp", type=int, default=200)
    parser.add_argument("--kmax", type=float, default=0.1)
    odil.util.add_arguments(parser)
    odil.linsolver.add_arguments(parser)
    parser.set_defaults(outdir="out_heat2d", optimizer="adam", lr=0.001, double=0, multigrid=1, plot_every=2000,
                        report_every=500, history_full=10, history_every=100, frames=10)
    args = parser.parse_args(argv)
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
