#!/usr/bin/env python3
"""Velocity from tracer with THREE space dimensions, (t, x, y, z): the shape BASELINE.json names
for the flow-reconstruction workload (256^3 x 128 t).  The reference example is (t, x, y)
(examples/velocity_from_tracer/veltracer.py:34-130); this is the same discretisation with one more
transport direction: fields u, vx, vy, vz at 'nccc', first-order upwinding from the side the FROZEN
velocity comes from, initial / final tracer imposed in rows 0 and Nt, Laplacian and time
smoothness of the velocity.  Parity: this operator run by the reference's own core.py gives the
fixtures tests/golden/veltracer3d_*.npz.

    python examples/velocity_from_tracer/veltracer3d.py --Nt 16 --Nx 32 --epochs 200
"""

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import odil_amd as odil  # noqa: E402
from odil_amd import printlog  # noqa: E402

VEL = ("vx", "vy", "vz")
LOC = "nccc"


def blob(x, y, z, t):
    """Gaussian tracer blob carried along a diagonal."""
    c = 0.35 + 0.3 * t
    return np.exp(-((x - c) ** 2 + (y - c) ** 2 + (z - 0.5) ** 2) * 40)


def unit(axis, s, st=0):
    shift = [st, 0, 0, 0]
    shift[axis] = s
    return tuple(shift)


def operator(ctx):
    extra, mod = ctx.extra, ctx.mod
    args = extra.args
    steps = ctx.step()
    dt, dx = steps[0], steps[1]
    it = ctx.indices(loc=LOC)[0]
    nt = ctx.size()[0]
    zero = ctx.cast(0)

    def upwind(um, u, up, v):
        return mod.where(v > 0, u - um, mod.where(v < 0, up - u, (up - um) * 0.5))

    u = ctx.field("u")
    u_prev = ctx.field("u", -1, 0, 0, 0)
    transport = zero
    for a, key in zip((1, 2, 3), VEL):
        v, v_frozen = ctx.field(key), ctx.field(key, frozen=True)
        um, up = ctx.field("u", *unit(a, -1, st=-1)), ctx.field("u", *unit(a, 1, st=-1))
        transport = transport + v * upwind(um, u_prev, up, v_frozen) / steps[a]
    u_old = mod.where(it == 1, extra.u_init[None], u_prev)
    fu = (u - u_old) / dt + transport
    fu = mod.where(it == 0, (u - extra.u_init[None]) / dx, fu)
    res = [fu, mod.where(it == nt - 1, (u - extra.u_final[None]) / dx, zero) * args.kimp]
    if args.kxreg:
        for key in VEL:
            q = ctx.field(key)
            lap = zero
            for a in (1, 2, 3):
                lap = lap + (ctx.field(key, *unit(a, 1)) - 2 * q + ctx.field(key, *unit(a, -1))) / steps[a] ** 2
            res.append(lap * args.kxreg)
    if args.ktreg:
        for key in VEL:
            d = (ctx.field(key) - ctx.field(key, -1, 0, 0, 0)) * (args.ktreg / dt)
            res.append(mod.where(it == 0, zero, d))
    return res


def make_problem(args):
    dtype = np.float64 if args.double else np.float32
    domain = odil.Domain(cshape=(args.Nt, args.Nx, args.Ny, args.Nz), dimnames=("t", "x", "y", "z"),
                         lower=(0, 0, 0, 0), upper=(1, 1, 1, 1), dtype=dtype, multigrid=args.multigrid,
                         mg_interp=args.mg_interp, mg_nlvl=args.nlvl)
    if domain.multigrid:
        printlog("multigrid levels:", domain.mg_cshapes)
    mod = domain.mod
    x, y, z = np.meshgrid(*domain.points_1d("x", "y", "z"), indexing="ij")
    extra = argparse.Namespace(args=args)
    extra.u_init = mod.cast(blob(x, y, z, 0), dtype)
    extra.u_final = mod.cast(blob(x, y, z, 1), dtype)
    state = odil.State()
    for key in ("u",) + VEL:
        state.fields[key] = odil.Field(None, loc=LOC)
    state = domain.init_state(state)
    return odil.Problem(operator, domain, extra), state


def parse_args(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--Nt", type=int, default=None)
    parser.add_argument("--Nx", type=int, default=32)
    parser.add_argument("--Ny", type=int, default=None)
    parser.add_argument("--Nz", type=int, default=None)
    parser.add_argument("--kxreg", type=float, default=0.01, help="Laplacian regularization weight")
    parser.add_argument("--ktreg", type=float, default=1, help="Time regularization weight")
    parser.add_argument("--kimp", type=float, default=10, help="Imposed values weight")
    parser.add_argument("--slab", type=int, default=0, help="Slab decomposition along x over the ranks of torch.distributed")
    odil.util.add_arguments(parser)
    odil.linsolver.add_arguments(parser)
    parser.set_defaults(outdir="out_veltracer3d", frames=5, plot_every=100, report_every=100, history_every=10,
                        optimizer="adam", lr=0.01, multigrid=1)
    args = parser.parse_args(argv)
    args.Nt = args.Nt or args.Nx
    args.Ny = args.Ny or args.Nx
    args.Nz = args.Nz or args.Nx
    return args


def main():
    args = parse_args()
    if args.slab:
        # one process per GPU (python -m torch.distributed.run --nproc-per-node N veltracer3d.py --slab 1 ...): every
        # rank builds the global problem, owns a slab of x, logs through rank 0, and all ranks write their part of the
        # final tracer field into one raw + XDMF2 file
        from odil_amd.slab import init_distributed
        from odil_amd.slab_traced import optimize_slab

        import torch.distributed as dist

        rank, world, _ = init_distributed()
        outdir = os.path.abspath(args.outdir)
        if rank == 0:
            odil.setup_outdir(args)  # (may clear the directory: the other ranks touch it only after the barrier)
        if world > 1:
            dist.barrier()
        if rank != 0:
            os.makedirs(outdir, exist_ok=True)
            odil.util.set_log_file(open(os.devnull, "w"))
            args.epochs = args.epochs or args.frames * args.plot_every
        problem, state = make_problem(args)
        run = optimize_slab(args, problem, state)
        u_last = run.owned_arrays()[0][-1]  # finest level of u at the final time: this rank's (x, y, z) planes
        odil.write_raw_slab(u_last, os.path.join(outdir, "u_final.xmf"), rank, world, axis=0,
                            spacing=[float(h) for h in problem.domain.step()[1:]][::-1], name="u",
                            barrier=dist.barrier if world > 1 else None)
        return run
    odil.setup_outdir(args)
    problem, state = make_problem(args)
    callback = odil.make_callback(problem, args)
    odil.util.optimize(args, args.optimizer, problem, state, callback)


if __name__ == "__main__":
    main()
