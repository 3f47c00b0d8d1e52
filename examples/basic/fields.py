#!/usr/bin/env python3
"""One unknown field per location of a 2-D grid -- cell centres, nodes, x-faces, y-faces -- each fitted to the same
linear function of its own points (the reference's tutorial examples/basic/fields.py:16-40; same outputs and state,
including the network the operator never evaluates).  Four multigrid fields of four shapes, four outputs of four
shapes: parity fixture tests/golden/basic_fields_*.npz.

    python examples/basic/fields.py --Nx 8 --Ny 4 --epochs 500
"""

import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import odil_amd as odil  # noqa: E402
from odil_amd import printlog  # noqa: E402

FIELDS = (("uc", "cc"), ("un", "nn"), ("ufx", "nc"), ("ufy", "cn"))  # (key, location) in state order


def target(x, y):
    return x * 0.25 + y * 0.5


def operator(ctx):
    out = []
    for key, loc in FIELDS:
        x, y = ctx.points(loc=loc)
        out.append((key, ctx.field(key) - target(x, y)))
    return out


def make_problem(args):
    dtype = np.float64 if args.double else np.float32
    domain = odil.Domain(cshape=(args.Nx, args.Ny), dimnames=["x", "y"], lower=(0, 0), upper=(2, 1), dtype=dtype,
                         multigrid=args.multigrid, mg_interp=args.mg_interp, mg_axes=[True, True], mg_nlvl=args.nlvl)
    fields = {key: odil.Field(np.zeros(domain.size(loc=loc)), loc=loc) for key, loc in FIELDS}
    fields["net"] = domain.make_neural_net([2, 4, 2])
    state = domain.init_state(odil.State(fields=fields))
    return odil.Problem(operator, domain), state


def parse_args(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--Nx", type=int, default=8, help="Cells in x")
    parser.add_argument("--Ny", type=int, default=4, help="Cells in y")
    odil.util.add_arguments(parser)
    odil.linsolver.add_arguments(parser)
    parser.set_defaults(outdir="out_fields", echo=1, frames=1, plot_every=100, report_every=50, history_every=10,
                        optimizer="adam", lr=1e-2, multigrid=1)
    return parser.parse_args(argv)


def max_errors(problem, state):
    """Largest deviation of each field from the target on its own points."""
    domain = problem.domain
    res = dict()
    for key, loc in FIELDS:
        x, y = domain.points(loc=loc)
        u = np.asarray(domain.mod.numpy(domain.field(state, key)))
        res[key] = float(np.max(np.abs(u - target(np.asarray(domain.mod.numpy(x)), np.asarray(domain.mod.numpy(y))))))
    return res


def main():
    args = parse_args()
    odil.setup_outdir(args)
    problem, state = make_problem(args)
    callback = odil.make_callback(problem, args)
    odil.util.optimize_grad(args, args.optimizer, problem, state, callback)
    printlog("max errors:", max_errors(problem, state))


if __name__ == "__main__":
    main()
