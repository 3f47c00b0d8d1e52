#!/usr/bin/env python3
"""Throughput of the five BASELINE.json configs THROUGH THE PUBLIC OPERATOR API
(`import odil_amd as odil`, the example operators unchanged), as the reference measures it:
grid-point-updates/s = prod(domain.cshape) * epochs / wall (reference src/odil/util.py:408-419).

    python bench_configs.py [--configs 1 2 3 4a 4b 5] [--scale 1.0]

Not the driver contract (that is bench.py); results are recorded in DESIGN.md.  --scale < 1
shrinks every grid (smoke runs).
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
for sub in ("poisson", "heat", "velocity_from_tracer"):
    sys.path.insert(0, os.path.join(ROOT, "examples", sub))

import odil_amd as odil  # noqa: E402


def run(problem, state, args, optname, epochs, warmup=2):
    # blocks cached for the previous (differently sized) configuration make the allocator split and retry
    # under the large ones that follow: the 4-D tracer measured 95 instead of 55 ms / epoch after the others
    torch.cuda.empty_cache()
    args.epoch_start, args.epochs = 0, warmup
    odil.util.set_log_file(open(os.devnull, "w"))
    try:
        odil.util.optimize(args, optname, problem, state, None)
    except odil.EarlyStopError:
        pass
    torch.cuda.synchronize()
    args.epochs = epochs
    t0 = time.perf_counter()
    try:
        odil.util.optimize(args, optname, problem, state, None)
    except odil.EarlyStopError:
        pass
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    loss = float(problem.eval_loss_grad(state)[0])
    cells = int(np.prod(problem.domain.cshape))
    return dict(cells=cells, epochs=epochs, wall_s=wall, ms_per_epoch=1e3 * wall / epochs,
                updates_per_s=cells * epochs / wall, loss=loss, fused=problem._fused is not None)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--configs", nargs="*", default=["1", "2", "3", "3b", "4a", "4b", "5", "5b"])
    p.add_argument("--scale", type=float, default=1.0)
    a = p.parse_args()
    sc = lambda n: max(8, int(round(n * a.scale)) // 8 * 8)
    out = {}
    if "1" in a.configs:  # Poisson 1-D N=256, Adam (launch-bound plumbing case)
        import poisson

        args = poisson.parse_args(["--ndim", "1", "--N", "256"])
        problem, state = poisson.make_problem(args)
        out["1: poisson 1D N=256 adam f64 mg"] = run(problem, state, args, "adam", 400)
    if "2" in a.configs:  # Poisson 2-D 1024^2, multigrid decomposition, L-BFGS-B (m=50)
        import poisson

        n = sc(1024)
        args = poisson.parse_args(["--ndim", "2", "--N", str(n)])
        problem, state = poisson.make_problem(args)
        out[f"2: poisson 2D {n}^2 lbfgsb f64 mg"] = run(problem, state, args, "lbfgsb", 100, warmup=1)
    if "3" in a.configs:  # heat inverse (t, x) = 256 x 512, f32, Adam, infer_k
        import heat

        nt, nx = sc(256), sc(512)
        args = heat.parse_args(["--Nt", str(nt), "--Nx", str(nx), "--infer_k", "1", "--imposed", "stripe"])
        problem, state = heat.make_problem(args)
        out[f"3: heat inverse {nt}x{nx} adam f32 mg (traced operator)"] = run(problem, state, args, "adam", 50)
    if "3b" in a.configs:  # heat inverse with two space dimensions: the shape BASELINE names, (t, x, y) = 256 x 512^2
        import heat2d

        nt, nx = sc(256), sc(512)
        args = heat2d.parse_args(["--Nt", str(nt), "--Nx", str(nx), "--Ny", str(nx), "--infer_k", "1", "--imposed", "stripe"])
        problem, state = heat2d.make_problem(args)
        out[f"3b: heat inverse {nt}x{nx}x{nx} adam f32 mg (traced operator)"] = run(problem, state, args, "adam", 20)
    if "4a" in a.configs:  # Poisson 3-D 512^3 multigrid, Adam (the bench.py workload through the API)
        import poisson

        n = sc(512)
        args = poisson.parse_args(["--ndim", "3", "--N", str(n)])
        problem, state = poisson.make_problem(args)
        out[f"4a: poisson 3D {n}^3 adam f64 mg"] = run(problem, state, args, "adam", 20)
    if "4b" in a.configs:  # Poisson 3-D 512^3 Newton (no decomposition), sparse Jacobian + multigrid solve
        import poisson

        n = sc(512)
        args = poisson.parse_args(["--ndim", "3", "--N", str(n), "--multigrid", "0", "--linsolver", "multigrid",
                                   "--linsolver_tol", "1e-10"])
        problem, state = poisson.make_problem(args)
        out[f"4b: poisson 3D {n}^3 newton + gmg f64"] = run(problem, state, args, "newton", 1, warmup=0)
        # the same solve (relative tolerance) once the work buffers exist: the first step also pays for
        # the first-touch device allocations of the solver (several GB)
        out[f"4b: poisson 3D {n}^3 newton + gmg f64, second step"] = run(problem, state, args, "newton", 1, warmup=0)
    if "5" in a.configs:  # velocity from tracer (t, x, y) = 128 x 256 x 256, f32, Adam
        import veltracer

        nt, nx = sc(128), sc(256)
        args = veltracer.parse_args(["--Nt", str(nt), "--Nx", str(nx), "--Ny", str(nx)])
        problem, state = veltracer.make_problem(args)
        out[f"5: veltracer {nt}x{nx}x{nx} adam f32 mg (traced operator)"] = run(problem, state, args, "adam", 20)
    if "5b" in a.configs:  # tracer with three space dimensions (t, x, y, z), 4 fields 'nccc'; 32 x 256^3 fits one GPU
        import veltracer3d

        nt, nx = sc(32), sc(256)
        args = veltracer3d.parse_args(["--Nt", str(nt), "--Nx", str(nx)])
        problem, state = veltracer3d.make_problem(args)
        out[f"5b: veltracer3d {nt}x{nx}^3 adam f32 mg (traced operator)"] = run(problem, state, args, "adam", 5, warmup=1)
    for k, v in out.items():
        print(json.dumps({"config": k, **v}))


if __name__ == "__main__":
    main()
