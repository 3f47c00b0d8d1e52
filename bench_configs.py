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
for sub in ("poisson", "heat", "velocity_from_tracer", "diffusion"):
    sys.path.insert(0, os.path.join(ROOT, "examples", sub))

import odil_amd as odil  # noqa: E402


def run(problem, state, args, optname, epochs, warmup=2):
    """`epochs` epochs of `odil.util.optimize` (the public driver loop) timed EPOCH BY EPOCH with HIP events recorded in the
    driver's own per-epoch callback, after `warmup` untimed epochs of a first call.  The call's set-up -- the optimizer's
    packed copy of the unknowns, its moment arrays allocated and zeroed, the first evaluation for the callback: tens of GB
    of fresh allocations at the space-time sizes -- happens before the first event and is reported apart (`setup_s`):
    round 4's driver run of config 5b measured 153 ms per epoch for 10 epochs where the epochs themselves take 30, because
    a second and a half of allocation on a fresh box sat inside the wall clock this used to divide.  ms_per_epoch is the
    MEDIAN of the epoch times (max and mean beside it); peak VRAM from torch's allocator statistics."""
    # blocks cached for the previous (differently sized) configuration make the allocator split and retry
    # under the large ones that follow: the 4-D tracer measured 95 instead of 55 ms / epoch after the others
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    odil.util.set_log_file(open(os.devnull, "w"))
    if warmup:  # (Newton too: the first step builds its Jacobian kernel and the solver's work buffers)
        args.epoch_start, args.epochs = 0, warmup if optname != "newton" else 1
        try:
            odil.util.optimize(args, optname, problem, state, None)
        except odil.EarlyStopError:
            pass
        torch.cuda.synchronize()
    events = []

    def callback(state_, epoch, pinfo):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        events.append(e)

    for _ in range(epochs + 2):  # (the runtime grows its event pool in chunks: make the events exist before they are needed)
        callback(None, 0, None)
    torch.cuda.synchronize()
    events.clear()
    args.epoch_start, args.epochs = 0, epochs
    t0 = time.perf_counter()
    try:
        odil.util.optimize(args, optname, problem, state, callback)
    except odil.EarlyStopError:
        pass
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    # events[0] follows the call's initial evaluation (reference util.py:223-225), events[k] epoch k
    times = [a.elapsed_time(b) for a, b in zip(events[:-1], events[1:])]
    done = len(times)
    loss = float(problem.eval_loss_grad(state)[0])
    cells = int(np.prod(problem.domain.cshape))
    med = float(np.median(times)) if times else 1e3 * wall / max(epochs, 1)
    extra = dict()
    ev = getattr(problem, "_fused", None)
    if optname == "adam" and ev is not None and ev.__dict__.get("_small_u") is not None:
        # a problem small enough for whole epochs in one launch (odil_poisson_small_epochs): the per-epoch callback above
        # forces one launch PER epoch; without a callback (or with one that tells its cadence: util.make_callback) the
        # optimizer runs them in one launch -- 4000 epochs, the call's set-up included in the time
        n = 4000
        args.epoch_start, args.epochs = 0, n
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        odil.util.optimize(args, optname, problem, state, None)
        torch.cuda.synchronize()
        extra["us_per_epoch_whole_epochs_in_one_launch"] = 1e6 * (time.perf_counter() - t1) / n
        extra["whole_epochs_kernel"] = True
    return dict(**extra, cells=cells, epochs=done, wall_s=wall, ms_per_epoch=med, ms_mean=float(np.mean(times)) if times else med,
                ms_max=float(np.max(times)) if times else med, ms_min=float(np.min(times)) if times else med,
                setup_s=wall - 1e-3 * float(np.sum(times)) if times else 0.0,
                updates_per_s=cells / (med * 1e-3), loss=loss, fused=problem._fused is not None,
                traced=getattr(problem, "_traced", None) is not None,
                vram_peak_gb=torch.cuda.max_memory_allocated() / 1e9, vram_reserved_gb=torch.cuda.memory_reserved() / 1e9)


CONFIGS = {
    # key: (example module, argv builder(sc), optimizer, default epochs, warmup, name)
    "1": ("poisson", lambda sc: ["--ndim", "1", "--N", "256"], "adam", 400, 2, "poisson 1D N=256 adam f64 mg"),
    "2": ("poisson", lambda sc: ["--ndim", "2", "--N", str(sc(1024))], "lbfgsb", 100, 1, "poisson 2D {0}^2 lbfgsb f64 mg"),
    "3": ("heat", lambda sc: ["--Nt", str(sc(256)), "--Nx", str(sc(512)), "--infer_k", "1", "--imposed", "stripe"], "adam", 50, 2,
          "heat inverse {0}x{1} adam f32 mg (traced operator)"),
    "3b": ("heat2d", lambda sc: ["--Nt", str(sc(256)), "--Nx", str(sc(512)), "--Ny", str(sc(512)), "--infer_k", "1",
                                 "--imposed", "stripe"], "adam", 20, 2, "heat inverse {0}x{1}x{1} adam f32 mg (traced operator)"),
    "4a": ("poisson", lambda sc: ["--ndim", "3", "--N", str(sc(512))], "adam", 20, 2, "poisson 3D {0}^3 adam f64 mg"),
    "4b": ("poisson", lambda sc: ["--ndim", "3", "--N", str(sc(512)), "--multigrid", "0", "--linsolver", "multigrid",
                                  "--linsolver_tol", "1e-10"], "newton", 1, 1, "poisson 3D {0}^3 newton + gmg f64 (first step from the zero state, work buffers warm)"),
    # not a BASELINE config: the general Newton route on an operator with VARIABLE coefficients (examples/diffusion)
    "4c": ("diffusion", lambda sc: ["--ndim", "3", "--N", str(sc(256)), "--kind", "jump", "--linsolver", "multigrid",
                                   "--linsolver_tol", "1e-10"], "newton", 1, 1,
           "diffusion div(k grad u), k jumps 1 : 1000, 3D {0}^3 newton + variable-coefficient gmg f64"),
    "5": ("veltracer", lambda sc: ["--Nt", str(sc(128)), "--Nx", str(sc(256)), "--Ny", str(sc(256))], "adam", 20, 2,
          "veltracer {0}x{1}x{1} adam f32 mg (traced operator)"),
    # (20 epochs per `optimize` call: the call's own set-up -- 17 GB of moment arrays allocated and zeroed, the packed
    # vector filled and written back -- is ~23 ms, 4.6 ms per epoch of a 5-epoch call: 38.6 against 33.9 ms / epoch)
    "5b": ("veltracer3d", lambda sc: ["--Nt", str(sc(32)), "--Nx", str(sc(256))], "adam", 20, 2,
           "veltracer3d {0}x{1}^3 adam f32 mg (traced operator)"),
}


def model_bytes_per_update(problem, state, optname, nout):
    """SURVEY.md 8(d) minimum-traffic model for k multigrid fields and n_out outputs under Adam:
    synthesis k (S + 1) + residual (k + 1 constant + n_out) + adjoint (n_out + k) + P^T chain k (2 S - 1) + Adam 7 k S
    = k (10 S + 2) + 2 n_out + 1 words per grid cell; for L-BFGS-B the evaluation + the two-loop recursion (below); None for
    Newton (bench.py prices its V-cycles)."""
    domain = problem.domain
    k = sum(1 for f in state.fields.values() if isinstance(f, (odil.Field, odil.MultigridField)))
    nl = domain.mg_nlvl if domain.multigrid else 1
    axes = sum(1 for a in (domain.mg_axes if domain.multigrid else [])) or domain.ndim
    S = sum(2.0 ** (-axes * l) for l in range(nl))
    wordsize = np.dtype(domain.dtype).itemsize
    if optname == "lbfgsb":
        # per iteration: ONE loss + gradient evaluation (synthesis S + 1, residual 3, adjoint 2, P^T chain 2 S - 1 words)
        # + the two-loop recursion over m = 50 (s, y) pairs, 4 m S words + ~6 S of vector updates; the
        # extra evaluations of the line search (about one in ten iterations here) are not priced
        m = 50
        return (k * (3 * S + (4 * m + 6) * S) + 2 * nout + 1 + 2 * k) * wordsize
    if optname != "adam":
        return None
    return (k * (10 * S + 2) + 2 * nout + 1) * wordsize


def run_config(key, scale=1.0, epochs=None, warmup=None):
    """One BASELINE config through the public operator API; returns a dict (see `run`) + name / optimizer / dtype."""
    import importlib

    sc = lambda n: max(8, int(round(n * scale)) // 8 * 8)
    modname, argv, optname, nepochs, nwarm, name = CONFIGS[key]
    ex = importlib.import_module(modname)
    args = ex.parse_args(argv(sc))
    problem, state = ex.make_problem(args)
    res = run(problem, state, args, optname, epochs or nepochs, warmup=nwarm if warmup is None else min(warmup, nwarm))
    cs = problem.domain.cshape
    res["name"] = name.format(cs[0], cs[-1])
    res["optimizer"] = optname
    res["dtype"] = "f64" if np.dtype(problem.domain.dtype) == np.float64 else "f32"
    nout = len(problem.eval_loss_grad(state)[2])
    res["model_bytes_per_update"] = model_bytes_per_update(problem, state, optname, nout)
    return res


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--configs", nargs="*", default=["1", "2", "3", "3b", "4a", "4b", "5", "5b"])
    p.add_argument("--scale", type=float, default=1.0)
    a = p.parse_args()
    for key in a.configs:
        res = run_config(key, a.scale)
        print(json.dumps({"config": key + ": " + res.pop("name"), **res}))


if __name__ == "__main__":
    main()
