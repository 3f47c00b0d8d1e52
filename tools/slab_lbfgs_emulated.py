#!/usr/bin/env python3
"""L-BFGS-B on the slab-decomposed Poisson path, ranks emulated as threads on one GPU: time per iteration and rank
against one rank alone (the same driver with the local closure).   python3 tools/slab_lbfgs_emulated.py [world] [N] [iters]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from odil_amd.slab import LocalComm  # noqa: E402
from odil_amd.slab_solvers import SlabPoissonLbfgs, run_threads  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
dev = torch.device("cuda:0")


def one(rank, nranks, comm):
    torch.cuda.set_device(dev)
    run = SlabPoissonLbfgs(N, rank, nranks, dtype=torch.float64, device=dev)
    run.minimize(comm, 3, m=50)  # warm-up: kernels loaded, buffers allocated
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run.minimize(comm, iters, m=50)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / res["nit"], res


alone, res1 = one(0, 1, LocalComm())
print("one rank alone, {}^3: {:.3f} ms per iteration ({} evaluations in {} iterations), loss {:.6e}".format(
    N, 1e3 * alone, res1["funcalls"], res1["nit"], res1["f"]))
torch.cuda.empty_cache()
out = run_threads(world, lambda rank, comm: one(rank, world, comm))
per_rank = max(o[0] for o in out) / world
print("{} emulated ranks of {}^3: {:.3f} ms per iteration and rank ({:+.1f} %), {} evaluations, loss {:.6e}".format(
    world, N, 1e3 * per_rank, 100 * (per_rank / alone - 1), out[0][1]["funcalls"], out[0][1]["f"]))
