#!/usr/bin/env python3
"""Headline benchmark: grid-point-updates/s of the ODIL hot path on MI355X.

Workload (BASELINE.json metric): 3-D Poisson 512^3, multigrid decomposition (9 levels),
f64, Adam -- one "step" is one optimizer epoch of the reference's hot loop
(reference src/odil/optimizer.py:331-336): multigrid synthesis -> residual + loss ->
adjoint -> P^T chain -> Adam update.  Metric = prod(cshape) * steps / wall
(reference src/odil/util.py:408-419), inputs resident in HBM, synthetic (`hat`
reference solution, discrete rhs, zero initial state; poisson.py:21-24,71-86,264-266).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--N 512] [--ndim 3]

N > 1: one rank per GPU over RCCL.  Either the caller starts the ranks (`python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`: RANK / WORLD_SIZE are then in the environment), or plain
`python bench.py --gpus N` starts them itself as FRESH child processes before this process has touched
the GPU, relays rank 0's JSON line and exits with the children's status.
Prints ONE JSON line on rank 0.
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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=40)
    p.add_argument("--warmup", type=int, default=10)
    p.add_argument("--N", type=int, default=512)
    p.add_argument("--ndim", type=int, default=3)
    p.add_argument("--dtype", type=str, default="f64", choices=["f64", "f32"])
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--cpu_N", type=int, default=128, help="grid size of the CPU-baseline sample")
    return p.parse_args()


def algorithmic_bytes_per_update(ndim, nlvl, wordsize):
    """(10 S + 5) words per fine cell per epoch, S = sum_l 2^(-d l)."""
    S = sum(2.0 ** (-ndim * l) for l in range(nlvl))
    return (10 * S + 5) * wordsize, S


def cpu_baseline(ndim, N, budget_s=20.0):
    """The oracle (NumPy port of the reference op sequence) timed on this host, 1 thread."""
    from oracle import odil_np as onp

    cshape = (N,) * ndim
    dw = onp.step(cshape)
    rhs = onp.poisson_discrete_rhs(onp.poisson_ref_u(cshape), dw)
    x = [np.zeros(s) for s in onp.mg_cshapes(cshape)]
    m = [np.zeros_like(a) for a in x]
    v = [np.zeros_like(a) for a in x]

    def epoch(k):
        nonlocal x, m, v
        loss, grads, _ = onp.poisson_loss_grad(x, rhs, dw)
        x, m, v = onp.adam_step(x, m, v, grads, k, 0.005)

    epoch(1)
    t0 = time.perf_counter()
    k = 0
    while True:
        epoch(k + 2)
        k += 1
        el = time.perf_counter() - t0
        if el > budget_s or k >= 50:
            break
    return {
        "value": N**ndim * k / el,
        "unit": "grid-point-updates/s",
        "cores": 1,
        "kind": "port",
        "sample": "oracle/odil_np.py, Poisson {}-D {}^{} f64 multigrid Adam, {} epochs in {:.1f} s".format(
            ndim, N, ndim, k, el
        ),
    }


def measured_traffic(kernel, ndim, N, dtype):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/
    (collected separately with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950 correction applied);
    None when no profile of this kernel / workload is on record."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
    except OSError:
        return None
    if rec.get("kernel") == kernel and (ndim, N, dtype) == (3, 512, "f64"):
        return rec["traffic_bytes_per_launch"]
    return None


class Timers:
    """HIP-event pairs per kernel family, recorded on the stream the kernels run on."""

    def __init__(self):
        self.pairs = {}

    def section(self, name):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        self.pairs.setdefault(name, []).append((a, b))
        return a, b

    def summary(self):
        return {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in self.pairs.items()}


def spawn_ranks(ngpus):
    """`python bench.py --gpus N` outside a launcher: start N ranks with torch.distributed.run as a child
    process (this process has not initialised the GPU and never does), pass its output through, return its
    exit status."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        import torch.distributed as dist

        backend = os.environ.get("ODIL_DIST_BACKEND", "nccl")  # nccl == RCCL on ROCm
        ngpu = torch.cuda.device_count()
        local_rank = local_rank % max(ngpu, 1)  # (tests may oversubscribe one GPU with gloo)
        torch.cuda.set_device(local_rank)
        # No silent fallback: a rank that cannot initialise RCCL fails the job (gloo, which stages the planes
        # through the host, only when ODIL_DIST_BACKEND asks for it -- the CPU-side tests do).
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == world
        if args.gpus != world:
            raise SystemExit("bench.py: --gpus {} but WORLD_SIZE={}".format(args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    dev = torch.device("cuda", local_rank)

    from odil_amd.poisson_path import PoissonMultigridAdam
    from odil_amd.slab import SlabPoissonAdam, TorchDistComm

    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    ndim, N = args.ndim, args.N
    comm = None
    if world > 1:
        # weak scaling: every rank owns an N^3 slab of the (world*N, N, N) grid
        assert ndim == 3, "the slab decomposition is 3-D"
        run = SlabPoissonAdam(N, rank, world, dtype=dtype, device=dev)
        comm = TorchDistComm(rank, world)
        step = lambda timers=None: run.epoch(comm, timers)
    else:
        run = PoissonMultigridAdam(ndim, N, dtype=dtype, device=dev)
        step = lambda timers=None: run.epoch(timers)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    timers = Timers()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(timers)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss = run.last_loss(comm) if world > 1 else run.last_loss()
    dist_world, comm_backend = 1, None
    if world > 1:
        dist_world = dist.get_world_size()
        comm_backend = "rccl" if dist.get_backend() == "nccl" else dist.get_backend()

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        cells_total = run.global_cells
        value = cells_total * args.steps / elapsed
        wordsize = 8 if dtype == torch.float64 else 4
        abytes, S = algorithmic_bytes_per_update(ndim, run.nlvl, wordsize)
        kt = timers.summary()
        n_unknowns = run.n_unknowns_local
        moved_bytes = None
        if "adjoint_transpose" in kt:
            # Dominant kernel: stencil adjoint + first transposed prolongation + the Adam updates of levels 0 and
            # 1 in one launch.  ALGORITHMIC bytes by the minimum-traffic model of SURVEY.md 8(d) for the work this
            # launch covers: stencil adjoint 2 words (read r, write g) + first level of the P^T chain 1 + 1/8
            # (read g, write g1) + Adam 7 words per unknown of levels 0 and 1 (7 + 7/8) = 11 words per fine cell.
            # The launch itself moves less -- g never reaches memory: read fu; read + write x, m, v of level 0;
            # write g1 and read + write x, m, v of level 1 = 7 + 7/8 words -- reported beside it.
            dom_name = "k_poisson_adjoint_tile<{}> (adjoint + first P^T + Adam of levels 0 and 1)"
            adam_bytes = 11.0 * run.local_cells * wordsize
            moved_bytes = (7.0 + 7.0 / 8.0) * run.local_cells * wordsize
            adam_ms = kt["adjoint_transpose"]
        elif kt.get("adjoint", 0) > kt.get("adam", 0):
            # Dominant kernel: adjoint with the finest-level Adam update fused in:
            # read fu, x, m, v; write gu, x, m, v = 8 words per fine cell.
            dom_name = "k_poisson_adjoint<{}, true> (+Adam of the finest level)"
            adam_bytes = 8.0 * run.local_cells * wordsize
            adam_ms = kt["adjoint"]
        else:
            # Dominant kernel: Adam over the packed multigrid state (7 words per unknown).
            dom_name = "k_adam<{}>"
            adam_bytes = 7.0 * n_unknowns * wordsize
            adam_ms = kt["adam"]
        achieved = adam_bytes / (adam_ms * 1e-3) / 1e9
        out = {
            "metric": "grid-point-updates/s, Poisson {}^{} multigrid".format(N, ndim),
            "value": value,
            "unit": "grid-point-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": "3D Poisson 512^3 multigrid (9 levels) Adam epoch, {}".format(
                    "1xMI355X" if world == 1 else "512^3 slab per GPU x {} MI355X".format(world))
                if (ndim, N) == (3, 512)
                else "{}D Poisson {}^{} multigrid Adam epoch".format(ndim, N, ndim),
                "cells_per_gpu": run.local_cells,
                "levels": run.nlvl,
                "optimizer": "adam lr=0.005",
                "decomposition": "slab x{}".format(world) if world > 1 else "none",
                "rccl_ranks": dist_world,
                "comm_backend": comm_backend,
            },
            "roofline": {
                "kernel": dom_name.format("double" if wordsize == 8 else "float"),
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(dom_name.format("double" if wordsize == 8 else "float"), ndim, N, args.dtype),
                "algorithmic_bytes_per_launch": adam_bytes,
                "avg_launch_ms": adam_ms,
                "compulsory_bytes_per_launch": moved_bytes,
                "achieved_on_compulsory_bytes": None if moved_bytes is None else moved_bytes / (adam_ms * 1e-3) / 1e9,
            },
            "epoch_roofline": {
                "algorithmic_bytes_per_update": abytes,
                "achieved": value * abytes / world / 1e9,
                "frac": value * abytes / world / 1e9 / HBM_PEAK_GBS,
                "unit": "GB/s per GPU",
            },
            "kernel_ms": kt,
            "loss_after": loss,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(ndim, args.cpu_N if ndim == 3 else min(N, 2048))
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
