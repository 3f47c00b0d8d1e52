"""Multi-rank path on CPU: world_size 2 over gloo.  The slab driver (odil_amd/slab.py) runs
unchanged with an oracle-backed `ops` double; its result must equal the undivided-domain
oracle.  This covers ghost layout, packed plane exchanges, the P^T cut rule, the loss
partial sums and the all-reduce -- everything in the N>1 path except the HIP kernels
themselves, which tests/test_slab_gpu.py checks with ranks emulated on one GPU."""

import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def global_reference(N, world, epochs, rhs):
    from oracle import odil_np as onp

    cshape = (N * world, N, N)
    dw = (1.0 / N,) * 3
    x = [np.zeros(s) for s in onp.mg_cshapes(cshape)]

    def loss_grad(x):
        loss, grads, _ = onp.poisson_loss_grad(x, rhs, dw)
        return loss, grads

    x, losses = onp.adam_run(x, loss_grad, epochs, 0.005)
    return x, losses


def make_rhs(N, world):
    rng = np.random.default_rng(42)
    return rng.standard_normal((N * world, N, N))


def worker(rank, world, N, epochs, port, out):
    import slab_oracle_ops

    from odil_amd import slab
    from odil_amd.slab import SlabPoissonAdam, TorchDistComm

    slab.hip_ops = slab_oracle_ops  # the exchange logic without a GPU: NumPy-oracle doubles of the kernels

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rhs = torch.from_numpy(make_rhs(N, world))
        run = SlabPoissonAdam(N, rank, world, dtype=torch.float64, device=torch.device("cpu"), rhs_global=rhs)
        comm = TorchDistComm(rank, world)
        losses = []
        for _ in range(epochs):
            run.epoch(comm)
            losses.append(run.last_loss(comm))
        owned = [w.clone().numpy() for w in run.owned_levels()]
        torch.save({"losses": losses, "owned": owned}, os.path.join(out, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_slab_two_ranks_equals_global_oracle(tmp_path, world):
    N, epochs = 8, 3
    port = 29500 + (os.getpid() + world) % 2000
    mp.spawn(worker, args=(world, N, epochs, port, str(tmp_path)), nprocs=world, join=True)
    rhs = make_rhs(N, world)
    x_ref, losses_ref = global_reference(N, world, epochs, rhs)
    results = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=False) for r in range(world)]
    for r in range(world):
        assert np.max(np.abs(np.array(results[r]["losses"]) - np.array(losses_ref)) / np.array(losses_ref)) < 1e-12
    for lvl, ref in enumerate(x_ref):
        nz = ref.shape[0] // world
        for r in range(world):
            got = results[r]["owned"][lvl]
            want = ref[r * nz : (r + 1) * nz]
            assert np.max(np.abs(got - want)) < 1e-12 * max(1.0, np.max(np.abs(want))), (lvl, r)
