"""Slab decomposition of TRACED operators on CPU, world_size 2 and 3 over gloo: odil_amd/slab_traced.py runs
unchanged -- its packed exchanges, the periodic wrap planes of `Context.field`, the exchange-free P^T chain with
one deferred halo-add, partial loss sums -- with CPU doubles for the kernels (tests/slab_traced_double.py for the
generated ones, tests/slab_oracle_ops.py for the transfers and Adam).  The result must equal the UNDIVIDED
problem evaluated by the generic oracle (oracle/odil_generic.py, itself pinned against the reference's
fixtures) under the oracle's Adam: tracer velocity in two and in three space dimensions."""

import argparse
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
sys.path.insert(0, os.path.join(ROOT, "examples", "velocity_from_tracer"))
sys.path.insert(0, os.path.join(ROOT, "examples", "heat"))


FACTORS = [1.0, 0.5, 2.0]  # multigrid factors of the `-factors` cases: u = f_0 w_0 + P(f_1 w_1 + P(f_2 w_2))


def make_problem(which, world, nx_rank=8):
    """The GLOBAL problem on CPU tensors (no kernel runs here) with a random state: (problem, state)."""
    import odil_amd as odil

    odil.runtime._mod = odil.ModRocm(device="cpu")
    odil.util.set_log_file(open(os.devnull, "w"))
    scaled = which.endswith("-factors")
    which = which.split("-")[0]
    ex = __import__(which)
    nx = nx_rank * world
    argv = ["--Nt", "8", "--Nx", str(nx), "--Ny", "8", "--double", "1"] + (["--Nz", "8"] if which == "veltracer3d" else [])
    if which == "heat2d":
        # (keep_init 0: with it the operator rolls the constant initial row along x, which the CPU double's per-rank
        # slices of the constants cannot express -- the generated kernel indexes the GLOBAL constants and is checked
        # with keep_init 1 by the emulated ranks of tests/test_slab_gpu.py)
        argv += ["--infer_k", "1", "--imposed", "stripe", "--keep_init", "0"]
    args = ex.parse_args(argv)
    problem, state = ex.make_problem(args)
    if scaled:
        for f in state.fields.values():
            if isinstance(f, odil.MultigridField):
                assert len(f.terms) == len(FACTORS)
                f.factors = list(FACTORS)
    rng = np.random.default_rng(7)
    arrays = [torch.tensor(rng.standard_normal(tuple(a.shape)) * 0.1) for a in problem.domain.arrays_from_state(state)]
    problem.domain.arrays_to_state(arrays, state)
    return problem, state


def local_extra(extra, off, n):
    """The rank's x range of the operator's constant arrays: u_init / u_final are (x, y[, z]) arrays; heat2d's
    init_u is (x, y), its imposed values and mask (t, x, y) (imp_size, a global count, stays)."""
    if hasattr(extra, "imp_mask"):
        return argparse.Namespace(args=extra.args, init_u=extra.init_u[off:off + n], imp_size=extra.imp_size,
                                  imp_mask=extra.imp_mask[:, off:off + n], imp_u=extra.imp_u[:, off:off + n])
    return argparse.Namespace(args=extra.args, u_init=extra.u_init[off:off + n], u_final=extra.u_final[off:off + n])


def worker(rank, world, which, epochs, port, out, nx_rank=8):
    import slab_oracle_ops
    import slab_traced_double

    from odil_amd import slab_traced
    from odil_amd.slab import TorchDistComm

    slab_traced.hip_ops = slab_oracle_ops
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        problem, state = make_problem(which, world, nx_rank)
        run = slab_traced.SlabTracedAdam(problem, state, rank, world, axis=1, lr=0.01, device=torch.device("cpu"),
                                         kernels=slab_traced_double.make_kernels(local_extra))
        comm = TorchDistComm(rank, world)
        losses = []
        for _ in range(epochs):
            run.epoch(comm)
            losses.append(run.last_loss(comm))
        # the planes two ranks share: the boundary planes of a rank and the inner ghost planes of its neighbours
        shared = []
        for e in run.entries:
            for lv, a in zip(e.get("levels", []), e.get("x", [])):
                if lv.replicated:
                    continue
                shared.append(dict(own_lo=lv.planes(a, 0).clone().numpy(), own_hi=lv.planes(a, lv.n - 1).clone().numpy(),
                                   ghost_lo=lv.planes(a, -1).clone().numpy() if lv.g_lo else None,
                                   ghost_hi=lv.planes(a, lv.n).clone().numpy() if lv.g_hi else None))
        torch.save({"losses": losses, "owned": [a.clone().numpy() for a in run.owned_arrays()], "shared": shared,
                    "redundant": run.redundant},
                   os.path.join(out, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def undivided(which, world, epochs, nx_rank=8):
    from oracle import odil_generic as og
    from oracle import odil_np as onp

    problem, state = make_problem(which, world, nx_rank)
    domain = problem.domain
    geom = og.Geometry.of(domain)
    fields = og.fields_of_state(domain, state)
    def get(fields):  # arrays in `Domain.arrays_from_state` order
        out = []
        for f in fields.values():
            out += list(f["terms"]) if f["kind"] == "mg" else (
                [f["array"]] if f["kind"] in ("field", "array") else list(f["weights"]) + list(f["biases"]))
        return out

    def put(fields, x):
        k = 0
        for f in fields.values():
            if f["kind"] == "mg":
                f["terms"] = x[k:k + len(f["terms"])]
                k += len(f["terms"])
            elif f["kind"] in ("field", "array"):
                f["array"] = x[k]
                k += 1
            else:
                nw, nb = len(f["weights"]), len(f["biases"])
                f["weights"], f["biases"] = x[k:k + nw], x[k + nw:k + nw + nb]
                k += nw + nb

    x0 = get(fields)

    def loss_grad(x):
        put(fields, x)
        loss, grads = og.eval_loss_grad(problem.operator, geom, fields, problem.extra, tracers=problem.tracers)[:2]
        return loss, grads

    x, losses = onp.adam_run(x0, loss_grad, epochs, 0.01)
    return x, losses, domain


@pytest.mark.parametrize("which,world,nx_rank", [("veltracer", 2, 8), ("veltracer", 3, 8), ("veltracer3d", 2, 8),
                                                  ("veltracer", 4, 2), ("veltracer", 2, 4), ("heat2d", 2, 8),
                                                  ("veltracer3d", 8, 4), ("veltracer-factors", 2, 8),
                                                  ("veltracer-factors", 4, 2)])
def test_slab_traced_ranks_equal_undivided_oracle(tmp_path, which, world, nx_rank):
    """nx_rank = 2, 4: the three multigrid levels (8 cells of t) leave 2, 1, 0.5 / 4, 2, 1 cells of x per rank: the
    coarsest levels are AGGLOMERATED (whole array on every rank, gradient shares summed by an all-reduce).  heat2d:
    a pointwise network inside the stencil -- its parameters are replicated, their gradients summed over the ranks.
    -factors: multigrid factors other than 1 (reference core.py:245-263): scaled terms in the synthesis, unscaled
    cotangents down the levels, every level's gradient scaled by its factor."""
    epochs = 3
    port = 29500 + (os.getpid() * 7 + world * 13 + nx_rank + len(which)) % 2000
    mp.spawn(worker, args=(world, which, epochs, port, str(tmp_path), nx_rank), nprocs=world, join=True)
    x_ref, losses_ref, domain = undivided(which, world, epochs, nx_rank)
    results = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=False) for r in range(world)]
    for r in range(world):
        assert np.max(np.abs(np.array(results[r]["losses"]) - np.array(losses_ref)) / np.array(losses_ref)) < 1e-12
    # redundant ghost updates: after `epochs` updates without any exchange of the unknowns, a rank's inner ghost planes
    # are BIT-identical to the neighbour's boundary planes (same gradient bits, same update on both sides)
    assert all(res["redundant"] for res in results)
    for r in range(world - 1):
        for lo_side, hi_side in zip(results[r + 1]["shared"], results[r]["shared"]):
            assert np.array_equal(lo_side["ghost_lo"], hi_side["own_hi"])
            assert np.array_equal(hi_side["ghost_hi"], lo_side["own_lo"])
    for i, ref in enumerate(x_ref):
        for r in range(world):
            got = results[r]["owned"][i]
            if got.shape == ref.shape:  # agglomerated levels and parameter arrays are whole on every rank
                want = ref
            else:
                n = ref.shape[1] // world
                want = ref[:, r * n:(r + 1) * n]
            assert got.shape == want.shape
            assert np.max(np.abs(got - want)) < 1e-12 * max(1.0, np.max(np.abs(want))), (i, r)
