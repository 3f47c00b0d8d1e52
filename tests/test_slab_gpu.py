"""Slab decomposition on the GPU: several ranks emulated in ONE process on one device
(`slab.run_lockstep`), HIP kernels throughout, against the undivided-domain HIP path.
Checks exactly what the multi-GPU run does per rank: ghost-extended arrays, cut-aware P^T,
plane-range loss, packed plane exchanges."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,N", [(2, 16), (3, 8), (4, 32), (2, 128), (3, 128)])  # 128: the one-launch adjoint + P^T path
def test_emulated_ranks_equal_single_domain(world, N):
    from odil_amd import ops
    from odil_amd.fused import PoissonEvaluator
    from odil_amd.poisson_path import mg_cshapes
    from odil_amd.slab import SlabPoissonAdam, run_lockstep

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cglobal = (N * world, N, N)
    rhs = torch.randn(cglobal, dtype=torch.float64, device=dev)
    ranks = [SlabPoissonAdam(N, r, world, device=dev, rhs_global=rhs) for r in range(world)]
    epochs = 4
    losses = []
    for _ in range(epochs):
        run_lockstep(ranks, 1)
        losses.append(sum(r.last_loss() for r in ranks))

    # undivided domain with the single-GPU kernels
    shapes = mg_cshapes(cglobal)
    h2 = [(1.0 / N) ** 2] * 3
    ev = PoissonEvaluator(cglobal, shapes, rhs, h2, device=dev)
    sizes = [int(np.prod(s)) for s in shapes]
    x = torch.zeros(sum(sizes), dtype=torch.float64, device=dev)
    m, v = torch.zeros_like(x), torch.zeros_like(x)
    w = [t.view(s) for t, s in zip(x.split(sizes), shapes)]
    ref_losses = []
    lr, b1, b2 = np.float64(0.005), np.float64(0.9), np.float64(0.999)
    for epoch in range(1, epochs + 1):
        loss, _ = ev.loss_grad_arrays(w)
        ref_losses.append(float(loss))
        e = np.float64(epoch)
        ops.adam_step(x, m, v, ev.g, lr * np.sqrt(1 - b2**e) / (1 - b1**e), 1 - b1, 1 - b2, 1e-7)
    assert np.max(np.abs(np.array(losses) - np.array(ref_losses)) / np.array(ref_losses)) < 1e-12
    for lvl, ref in enumerate(w):
        nz = ref.shape[0] // world
        for r in range(world):
            got = ranks[r].owned_levels()[lvl]
            want = ref[r * nz : (r + 1) * nz]
            assert float((got - want).abs().max()) <= 1e-12 * max(1.0, float(want.abs().max())), (lvl, r)


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_two_ranks_over_torch_distributed_on_one_gpu(tmp_path, launcher):
    """bench.py at N = 2 both ways the driver may start it -- plain `python bench.py --gpus 2` (bench.py starts
    its ranks itself, as fresh child processes) and under torch.distributed.run -- with the gloo backend so that
    two ranks can share this box's single GPU: exercises init, the TorchDistComm exchanges with real HIP
    kernels, barrier / max-over-ranks timing and the JSON line."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ODIL_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(key, None)
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--N", "64", "--no_cpu_baseline"]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(29600 + os.getpid() % 300)] + tail
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["decomposition"] == "slab x2"
    assert d["config"]["rccl_ranks"] == 2 and d["config"]["comm_backend"] == "gloo"
    assert d["value"] > 0 and np.isfinite(d["loss_after"])
    # same problem on one rank pair emulated in-process
    from odil_amd.slab import SlabPoissonAdam, run_lockstep

    dev = torch.device("cuda:0")
    ranks = [SlabPoissonAdam(64, r, 2, device=dev) for r in range(2)]
    run_lockstep(ranks, 4)
    want = sum(r.last_loss() for r in ranks)
    assert abs(d["loss_after"] - want) <= 1e-10 * abs(want)


def test_bench_single_gpu_json_contract():
    """`python bench.py` at N = 1 on a small grid: ONE JSON line with every key of the driver's
    contract, the roofline object of the dominant kernel and the CPU baseline."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--N", "64",
                          "--cpu_N", "16", "--cpu_N_all", "16", "--cpu_budget", "1"], capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]:
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "grid-point-updates/s" and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 64**3 * 4 / (d["ms_per_step"] * 4e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and "traffic" in r and r["traffic_source"]
    assert abs(r["frac_model"] - r["achieved_model"] / r["peak"]) < 1e-12 and r["frac_model"] >= r["frac"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    assert c["all_cores"]["cores"] >= 1 and c["all_cores"]["value"] > 0 and c["all_cores"]["sample"]


# ---- traced operators (odil_amd/slab_traced.py): ranks emulated on one GPU == the undivided HIP path ------------
def _traced_problem(which, world, dtype_flag, nx_rank=16):
    import os
    import sys

    import odil_amd as odil

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for sub in ("heat", "velocity_from_tracer"):
        sys.path.insert(0, os.path.join(root, "examples", sub))
    for sub in ("wave", "infer_constant"):
        sys.path.insert(0, os.path.join(root, "examples", sub))
    scaled, decay = which.endswith("-factors"), which.endswith("-decay")
    which = which.split("-")[0]
    ex = __import__(which)
    odil.util.set_log_file(open(os.devnull, "w"))
    nx = nx_rank * world
    if which == "wave":  # plain Field unknown (no multigrid decomposition), walls masked on the GLOBAL x index
        argv = ["--Nt", "16", "--Nx", str(nx), "--multigrid", "0"]
    elif which == "infer_constant":  # an Array unknown inside the stencil: its gradient is summed over the ranks
        argv = ["--Nt", "16", "--Nx", str(nx)]
    elif which == "veltracer":
        argv = ["--Nt", "16", "--Nx", str(nx), "--Ny", "32"]
    elif which == "veltracer3d":
        argv = ["--Nt", "8", "--Nx", str(nx), "--Ny", "16", "--Nz", "24"]
    else:
        argv = ["--Nt", "16", "--Nx", str(nx), "--Ny", "32", "--infer_k", "1", "--imposed", "stripe"]
    problem, state = ex.make_problem(ex.parse_args(argv + ["--double", str(dtype_flag)]))
    if decay:
        # outputs in PARAMETER space on top of the example's: the reference's weight regulariser (identically zero) and a
        # weight decay whose value and gradient are not -- evaluated by the generated kernel of odil_amd/param_expr.py
        base = problem.operator

        def operator(ctx):
            m = ctx.mod
            ww = ctx.domain.arrays_from_field(ctx.state.fields["k_net"])
            flat = m.concatenate([m.flatten(w) for w in ww], axis=0)
            k = 0.3 * 0.5 ** (ctx.tracers["epoch"] / 4)
            return base(ctx) + [("wreg", (m.stop_gradient(flat) - flat) * k), ("wdecay", flat * k), ("bias0", (ww[-1] - 0.25) / (1 + k))]

        problem = odil.Problem(operator, problem.domain, problem.extra, tracers={"epoch": 2})
    if scaled:  # multigrid factors other than 1 (reference core.py:245-263)
        for f in state.fields.values():
            if isinstance(f, odil.MultigridField):
                f.factors = [1.0, 0.5, 2.0, 1.5, 0.25][: len(f.terms)]
    gen = torch.Generator(device="cpu").manual_seed(11)
    arrays = problem.domain.arrays_from_state(state)
    new = [(torch.randn(tuple(a.shape), generator=gen, dtype=torch.float64) * 0.1).to(device=a.device, dtype=a.dtype)
           for a in arrays]
    problem.domain.arrays_to_state(new, state)
    return problem, state


@pytest.mark.parametrize("which,world,double,nx_rank", [("veltracer", 2, 1, 16), ("veltracer", 3, 1, 16),
                                                        ("veltracer3d", 2, 1, 16), ("heat2d", 2, 1, 16),
                                                        ("heat2d-decay", 2, 1, 16),
                                                        ("veltracer", 4, 0, 16), ("veltracer", 4, 1, 8),
                                                        ("veltracer3d", 4, 1, 4), ("wave", 2, 1, 16),
                                                        ("infer_constant", 2, 1, 16),
                                                        ("veltracer-factors", 2, 1, 16), ("veltracer3d-factors", 4, 1, 4)])
def test_slab_traced_emulated_ranks_equal_single_gpu(which, world, double, nx_rank):
    """The slab-decomposed Adam loop of a traced operator -- generated kernels in slab mode (global indices,
    ghost-extended sources, periodic wrap planes, ghost-writing gathers), exchange-free P^T chain, deferred
    halo-add, summed parameter gradients of the pointwise network -- against the single-GPU traced path of the
    same problem: loss and every level array after 3 epochs.  nx_rank = 8, 4: the coarsest of the four / three
    levels leave a rank 1 (0.5) cells of x and are agglomerated (whole arrays, gradient shares all-reduced)."""
    import argparse

    import odil_amd as odil
    from odil_amd.slab import run_lockstep
    from odil_amd.slab_traced import SlabTracedAdam

    problem, state = _traced_problem(which, world, double, nx_rank)
    lr, epochs = 0.01, 3
    # heat2d's fields are cell-centred in t too; x is the axis the workloads shard
    ranks = [SlabTracedAdam(problem, state, r, world, axis=1, lr=lr) for r in range(world)]
    assert ranks[0].h == 1
    run_lockstep(ranks, epochs)
    run_lockstep(ranks, 1)  # evaluates the loss at the state after `epochs` updates
    got_loss = sum(r.last_loss() for r in ranks)
    a = argparse.Namespace(epoch_start=0, epochs=epochs, lr=lr, bfgs_m=None, bfgs_pgtol=None, bfgs_maxls=None,
                           adam_epsilon=None, adam_beta_1=None, adam_beta_2=None, callback_update_state=0)
    start = [t.clone() for t in problem.domain.arrays_from_state(state)]
    odil.util.optimize_grad(a, "adam", problem, state, None)
    assert problem._traced is not None
    want_loss = float(problem.eval_loss_grad(state)[0])
    tol = 1e-10 if double else 2e-4
    assert abs(got_loss - want_loss) <= tol * abs(want_loss), (got_loss, want_loss)
    # the ranks have made epochs + 1 updates: the undivided problem again from the start (a new optimizer run
    # starts with fresh moments)
    problem.domain.arrays_to_state(start, state)
    a.epochs = epochs + 1
    odil.util.optimize_grad(a, "adam", problem, state, None)
    want = problem.domain.arrays_from_state(state)
    for r, run in enumerate(ranks):
        for i, (got, ref) in enumerate(zip(run.owned_arrays(), want)):
            if got.shape != ref.shape:  # a grid array: this rank's planes of the sharded axis
                n = ref.shape[1] // world
                ref = ref[:, r * n:(r + 1) * n]
            scale = max(1.0, float(ref.abs().max()))
            assert float((got - ref).abs().max()) <= (1e-10 if double else 1e-4) * scale, (r, i)


def test_slab_ranks_dump_their_device_fields_into_one_raw_file(tmp_path):
    """Row F4: every rank of the slab-decomposed Poisson run writes the owned planes of its (device-resident)
    finest-level array into ONE raw + XDMF2 file (odil.write_raw_slab); reading the file back gives the field of
    the undivided run of the same problem, and the single-GPU dump of a device tensor round-trips."""
    import odil_amd as odil
    from odil_amd.poisson_path import PoissonMultigridAdam
    from odil_amd.slab import SlabPoissonAdam, run_lockstep

    dev = torch.device("cuda:0")
    N, world = 32, 2
    ranks = [SlabPoissonAdam(N, r, world, device=dev) for r in range(world)]
    run_lockstep(ranks, 3)
    path = str(tmp_path / "w0.xmf")
    for r in reversed(range(world)):
        odil.write_raw_slab(ranks[r].owned_levels()[0], path, r, world, axis=0, spacing=(1.0 / N,) * 3, name="w0")
    got, meta = odil.read_raw_with_xmf(path)
    assert got.shape == (world * N, N, N) and meta["name"] == "w0" and meta["cell"] and meta["precision"] == 8
    want = torch.cat([r.owned_levels()[0] for r in ranks]).cpu().numpy()
    assert np.array_equal(got, want)
    # a device tensor through the single-process writer
    single = str(tmp_path / "single.xmf")
    odil.write_raw_with_xmf(ranks[0].owned_levels()[0], single, spacing=(1.0 / N,) * 3)
    back, _ = odil.read_raw_with_xmf(single)
    assert np.array_equal(back, ranks[0].owned_levels()[0].cpu().numpy())


def test_bench_config5_two_ranks_over_torch_distributed(tmp_path):
    """BASELINE's slab-decomposed tracer workload through `python bench.py --config 5 --gpus 2` (ranks started by
    bench.py itself, gloo so that both can share this box's one GPU; grids scaled down): the JSON line of the
    driver's contract, and the loss of the two-rank run equals the one-rank run of the same global problem."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ODIL_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(key, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "5", "--gpus", "2", "--scale", "0.25",
                          "--steps", "3", "--warmup", "1", "--no_cpu_baseline"], env=env, capture_output=True, text=True,
                         timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["decomposition"] == "slab x2" and d["config"]["rccl_ranks"] == 2
    assert d["dtype"] == "f32" and d["unit"] == "grid-point-updates/s" and d["scaling"] == "weak"
    assert "velocity_from_tracer" in d["config"]["workload"] and d["config"]["fields"] == 4
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["kernel"].startswith("k_fwd")
    assert d["value"] > 0 and np.isfinite(d["loss_after"])
    cells = d["config"]["cells_per_gpu"]
    assert abs(d["value"] - 2 * cells * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]


@pytest.mark.parametrize("optimizer", ["adam", "lbfgsb"])
def test_example_runs_slab_decomposed_under_torch_distributed(tmp_path, optimizer):
    """`python -m torch.distributed.run --nproc-per-node 2 examples/velocity_from_tracer/veltracer3d.py --slab 1`
    (gloo, both ranks on this box's GPU): the user-level entry of the slab path -- rank 0 logs the all-reduced loss,
    both ranks write their planes of the final tracer field into one raw + XDMF2 file -- against the same example
    run undivided: same loss after the same epochs, same field."""
    import os
    import re
    import subprocess
    import sys

    import odil_amd as odil

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "examples", "velocity_from_tracer", "veltracer3d.py")
    common = ["--Nt", "8", "--Nx", "32", "--Ny", "16", "--Nz", "16", "--double", "1", "--epochs", "6", "--report_every", "3",
              "--plot_every", "100", "--history_every", "0", "--checkpoint_every", "0", "--frames", "1",
              "--optimizer", optimizer]
    env = dict(os.environ, ODIL_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(key, None)
    slab_dir, one_dir = str(tmp_path / "slab"), str(tmp_path / "one")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + os.getpid() % 200), script, "--slab", "1", "--outdir", slab_dir] + common
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    log = open(os.path.join(slab_dir, "train.log")).read()
    losses = [float(v) for v in re.findall(r"ranks=2 loss=([0-9.eE+-]+)", log)]
    assert len(losses) == 2
    # the undivided run through the same entry (slab path with one rank: the periodic closure is local)
    out = subprocess.run([sys.executable, script, "--slab", "1", "--outdir", one_dir] + common, env=env, capture_output=True,
                         text=True, timeout=900, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    ref = [float(v) for v in re.findall(r"ranks=1 loss=([0-9.eE+-]+)", open(os.path.join(one_dir, "train.log")).read())]
    assert len(ref) == 2 and all(abs(a - b) <= 1e-7 * abs(b) for a, b in zip(losses, ref)), (losses, ref)
    got, meta = odil.read_raw_with_xmf(os.path.join(slab_dir, "u_final.xmf"))
    want, _ = odil.read_raw_with_xmf(os.path.join(one_dir, "u_final.xmf"))
    assert got.shape == (32, 16, 16) and meta["name"] == "u"
    assert np.max(np.abs(got - want)) <= 1e-10 * max(1.0, np.max(np.abs(want)))
