"""The Adam that runs for the TRACED workloads pinned on the reference's optimizer.

Fixtures tests/golden/traj_adam_states_{heat,heat2d,veltracer,veltracer3d}_{f64,f32}.npz hold 20 epochs of the
reference's own AdamNativeOptimizer (reference src/odil/optimizer.py:311-336) on the reference's operators (heat.py,
veltracer.py; the generalised operators of this repo evaluated by the reference's core.py), every array of the state an
unknown (network weights included): its (x, m, v) at sampled epochs k and k + 1 and the loss of every epoch
(tests/golden/make_golden_traj.py:workload_adam_states).  One epoch of THIS package started from the reference's state at
k must give the reference's loss at k and its state at k + 1 -- through

  fused     the update inside the generated gather (`k_gat_*` / `adam_apply*` of stencil_codegen.py; forced for these small
            grids with ODIL_FUSE_ADAM_SMALL=1), coarser levels by `odil_adam_step`;
  separate  generated kernels for the gradient, one `odil_adam_step` over the packed vector;
  graph     ten epochs replayed as a hipGraph from the state at epoch 1 (free-running: compared at epoch 11);
  slab      two emulated ranks of odil_amd/slab_traced.py (update inside the slab gathers), ten epochs from epoch 1.

Tolerances.  f64: loss 1e-12, state 1e-11 (teacher-forced), 1e-9 after ten free epochs.  f32: the float kernels use
v_rcp_f32 / v_sqrt_f32 / v_exp_f32 forms (~1 ulp each) and FMA contraction, the reference run is torch float32: the
gradient agrees to the stated 2e-5; one update is lr * m / (sqrt(v) + eps) with |update| <= ~lr, so x is held to 4
float32 ulps of its own magnitude PLUS 1e-3 of the step size lr, m to 2e-5 and v to 4e-5 of their largest entry.
"""

import argparse
import os
import sys

import numpy as np
import pytest
import torch
from conftest import ROOT, load_golden

import odil_amd as odil

pytestmark = pytest.mark.gpu

for sub in ("heat", "velocity_from_tracer"):
    sys.path.insert(0, os.path.join(ROOT, "examples", sub))

WORKLOADS = ["heat", "heat2d", "veltracer", "veltracer3d"]
EPS32 = float(np.finfo(np.float32).eps)


def _args(g):
    return argparse.Namespace(**{k[5:]: g[k].item() for k in g.files if k.startswith("args/")})


def build(which, tag, kwreg=True):
    """(problem, state, number of arrays) of a workload on the inputs of its loss / gradient fixture.  kwreg=False: without
    heat's weight regulariser `(stop_gradient(w) - w) k` (reference heat.py:131-136) -- identically zero with zero
    gradient, an output in parameter space that the slab path refuses."""
    g = load_golden("{}_{}".format(which, tag))
    dtype = np.float64 if tag == "f64" else np.float32
    mod = odil.runtime.get_mod()
    odil.util.set_log_file(open(os.devnull, "w"))
    if which in ("heat", "heat2d"):
        ex = __import__(which)
        cshape = (int(g["Nt"]), int(g["Nx"])) + ((int(g["Ny"]),) if which == "heat2d" else ())
        domain = odil.Domain(cshape=cshape, dimnames=("t", "x", "y")[:len(cshape)], multigrid=True, dtype=dtype)
        nlvl = int(g["nlvl"])
        extra = argparse.Namespace(args=_args(g), init_u=mod.array(g["init_u"]), imp_mask=mod.array(g["imp_mask"]),
                                   imp_u=mod.array(g["imp_u"]), imp_size=int(g["imp_size"]))
        if not kwreg:
            extra.args.kwreg = 0
        state = odil.State()
        state.fields["u"] = np.zeros(domain.cshape)
        state.fields["k_net"] = odil.NeuralNet([g[f"x{nlvl + i}"] for i in range(3)], [g[f"x{nlvl + 3 + i}"] for i in range(3)])
        state = domain.init_state(state)
        problem = odil.Problem(ex.operator, domain, extra, tracers={"epoch": 1})
        return problem, state, nlvl + 6
    ex = __import__(which)
    if which == "veltracer":
        domain = odil.Domain(cshape=(int(g["Nt"]), int(g["Nx"]), int(g["Ny"])), dimnames=("t", "x", "y"), lower=(0, 0, 0),
                             upper=(1, 1, 1), dtype=dtype, multigrid=True, mg_interp="conv")
        keys, loc = ("u", "vx", "vy"), "ncc"
    else:
        n = int(g["Nx"])
        domain = odil.Domain(cshape=(int(g["Nt"]), n, n, n), dimnames=("t", "x", "y", "z"), lower=(0, 0, 0, 0),
                             upper=(1, 1, 1, 1), dtype=dtype, multigrid=True)
        keys, loc = ("u",) + ex.VEL, "nccc"
    extra = argparse.Namespace(args=_args(g), u_init=mod.array(g["u_init"]), u_final=mod.array(g["u_final"]))
    state = odil.State()
    for key in keys:
        state.fields[key] = odil.Field(None, loc=loc)
    state = domain.init_state(state)
    problem = odil.Problem(ex.operator, domain, extra, tracers={"epoch": 1})
    return problem, state, len(keys) * int(g["nlvl"])


def state_at(t, k, n):
    return [[t[f"{name}{i}_e{k}"] for i in range(n)] for name in ("x", "m", "v")]


def check_state(got, want, tag, lr, where):
    """got / want: (x, m, v) lists of arrays."""
    to = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    (x, m, v), (xr, mr, vr) = got, want
    if tag == "f64":
        tol = 1e-11 if where != "free" else 1e-9
        for part, ref in ((x, xr), (m, mr), (v, vr)):
            for i, (a, b) in enumerate(zip(part, ref)):
                assert float(np.max(np.abs(to(a) - b))) <= tol * max(1.0, float(np.max(np.abs(b)))), (where, i)
        return
    steps = 1 if where != "free" else 10
    for i, (a, b) in enumerate(zip(x, xr)):
        bound = 4 * EPS32 * max(1.0, float(np.max(np.abs(b)))) + steps * 1e-3 * lr
        assert float(np.max(np.abs(to(a) - b))) <= bound, (where, "x", i, float(np.max(np.abs(to(a) - b))), bound)
    for name, part, ref, tol in (("m", m, mr, 2e-5), ("v", v, vr, 4e-5)):
        scale = max(float(np.max(np.abs(b))) for b in ref)
        for i, (a, b) in enumerate(zip(part, ref)):
            err = float(np.max(np.abs(to(a) - b)))
            assert err <= steps * tol * scale, (where, name, i, err, scale)


def loss_grad_recording(problem, state, seen):
    inner = odil.util.make_loss_grad(problem, state)

    def loss_grad(arrays):
        loss, grads, pinfo = inner(arrays)
        seen.append(float(loss))
        return loss, grads, pinfo

    fused = getattr(inner, "fused_adam", None)
    if fused is not None:
        def fused_adam(*a):
            res = fused(*a)
            if res is not None:
                seen.append(float(res[0]))
            return res

        loss_grad.fused_adam = fused_adam
    return loss_grad


@pytest.mark.parametrize("mode", ["fused", "separate"])
@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("which", WORKLOADS)
def test_adam_epoch_from_the_reference_state(which, tag, mode, monkeypatch):
    monkeypatch.setenv("ODIL_GRAPH", "0")
    monkeypatch.setenv("ODIL_FUSE_ADAM_SMALL", "1" if mode == "fused" else "0")
    t = load_golden("traj_adam_states_{}_{}".format(which, tag))
    lr = float(t["lr"])
    problem, state, n = build(which, tag)
    mod, domain = odil.runtime.get_mod(), problem.domain
    fused_seen = False
    for k in [int(k) for k in t["sample"]]:
        x, m, v = [[mod.array(a) for a in part] for part in state_at(t, k, n)]
        domain.arrays_to_state(x, state)
        problem.tracers["epoch"] = k
        seen = []
        opt = odil.optimizer.AdamNativeOptimizer(dtype=domain.dtype, mod=mod)
        lg = loss_grad_recording(problem, state, seen)
        x1, info = opt.run(domain.arrays_from_state(state), lg, epochs=1, lr=lr, moments=(m, v), steps_done=k - 1)
        assert problem._traced is not None
        ref = float(t["losses"][k - 1])
        assert abs(seen[0] - ref) <= (1e-12 if tag == "f64" else 2e-5) * abs(ref), (k, seen[0], ref)
        check_state((x1, info.m, info.v), state_at(t, k + 1, n), tag, lr, "epoch {}".format(k))
        fused_seen = fused_seen or hasattr(lg, "fused_adam")
    if mode == "fused":
        # the update of the leading grid field really ran inside the generated gather
        res = problem._traced.eval_loss_grad_adam(state, info.m, info.v, 0.0, 0.1, 0.001, 1e-7)
        assert fused_seen and res is not None and res[-1] >= 1


@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("which", WORKLOADS)
def test_ten_epochs_replayed_as_a_graph(which, tag, monkeypatch):
    """Epochs 1 .. 10 from the reference's state at epoch 1 (= the fixture's start, zero moments) with the epoch replayed
    as a hipGraph from the third one on -- step size and the `epoch` tracer arrive through device memory -- against the
    reference's state at epoch 11 and its loss of every epoch."""
    monkeypatch.setenv("ODIL_GRAPH", "1")
    monkeypatch.setenv("ODIL_FUSE_ADAM_SMALL", "1")
    t = load_golden("traj_adam_states_{}_{}".format(which, tag))
    if 11 not in {int(k) + 1 for k in t["sample"]}:
        pytest.skip("fixture holds no state at epoch 11")
    lr = float(t["lr"])
    problem, state, n = build(which, tag)
    mod, domain = odil.runtime.get_mod(), problem.domain
    x, m, v = [[mod.array(a) for a in part] for part in state_at(t, 1, n)]
    domain.arrays_to_state(x, state)
    problem.tracers["epoch"] = 1
    seen = []
    opt = odil.optimizer.AdamNativeOptimizer(dtype=domain.dtype, mod=mod)

    def callback(arrays, epoch, pinfo):
        problem.tracers["epoch"] = epoch + 1  # what the next evaluation sees (the examples' callbacks set it)

    x1, info = opt.run(domain.arrays_from_state(state), loss_grad_recording(problem, state, seen), epochs=10, lr=lr,
                       callback=callback)
    check_state((x1, info.m, info.v), state_at(t, 11, n), tag, lr, "free")
    if seen:  # (replays do not pass through the recording wrapper)
        ref = float(t["losses"][0])
        assert abs(seen[0] - ref) <= (1e-12 if tag == "f64" else 2e-5) * abs(ref)


@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("which", WORKLOADS)
def test_two_emulated_slab_ranks_follow_the_reference(which, tag):
    """The slab-decomposed Adam loop (odil_amd/slab_traced.py, two ranks emulated on one GPU, axis 1, update inside the
    slab gathers) for ten epochs from the fixture's start: every epoch's loss and the owned parts of the state at epoch
    11 against the reference's undivided run."""
    from odil_amd.slab import run_lockstep
    from odil_amd.slab_traced import SlabTracedAdam

    t = load_golden("traj_adam_states_{}_{}".format(which, tag))
    if 11 not in {int(k) + 1 for k in t["sample"]}:
        pytest.skip("fixture holds no state at epoch 11")
    lr = float(t["lr"])
    problem, state, n = build(which, tag, kwreg=False)
    mod, domain = odil.runtime.get_mod(), problem.domain
    x = [mod.array(a) for a in state_at(t, 1, n)[0]]
    domain.arrays_to_state(x, state)
    world = 2
    ranks = [SlabTracedAdam(problem, state, r, world, axis=1, lr=lr) for r in range(world)]
    losses = []
    for k in range(1, 11):
        problem.tracers["epoch"] = k
        run_lockstep(ranks, 1)
        losses.append(sum(r.last_loss() for r in ranks))
    ref = np.asarray(t["losses"][:10])
    tol = 1e-11 if tag == "f64" else 2e-5
    assert np.max(np.abs(np.asarray(losses) - ref) / np.abs(ref)) <= tol, losses
    xr = state_at(t, 11, n)[0]
    for r, run in enumerate(ranks):
        for i, (got, want) in enumerate(zip(run.owned_arrays(), xr)):
            want = torch.as_tensor(want)
            if tuple(got.shape) != tuple(want.shape):
                per = want.shape[1] // world
                want = want[:, r * per:(r + 1) * per]
            bound = (1e-9 if tag == "f64" else 4 * EPS32) * max(1.0, float(want.abs().max())) + (0 if tag == "f64" else 1e-2 * lr)
            assert float((got.cpu() - want).abs().max()) <= bound, (r, i, float((got.cpu() - want).abs().max()), bound)


@pytest.mark.parametrize("path", ["traced", "generic"])
@pytest.mark.parametrize("which", ["wave", "heat_tmax", "infer_constant"])
def test_lbfgsb_on_the_example_workloads_follows_the_reference(which, path, monkeypatch):
    """The three reference examples whose default optimizer is L-BFGS-B (wave, heat_tmax, infer_constant): this package's
    on-device L-BFGS-B (`odil.util.optimize_grad(..., "lbfgsb")`, traced operator and autograd path) from the fixtures'
    random states against the reference's LbfgsbOptimizer -> SciPy run (fixtures traj_lbfgsb_pair_*: two reference runs one
    ulp apart agree to 1e-6 over all 40 iterations, so every iteration is held to 1e-6)."""
    sys.path.insert(0, os.path.join(ROOT, "examples", which))
    ex = __import__(which if which != "wave" else "wave")
    monkeypatch.setattr(odil.runtime, "enable_trace", path == "traced")
    g = load_golden(which + "_f64")
    t = load_golden("traj_lbfgsb_pair_{}_f64".format(which))
    a, b = t["iter_losses_a"], t["iter_losses_b"]
    n = min(len(a), len(b))
    assert n >= 30 and np.max(np.abs(a[:n] - b[:n]) / np.abs(a[:n])) < 1e-6  # the reference agrees with itself
    mod = odil.runtime.get_mod()
    odil.util.set_log_file(open(os.devnull, "w"))
    argv = ["--Nt", str(int(g["Nt"])), "--Nx", str(int(g["Nx"]))] + (["--kimp", str(float(g["kimp"]))] if "kimp" in g.files else [])
    args = ex.parse_args(argv)
    problem, state = ex.make_problem(args)
    for k in ("left_u", "right_u", "init_u", "init_ut", "u_init", "u_final"):
        if k in g.files:
            setattr(problem.extra, k, mod.array(g[k]))
    arrays = [mod.array(g[f"x{i}"]) for i in range(len(problem.domain.arrays_from_state(state)))]
    problem.domain.arrays_to_state(arrays, state)
    args.epoch_start, args.epochs = 0, int(t["epochs"])
    args.bfgs_m, args.bfgs_maxls = int(t["m"]), int(t["maxls"])
    losses = []
    try:
        odil.util.optimize_grad(args, "lbfgsb", problem, state, lambda s, e, p: losses.append(float(p["loss"])))
    except odil.EarlyStopError:
        pass
    assert (problem._traced is not None) == (path == "traced")
    got = np.array(losses[1:])  # (the callback also sees the initial evaluation)
    k = min(len(got), n)
    assert k >= 30, (len(got), n)
    assert np.max(np.abs(got[:k] - a[:k]) / np.abs(a[:k])) < 1e-6, np.abs(got[:k] - a[:k]) / np.abs(a[:k])
