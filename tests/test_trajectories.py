"""Optimizer trajectories at the lengths the configs really run (north_star: "loss trajectory matching reference to
1e-6 rel"), against fixtures the REFERENCE's own optimizers produced (tests/golden/make_golden_traj.py):
Adam 400 epochs 1-D N=256 (reference examples/poisson/poisson.py:142), 300 epochs 2-D N=32, 100 epochs 3-D N=16;
gradient descent 60 epochs; L-BFGS-B 60 iterations.  These iterations amplify rounding-level differences
exponentially, so every fixture holds TWO reference runs one ulp apart (right-hand side for Adam, start for
L-BFGS-B): the epoch up to which the reference agrees with ITSELF to 1e-6 is the horizon an implementation can
be held to at 1e-6 (asserted, so a change of the fixture shows: Adam 1-D 24 epochs, 2-D 124, 3-D all 100;
L-BFGS-B 18 iterations); beyond it the implementation must stay within the envelope of the reference pair.

The epochs beyond those horizons are pinned by TEACHER FORCING (fixtures traj_adam_states_*, traj_lbfgsb_*_iterates:
the reference optimizer's own state at sampled epochs): one epoch of the implementation started from the
reference's state at epoch k is compared with the reference's state at k + 1 at round-off level -- no trajectory
along which a difference could be amplified, so the bound is as strict at epoch 399 as at epoch 1.

CPU tests hold the NumPy oracle to the fixtures; `-m gpu` tests hold the HIP path (fused kernels and the generic
operator path, through the public API and the C-ABI) to them."""

import os
import sys

import numpy as np
import pytest
import torch
from conftest import ROOT, load_golden

from oracle import odil_np as onp

ADAM = ["traj_adam_1d_N256", "traj_adam_2d_N32", "traj_adam_3d_N16"]


def lbfgsb_horizon(g, tol=1e-6):
    """First iteration at which the two reference runs (starts one ulp apart) differ by more than tol."""
    a, b = g["iter_losses_a"], g["iter_losses_b"]
    n = min(len(a), len(b))
    bad = np.nonzero(np.abs(a[:n] - b[:n]) / np.abs(a[:n]) > tol)[0]
    return int(bad[0]) if len(bad) else n


def adam_horizon(g, tol=1e-6):
    rel = np.abs(g["losses_b"] - g["losses"]) / g["losses"]
    bad = np.nonzero(rel > tol)[0]
    return (int(bad[0]) if len(bad) else len(rel)), np.maximum.accumulate(rel)


def check_adam_trajectory(got, g):
    """|loss - reference| / reference <= max(1e-6, 100 x spread of the reference pair so far) at every epoch.
    Returns the number of leading epochs held to 1e-6 exactly."""
    ref = g["losses"]
    got = np.asarray(got)
    assert got.shape == ref.shape
    h, env = adam_horizon(g)
    rel = np.abs(got - ref) / ref
    tol = np.maximum(1e-6, 100 * env)
    assert np.all(rel <= tol), [(int(k), float(rel[k]), float(tol[k])) for k in np.nonzero(rel > tol)[0][:3]]
    strict = np.nonzero(100 * env > 1e-6)[0]
    return int(strict[0]) if len(strict) else len(ref)


def test_reference_adam_agrees_with_itself_up_to_a_horizon():
    assert [adam_horizon(load_golden(n))[0] for n in ADAM] == [24, 124, 100]


def test_reference_lbfgsb_agrees_with_itself_for_18_iterations():
    """The reference's L-BFGS-B trajectory on the ill-conditioned multigrid Poisson problem is sensitive to the
    last bit of its start: two reference runs one ulp apart part ways (1e-6) after 18 iterations and differ by
    percent a few iterations later."""
    g = load_golden("traj_lbfgsb_2d_N32_pair")
    assert lbfgsb_horizon(g) == 18
    a, b = g["iter_losses_a"], g["iter_losses_b"]
    assert np.max(np.abs(a[25:50] - b[25:50]) / a[25:50]) > 1e-2


@pytest.mark.parametrize("name", ADAM)
def test_oracle_adam_full_length(name):
    g = load_golden(name)
    rhs = g["rhs"]
    cshape = rhs.shape
    dw = onp.step(cshape)
    x0 = [np.zeros(cs) for cs in onp.mg_cshapes(cshape)]
    x, losses = onp.adam_run(x0, lambda x: onp.poisson_loss_grad(x, rhs, dw)[:2], int(g["epochs"]), float(g["lr"]))
    assert len(losses) == int(g["epochs"])
    if check_adam_trajectory(losses, g) == len(losses):  # the whole run is inside the horizon: the final state too
        for i, a in enumerate(x):
            assert np.max(np.abs(a - g[f"w{i}"])) < 1e-7 * max(1.0, np.max(np.abs(g[f"w{i}"])))


STATES = ["traj_adam_states_1d_N256", "traj_adam_states_2d_N32", "traj_adam_states_3d_N16"]


def _state_at(g, k, nlvl):
    return [[g[f"{name}{i}_e{k}"] for i in range(nlvl)] for name in ("x", "m", "v")]


def _state_err(got, want):
    """max over arrays of |got - want| / max(|want|): per array, so that the small coarse levels count."""
    return max(float(np.max(np.abs(np.asarray(a) - b))) / max(float(np.max(np.abs(b))), 1e-300) for a, b in zip(got, want))


@pytest.mark.parametrize("name", STATES)
def test_oracle_adam_teacher_forced(name):
    """One oracle epoch from the reference's (x, m, v) at every sampled epoch k against the reference's loss at k
    and state at k + 1 (reference optimizer.py:311-319, 331-336)."""
    g = load_golden(name)
    rhs = g["rhs"]
    dw = onp.step(rhs.shape)
    nlvl = len(onp.mg_cshapes(rhs.shape))
    for k in [int(k) for k in g["sample"]]:
        x, m, v = _state_at(g, k, nlvl)
        loss, grads = onp.poisson_loss_grad(x, rhs, dw)[:2]
        x1, m1, v1 = onp.adam_step(x, m, v, grads, k, float(g["lr"]))
        xr, mr, vr = _state_at(g, k + 1, nlvl)
        assert abs(loss - g["losses"][k - 1]) <= 1e-13 * g["losses"][k - 1], k
        assert _state_err(x1, xr) < 1e-12 and _state_err(m1, mr) < 1e-11 and _state_err(v1, vr) < 1e-11, k


def test_oracle_gd_and_lbfgsb():
    g = load_golden("traj_gd_2d_N16")
    rhs = g["rhs"]
    dw = onp.step(rhs.shape)
    x0 = [np.zeros(cs) for cs in onp.mg_cshapes(rhs.shape)]
    x, losses = onp.gd_run(x0, lambda x: onp.poisson_loss_grad(x, rhs, dw)[:2], int(g["epochs"]), float(g["lr"]))
    assert np.max(np.abs(np.array(losses) - g["losses"]) / g["losses"]) < 1e-12
    import scipy

    g = load_golden("traj_lbfgsb_2d_N32_pair")
    rhs = g["rhs"]
    dw = onp.step(rhs.shape)
    nlvl = len(onp.mg_cshapes(rhs.shape))
    x0 = [g[f"start{i}"] for i in range(nlvl)]
    x, losses, iters, info = onp.lbfgsb_run(x0, lambda x: onp.poisson_loss_grad(x, rhs, dw)[:2], int(g["epochs"]))
    if str(g["scipy_version"]) == scipy.__version__:
        h = lbfgsb_horizon(g)
        ref = g["iter_losses_a"]
        assert np.max(np.abs(np.array(iters[:h]) - ref[:h]) / ref[:h]) < 1e-6


# ------------------------------------------------------------------------------------------------- GPU
def _api(ndim, N, **kw):
    sys.path.insert(0, os.path.join(ROOT, "examples", "poisson"))
    import poisson

    import odil_amd as odil

    odil.util.set_log_file(open(os.devnull, "w"))
    args = poisson.parse_args([])
    args.ndim, args.N, args.multigrid, args.epoch_start = ndim, N, 1, 0
    for k, v in kw.items():
        setattr(args, k, v)
    return odil, poisson, args


def _rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return float(np.max(np.abs(a - b))) / max(1.0, float(np.max(np.abs(b))))


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("name", ADAM)
def test_hip_adam_full_length(name, fuse, monkeypatch):
    """`odil.optimize` with Adam on the Poisson example, every epoch of the reference's trajectory to 1e-6: the
    recognised-operator kernels (fuse: Adam inside the gradient launches, eager and replayed as a hipGraph) and
    the generic operator path."""
    g = load_golden(name)
    ref = g["losses"]
    odil, poisson, args = _api(int(g["ndim"]), int(g["N"]), epochs=len(ref), lr=float(g["lr"]))
    monkeypatch.setattr(odil.runtime, "enable_fuse", fuse)
    problem, state = poisson.make_problem(args)
    problem.extra.rhs = odil.runtime.get_mod().array(g["rhs"])
    losses = []
    odil.util.optimize_grad(args, "adam", problem, state, lambda state, epoch, pinfo: losses.append(float(pinfo["loss"])))
    got = np.array(losses[1:])  # the callback also sees the initial evaluation (epoch 0)
    if check_adam_trajectory(got, g) == len(ref):
        for i, a in enumerate(problem.domain.arrays_from_state(state)):
            assert _rel(a, g[f"w{i}"]) < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("name", STATES)
def test_hip_adam_teacher_forced(name, fuse, monkeypatch):
    """EVERY sampled epoch of the reference's Adam runs (1-D N=256: 40 epochs up to 390 of 400; 2-D; 3-D) pinned at
    round-off level: `AdamNativeOptimizer.run(epochs=1)` of this package resumed from the reference's own
    (x, m, v) at epoch k -- through the recognised-operator kernels with the update inside the gradient launches
    (fuse) and through the generic operator path + odil_adam_step -- against the reference's loss at k and its
    state at k + 1 (reference optimizer.py:311-336)."""
    g = load_golden(name)
    odil, poisson, args = _api(int(g["ndim"]), int(g["N"]), epochs=1, lr=float(g["lr"]))
    monkeypatch.setattr(odil.runtime, "enable_fuse", fuse)
    monkeypatch.setenv("ODIL_GRAPH", "0")
    problem, state = poisson.make_problem(args)
    mod = odil.runtime.get_mod()
    problem.extra.rhs = mod.array(g["rhs"])
    domain = problem.domain
    nlvl = len(domain.arrays_from_state(state))
    worst = 0.0
    for k in [int(k) for k in g["sample"]]:
        x, m, v = [[mod.array(a) for a in part] for part in _state_at(g, k, nlvl)]
        domain.arrays_to_state(x, state)
        seen = []
        opt = odil.optimizer.AdamNativeOptimizer(dtype=domain.dtype, mod=mod)
        x1, info = opt.run(domain.arrays_from_state(state), _loss_grad_of(odil, problem, state, seen), epochs=1,
                           lr=float(g["lr"]), moments=(m, v), steps_done=k - 1)
        xr, mr, vr = _state_at(g, k + 1, nlvl)
        ref = g["losses"][k - 1]
        assert abs(seen[0] - ref) <= 1e-12 * ref, (k, seen[0], ref)
        to = lambda arrs: [a.detach().cpu().numpy() for a in arrs]
        ex, em, ev = _state_err(to(x1), xr), _state_err(to(info.m), mr), _state_err(to(info.v), vr)
        assert ex < 1e-11 and em < 1e-11 and ev < 1e-11, (k, ex, em, ev)
        worst = max(worst, ex)
    print(name, "fuse" if fuse else "generic", "worst state error over", len(g["sample"]), "epochs:", worst)


def _loss_grad_of(odil, problem, state, seen):
    """The loss_grad callable `util.optimize_grad` hands to the optimizers (with its `fused_adam` hook), recording
    the loss of every evaluation."""
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


@pytest.mark.gpu
def test_hip_lbfgsb_teacher_forced_all_60_iterations():
    """The on-device L-BFGS-B (HIP vector algebra, fused Poisson loss + gradient) reproduces EACH of the reference's
    60 iterations from the reference's own iterates: step x_{k+1} - x_k to 1e-9 of its size (see
    tests/test_lbfgs_host_logic.py for the host-logic twin, measured 6e-14 there)."""
    from test_lbfgs_host_logic import teacher_forced_lbfgsb

    g = load_golden("traj_lbfgsb_2d_N32_iterates")
    xs = g["x"]
    odil, poisson, args = _api(2, 32, epochs=1)
    problem, state = poisson.make_problem(args)
    mod = odil.runtime.get_mod()
    problem.extra.rhs = mod.array(g["rhs"])
    domain = problem.domain
    arrays = domain.arrays_from_state(state)
    sizes = [int(a.numel()) for a in arrays]
    inner = odil.util.make_loss_grad(problem, state)
    dev = arrays[0].device

    def fun(x):
        parts = torch.as_tensor(np.asarray(x), device=dev).split(sizes)
        loss, grads, _ = inner([p.view(a.shape) for p, a in zip(parts, arrays)])
        return float(loss), torch.cat([gr.reshape(-1) for gr in grads]).cpu().numpy()

    def factory():
        vec = odil.optimizer.LbfgsVectors(xs.shape[1], 50, dev)
        return vec, (lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64, device=dev).clone()), (
            lambda t: t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t))

    res = teacher_forced_lbfgsb(fun, xs, factory, range(0, len(xs) - 1))
    worst = max(e for _, e in res)
    assert len(res) == 60 and worst < 1e-9, sorted(res, key=lambda r: -r[1])[:5]
    print("L-BFGS-B teacher-forced, worst step error:", worst)


@pytest.mark.gpu
def test_hip_gd_trajectory():
    """GdOptimizer.run (reference optimizer.py:262-277: x -= lr g through odil_axpy) against the reference's."""
    g = load_golden("traj_gd_2d_N16")
    odil, poisson, args = _api(2, 16, epochs=int(g["epochs"]), lr=float(g["lr"]))
    problem, state = poisson.make_problem(args)
    problem.extra.rhs = odil.runtime.get_mod().array(g["rhs"])
    losses = []
    odil.util.optimize_grad(args, "gd", problem, state, lambda state, epoch, pinfo: losses.append(float(pinfo["loss"])))
    got = np.array(losses[1:])
    assert got.shape == g["losses"].shape
    assert np.max(np.abs(got - g["losses"]) / g["losses"]) < 1e-10
    for i, a in enumerate(problem.domain.arrays_from_state(state)):
        assert _rel(a, g[f"w{i}"]) < 1e-10


@pytest.mark.gpu
def test_hip_lbfgsb_to_the_reference_horizon():
    """The on-device L-BFGS-B against the reference's (SciPy) run, from the same random start, for as long as the
    reference agrees with its own one-ulp twin."""
    g = load_golden("traj_lbfgsb_2d_N32_pair")
    h = lbfgsb_horizon(g)
    assert h >= 15
    odil, poisson, args = _api(2, 32, epochs=int(g["epochs"]))
    problem, state = poisson.make_problem(args)
    mod = odil.runtime.get_mod()
    problem.extra.rhs = mod.array(g["rhs"])
    domain = problem.domain
    n = len(domain.arrays_from_state(state))
    domain.arrays_to_state([mod.array(g[f"start{i}"]) for i in range(n)], state)
    losses = []
    try:
        odil.util.optimize_grad(args, "lbfgsb", problem, state, lambda state, epoch, pinfo: losses.append(float(pinfo["loss"])))
    except odil.EarlyStopError:
        pass
    got, ref = np.array(losses[1:]), g["iter_losses_a"]
    assert len(got) >= h
    assert np.max(np.abs(got[:h] - ref[:h]) / ref[:h]) < 1e-6
    # beyond the horizon: no closer than the reference is to itself is asked, but the run must keep descending
    k = min(len(got), len(ref))
    assert got[k - 1] < 5 * ref[k - 1]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("fine", [(16, 32, 64), (8, 8, 16), (12, 20, 36)])
def test_one_launch_adjoint_transpose_adam_vs_oracle(dtype, fine):
    """odil_poisson_adjoint_transpose_adam DIRECTLY against the NumPy oracle at sizes the oracle does in
    milliseconds (the larger-size tests hold it bit-for-bit to the separate kernels): g0 = scale A^T fu,
    g1 = P^T g0, Adam of both levels (reference optimizer.py:311-319)."""
    from odil_amd import ops

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(61)
    coarse = tuple(n // 2 for n in fine)
    dw = [0.25, 0.1, 0.3]
    h2 = [dtype(d) ** 2 for d in dw]
    scale = 2.0 / np.prod(fine)
    fu = rng.standard_normal(fine).astype(dtype)
    x = [rng.standard_normal(s).astype(dtype) for s in (fine, coarse)]
    m = [rng.standard_normal(s).astype(dtype) * 0.1 for s in (fine, coarse)]
    v = [np.abs(rng.standard_normal(s)).astype(dtype) for s in (fine, coarse)]
    to = lambda a: torch.tensor(a, device=dev)
    tx, tm, tv = [to(a) for a in x], [to(a) for a in m], [to(a) for a in v]
    g0 = torch.empty(fine, dtype=tx[0].dtype, device=dev)
    g1 = torch.empty(coarse, dtype=tx[0].dtype, device=dev)
    alpha, omb1, omb2, eps = 0.01, 0.1, 0.001, 1e-7
    ops.poisson_adjoint_transpose(to(fu), h2, scale, g1, g0=g0, adam0=(tx[0], tm[0], tv[0]), adam1=(tx[1], tm[1], tv[1]),
                                  alpha=alpha, one_minus_b1=omb1, one_minus_b2=omb2, eps=eps)
    f64 = lambda a: np.asarray(a, dtype=np.float64)
    g0_ref = onp.poisson_adjoint(f64(fu) * scale, [np.float64(np.sqrt(f64(h))) for h in h2])
    g1_ref = onp.interp_to_finer_adj(g0_ref, "ccc", coarse)
    tol = 1e-13 if dtype == np.float64 else 2e-6
    assert _rel(g0, g0_ref) < tol and _rel(g1, g1_ref) < tol
    for lvl, gr in enumerate((g0_ref, g1_ref)):
        mr = f64(m[lvl]) + (gr - f64(m[lvl])) * omb1
        vr = f64(v[lvl]) + (gr * gr - f64(v[lvl])) * omb2
        xr = f64(x[lvl]) - mr * alpha / (np.sqrt(vr) + eps)
        assert _rel(tm[lvl], mr) < tol and _rel(tv[lvl], vr) < tol and _rel(tx[lvl], xr) < 10 * tol


@pytest.mark.gpu
def test_lbfgsb_with_the_evaluation_replayed_as_a_graph_follows_the_eager_run(monkeypatch):
    """Launch-bound problems replay the loss + gradient evaluation of L-BFGS-B as a hipGraph (optimizer.LbfgsbOptimizer):
    the same kernels on the same buffers, so every iterate, every loss and the evaluation count are those of the eager run."""
    import torch

    import odil_amd as odil

    sys.path.insert(0, os.path.join(ROOT, "examples", "poisson"))
    import poisson

    odil.util.set_log_file(open(os.devnull, "w"))

    def run(graph):
        monkeypatch.setenv("ODIL_GRAPH", graph)
        args = poisson.parse_args(["--ndim", "2", "--N", "64"])
        problem, state = poisson.make_problem(args)
        args.epoch_start, args.epochs = 0, 40
        losses = []
        try:
            _, info = odil.util.optimize(args, "lbfgsb", problem, state, lambda s, e, p: losses.append(float(np.array(p["loss"]))))
        except odil.EarlyStopError as e:
            info = e.optinfo
        return losses, info.evals, [a.clone() for a in problem.domain.arrays_from_state(state)]

    l1, e1, x1 = run("1")
    l0, e0, x0 = run("0")
    assert len(l1) == len(l0) >= 30 and e1 == e0
    assert l1 == l0
    for a, b in zip(x1, x0):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("double", [1, 0])
@pytest.mark.parametrize("ndim,N,epochs", [(1, 256, 60), (1, 4096, 12), (1, 8, 8), (2, 32, 40), (2, 256, 6), (2, 4, 8)])
def test_whole_epochs_in_one_launch_are_bit_identical(ndim, N, epochs, double, monkeypatch):
    """odil_poisson_small_epochs (ONE workgroup walks synthesis, residual + loss, adjoint, transposes and every level's Adam
    update; state in LDS when it fits) against the same epochs as separate kernels: every loss and the final x, m, v BIT
    for bit -- one launch per epoch (a callback that must see every epoch) and all epochs in ONE launch (no callback) --
    1-D and 2-D, one and several virtual workgroups of the loss reduction (4096 cells; 256 rows), LDS-resident and
    global-memory states, both precisions."""
    from odil_amd import fused

    monkeypatch.setenv("ODIL_GRAPH", "0")
    runs = {}
    for mode in ("separate", "per-epoch", "one-launch"):
        odil, poisson, args = _api(ndim, N, epochs=epochs, lr=0.005, double=double)
        monkeypatch.setattr(fused.PoissonEvaluator, "small_max_cells", 0 if mode == "separate" else 4096)
        monkeypatch.setattr(fused.PoissonEvaluator, "small_force", mode != "separate")
        problem, state = poisson.make_problem(args)
        losses = []
        cb = None if mode == "one-launch" else (lambda st, ep, pinfo: losses.append(float(np.array(pinfo["loss"]))))
        arrays, info = odil.util.optimize_grad(args, "adam", problem, state, cb)
        assert getattr(problem, "_fused", None) is not None
        used = problem._fused.__dict__.get("_small_u") is not None
        assert used == (mode != "separate"), (mode, used)
        runs[mode] = (losses, [a.clone() for a in arrays], [a.clone() for a in info.m], [a.clone() for a in info.v])
    ref = runs["separate"]
    assert len(ref[0]) == epochs + 1 and runs["per-epoch"][0] == ref[0], "losses differ"
    for mode in ("per-epoch", "one-launch"):
        for part, name in zip(runs[mode][1:], "xmv"):
            for lvl, (a, b) in enumerate(zip(part, ref[1:]["xmv".index(name)])):
                assert torch.equal(a, b), (mode, name, lvl, float((a - b).abs().max()))
