"""VALUE-level parity of the headline workload at the size the metric is quoted on (BASELINE config 4a: 3-D Poisson
512^3, f64, 9 multigrid levels, Adam from the zero state) -- not properties, not a self-comparison:

`tests/golden/fullsize_poisson_N512.npz` holds three epochs of `oracle/poisson_epoch.c` (the plain-C restatement pinned to
the NumPy oracle, which is pinned on the reference's golden vectors): every loss, and per multigrid level the sum, the sum of
squares and 64 sampled entries of x, m, v after every epoch (`tests/golden/make_golden_fullsize.py`).  The HIP epoch must
reproduce them

  * through the bespoke driver `bench.py` times (`PoissonMultigridAdam`: fused residual / adjoint + P^T + Adam launches), and
  * through the PUBLIC API (`examples/poisson/poisson.py` operator -> `odil.util.optimize_grad(args, "adam", ...)`),

loss to 1e-12 relative at every epoch; the state to 1e-13 after epoch 1 and to 1e-7 (samples, of the level's largest) /
1e-9 (sums, of their Cauchy-Schwarz scale) after epochs 2 and 3 -- see the note at the tolerances.  Inputs: ref_u by the
generator's NumPy function on this host, rhs from it by the HIP residual kernel, both REQUIRED to equal the fixture's
sampled entries bit for bit.
Reference arithmetic: src/odil/core.py:245-263,606-700, examples/poisson/poisson.py:57-113, optimizer.py:311-319."""

import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("make_golden_fullsize", os.path.join(ROOT, "tests", "golden", "make_golden_fullsize.py"))
mk = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mk)

# Observed at 512^3 (r05): loss 3e-14 at all three epochs; epoch 1 state 2e-16; epochs 2-3: x 7e-10, m / v 6e-9 of the
# level's largest sample, sums 1e-10.  The state tolerances of epochs >= 2 are what the problem's conditioning leaves of
# two correct float64 implementations: x_1 = -lr g / (|g| + eps / sqrt(1 - b2)) has slope lr / 3e-6 where |g| is small,
# and g_2 sees x_1 through (2 / n) A^T A with |A| ~ 1 / h^2 = 2.6e5 -- last-bit differences of g_1 come back ~1e8 times
# larger in g_2 (the same run with inputs that differ in the last bit is off by 49 % in x after epoch 3).
LOSS_RTOL = 1e-12
SAMPLE_RTOL = {1: 1e-13, 2: 1e-7, 3: 1e-7}
SUM_RTOL = {1: 1e-13, 2: 1e-9, 3: 1e-9}


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def load_fixture(N):
    path = os.path.join(ROOT, "tests", "golden", "fullsize_poisson_N{}.npz".format(N))
    fx = dict(np.load(path))
    counts = fx["sample_count"]
    fx["idx"] = np.split(fx["sample_index"], np.cumsum(counts)[:-1])
    return fx


def device_inputs(N, fx, dev):
    """(ref_u, rhs) on the device, formed as the product forms them -- ref_u by the generator's NumPy function on this
    host, rhs = its discrete Laplacian by the HIP residual kernel -- and REQUIRED to be the fixture's inputs bit for bit
    on the sampled entries (the NumPy oracle made the fixture's rhs; the kernel reproduces its operation order)."""
    from odil_amd import ops

    ref_u = torch.as_tensor(mk.reference_u(N)).to(dev)
    h2 = [np.float64(1.0 / N) ** 2] * 3
    rhs, _ = ops.poisson_residual(ref_u, torch.zeros_like(ref_u), h2)
    i0 = torch.as_tensor(fx["idx"][0], device=dev)
    assert np.array_equal(ref_u.reshape(-1)[i0].cpu().numpy(), fx["ref_u_samples"]), "ref_u drifted (another libm?)"
    assert np.array_equal(rhs.reshape(-1)[i0].cpu().numpy(), fx["rhs_samples"]), "rhs is not the fixture's bit for bit"
    ssq = float((rhs * rhs).sum())
    assert abs(ssq - fx["rhs_stats"][0, 1]) <= 1e-13 * ssq
    return ref_u, rhs


def check_state(fx, epoch, name, arrs, report):
    """`arrs`: per-level device tensors of x, m, v or g after `epoch` epochs."""
    ref_stats = fx["{}_stats_e{}".format(name, epoch)]
    ref_samples = np.split(fx["{}_samples_e{}".format(name, epoch)], np.cumsum(fx["sample_count"])[:-1])
    for lvl, (a, idx, rs) in enumerate(zip(arrs, fx["idx"], ref_samples)):
        flat = a.reshape(-1)
        got = flat[torch.as_tensor(idx, device=flat.device)].cpu().numpy()
        scale = max(np.abs(rs).max(), 1e-300)
        err = np.abs(got - rs).max() / scale
        report.append((epoch, name, lvl, "samples", err, SAMPLE_RTOL[epoch]))
        n = flat.numel()
        ssum, ssq = float(flat.sum()), float((flat * flat).sum())
        cs = np.sqrt(max(ref_stats[lvl, 1], 1e-300) * n)  # |sum| <= sqrt(n * sum of squares)
        e1, e2 = abs(ssum - ref_stats[lvl, 0]) / cs, abs(ssq - ref_stats[lvl, 1]) / max(ref_stats[lvl, 1], 1e-300)
        report.append((epoch, name, lvl, "sum", e1, SUM_RTOL[epoch]))
        report.append((epoch, name, lvl, "sumsq", e2, 10 * SUM_RTOL[epoch]))


def summarize(report, tag):
    """Prints the worst deviation per (array, kind) and fails on every entry above its tolerance (all of them listed)."""
    worst = {}
    for epoch, name, lvl, kind, err, tol in report:
        key = (name, kind)
        worst[key] = max(worst.get(key, 0.0), err)
    print("\n[{}] worst relative deviations: ".format(tag) + ", ".join("{}.{} {:.1e}".format(k[0], k[1], v) for k, v in sorted(worst.items())))
    bad = [(e, n, l, k, "{:.2e} > {:.0e}".format(err, tol)) for e, n, l, k, err, tol in report if not err <= tol]
    assert not bad, bad


@pytest.mark.parametrize("N", [64, 512])
def test_bespoke_driver_reproduces_the_c_oracle(dev, N):
    from odil_amd.poisson_path import PoissonMultigridAdam

    fx = load_fixture(N)
    ref_u, rhs = device_inputs(N, fx, dev)
    run = PoissonMultigridAdam(3, N, dtype=torch.float64, device=dev, rhs=rhs, ref_u=ref_u)
    del rhs, ref_u
    report = []
    for epoch in range(1, int(fx["epochs"]) + 1):
        run.epoch()
        loss = run.last_loss()
        ref = float(fx["losses"][epoch - 1])
        report.append((epoch, "loss", 0, "value", abs(loss - ref) / abs(ref), LOSS_RTOL))
        for name, arrs in (("x", run.w), ("m", run.mw), ("v", run.vw)):
            check_state(fx, epoch, name, arrs, report)
    summarize(report, "bespoke N={}".format(N))
    del run
    torch.cuda.empty_cache()


@pytest.mark.parametrize("N", [64, 512])
def test_public_api_reproduces_the_c_oracle(dev, N):
    """The same three epochs through `import odil_amd as odil`: the example's operator callback, `Problem`,
    `optimize_grad(args, "adam", ...)`, one call per epoch resumed with the previous call's moments."""
    sys.path.insert(0, os.path.join(ROOT, "examples", "poisson"))
    import odil_amd as odil
    import poisson

    odil.util.set_log_file(open(os.devnull, "w"))
    fx = load_fixture(N)
    _, rhs = device_inputs(N, fx, dev)
    args = poisson.parse_args(["--ndim", "3", "--N", str(N)])
    problem, state = poisson.make_problem(args)
    # the example forms its reference solution on the device (torch pow: last-bit differences from NumPy's, which this
    # problem amplifies to O(1) within three epochs): replace its right-hand side by the fixture's bits BEFORE the first
    # evaluation (the operator is recognised then, and the evaluator takes rhs = -f(0) from the callback itself)
    assert not problem._fused_checked
    problem.extra.rhs = rhs
    del rhs
    report, losses = [], []
    moments = None
    for epoch in range(1, int(fx["epochs"]) + 1):
        args.epoch_start, args.epochs = 0, 1
        cb = lambda st, ep, pinfo: losses.append(float(np.array(pinfo["loss"])))  # noqa: E731
        arrays, info = odil.util.optimize_grad(args, "adam", problem, state, cb, moments=moments, steps_done=epoch - 1)
        moments = ([a.clone() for a in info.m], [a.clone() for a in info.v])
        ref = float(fx["losses"][epoch - 1])
        loss = losses[-1]  # the loss of the epoch's own evaluation (before its update), as the oracle returns it
        report.append((epoch, "loss", 0, "value", abs(loss - ref) / abs(ref), LOSS_RTOL))
        for name, arrs in (("x", arrays), ("m", moments[0]), ("v", moments[1])):
            check_state(fx, epoch, name, arrs, report)
    assert getattr(problem, "_fused", None) is not None, "the Poisson operator must have taken the fused HIP route"
    summarize(report, "public API N={}".format(N))
    del problem, state, arrays, moments
    torch.cuda.empty_cache()
