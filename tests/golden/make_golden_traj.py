#!/usr/bin/env python3
"""Long optimizer trajectories FROM THE REFERENCE (build container only, needs /root/reference):

    python3 tests/golden/make_golden_traj.py

* Adam (reference src/odil/optimizer.py:286-341, AdamNativeOptimizer) on the Poisson example at the lengths
  the configs really run: 400 epochs 1-D N=256 (examples/poisson/poisson.py:142), 300 epochs 2-D N=32,
  100 epochs 3-D N=16 -- loss of every evaluation + final state;
* gradient descent (optimizer.py:262-277, GdOptimizer), 2-D N=16, 60 epochs;
* L-BFGS-B (optimizer.py:54-117 -> SciPy) 60 iterations, 2-D N=32, from a random start -- TWICE, the second
  run from a start moved by one ulp, so that the fixture carries the reference's OWN sensitivity: the
  iteration up to which two reference runs agree to 1e-6 is the horizon any implementation can be held to.
The reference's code runs unchanged on the torch-CPU shim (ref_shim.py), as in make_golden.py.
"""
import numpy as np
import torch

from make_golden import T, make_poisson, mod, npy, odil, poisson, ref_loss_grad, save


def adam_traj(ndim, N, epochs):
    """Two reference runs: the config itself, and the same with every entry of the right-hand side moved by one ulp up or down (`losses_b`)
    -- the reference's own sensitivity to rounding-level differences over these many epochs."""
    data = dict(ndim=np.array(ndim), N=np.array(N), lr=np.array(0.005), epochs=np.array(epochs))
    for tag in ("", "_b"):
        domain, state, extra = make_poisson(ndim, N)
        if tag:  # every entry one ulp up or down at random: what a different summation order does to a result
            r = npy(extra.rhs)
            sign = np.random.default_rng(99).integers(0, 2, r.shape) * 2.0 - 1.0
            extra.rhs = T(np.nextafter(r, sign * np.inf))
        else:
            data["rhs"] = npy(extra.rhs)
        losses = []

        def loss_grad(arrays):
            loss, grads, terms, _ = ref_loss_grad(domain, state, extra, poisson.operator, arrays)
            losses.append(float(loss))
            return loss, grads, None

        opt = odil.optimizer.AdamNativeOptimizer(dtype=np.float64, mod=mod)
        x, _ = opt.run(domain.arrays_from_state(state), loss_grad, epochs=epochs, lr=0.005, jit=False)
        data["losses" + tag] = np.array(losses)
        if not tag:
            for i, a in enumerate(x):
                data[f"w{i}"] = npy(a)
    rel = np.abs(data["losses"] - data["losses_b"]) / data["losses"]
    bad = np.nonzero(rel > 1e-6)[0]
    print("adam {}d N={}: reference vs reference (rhs one ulp apart): max rel {:.1e}, first epoch beyond 1e-6: {}".format(
        ndim, N, rel.max(), bad[0] if len(bad) else None))
    save(f"traj_adam_{ndim}d_N{N}", **data)


def gd_traj():
    domain, state, extra = make_poisson(2, 16)
    losses = []

    def loss_grad(arrays):
        loss, grads, terms, _ = ref_loss_grad(domain, state, extra, poisson.operator, arrays)
        losses.append(float(loss))
        return loss, grads, None

    lr, epochs = 1e-6, 60
    opt = odil.optimizer.GdOptimizer(dtype=np.float64, mod=mod)
    x, _ = opt.run(domain.arrays_from_state(state), loss_grad, epochs=epochs, lr=lr)
    data = dict(lr=np.array(lr), epochs=np.array(epochs), rhs=npy(extra.rhs), losses=np.array(losses))
    for i, a in enumerate(x):
        data[f"w{i}"] = npy(a)
    save("traj_gd_2d_N16", **data)


def lbfgsb_pair():
    import scipy

    rng = np.random.default_rng(2024)
    domain, state, extra = make_poisson(2, 32)
    start = [rng.standard_normal(tuple(a.shape)) * 0.01 for a in domain.arrays_from_state(state)]
    epochs = 60
    data = dict(rhs=npy(extra.rhs), epochs=np.array(epochs), m=np.array(50), maxls=np.array(50),
                scipy_version=np.array(scipy.__version__))
    for i, a in enumerate(start):
        data[f"start{i}"] = a
    for tag, x0 in (("a", start), ("b", [np.nextafter(a, np.inf) for a in start])):
        domain, state, extra = make_poisson(2, 32)
        evals, iters = [], []

        def loss_grad(arrays):
            loss, grads, terms, _ = ref_loss_grad(domain, state, extra, poisson.operator, arrays)
            evals.append(float(loss))
            return loss, grads, None

        def callback(arrays, epoch, pinfo):
            iters.append(evals[-1])

        opt = odil.optimizer.LbfgsbOptimizer(dtype=np.float64, mod=mod, m=50, maxls=50)
        try:
            opt.run([T(a) for a in x0], loss_grad, epochs=epochs, callback=callback)
        except odil.EarlyStopError as e:
            print("early stop", e)
        data[f"iter_losses_{tag}"] = np.array(iters)
        data[f"eval_losses_{tag}"] = np.array(evals)
    a, b = data["iter_losses_a"], data["iter_losses_b"]
    n = min(len(a), len(b))
    rel = np.abs(a[:n] - b[:n]) / np.abs(a[:n])
    print("reference vs reference (1 ulp apart): rel. difference per iteration\n", np.array2string(rel, precision=1))
    save("traj_lbfgsb_2d_N32_pair", **data)


def adam_states(ndim, N, epochs, sample):
    """Teacher-forcing fixture: the reference optimizer's OWN state (x, m, v) at the start of the sampled epochs k
    and k + 1 and the loss of every epoch, so that ONE epoch of an implementation started from the reference's
    state at k can be compared with the reference's state at k + 1 -- every sampled epoch is pinned at round-off
    level, with no amplification along the trajectory.  m and v are locals of the reference's `run`
    (optimizer.py:327-335); they are read from its frame when it calls `loss_grad`, the reference code itself is
    not touched."""
    import sys

    domain, state, extra = make_poisson(ndim, N)
    want = sorted(set(sample) | {k + 1 for k in sample})
    data = dict(ndim=np.array(ndim), N=np.array(N), lr=np.array(0.005), epochs=np.array(epochs),
                rhs=npy(extra.rhs), sample=np.array(sorted(sample)))
    losses = []

    def loss_grad(arrays):
        frame = sys._getframe(1)
        assert frame.f_code.co_name == "run" and "m" in frame.f_locals and "v" in frame.f_locals
        k = len(losses) + 1  # the epoch about to be evaluated (1-based, = the reference's local_epoch)
        if k in want:
            for name in ("x", "m", "v"):
                for i, a in enumerate(frame.f_locals[name]):
                    data[f"{name}{i}_e{k}"] = npy(a).copy()
        loss, grads, terms, _ = ref_loss_grad(domain, state, extra, poisson.operator, arrays)
        losses.append(float(loss))
        return loss, grads, None

    opt = odil.optimizer.AdamNativeOptimizer(dtype=np.float64, mod=mod)
    opt.run(domain.arrays_from_state(state), loss_grad, epochs=epochs, lr=0.005, jit=False)
    data["losses"] = np.array(losses)
    ref = load_losses(f"traj_adam_{ndim}d_N{N}")
    assert np.array_equal(ref[: len(losses)], data["losses"]), "not the run the trajectory fixture holds"
    save(f"traj_adam_states_{ndim}d_N{N}", **data)


def lbfgsb_iterates():
    """Teacher-forcing fixture for L-BFGS-B: EVERY iterate x_k of the reference run `a` of lbfgsb_pair() (the
    arrays the reference's callback receives, optimizer.py:84-88).  The limited-memory pairs of iteration k are
    s_i = x_{i+1} - x_i, y_i = g(x_{i+1}) - g(x_i), i < k: an implementation handed that history and x_k must
    produce the reference's x_{k+1} -- one iteration at a time, far beyond the 18 iterations over which two
    reference runs agree with each other."""
    import os

    pair = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "traj_lbfgsb_2d_N32_pair.npz"))
    domain, state, extra = make_poisson(2, 32)
    nlvl = len(domain.arrays_from_state(state))
    start = [pair[f"start{i}"] for i in range(nlvl)]
    epochs = int(pair["epochs"])
    evals, iters, xs = [], [], [np.concatenate([a.ravel() for a in start])]

    def loss_grad(arrays):
        loss, grads, terms, _ = ref_loss_grad(domain, state, extra, poisson.operator, arrays)
        evals.append(float(loss))
        return loss, grads, None

    def callback(arrays, epoch, pinfo):
        iters.append(evals[-1])
        xs.append(np.concatenate([npy(a).ravel() for a in arrays]))

    opt = odil.optimizer.LbfgsbOptimizer(dtype=np.float64, mod=mod, m=50, maxls=50)
    try:
        opt.run([T(a) for a in start], loss_grad, epochs=epochs, callback=callback)
    except odil.EarlyStopError as e:
        print("early stop", e)
    assert np.array_equal(np.array(iters), pair["iter_losses_a"]), "not the run the pair fixture holds"
    save("traj_lbfgsb_2d_N32_iterates", x=np.array(xs), iter_losses=np.array(iters), evals_per_iter=np.array(len(evals)),
         rhs=npy(extra.rhs), m=np.array(50), maxls=np.array(50), shapes=np.array([a.shape for a in start]))


def workload_adam_states(epochs=20, sample=(1, 2, 5, 10, 19)):
    """Teacher-forcing fixtures for the TRACED workloads (heat, veltracer and their generalisations with one more space
    dimension; f64 and f32): the reference's AdamNativeOptimizer (optimizer.py:311-336) run for `epochs` epochs on the
    set-ups of make_golden.py (the reference's operators, its core.py, its optimizer, all on the shim; every array of
    the state, network weights included, is an unknown), started from the fixture's random state.  Stored: the
    optimizer's own (x, m, v) at the sampled epochs k and at k + 1, and the loss of every epoch.  `tracers["epoch"]` is
    k at evaluation k (what the examples' callbacks set)."""
    import sys

    import make_golden as mg

    mg.gen_heat()
    mg.gen_veltracer()
    mg.gen_generalised()
    for name, case in mg.CASES.items():
        smp = (1, 10) if name.startswith("veltracer3d") else sample  # (four 4-D fields: 0.2 MB per stored state)
        want = sorted(set(smp) | {k + 1 for k in smp})
        domain, state, extra, operator = case["domain"], case["state"], case["extra"], case["operator"]
        dtype = domain.dtype
        data = dict(lr=np.array(case["lr"]), epochs=np.array(epochs), sample=np.array(sorted(smp)))
        losses = []

        def loss_grad(arrays, want=want):
            frame = sys._getframe(1)
            assert frame.f_code.co_name == "run" and "m" in frame.f_locals and "v" in frame.f_locals
            k = len(losses) + 1
            if k in want:
                for nm in ("x", "m", "v"):
                    for i, a in enumerate(frame.f_locals[nm]):
                        data[f"{nm}{i}_e{k}"] = npy(a).copy()
            leaves = [a.detach().clone().requires_grad_(True) for a in arrays]
            domain.arrays_to_state(leaves, state)
            ctx = odil.core.Context(domain, state, extra=extra, tracers={"epoch": k})
            ff = operator(ctx)
            values = [f[1] if isinstance(f, tuple) else f for f in ff]
            loss = sum(mod.mean(mod.square(v)) for v in values)
            grads = torch.autograd.grad(loss, leaves, allow_unused=True)
            grads = [g if g is not None else torch.zeros_like(a) for g, a in zip(grads, leaves)]
            losses.append(float(loss))
            return loss.detach(), grads, None

        opt = odil.optimizer.AdamNativeOptimizer(dtype=dtype, mod=mod)
        opt.run([a.detach().clone() for a in case["arrays"]], loss_grad, epochs=epochs, lr=case["lr"], jit=False)
        data["losses"] = np.array(losses)
        save("traj_adam_states_" + name, **data)


def workload_lbfgsb_pairs(iters=40):
    """The reference's LbfgsbOptimizer (optimizer.py:54-117 -> SciPy) on the three example workloads whose default
    optimizer it is -- wave, heat_tmax, infer_constant (reference examples/*/: `optimizer="lbfgsb"`) -- from the fixtures'
    random states, TWICE (the second start one ulp away): loss of every iteration of both runs, so that an
    implementation can be held to the reference for as long as the reference agrees with itself."""
    import make_golden as mg
    import scipy

    mg.gen_examples_f2()
    for name in ("wave_f64", "heat_tmax_f64", "infer_constant_f64"):
        case = mg.CASES[name]
        domain, state, extra, operator = case["domain"], case["state"], case["extra"], case["operator"]
        data = dict(epochs=np.array(iters), m=np.array(50), maxls=np.array(50), scipy_version=np.array(scipy.__version__))
        for tag in ("a", "b"):
            start = [npy(a) for a in case["arrays"]]
            if tag == "b":
                start = [np.nextafter(a, np.inf) for a in start]
            evals, its = [], []

            def loss_grad(arrays):
                leaves = [T(npy(a)).requires_grad_(True) for a in arrays]
                domain.arrays_to_state(leaves, state)
                ctx = odil.core.Context(domain, state, extra=extra, tracers={"epoch": 0})
                values = [f[1] if isinstance(f, tuple) else f for f in operator(ctx)]
                loss = sum(mod.mean(mod.square(v)) for v in values)
                grads = torch.autograd.grad(loss, leaves, allow_unused=True)
                grads = [g if g is not None else torch.zeros_like(a) for g, a in zip(grads, leaves)]
                evals.append(float(loss))
                return loss.detach(), grads, None

            opt = odil.optimizer.LbfgsbOptimizer(dtype=np.float64, mod=mod, m=50, maxls=50)
            try:
                opt.run([T(a) for a in start], loss_grad, epochs=iters, callback=lambda arrays, epoch, pinfo: its.append(evals[-1]))
            except odil.EarlyStopError as e:
                print("early stop", e)
            data["iter_losses_" + tag] = np.array(its)
        a, b = data["iter_losses_a"], data["iter_losses_b"]
        n = min(len(a), len(b))
        rel = np.abs(a[:n] - b[:n]) / np.abs(a[:n])
        bad = np.nonzero(rel > 1e-6)[0]
        print(name, "lbfgsb: reference vs reference one ulp apart agree to 1e-6 for", int(bad[0]) if len(bad) else n, "of", n, "iterations")
        save("traj_lbfgsb_pair_" + name, **data)


def load_losses(name):
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz"))["losses"]


if __name__ == "__main__":
    adam_traj(1, 256, 400)
    adam_traj(2, 32, 300)
    adam_traj(3, 16, 100)
    gd_traj()
    lbfgsb_pair()
    adam_states(1, 256, 400, [1, 2, 3, 5, 8, 12, 16, 20, 24, 25, 26, 30] + list(range(40, 400, 13)))
    adam_states(2, 32, 300, [1, 2, 3, 10, 50, 100, 124, 125, 126, 150, 200, 250, 299])
    adam_states(3, 16, 100, [1, 2, 50, 99])
    lbfgsb_iterates()
    workload_adam_states()
    workload_lbfgsb_pairs()
