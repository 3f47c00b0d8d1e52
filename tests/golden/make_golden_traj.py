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


if __name__ == "__main__":
    adam_traj(1, 256, 400)
    adam_traj(2, 32, 300)
    adam_traj(3, 16, 100)
    gd_traj()
    lbfgsb_pair()
