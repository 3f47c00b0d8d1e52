#!/usr/bin/env python3
"""Generates the golden fixtures tests/golden/*.npz FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference; never on the GPU box):

    python3 tests/golden/make_golden.py

Everything numerical below is computed by the reference's own code
(`odil.core`, `odil.optimizer`, `odil.linsolver`, `examples/poisson/poisson.py`,
`tests/test_newton.py`) running unchanged on the torch-CPU `mod` shim of
ref_shim.py; `torch.autograd` replaces `jax.value_and_grad` / `tf.GradientTape`
exactly where the reference calls them (core.py:1100, :1062, :1346-1349).
The fixtures are data only: inputs and expected outputs.
"""

import argparse
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_shim import T, import_reference  # noqa: E402

odil = import_reference()
mod = odil.runtime.mod
torch.set_num_threads(1)


def load_module(name, path):
    import matplotlib

    matplotlib.use("Agg")
    spec = importlib.util.spec_from_file_location(name, path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def npy(x):
    return x.detach().numpy().copy() if isinstance(x, torch.Tensor) else np.array(x)


# set-ups of the example workloads as the generators below build them (make_golden_traj.py runs the reference's
# optimizers on them): name -> dict(domain, state, extra, arrays, operator, lr)
CASES = dict()


def save(name, **data):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **data)
    print("{:<28} {:>8.1f} kB  {} arrays".format(name + ".npz", os.path.getsize(path) / 1024, len(data)))


# ---------------------------------------------------------------- interp (A3, A3^T)
def gen_interp():
    rng = np.random.default_rng(101)
    data = dict()
    cases = []
    locs = ["c", "n", "cc", "nn", "cn", "nc", "c.", ".n", "ccc", "nnn", "cnn", "ncc", "c.n", "cccc", "nnnn", "cnnn", "nccc"]
    for loc in locs:
        ndim = len(loc)
        shape = tuple(int(3 + i + (1 if l == "n" else 0)) for i, l in enumerate(loc))
        u = rng.standard_normal(shape)
        ut = T(u).requires_grad_(True)
        fine = odil.core.interp_to_finer(ut, loc=loc, method="stack", mod=mod)
        gf = rng.standard_normal(tuple(fine.shape))
        (gu,) = torch.autograd.grad(fine, ut, T(gf))
        cases.append(loc)
        data[f"{loc}/u"] = u
        data[f"{loc}/fine"] = npy(fine)
        data[f"{loc}/gfine"] = gf
        data[f"{loc}/gu"] = npy(gu)
        if ndim <= 2:  # depth 2
            fine2 = odil.core.interp_to_finer(T(u), loc=loc, method="stack", mod=mod, depth=2)
            data[f"{loc}/fine2"] = npy(fine2)
    data["cases"] = np.array(cases)
    save("interp", **data)


# ---------------------------------------------------------------- conv transfers (A3 method="conv", A5)
def gen_conv_transfers():
    """The reference's own `interp_to_finer(method="conv")` (core.py:645-667; the default of the tracer workload,
    veltracer.py:150) and `restrict_to_coarser` (core.py:703-755), values and cotangents, 1-3-D, depth 1-2.  The loc
    combinations are those of the reference's tests (test_mg_interp.py / test_mg_restrict.py: prefixes of cccc, nnnn,
    cnnn, nccc) plus mixed ones and '.' axes -- on which the integer stride 2 of backend.py:118-119 subsamples."""
    rng = np.random.default_rng(303)
    locs = ["c", "n", "cc", "nn", "cn", "nc", "c.", ".n", "ccc", "nnn", "cnn", "ncc", "c.n"]
    data, cases = dict(), []
    for loc in locs:
        shape = tuple(int(3 + i + (1 if l == "n" else 0)) for i, l in enumerate(loc))
        u = rng.standard_normal(shape)
        ut = T(u).requires_grad_(True)
        fine = odil.core.interp_to_finer(ut, loc=loc, method="conv", mod=mod)
        stack = odil.core.interp_to_finer(T(u), loc=loc, method="stack", mod=mod)
        gf = rng.standard_normal(tuple(fine.shape))
        (gu,) = torch.autograd.grad(fine, ut, T(gf))
        cases.append(loc)
        data[f"{loc}/u"], data[f"{loc}/fine"], data[f"{loc}/gfine"], data[f"{loc}/gu"] = u, npy(fine), gf, npy(gu)
        data[f"{loc}/stack_minus_conv"] = np.array(float(torch.max(torch.abs(stack - fine.detach()))))
        if len(loc) <= 2:
            data[f"{loc}/fine2"] = npy(odil.core.interp_to_finer(T(u), loc=loc, method="conv", mod=mod, depth=2))
    data["cases"] = np.array(cases)
    save("interp_conv", **data)

    data, cases = dict(), []
    for loc in locs:
        # fine arrays whose restriction exists twice: 4 m cells / 4 m + 1 nodes per axis ('.': 5, subsampled to 3, 2)
        shape = tuple(5 if l == "." else 4 * (2 + i) + (1 if l == "n" else 0) for i, l in enumerate(loc))
        u = rng.standard_normal(shape)
        ut = T(u).requires_grad_(True)
        coarse = odil.core.restrict_to_coarser(ut, loc=loc, mod=mod)
        gc = rng.standard_normal(tuple(coarse.shape))
        (gu,) = torch.autograd.grad(coarse, ut, T(gc))
        cases.append(loc)
        data[f"{loc}/u"], data[f"{loc}/coarse"], data[f"{loc}/gcoarse"], data[f"{loc}/gu"] = u, npy(coarse), gc, npy(gu)
        data[f"{loc}/coarse2"] = npy(odil.core.restrict_to_coarser(T(u), loc=loc, mod=mod, depth=2))
    data["cases"] = np.array(cases)
    save("restrict", **data)


# ---------------------------------------------------------------- multigrid synthesis (A2) + adjoint
def gen_mg():
    rng = np.random.default_rng(202)
    data = dict()
    cases = []
    for name, cshape, loc, axes, factors in [
        ("1d_c", (16,), "c", None, None),
        ("1d_n", (16,), "n", None, None),
        ("2d_cc", (8, 16), "cc", None, None),
        ("2d_nc", (8, 16), "nc", None, None),
        ("2d_cn_axes", (8, 16), "cn", [True, False], None),
        ("3d_ccc", (8, 8, 8), "ccc", None, [1.0, 2.0, 0.5]),
        ("3d_ncc", (4, 8, 8), "ncc", None, None),
    ]:
        domain = odil.Domain(
            cshape=cshape, dimnames=["x", "y", "z"][: len(cshape)], multigrid=True, dtype=np.float64, mod=mod,
            mg_axes=axes, mg_factors=factors,
        )
        terms = []
        for cs in domain.mg_cshapes:
            shape = domain._get_field_shape(cs, loc)
            terms.append(odil.Field(T(rng.standard_normal(shape)).requires_grad_(True), loc=loc, cshape=cs))
        mgf = odil.MultigridField(terms=terms, loc=loc, factors=factors or [1] * len(terms), axes=axes, method="stack")
        u = domain.multigrid_to_regular(mgf).array
        gu = rng.standard_normal(tuple(u.shape))
        grads = torch.autograd.grad(u, [t.array for t in terms], T(gu))
        cases.append(name)
        data[f"{name}/cshape"] = np.array(cshape)
        data[f"{name}/loc"] = np.array(loc)
        data[f"{name}/axes"] = np.array(axes if axes else [True] * len(cshape))
        data[f"{name}/factors"] = np.array(factors or [1.0] * len(terms))
        data[f"{name}/nlvl"] = np.array(len(terms))
        for i, t in enumerate(terms):
            data[f"{name}/w{i}"] = npy(t.array)
            data[f"{name}/g{i}"] = npy(grads[i])
        data[f"{name}/u"] = npy(u)
        data[f"{name}/gu"] = gu
    data["cases"] = np.array(cases)
    save("mg", **data)


# ---------------------------------------------------------------- ctx.field access (A4)
def gen_field_access():
    rng = np.random.default_rng(303)
    data = dict()
    cases = []
    domain = odil.Domain(cshape=(4, 5), dimnames=["x", "y"], dtype=np.float64, mod=mod)
    specs = [
        ("cc", (0, 0), None),
        ("cc", (1, 0), None),
        ("cc", (-1, 2), None),
        ("nc", (0, 0), "cc"),
        ("nc", (1, 0), "cc"),
        ("cc", (0, 0), "nc"),
        ("cc", (-1, 0), "nc"),
        ("nn", (1, -1), "cc"),
        ("cn", (0, 1), "nc"),
    ]
    for k, (floc, shift, loc) in enumerate(specs):
        shape = domain.get_field_shape(floc)
        a = T(rng.standard_normal(shape)).requires_grad_(True)
        state = odil.State(fields={"f": odil.Field(a, loc=floc, cshape=domain.cshape)}, initialized=True)
        ctx = odil.core.Context(domain, state)
        out = ctx.field("f", *shift, loc=loc)
        g = rng.standard_normal(tuple(out.shape))
        (ga,) = torch.autograd.grad(out, a, T(g))
        name = f"case{k}"
        cases.append(name)
        data[f"{name}/field_loc"] = np.array(floc)
        data[f"{name}/shift"] = np.array(shift)
        data[f"{name}/loc"] = np.array(loc or floc)
        data[f"{name}/a"] = npy(a)
        data[f"{name}/out"] = npy(out)
        data[f"{name}/g"] = g
        data[f"{name}/ga"] = npy(ga)
    data["cases"] = np.array(cases)
    save("field_access", **data)


# ---------------------------------------------------------------- Poisson (A1, A6, A7, A10, A11)
poisson = load_module("ref_poisson", "/root/reference/examples/poisson/poisson.py")


def make_poisson(ndim, N, multigrid=1, dtype=np.float64):
    args = argparse.Namespace(
        ndim=ndim, N=N, multigrid=multigrid, double=1, cellbased=1, ref="hat", rhs="discrete", plot=0, mgloss=0
    )
    domain = odil.Domain(
        cshape=[N] * ndim, dimnames=["x", "y", "z"][:ndim], multigrid=multigrid, dtype=dtype, mod=mod
    )
    ref_u = poisson.get_ref_u("hat", args, domain)
    rhs = poisson.get_discrete_rhs(ref_u, domain, mod)
    state = odil.State()
    state.fields["u"] = None
    state = domain.init_state(state)
    extra = argparse.Namespace(ref_u=ref_u, rhs=rhs, args=args)
    return domain, state, extra


def ref_loss_grad(domain, state, extra, operator, arrays):
    """core.py:1082-1104 with torch.autograd in place of jax.value_and_grad."""
    arrays = [T(a).detach().clone().requires_grad_(True) for a in arrays]
    domain.arrays_to_state(arrays, state)
    ctx = odil.core.Context(domain, state, extra=extra, tracers={"epoch": 0})
    ff = operator(ctx)
    values = [f[1] if isinstance(f, tuple) else f for f in ff]
    terms = [mod.mean(mod.square(f)) for f in values]
    loss = sum(terms)
    grads = torch.autograd.grad(loss, arrays, allow_unused=True)
    grads = [g if g is not None else torch.zeros_like(a) for g, a in zip(grads, arrays)]
    return loss.detach(), [g.detach() for g in grads], [t.detach() for t in terms], values


def gen_poisson():
    rng = np.random.default_rng(404)
    for ndim, N, epochs in [(1, 256, 20), (2, 32, 10), (3, 16, 10), (2, 8, 5), (3, 8, 5)]:
        domain, state, extra = make_poisson(ndim, N)
        data = dict(ndim=np.array(ndim), N=np.array(N), lr=np.array(0.005))
        data["ref_u"] = npy(extra.ref_u)
        data["rhs"] = npy(extra.rhs)
        data["nlvl"] = np.array(domain.mg_nlvl)
        # loss + grads at a random multigrid state.
        arrays = [T(rng.standard_normal(tuple(a.shape)) * 0.1) for a in domain.arrays_from_state(state)]
        loss, grads, terms, values = ref_loss_grad(domain, state, extra, poisson.operator, arrays)
        for i, (a, g) in enumerate(zip(arrays, grads)):
            data[f"rand/w{i}"] = npy(a)
            data[f"rand/g{i}"] = npy(g)
        data["rand/loss"] = npy(loss)
        data["rand/fu"] = npy(values[0])
        # Adam trajectory from the zero state with the reference optimizer.
        domain, state, extra = make_poisson(ndim, N)
        losses = []

        def loss_grad(arrays):
            loss, grads, terms, _ = ref_loss_grad(domain, state, extra, poisson.operator, arrays)
            losses.append(float(loss))
            return loss, grads, None

        opt = odil.optimizer.AdamNativeOptimizer(dtype=np.float64, mod=mod)
        arrays = domain.arrays_from_state(state)
        x, _ = opt.run(arrays, loss_grad, epochs=epochs, lr=0.005, jit=False)
        data["adam/losses"] = np.array(losses)
        for i, a in enumerate(x):
            data[f"adam/w{i}"] = npy(a)
        save(f"poisson_{ndim}d_N{N}", **data)

    # float32, no multigrid (exercise dtype + plain Field path), GD
    domain, state, extra = make_poisson(2, 16, multigrid=0, dtype=np.float32)
    data = dict(rhs=npy(extra.rhs), ref_u=npy(extra.ref_u))
    arrays = [T(rng.standard_normal((16, 16)).astype(np.float32))]
    loss, grads, terms, values = ref_loss_grad(domain, state, extra, poisson.operator, arrays)
    data["rand/w0"] = npy(arrays[0])
    data["rand/g0"] = npy(grads[0])
    data["rand/loss"] = npy(loss)
    data["rand/fu"] = npy(values[0])
    save("poisson_2d_N16_f32_nomg", **data)


def gen_lbfgsb():
    import scipy

    domain, state, extra = make_poisson(2, 32)
    evals = []
    iters = []
    xs = []

    def loss_grad(arrays):
        loss, grads, terms, _ = ref_loss_grad(domain, state, extra, poisson.operator, arrays)
        evals.append(float(loss))
        return loss, grads, None

    def callback(arrays, epoch, pinfo):
        iters.append(evals[-1])
        xs.append(np.concatenate([npy(a).ravel() for a in arrays]))

    opt = odil.optimizer.LbfgsbOptimizer(dtype=np.float64, mod=mod, m=50, maxls=50)
    arrays = domain.arrays_from_state(state)
    epochs = 25
    try:
        x, optinfo = opt.run(arrays, loss_grad, epochs=epochs, callback=callback)
    except odil.EarlyStopError as e:
        print("early stop", e)
    data = dict(
        rhs=npy(extra.rhs), eval_losses=np.array(evals), iter_losses=np.array(iters), x_iters=np.array(xs),
        scipy_version=np.array(scipy.__version__), epochs=np.array(epochs), m=np.array(50), maxls=np.array(50),
    )
    save("lbfgsb_poisson_2d_N32", **data)


# ---------------------------------------------------------------- Newton (A13-A16)
def make_ref_problem(operator, domain, extra):
    """A reference `Problem` whose TF-only `_eval_operator_grad` (core.py:1313-1361) is
    re-expressed with torch.autograd; `Problem.linearize` (core.py:1113-1217) itself
    then runs unchanged."""
    problem = object.__new__(odil.Problem)
    problem.domain = domain
    problem.operator = operator
    problem.extra = extra
    problem.tracers = {"epoch": 0}

    def eval_operator_grad(state):
        watched = []

        def watch(x):
            for a in x if isinstance(x, (list, tuple)) else [x]:
                if not a.requires_grad:
                    a.requires_grad_(True)
                watched.append(a)

        # Leaves must be detached copies so that watched shifted copies are independent symbols.
        for field in state.fields.values():
            for a in domain.arrays_from_field(field):
                a.requires_grad_(True)
        ctx = odil.core.Context(domain, state, watch_func=watch, extra=extra, tracers=problem.tracers, distinct_shift=True)
        ff = operator(ctx)
        names = [f[0] if isinstance(f, tuple) else "" for f in ff]
        values = [f[1] if isinstance(f, tuple) else f for f in ff]
        grads = []
        for v in values:
            s = v.sum()
            descs = list(ctx.desc_to_array.keys())
            arrs = [ctx.desc_to_array[d] for d in descs]
            gg = torch.autograd.grad(s, arrs, retain_graph=True, allow_unused=True)
            g = dict(zip(descs, gg))
            for key, arr in ctx.key_to_array_jac.items():
                # tape.jacobian(v, arr): shape v.shape + arr.shape (core.py:1347-1349)
                def jac_of(a):
                    rows = []
                    for e in v.reshape(-1):
                        (ge,) = torch.autograd.grad(e, a, retain_graph=True, allow_unused=True)
                        rows.append(ge if ge is not None else torch.zeros_like(a))
                    return torch.stack(rows).reshape(tuple(v.shape) + tuple(a.shape))

                if isinstance(arr, list):
                    jac = [jac_of(a) for a in arr]
                    if all(float(j.abs().max()) == 0 for j in jac):
                        jac = [None for _ in jac]
                    g[key] = jac
                else:
                    j = jac_of(arr)
                    g[key] = j if float(j.abs().max()) != 0 else None
            grads.append(g)
        values = [v.detach() for v in values]
        return values, grads, names

    problem.eval_operator_grad = eval_operator_grad
    return problem


def linsolver_args():
    return argparse.Namespace(
        linsolver="direct", linsolver_maxiter=None, linsolver_damp=0, linsolver_dampdiag=0, linsolver_tol=1e-10
    )


def gen_newton():
    rng = np.random.default_rng(505)
    # --- Poisson 2-D and 3-D, plain Field (multigrid is rejected by linearize, core.py:1208-1209)
    for ndim, N in [(1, 8), (2, 6), (3, 4)]:
        domain, state, extra = make_poisson(ndim, N, multigrid=0)
        u0 = rng.standard_normal((N,) * ndim) * 0.1
        state.fields["u"].array = T(u0).clone()
        problem = make_ref_problem(poisson.operator, domain, extra)
        values, grads, names = problem.eval_operator_grad(state)
        data = dict(u0=u0, rhs=npy(extra.rhs), ref_u=npy(extra.ref_u), fu=npy(values[0]))
        shifts = []
        for (key, shift, loc), g in grads[0].items():
            sname = ",".join(str(s) for s in shift)
            shifts.append(sname)
            data[f"coeff/{sname}"] = npy(g)
        data["shifts"] = np.array(shifts)
        vector, matrix = problem.linearize(state)
        data["vector"] = npy(vector)
        data["matrix"] = matrix.toarray()
        delta = odil.linsolver.solve(matrix, -npy(vector), linsolver_args(), dict(), "direct")
        data["delta"] = delta
        packed = npy(domain.pack_state(state))
        data["u1"] = (packed + delta).reshape(u0.shape)
        save(f"newton_poisson_{ndim}d_N{N}", **data)


def gen_test_newton():
    """tests/test_newton.py:60-148 re-run with the shim (Problem constructed by hand
    because Problem.__init__ rejects non-TF/JAX mods, core.py:1035-1036)."""
    tn = load_module("ref_test_newton", "/root/reference/tests/test_newton.py")
    args = argparse.Namespace(Nx=3, Ny=2, Na=5, Nnet=5, multigrid=0, mg_interp="stack", nlvl=None)
    np.random.seed(1000)
    domain = odil.Domain(
        cshape=(args.Nx, args.Ny), dimnames=["x", "y"], lower=(0, 0), dtype=np.float64, upper=(args.Nx, args.Ny),
        multigrid=0, mod=mod,
    )
    dtype = domain.dtype
    net = domain.make_neural_net([args.Nnet, args.Nnet], activation="none")
    state = odil.State(
        fields={
            "uc": odil.Field(np.ones(domain.size(loc="cc")), loc="cc"),
            "ufx": odil.Field(np.ones(domain.size(loc="nc")), loc="nc"),
            "a": odil.Array(np.zeros(args.Na, dtype=dtype)),
            "net": net,
        }
    )
    state = domain.init_state(state)
    xc, yc = [npy(x) for x in domain.points(loc="cc")]
    xfx, yfx = [npy(x) for x in domain.points(loc="nc")]
    extra = argparse.Namespace()
    extra.ref = {
        "uc": 0.25 * xc * yc,
        "ufx": 0.25 * xfx * yfx,
        "dudx": 0.25 * yc,
        "a": np.linspace(0, 1, args.Na, dtype=dtype),
    }
    extra.ref["net_in"] = np.random.rand(args.Nnet, args.Nnet + 1)
    extra.ref["net_out"] = np.random.rand(args.Nnet, args.Nnet + 1)
    extra.args = args
    data = dict()
    for k, a in enumerate(domain.arrays_from_state(state)):
        data[f"x0/{k}"] = npy(a)
    for k, v in extra.ref.items():
        data[f"ref/{k}"] = np.array(v)
    # The operator mixes torch tensors with NumPy reference arrays; give it tensors.
    extra_t = argparse.Namespace(args=args, ref={k: T(v) for k, v in extra.ref.items()})
    problem = make_ref_problem(tn.operator, domain, extra_t)
    vector, matrix = problem.linearize(state)
    data["vector"] = npy(vector)
    data["matrix"] = matrix.toarray()
    import scipy.sparse as sp

    vector = npy(vector)
    delta = sp.linalg.spsolve((matrix.T @ matrix).tocsc(), -matrix.T @ vector)
    data["delta"] = delta
    packed = npy(domain.pack_state(state))
    with torch.no_grad():
        domain.unpack_state(T(packed + delta), state)
    for k, a in enumerate(domain.arrays_from_state(state)):
        data[f"x1/{k}"] = npy(a)
    errors = []
    for key in ["ufx", "uc", "a", "net_out"]:
        if key == "net_out":
            value = torch.stack(domain.neural_net(state, "net")(*T(extra.ref["net_in"])))
        else:
            value = domain.field(state, key)
        error = npy(value) - extra.ref[key]
        errors.append(np.sqrt(np.mean(np.square(error))))
    data["errors"] = np.array(errors)
    print("test_newton errors (must be < 1e-6):", errors)
    assert max(errors) < 1e-6
    save("test_newton", **data)


# ---------------------------------------------------------------- heat (A8, A17) and veltracer (A9)
def gen_heat():
    """reference examples/heat/heat.py:operator_odil with the inverse-problem terms switched on
    (infer_k: conductivity MLP [1,5,5,1] evaluated inside the stencil on frozen u; imposed
    points; annealed regularisation via ctx.tracers['epoch'])."""
    heat = load_module("ref_heat", "/root/reference/examples/heat/heat.py")
    rng = np.random.default_rng(606)
    for tag, dtype in [("f64", np.float64), ("f32", np.float32)]:
        Nt, Nx = 8, 16
        domain = odil.Domain(cshape=(Nt, Nx), dimnames=("t", "x"), multigrid=True, dtype=dtype, mod=mod)
        args = argparse.Namespace(
            keep_frozen=1, keep_init=1, infer_k=1, kmax=0.1, kimp=2.0, kxreg=0.3, kxregdecay=50.0, ktreg=0.2,
            ktregdecay=0.0, kwreg=0.05, kwregdecay=10.0,
        )
        tt, xx = domain.points()
        t1, x1 = domain.points_1d()
        init_u = heat.get_init_u(T(x1 * 0), T(x1), mod)
        ref_u = heat.get_init_u(tt, xx, mod)
        imp_mask = (rng.random((Nt, Nx)) < 0.2).astype(dtype)
        extra = argparse.Namespace(
            args=args, init_u=init_u, imp_mask=T(imp_mask), imp_size=int(imp_mask.sum()), imp_u=ref_u,
        )
        state = odil.State()
        state.fields["u"] = np.zeros(domain.cshape, dtype=dtype)
        layers = [1, 5, 5, 1]
        weights = [rng.uniform(-1, 1, (no, ni)).astype(dtype) for ni, no in zip(layers[:-1], layers[1:])]
        biases = [rng.uniform(-0.5, 0.5, (no,)).astype(dtype) for no in layers[1:]]
        state.fields["k_net"] = odil.NeuralNet([T(w) for w in weights], [T(b) for b in biases])
        state = domain.init_state(state)
        arrays = [T((rng.standard_normal(tuple(a.shape)) * 0.3).astype(dtype)) if i < domain.mg_nlvl else a
                  for i, a in enumerate(domain.arrays_from_state(state))]
        epoch = 7
        arrays_l = [a.detach().clone().requires_grad_(True) for a in arrays]
        domain.arrays_to_state(arrays_l, state)
        ctx = odil.core.Context(domain, state, extra=extra, tracers={"epoch": epoch})
        ff = heat.operator_odil(ctx)
        names = [f[0] for f in ff]
        values = [f[1] for f in ff]
        terms = [mod.mean(mod.square(v)) for v in values]
        loss = sum(terms)
        grads = torch.autograd.grad(loss, arrays_l, allow_unused=True)
        data = dict(Nt=np.array(Nt), Nx=np.array(Nx), epoch=np.array(epoch), names=np.array(names),
                    init_u=npy(init_u), imp_mask=imp_mask, imp_u=npy(ref_u), imp_size=np.array(extra.imp_size),
                    loss=npy(loss), nlvl=np.array(domain.mg_nlvl))
        for k, v in vars(args).items():
            data[f"args/{k}"] = np.array(v)
        for i, (a, g) in enumerate(zip(arrays, grads)):
            data[f"x{i}"] = npy(a)
            data[f"g{i}"] = npy(g) if g is not None else np.zeros(tuple(a.shape), dtype=dtype)
        for n, v, t in zip(names, values, terms):
            data[f"value/{n}"] = npy(v)
            data[f"term/{n}"] = npy(t)
        save(f"heat_{tag}", **data)
        CASES[f"heat_{tag}"] = dict(domain=domain, state=state, extra=extra, arrays=arrays, operator=heat.operator_odil, lr=1e-3)


def gen_veltracer():
    """reference examples/velocity_from_tracer/veltracer.py:operator_advection: fields u, vx, vy
    at loc 'ncc', first-order upwinding selected by the sign of the FROZEN velocity."""
    vt = load_module("ref_veltracer", "/root/reference/examples/velocity_from_tracer/veltracer.py")
    rng = np.random.default_rng(707)
    for tag, dtype in [("f64", np.float64), ("f32", np.float32)]:
        Nt, Nx, Ny = 8, 8, 8
        # mg_interp="conv": the workload's own default (veltracer.py:150), through the shim's conv_transpose
        domain = odil.Domain(cshape=(Nt, Nx, Ny), dimnames=("t", "x", "y"), lower=(0, 0, 0), upper=(1, 1, 1),
                             dtype=dtype, multigrid=True, mg_interp="conv", mod=mod)
        args = argparse.Namespace(kxreg=0.01, ktreg=1.0, kimp=10.0)
        x, y = domain.points("x", "y", loc=".cc")
        u_init = vt.u_init_blob(npy(x), npy(y), 0).astype(dtype)
        u_final = vt.u_init_blob(npy(x), npy(y), 1).astype(dtype)
        extra = argparse.Namespace(args=args, u_init=T(u_init), u_final=T(u_final))
        state = odil.State()
        for key in ["u", "vx", "vy"]:
            state.fields[key] = odil.Field(None, loc="ncc")
        state = domain.init_state(state)
        arrays = [T((rng.standard_normal(tuple(a.shape)) * 0.3).astype(dtype)) for a in domain.arrays_from_state(state)]
        arrays_l = [a.detach().clone().requires_grad_(True) for a in arrays]
        domain.arrays_to_state(arrays_l, state)
        ctx = odil.core.Context(domain, state, extra=extra, tracers={"epoch": 0})
        ff = vt.operator_advection(ctx)
        values = list(ff)
        terms = [mod.mean(mod.square(v)) for v in values]
        loss = sum(terms)
        grads = torch.autograd.grad(loss, arrays_l, allow_unused=True)
        data = dict(Nt=np.array(Nt), Nx=np.array(Nx), Ny=np.array(Ny), u_init=u_init, u_final=u_final,
                    loss=npy(loss), nlvl=np.array(domain.mg_nlvl), nout=np.array(len(values)))
        for k, v in vars(args).items():
            data[f"args/{k}"] = np.array(v)
        for i, (a, g) in enumerate(zip(arrays, grads)):
            data[f"x{i}"] = npy(a)
            data[f"g{i}"] = npy(g) if g is not None else np.zeros(tuple(a.shape), dtype=dtype)
        for i, (v, t) in enumerate(zip(values, terms)):
            data[f"value/{i}"] = npy(v)
            data[f"term/{i}"] = npy(t)
        save(f"veltracer_{tag}", **data)
        CASES[f"veltracer_{tag}"] = dict(domain=domain, state=state, extra=extra, arrays=arrays, operator=vt.operator_advection, lr=0.01)



# ---------------------------------------------------------------- wave, heat_tmax, infer_constant
def _loss_grads(ff, arrays_l):
    values = [f[1] if isinstance(f, tuple) else f for f in ff]
    names = [f[0] if isinstance(f, tuple) else "" for f in ff]
    terms = [mod.mean(mod.square(v)) for v in values]
    loss = sum(terms)
    grads = torch.autograd.grad(loss, arrays_l, allow_unused=True)
    return names, values, terms, loss, grads


def _store(data, arrays, names, values, terms, loss, grads):
    data["loss"] = npy(loss)
    data["names"] = np.array(names)
    for i, (a, g) in enumerate(zip(arrays, grads)):
        data[f"x{i}"] = npy(a)
        data[f"g{i}"] = npy(g) if g is not None else np.zeros(tuple(a.shape))
    for i, (v, t) in enumerate(zip(values, terms)):
        data[f"value/{i}"] = npy(v)
        data[f"term/{i}"] = npy(t)


def gen_examples_f2():
    """The reference's remaining example operators on random states: three time levels (wave),
    an `Array` unknown scaling the time step with a scalar output (heat_tmax), three constants
    and a residual without its first row (infer_constant)."""
    rng = np.random.default_rng(808)
    Nt, Nx = 8, 16
    # wave.py:29-75; boundary data are inputs of the fixture (the reference builds them with TF)
    wave = load_module("ref_wave", "/root/reference/examples/wave/wave.py")
    domain = odil.Domain(cshape=(Nt, Nx), dimnames=("t", "x"), lower=(0, -1), upper=(1, 1), multigrid=True,
                         dtype=np.float64, mod=mod)
    extra = argparse.Namespace(args=argparse.Namespace(kimp=1.5), left_u=T(rng.standard_normal(Nt)),
                               right_u=T(rng.standard_normal(Nt)), init_u=T(rng.standard_normal(Nx)),
                               init_ut=T(rng.standard_normal(Nx)))
    state = odil.State()
    state.fields["u"] = np.zeros(domain.cshape)
    state = domain.init_state(state)
    arrays = [T(rng.standard_normal(tuple(a.shape)) * 0.3) for a in domain.arrays_from_state(state)]
    arrays_l = [a.detach().clone().requires_grad_(True) for a in arrays]
    domain.arrays_to_state(arrays_l, state)
    ctx = odil.core.Context(domain, state, extra=extra, tracers={"epoch": 0})
    data = dict(Nt=np.array(Nt), Nx=np.array(Nx), kimp=np.array(1.5), nlvl=np.array(domain.mg_nlvl))
    for k in ("left_u", "right_u", "init_u", "init_ut"):
        data[k] = npy(getattr(extra, k))
    _store(data, arrays, *_loss_grads(wave.operator_wave(ctx), arrays_l))
    save("wave_f64", **data)
    CASES["wave_f64"] = dict(domain=domain, state=state, extra=extra, arrays=arrays, operator=wave.operator_wave, lr=None)

    # heat_tmax.py:29-75
    ht = load_module("ref_heat_tmax", "/root/reference/examples/heat_tmax/heat_tmax.py")
    domain = odil.Domain(cshape=(Nt, Nx), dimnames=("t", "x"), lower=(0, 0), upper=(1, np.pi), multigrid=True,
                         dtype=np.float64, mod=mod)
    extra = argparse.Namespace(args=argparse.Namespace(kimp=2.0), u_init=T(rng.standard_normal(Nx)),
                               u_final=T(rng.standard_normal(Nx)))
    state = odil.State(fields={"u": odil.Field(None, loc="nc"), "coeff": odil.Array([1.7])})
    state = domain.init_state(state)
    arrays = [T(rng.standard_normal(tuple(a.shape)) * 0.3) if i < domain.mg_nlvl else T(np.array([1.7]))
              for i, a in enumerate(domain.arrays_from_state(state))]
    arrays_l = [a.detach().clone().requires_grad_(True) for a in arrays]
    domain.arrays_to_state(arrays_l, state)
    ctx = odil.core.Context(domain, state, extra=extra, tracers={"epoch": 0})
    data = dict(Nt=np.array(Nt), Nx=np.array(Nx), kimp=np.array(2.0), nlvl=np.array(domain.mg_nlvl),
                u_init=npy(extra.u_init), u_final=npy(extra.u_final))
    _store(data, arrays, *_loss_grads(ht.operator_heat(ctx), arrays_l))
    save("heat_tmax_f64", **data)
    CASES["heat_tmax_f64"] = dict(domain=domain, state=state, extra=extra, arrays=arrays, operator=ht.operator_heat, lr=None)

    # infer_constant.py:44-74 (fields in the reference's order: coeff first)
    ic = load_module("ref_infer_constant", "/root/reference/examples/infer_constant/infer_constant.py")
    domain = odil.Domain(cshape=(Nt, Nx), dimnames=("t", "x"), lower=(0, -1), upper=(1, 1), multigrid=True,
                         dtype=np.float64, mod=mod)
    extra = argparse.Namespace(u_init=T(rng.standard_normal(Nx)), u_final=T(rng.standard_normal(Nx)))
    state = odil.State(fields={"coeff": odil.Array([0, 0, 0.001]), "u": odil.Field(None, loc="nc")})
    state = domain.init_state(state)
    arrays = [T(np.array([0.03, 0.2, -0.4])) if i == 0 else T(rng.standard_normal(tuple(a.shape)) * 0.3)
              for i, a in enumerate(domain.arrays_from_state(state))]
    arrays_l = [a.detach().clone().requires_grad_(True) for a in arrays]
    domain.arrays_to_state(arrays_l, state)
    ctx = odil.core.Context(domain, state, extra=extra, tracers={"epoch": 0})
    data = dict(Nt=np.array(Nt), Nx=np.array(Nx), nlvl=np.array(domain.mg_nlvl), u_init=npy(extra.u_init),
                u_final=npy(extra.u_final))
    _store(data, arrays, *_loss_grads(ic.operator_adv(ctx), arrays_l))
    save("infer_constant_f64", **data)
    CASES["infer_constant_f64"] = dict(domain=domain, state=state, extra=extra, arrays=arrays, operator=ic.operator_adv, lr=None)


# ---------------------------------------------------------------- generalised workloads at BASELINE's named shapes
def gen_generalised():
    """BASELINE.json names heat with two space dimensions and the tracer problem with three; the
    reference examples have one and two.  examples/heat/heat2d.py and
    examples/velocity_from_tracer/veltracer3d.py (this repository) generalise the discretisations;
    here those operators are evaluated by the REFERENCE's framework (its Context.field,
    multigrid_to_regular, eval_neural_net, on the shim) to pin what odil_amd must reproduce."""
    root = os.path.dirname(os.path.dirname(HERE))
    heat2d = load_module("our_heat2d", os.path.join(root, "examples", "heat", "heat2d.py"))
    vt3 = load_module("our_veltracer3d", os.path.join(root, "examples", "velocity_from_tracer", "veltracer3d.py"))
    rng = np.random.default_rng(909)
    for tag, dtype in [("f64", np.float64), ("f32", np.float32)]:
        Nt, Nx, Ny = 8, 8, 16
        domain = odil.Domain(cshape=(Nt, Nx, Ny), dimnames=("t", "x", "y"), multigrid=True, dtype=dtype, mod=mod)
        args = argparse.Namespace(keep_frozen=1, keep_init=1, infer_k=1, kmax=0.1, kimp=2.0, kxreg=0.3, kxregdecay=50.0,
                                  ktreg=0.2, ktregdecay=0.0)
        imp_mask = (rng.random((Nt, Nx, Ny)) < 0.2).astype(dtype)
        extra = argparse.Namespace(args=args, init_u=T(rng.standard_normal((Nx, Ny)).astype(dtype) * 0.5),
                                   imp_mask=T(imp_mask), imp_size=int(imp_mask.sum()),
                                   imp_u=T(rng.standard_normal((Nt, Nx, Ny)).astype(dtype)))
        state = odil.State()
        state.fields["u"] = np.zeros(domain.cshape, dtype=dtype)
        layers = [1, 5, 5, 1]
        weights = [rng.uniform(-1, 1, (no, ni)).astype(dtype) for ni, no in zip(layers[:-1], layers[1:])]
        biases = [rng.uniform(-0.5, 0.5, (no,)).astype(dtype) for no in layers[1:]]
        state.fields["k_net"] = odil.NeuralNet([T(w) for w in weights], [T(b) for b in biases])
        state = domain.init_state(state)
        arrays = [T((rng.standard_normal(tuple(a.shape)) * 0.3).astype(dtype)) if i < domain.mg_nlvl else a
                  for i, a in enumerate(domain.arrays_from_state(state))]
        arrays_l = [a.detach().clone().requires_grad_(True) for a in arrays]
        domain.arrays_to_state(arrays_l, state)
        ctx = odil.core.Context(domain, state, extra=extra, tracers={"epoch": 7})
        data = dict(Nt=np.array(Nt), Nx=np.array(Nx), Ny=np.array(Ny), epoch=np.array(7), nlvl=np.array(domain.mg_nlvl),
                    init_u=npy(extra.init_u), imp_mask=imp_mask, imp_u=npy(extra.imp_u), imp_size=np.array(extra.imp_size))
        for k, v in vars(args).items():
            data[f"args/{k}"] = np.array(v)
        _store(data, arrays, *_loss_grads(heat2d.operator(ctx), arrays_l))
        save(f"heat2d_{tag}", **data)
        CASES[f"heat2d_{tag}"] = dict(domain=domain, state=state, extra=extra, arrays=arrays, operator=heat2d.operator, lr=1e-3)

        Nt, Nx = 4, 8
        domain = odil.Domain(cshape=(Nt, Nx, Nx, Nx), dimnames=("t", "x", "y", "z"), lower=(0, 0, 0, 0),
                             upper=(1, 1, 1, 1), dtype=dtype, multigrid=True, mg_interp="stack", mod=mod)
        args = argparse.Namespace(kxreg=0.01, ktreg=1.0, kimp=10.0)
        extra = argparse.Namespace(args=args, u_init=T(rng.standard_normal((Nx, Nx, Nx)).astype(dtype)),
                                   u_final=T(rng.standard_normal((Nx, Nx, Nx)).astype(dtype)))
        state = odil.State()
        for key in ("u",) + vt3.VEL:
            state.fields[key] = odil.Field(None, loc="nccc")
        state = domain.init_state(state)
        arrays = [T((rng.standard_normal(tuple(a.shape)) * 0.3).astype(dtype)) for a in domain.arrays_from_state(state)]
        arrays_l = [a.detach().clone().requires_grad_(True) for a in arrays]
        domain.arrays_to_state(arrays_l, state)
        ctx = odil.core.Context(domain, state, extra=extra, tracers={"epoch": 0})
        data = dict(Nt=np.array(Nt), Nx=np.array(Nx), nlvl=np.array(domain.mg_nlvl), u_init=npy(extra.u_init),
                    u_final=npy(extra.u_final))
        for k, v in vars(args).items():
            data[f"args/{k}"] = np.array(v)
        names, values, terms, loss, grads = _loss_grads(vt3.operator(ctx), arrays_l)
        _store(data, arrays, names, values[:2], terms, loss, grads)  # values of the first two outputs only (size)
        for i, t in enumerate(terms):
            data[f"term/{i}"] = npy(t)
        save(f"veltracer3d_{tag}", **data)
        CASES[f"veltracer3d_{tag}"] = dict(domain=domain, state=state, extra=extra, arrays=arrays, operator=vt3.operator, lr=0.01)

# ---------------------------------------------------------------- reference tests as known-answer checks
# ---------------------------------------------------------------- examples/basic/fields.py: one field per location
def gen_basic_fields():
    """The reference's tutorial example on fields centred in cells, nodes and faces (examples/basic/fields.py:16-40):
    four multigrid fields of four different shapes, four outputs of four different shapes, and a network in the state
    that the operator never evaluates (its gradient is absent: zeros)."""
    rng = np.random.default_rng(909)
    bf = load_module("ref_basic_fields", "/root/reference/examples/basic/fields.py")
    Nx, Ny = 8, 4
    for tag, dtype in (("f64", np.float64), ("f32", np.float32)):
        domain = odil.Domain(cshape=(Nx, Ny), dimnames=["x", "y"], lower=(0, 0), upper=(2, 1), dtype=dtype, multigrid=1,
                             mg_axes=[True, True], mod=mod)
        state = odil.State(fields={
            "uc": odil.Field(np.zeros(domain.size(loc="cc")), loc="cc"),
            "un": odil.Field(np.zeros(domain.size(loc="nn")), loc="nn"),
            "ufx": odil.Field(np.zeros(domain.size(loc="nc")), loc="nc"),
            "ufy": odil.Field(np.zeros(domain.size(loc="cn")), loc="cn"),
            "net": domain.make_neural_net([2, 4, 2]),
        })
        state = domain.init_state(state)
        arrays = [T(rng.standard_normal(tuple(a.shape)) * 0.3).to(torch.float64 if dtype == np.float64 else torch.float32)
                  for a in domain.arrays_from_state(state)]
        arrays_l = [a.detach().clone().requires_grad_(True) for a in arrays]
        domain.arrays_to_state(arrays_l, state)
        ctx = odil.core.Context(domain, state, extra=None, tracers={"epoch": 0})
        data = dict(Nx=np.array(Nx), Ny=np.array(Ny), nlvl=np.array(domain.mg_nlvl))
        _store(data, arrays, *_loss_grads(bf.operator(ctx), arrays_l))
        save("basic_fields_" + tag, **data)


def check_reference_tests():
    """tests/test_mg_interp.py:11-32 on the shim: exact on linear functions."""
    for ndim in [1, 2, 3, 4]:
        for loc in {s[:ndim] for s in ["cccc", "nnnn", "cnnn", "nccc"]}:
            cshapeh = 3 + np.array(range(ndim))
            cshape = cshapeh * 2
            dimnames = ["x", "y", "z", "w"][:ndim]
            domain = odil.Domain(cshape=cshape, dimnames=dimnames, dtype=np.float64, mod=mod)
            domainh = odil.Domain(cshape=cshapeh, dimnames=dimnames, dtype=np.float64, mod=mod)

            def func(xx):
                return sum(x * np.sqrt(i + 1) for i, x in enumerate(xx))

            u = func(domain.points(loc=loc))
            uh = func(domainh.points(loc=loc))
            ui = odil.core.interp_to_finer(uh, loc=loc, mod=mod, method="stack")
            error = float(mod.max(abs(ui - u)))
            assert error <= np.finfo(np.float64).eps * 100, (ndim, loc, error)
    print("reference test_mg_interp (stack) on shim: PASS")


if __name__ == "__main__":
    check_reference_tests()
    gen_interp()
    gen_conv_transfers()
    gen_mg()
    gen_field_access()
    gen_poisson()
    gen_lbfgsb()
    gen_newton()
    gen_test_newton()
    gen_heat()
    gen_veltracer()
    gen_examples_f2()
    gen_basic_fields()
    gen_generalised()
