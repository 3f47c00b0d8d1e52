"""The plain-C restatement of the headline epoch (oracle/poisson_epoch.c, the CPU baseline bench.py times at the
headline's own size) pinned against the NumPy oracle (oracle/odil_np.py, itself pinned on the reference's golden
vectors by test_oracle_golden.py).  CPU only; nothing here touches the product."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import odil_np as onp

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
P = ctypes.POINTER(ctypes.c_double)


def ptr(a):
    return None if a is None else a.ctypes.data_as(P)


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-C", ORACLE, "-s"])
    lib = ctypes.CDLL(os.path.join(ORACLE, "_build", "libpoisson_epoch.so"))
    i64 = ctypes.c_int64
    lib.odil_c_interp_add.argtypes = [P, i64, i64, i64, P, P, P]
    lib.odil_c_interp_adj.argtypes = [P, i64, i64, i64, P, P]
    lib.odil_c_residual.argtypes = [P, P, i64, i64, i64, P, P]
    lib.odil_c_residual.restype = ctypes.c_double
    lib.odil_c_adjoint.argtypes = [P, i64, i64, i64, P, ctypes.c_double, P]
    lib.odil_c_epoch.argtypes = [i64, ctypes.c_int] + [ctypes.POINTER(P)] * 4 + [P] * 6 + [ctypes.c_int, ctypes.c_double]
    lib.odil_c_epoch.restype = ctypes.c_double
    return lib


@pytest.mark.parametrize("shape", [(2, 2, 2), (3, 5, 4), (8, 2, 6)])
def test_prolongation_and_its_transpose(lib, shape):
    rng = np.random.default_rng(1)
    n0, n1, n2 = shape
    fshape = tuple(2 * n for n in shape)
    coarse, add = rng.standard_normal(shape), rng.standard_normal(fshape)
    fine, work = np.empty(fshape), np.empty((n0 + 2) * (n1 + 2) * (n2 + 2))
    lib.odil_c_interp_add(ptr(coarse), n0, n1, n2, ptr(add), ptr(fine), ptr(work))
    np.testing.assert_allclose(fine, add + onp.interp_to_finer(coarse, "ccc"), rtol=0, atol=1e-14)
    gfine, gcoarse = rng.standard_normal(fshape), np.empty(shape)
    lib.odil_c_interp_adj(ptr(gfine), n0, n1, n2, ptr(gcoarse), ptr(work))
    np.testing.assert_allclose(gcoarse, onp.interp_to_finer_adj(gfine, "ccc", shape), rtol=0, atol=1e-14)
    # <P c, g> == <c, P^T g>
    lib.odil_c_interp_add(ptr(coarse), n0, n1, n2, None, ptr(fine), ptr(work))
    assert abs(np.vdot(fine, gfine) - np.vdot(coarse, gcoarse)) < 1e-12


@pytest.mark.parametrize("shape", [(2, 3, 2), (6, 5, 7)])
def test_residual_and_adjoint(lib, shape):
    rng = np.random.default_rng(2)
    u, rhs = rng.standard_normal(shape), rng.standard_normal(shape)
    dw = (0.5, 0.25, 0.125)
    h2 = np.array([d * d for d in dw])
    fu, gu = np.empty(shape), np.empty(shape)
    ssum = lib.odil_c_residual(ptr(u), ptr(rhs), *shape, ptr(h2), ptr(fu))
    ref = onp.poisson_residual(u, rhs, dw)
    np.testing.assert_allclose(fu, ref, rtol=0, atol=1e-11)
    assert abs(ssum - np.sum(ref**2)) <= 1e-12 * np.sum(ref**2)
    lib.odil_c_adjoint(ptr(fu), *shape, ptr(h2), 0.3, ptr(gu))
    np.testing.assert_allclose(gu, onp.poisson_adjoint(0.3 * ref, dw), rtol=0, atol=1e-10)


@pytest.mark.parametrize("N", [4, 16])
def test_whole_epochs_follow_the_numpy_oracle(lib, N):
    cshape = (N,) * 3
    dw = onp.step(cshape)
    rhs = onp.poisson_discrete_rhs(onp.poisson_ref_u(cshape), dw)
    shapes = onp.mg_cshapes(cshape)
    rng = np.random.default_rng(3)
    x = [rng.standard_normal(s) * 0.1 for s in shapes]
    m = [np.zeros(s) for s in shapes]
    v = [np.zeros(s) for s in shapes]
    cx, cm, cv = ([a.copy() for a in arrs] for arrs in (x, m, v))
    cg = [np.zeros(s) for s in shapes]
    arr = lambda arrs: (P * len(arrs))(*[ptr(a) for a in arrs])  # noqa: E731
    half = N // 2
    u, fu = np.empty(cshape), np.empty(cshape)
    work, la, lb = np.empty((half + 2) ** 3), np.empty(half**3), np.empty(half**3)
    for epoch in (1, 2, 3):
        loss_ref, grads_ref, _ = onp.poisson_loss_grad(x, rhs, dw)
        x, m, v = onp.adam_step(x, m, v, grads_ref, epoch, 0.005)
        loss = lib.odil_c_epoch(N, len(shapes), arr(cx), arr(cm), arr(cv), arr(cg), ptr(rhs), ptr(u), ptr(fu), ptr(work),
                                ptr(la), ptr(lb), epoch, 0.005)
        assert abs(loss - loss_ref) <= 1e-12 * abs(loss_ref)
        for a, b in zip(cg, grads_ref):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-11 * max(1.0, np.abs(b).max()))
        for a, b in zip(cx + cm + cv, list(x) + list(m) + list(v)):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-9 * max(1.0, np.abs(b).max()))


def test_timing_binary_reports_one_json_line(lib):
    import json

    out = subprocess.check_output([os.path.join(ORACLE, "_build", "poisson_epoch"), "8", "0.05"], text=True)
    rec = json.loads(out.strip().splitlines()[-1])
    assert rec["cells"] == 512 and rec["levels"] == 3 and rec["epochs"] >= 1 and np.isfinite(rec["loss"])


def test_threaded_binary_follows_the_serial_one(lib):
    """oracle/_build/poisson_epoch_omp (the same source with -fopenmp: ONE problem on all host cores, bench.py's
    `cpu_baseline.all_cores`) reorders a few sums only: its loss after a fixed number of epochs equals the serial binary's
    to 1e-12, and with one thread to the last bit."""
    import json

    def loss(binary, threads):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads))
        out = subprocess.check_output([os.path.join(ORACLE, "_build", binary), "16", "1", "0", "4"], text=True, env=env)
        rec = json.loads(out.strip().splitlines()[-1])
        assert rec["epochs"] == 4
        return rec["loss"]

    serial = loss("poisson_epoch", 1)
    assert loss("poisson_epoch_omp", 1) == serial
    for threads in (2, 3, 8):
        assert abs(loss("poisson_epoch_omp", threads) - serial) <= 1e-12 * abs(serial)
