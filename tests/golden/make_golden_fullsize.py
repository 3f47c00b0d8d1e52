#!/usr/bin/env python3
"""Value-level fixture of the HEADLINE workload at the size the metric is quoted on (BASELINE config 4a: 3-D Poisson
512^3, f64, 9 multigrid levels, Adam lr = 0.005, zero start).

    python tests/golden/make_golden_fullsize.py [N ...]        (build container; ~9 GB of host memory at 512, ~1 min)

Runs EPOCHS epochs of `oracle/poisson_epoch.c` (the plain-C restatement of the epoch, pinned to the NumPy oracle by
tests/test_oracle_c.py, which is pinned on the reference's golden vectors by tests/test_oracle_golden.py) and stores,
per epoch: the loss; per level: sum, sum of squares and NSAMPLE sampled entries of x, m, v (and of the gradient).
The inputs are NOT stored (1 GB): `reference_u` below regenerates ref_u with NumPy on the host the test runs on (same
image => same libm), the device forms rhs from it, and the fixture carries samples and checksums of both so that a drifted
input is reported as such.  tests/test_fullsize_values_gpu.py compares the HIP epoch (bespoke driver AND public API) with it.

Reference arithmetic restated by the C file: src/odil/core.py:245-263, 606-700 (synthesis), examples/poisson/poisson.py:57-113
(residual), core.py:1093-1100 (loss, reverse mode), optimizer.py:311-319 (Adam).
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ORACLE = os.path.join(ROOT, "oracle")
EPOCHS = 3
NSAMPLE = 64
LR = 0.005
P = ctypes.POINTER(ctypes.c_double)


def ptr(a):
    return None if a is None else a.ctypes.data_as(P)


def load_lib():
    so = os.path.join(ORACLE, "_build", "libpoisson_epoch.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", ORACLE, "-s"])
    lib = ctypes.CDLL(so)
    i64 = ctypes.c_int64
    lib.odil_c_residual.argtypes = [P, P, i64, i64, i64, P, P]
    lib.odil_c_residual.restype = ctypes.c_double
    lib.odil_c_epoch.argtypes = [i64, ctypes.c_int] + [ctypes.POINTER(P)] * 4 + [P] * 6 + [ctypes.c_int, ctypes.c_double]
    lib.odil_c_epoch.restype = ctypes.c_double
    return lib


def levels(N):
    nlvl = int(round(np.log2(N)))  # reference core.py:66-73
    return [(N >> l,) * 3 for l in range(nlvl)]


def reference_u(N):
    """ref_u = 'hat' on the cell centres of the unit cube (reference examples/poisson/poisson.py:18-24), float64
    (N, N, N), plain NumPy on the host: the SAME function gives the same bits in the generator, in the GPU test and in
    bench.py (one image, one libm) -- which matters: Adam from the zero state on this problem amplifies a last-bit change
    of the inputs by (1 / h^2)^2 lr / eps ~ 1e14 within two epochs (see tests/test_fullsize_values_gpu.py)."""
    x = (np.arange(N, dtype=np.float64) + 0.5) / N
    p = (1 - x) * x * 5
    u = np.ones((N, N, N))
    u *= p[:, None, None]
    u *= p[None, :, None]
    u *= p[None, None, :]
    u5 = u**5
    return (u5 / (1 + u5)) ** (1 / 5)


def reference_inputs(N):
    """(ref_u, rhs): rhs = the discrete Laplacian of ref_u (poisson.py:71-86) by the NumPy oracle, whose operation order
    the HIP residual kernel reproduces bit for bit (asserted at full size by the GPU test: the device forms ITS rhs from
    ref_u with `odil_poisson_residual` and must find the fixture's sampled entries exactly)."""
    sys.path.insert(0, ROOT)
    from oracle import odil_np as onp

    u = reference_u(N)
    return u, onp.poisson_discrete_rhs(u, onp.step((N,) * 3))


def sample_indices(N):
    """NSAMPLE flat indices per level (all of a level smaller than that), seeded by the level's size."""
    out = []
    for s in levels(N):
        n = int(np.prod(s))
        rng = np.random.default_rng(1000 + n)
        out.append(np.sort(rng.choice(n, size=min(NSAMPLE, n), replace=False)).astype(np.int64))
    return out


def stats(arrs, idx):
    """(sum, sum of squares) per level in extended precision + the sampled entries, one flat vector."""
    s = np.array([[float(np.sum(a, dtype=np.longdouble)), float(np.sum(np.square(a, dtype=np.longdouble)))] for a in arrs])
    return s, np.concatenate([a.reshape(-1)[i] for a, i in zip(arrs, idx)])


def make(N):
    lib = load_lib()
    shapes = levels(N)
    ref_u, rhs = reference_inputs(N)
    x, m, v, g = ([np.zeros(s) for s in shapes] for _ in range(4))
    arr = lambda arrs: (P * len(arrs))(*[ptr(a) for a in arrs])  # noqa: E731
    half = N // 2
    u, fu = np.zeros((N,) * 3), np.zeros((N,) * 3)
    work, la, lb = np.zeros((half + 2) ** 3), np.zeros(half**3), np.zeros(half**3)
    idx = sample_indices(N)
    out = dict(N=np.int64(N), epochs=np.int64(EPOCHS), lr=np.float64(LR),
               sample_index=np.concatenate(idx), sample_count=np.array([len(i) for i in idx]),
               rhs_stats=stats([rhs], [idx[0]])[0], rhs_samples=stats([rhs], [idx[0]])[1],
               ref_u_samples=stats([ref_u], [idx[0]])[1])
    losses = []
    for epoch in range(1, EPOCHS + 1):
        loss = lib.odil_c_epoch(N, len(shapes), arr(x), arr(m), arr(v), arr(g), ptr(rhs), ptr(u), ptr(fu), ptr(work),
                                ptr(la), ptr(lb), epoch, LR)
        # the C epoch sums its 1.3e8 squares one after the other (rounding ~ sqrt(n) eps = 1e-12 of the sum): the fixture
        # holds the mean square of the epoch's OWN residual array summed in extended precision instead
        loss_seq, loss = loss, float(np.sum(np.square(fu, dtype=np.longdouble)) / fu.size)
        assert abs(loss - loss_seq) <= 1e-10 * abs(loss)
        losses.append(loss)
        for name, arrs in (("x", x), ("m", m), ("v", v), ("g", g)):
            s, smp = stats(arrs, idx)
            out["{}_stats_e{}".format(name, epoch)] = s
            out["{}_samples_e{}".format(name, epoch)] = smp
        print("N={} epoch {} loss {!r}".format(N, epoch, loss), flush=True)
    out["losses"] = np.array(losses)
    path = os.path.join(HERE, "fullsize_poisson_N{}.npz".format(N))
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [64, 512]:
        make(n)
