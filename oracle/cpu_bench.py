"""CPU-baseline worker of bench.py (TEST / MEASUREMENT INFRASTRUCTURE): times the NumPy oracle's Poisson multigrid
Adam epoch (oracle/odil_np.py -- the reference's op sequence, reference src/odil/optimizer.py:331-336 around
core.py:1076-1111) on ONE host thread and prints one JSON line.  bench.py starts one of these for the 1-core leg
and one per host core, concurrently, for the all-cores leg.

    python -m oracle.cpu_bench <ndim> <N> <budget_seconds> [start_at_unix_time]
"""
import json
import os
import sys
import time

for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):  # the reference's default: 1 thread
    os.environ[var] = "1"                                                     # (reference src/odil/runtime.py:8-12)
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import odil_np as onp  # noqa: E402


def main():
    ndim, N, budget = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    cshape = (N,) * ndim
    dw = onp.step(cshape)
    rhs = onp.poisson_discrete_rhs(onp.poisson_ref_u(cshape), dw)
    x = [np.zeros(s) for s in onp.mg_cshapes(cshape)]
    m = [np.zeros_like(a) for a in x]
    v = [np.zeros_like(a) for a in x]

    def epoch(k):
        nonlocal x, m, v
        loss, grads, _ = onp.poisson_loss_grad(x, rhs, dw)
        x, m, v = onp.adam_step(x, m, v, grads, k, 0.005)

    if N**ndim < 10**8:
        epoch(1)  # warm-up (at 512^3 an epoch takes minutes: the one timed epoch is the sample, first touches and all)
    if len(sys.argv) > 4:  # all workers of a leg start their timed region together
        time.sleep(max(0.0, float(sys.argv[4]) - time.time()))
    t0 = time.perf_counter()
    k = 0
    while True:
        epoch(k + 2)
        k += 1
        el = time.perf_counter() - t0
        if el > budget or k >= 200:
            break
    print(json.dumps({"cells": N**ndim, "epochs": k, "seconds": el}))


if __name__ == "__main__":
    main()
