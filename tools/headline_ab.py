"""The headline epoch (3-D Poisson 512^3, 9 levels, Adam) on ONE box: the bespoke driver bench.py timed up to round 5
(`poisson_path.PoissonMultigridAdam`) against the public API (`odil.util.optimize(args, "adam", problem, state, cb)` on
examples/poisson/poisson.py), each as bench.py times its headline (wall clock over K epochs between synchronisations)
and epoch by epoch with HIP events.   python tools/headline_ab.py [N] [K]"""
import gc
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples", "poisson"))
import bench  # noqa: E402
import odil_amd as odil  # noqa: E402
import poisson  # noqa: E402
from odil_amd.poisson_path import PoissonMultigridAdam  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
odil.util.set_log_file(open(os.devnull, "w"))


def bespoke():
    run = PoissonMultigridAdam(3, N, dtype=torch.float64, device=dev)
    for _ in range(5):
        run.epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        run.epoch()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / K


def api(callback_events, nogc):
    args = poisson.parse_args(["--ndim", "3", "--N", str(N)])
    problem, state = poisson.make_problem(args)
    args.epoch_start, args.epochs = 0, 5
    odil.util.optimize(args, "adam", problem, state, None)
    torch.cuda.synchronize()
    events = []

    def cb(st, epoch, pinfo):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        events.append(e)

    for _ in range(K + 4):
        cb(None, 0, None)
    torch.cuda.synchronize()
    events.clear()
    args.epoch_start, args.epochs = 0, K
    if nogc:
        gc.collect()
        gc.disable()
    t0 = time.perf_counter()
    odil.util.optimize(args, "adam", problem, state, cb if callback_events else None)
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - t0) / K
    gc.enable()
    times = [a.elapsed_time(b) for a, b in zip(events[:-1], events[1:])]
    return wall, (float(np.median(times)) if times else None)


for rnd in range(3):
    bench.spin_up(dev, 80)
    b = bespoke()
    bench.spin_up(dev, 80)
    w1, m1 = api(True, False)
    bench.spin_up(dev, 80)
    w2, m2 = api(True, True)
    bench.spin_up(dev, 80)
    w3, _ = api(False, True)
    print("round {}: bespoke wall {:.3f} | API events: wall {:.3f} median {:.3f} | API events, gc off: wall {:.3f} median {:.3f} | "
          "API no callback, gc off: wall {:.3f}  (ms per epoch; the API's wall includes the call's set-up)".format(rnd, b, w1, m1, w2, m2, w3),
          flush=True)
