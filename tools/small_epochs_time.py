"""Epoch time of small Poisson problems (BASELINE config 1: 1-D N = 256; 2-D 256^2 ...) through the public API:
separate kernels (eager / hipGraph replay) against whole epochs in one launch (one launch per epoch / all epochs in one).

    python tools/small_epochs_time.py [ndim N epochs] ...
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples", "poisson"))
import odil_amd as odil  # noqa: E402
import poisson  # noqa: E402
from odil_amd import fused  # noqa: E402

odil.util.set_log_file(open(os.devnull, "w"))
cases = [(1, 256, 400), (2, 256, 100), (2, 64, 400), (1, 4096, 400)]
if len(sys.argv) > 3:
    v = [int(a) for a in sys.argv[1:]]
    cases = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)]
for ndim, N, epochs in cases:
    line = "{}-D N={} ({} epochs):".format(ndim, N, epochs)
    for mode in ("separate eager", "separate graph", "one launch per epoch", "all epochs in one launch"):
        os.environ["ODIL_GRAPH"] = "1" if mode == "separate graph" else "0"
        fused.PoissonEvaluator.small_max_cells = 0 if mode.startswith("separate") else 4096
        if os.environ.get("FORCE"):
            fused.PoissonEvaluator.small_force = not mode.startswith("separate")
        best = 1e9
        for rep in range(3):
            args = poisson.parse_args(["--ndim", str(ndim), "--N", str(N)])
            args.epoch_start, args.epochs = 0, epochs
            problem, state = poisson.make_problem(args)
            cb = None if mode == "all epochs in one launch" else (lambda st, ep, pinfo: None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            odil.util.optimize_grad(args, "adam", problem, state, cb)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / epochs * 1e6)
        line += "  {} {:.1f} us".format(mode, best)
    print(line, flush=True)
