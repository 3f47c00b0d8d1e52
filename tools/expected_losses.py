#!/usr/bin/env python3
"""Loss after every epoch of the two bench.py workloads at N = 1, 2, 4, 8 ranks, produced with the ranks EMULATED on one
GPU (odil_amd.slab.run_lockstep: device copies instead of RCCL messages; the inputs are deterministic and so are the
kernels).  `bench.py --gpus N` compares the all-reduced loss of its real multi-process run with this table and reports
`parity_ok`: a wrong halo on the first RCCL run cannot pass as a plausible number.

    python3 tools/expected_losses.py [epochs] > profiles/expected_losses.json       (on a GPU box)
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples", "velocity_from_tracer"))


def poisson(world, epochs, dev):
    from odil_amd.poisson_path import PoissonMultigridAdam
    from odil_amd.slab import SlabPoissonAdam, run_lockstep

    losses = []
    if world == 1:
        run = PoissonMultigridAdam(3, 512, dtype=torch.float64, device=dev)
        for _ in range(epochs):
            run.epoch()
            losses.append(float(run.last_loss()))
        return losses
    ranks = [SlabPoissonAdam(512, r, world, dtype=torch.float64, device=dev) for r in range(world)]
    for _ in range(epochs):
        run_lockstep(ranks, 1)
        losses.append(float(sum(r.last_loss() for r in ranks)))
    return losses


def tracer(world, epochs, dev):
    import bench
    import odil_amd as odil
    from odil_amd.slab import run_lockstep

    odil.util.set_log_file(open(os.devnull, "w"))
    ranks = []
    for r in range(world):
        free, _ = torch.cuda.mem_get_info()
        if free < (34 << 30):
            raise MemoryError("{} GB free before rank {} of {}".format(free >> 30, r, world))
        ranks.append(bench.make_tracer_rank(argparse.Namespace(scale=1.0), r, world, dev)[0])
    losses = []
    for _ in range(epochs):
        run_lockstep(ranks, 1)
        losses.append(float(sum(r.last_loss() for r in ranks)))
    return losses


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dev = torch.device("cuda:0")
    out = dict(note="loss after epoch k (index k - 1), emulated ranks on one GPU, tools/expected_losses.py", epochs=epochs,
               poisson_512=dict(), tracer_cfg5=dict())
    for name, fn in (("poisson_512", poisson), ("tracer_cfg5", tracer)):
        for world in (1, 2, 4, 8):
            try:
                out[name][str(world)] = fn(world, epochs, dev)
            except MemoryError as e:
                out[name][str(world) + "_skipped"] = str(e)
            import gc

            gc.collect()
            torch.cuda.empty_cache()
            print(name, world, "done", file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
