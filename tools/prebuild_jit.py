#!/usr/bin/env python3
"""Fills odil_amd/_jit_cache with the generated kernels of the BASELINE configurations (and of the build / smoke example)
by tracing the example operators on CPU tensors and cross-compiling for gfx950 -- no GPU needed, nothing is launched.
A clean clone then does not spend its first GPU minutes in hipcc.  (The kernels of the GPU test suite's small fixtures --
a few hundred variants of a few seconds each -- are built by the suite's first run and kept in the same cache.)

    python3 tools/prebuild_jit.py [--quick]
"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for sub in ("poisson", "heat", "velocity_from_tracer", "wave", "heat_tmax", "infer_constant", "basic"):
    sys.path.insert(0, os.path.join(ROOT, "examples", sub))

CONFIGS = [  # (example module, argv, slab ranks or None)
    ("wave", ["--Nt", "8", "--Nx", "8"], None),                                                      # __graft_entry__.build()
    ("wave", ["--Nt", "8", "--Nx", "16"], None),                                                     # __graft_entry__.smoke()
    ("fields", [], None),                                                                            # one kernel set per field location
    ("heat", ["--Nt", "256", "--Nx", "512", "--infer_k", "1", "--imposed", "stripe"], None),          # config 3
    ("heat2d", ["--Nt", "256", "--Nx", "512", "--Ny", "512", "--infer_k", "1", "--imposed", "stripe"], None),  # config 3 at BASELINE's shape
    ("veltracer", ["--Nt", "128", "--Nx", "256", "--Ny", "256"], None),                                # config 5, reference-native
    ("veltracer3d", ["--Nt", "32", "--Nx", "256"], None),                                            # 5b
    ("veltracer3d", ["--Nt", "128", "--Nx", "32", "--Ny", "256", "--Nz", "256"], 1),                  # config 5: one rank's slab, world 1
    ("veltracer3d", ["--Nt", "128", "--Nx", "64", "--Ny", "256", "--Nz", "256"], 2),                  # ... and with interfaces (world >= 2)
]


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--quick", action="store_true", help="the build / smoke kernels only")
    a = p.parse_args()
    import odil_amd
    from odil_amd import runtime, slab_traced, stencil_jit

    runtime._mod = odil_amd.ModRocm(device="cpu")
    odil_amd.util.set_log_file(open(os.devnull, "w"))
    for modname, argv, world in CONFIGS[:2] if a.quick else CONFIGS:
        t0 = time.time()
        ex = importlib.import_module(modname)
        problem, state = ex.make_problem(ex.parse_args(argv))
        if world is None:
            path = stencil_jit.trace(problem, state).lib_path
            path = path[0] if isinstance(path, list) else path
        else:
            n = problem.domain.cshape[1] // world
            path = slab_traced.HipSlabKernels(problem, state, 1, n, "cpu").lib_path
        print("{:12s} {:60s} {}  {:.1f} s".format(modname, " ".join(argv), os.path.basename(path), time.time() - t0), flush=True)
        del problem, state
    if not a.quick:
        # config 5 with 4 and 8 ranks: the GLOBAL grid would not fit on the host -- the shape-only state bench.py builds
        import argparse as ap
        import numpy as np
        import veltracer3d

        for world in (4, 8):
            t0 = time.time()
            va = veltracer3d.parse_args(["--Nt", "128", "--Nx", str(32 * world), "--Ny", "256", "--Nz", "256"])
            domain = odil_amd.Domain(cshape=(va.Nt, va.Nx, va.Ny, va.Nz), dimnames=("t", "x", "y", "z"), lower=(0, 0, 0, 0),
                                     upper=(1, 1, 1, 1), dtype=np.float32, multigrid=va.multigrid, mg_interp=va.mg_interp,
                                     mg_nlvl=va.nlvl)
            x, y, z = np.meshgrid(*domain.points_1d("x", "y", "z"), indexing="ij")
            extra = ap.Namespace(args=va, u_init=domain.mod.cast(veltracer3d.blob(x, y, z, 0), np.float32),
                                 u_final=domain.mod.cast(veltracer3d.blob(x, y, z, 1), np.float32))
            del x, y, z
            state = odil_amd.State()
            for key in ("u",) + veltracer3d.VEL:
                state.fields[key] = odil_amd.Field(None, loc=veltracer3d.LOC)
            problem = odil_amd.Problem(veltracer3d.operator, domain, extra)
            path = slab_traced.HipSlabKernels(problem, slab_traced.shape_state(domain, state), 1, 32, "cpu").lib_path
            print("{:12s} {:60s} {}  {:.1f} s".format("veltracer3d", "config 5, {} ranks".format(world), os.path.basename(path),
                                                        time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
