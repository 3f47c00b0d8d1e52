"""The generic CPU oracle (oracle/odil_generic.py: any operator(ctx) through torch-CPU autograd + the NumPy
transfers) pinned against fixtures the REFERENCE produced for the same operators: tracer velocity (reference
examples/velocity_from_tracer/veltracer.py evaluated by the reference's core.py), its (t, x, y, z) generalisation
and the heat operators (pointwise MLP inside the stencil, tracers).  It is the undivided-domain checker of the
slab decomposition of traced operators (tests/test_slab_traced_cpu.py)."""

import argparse
import os
import sys

import numpy as np
import pytest
import torch
from conftest import ROOT, load_golden

from oracle import odil_generic as og

for sub in ("heat", "velocity_from_tracer"):
    sys.path.insert(0, os.path.join(ROOT, "examples", sub))


def golden_args(g):
    return argparse.Namespace(**{k[5:]: g[k].item() for k in g.files if k.startswith("args/")})


def mg_fields(g, keys, loc, nlvl, start=0):
    return {key: dict(kind="mg", loc=loc, terms=[g[f"x{start + i * nlvl + l}"] for l in range(nlvl)])
            for i, key in enumerate(keys)}


def check(g, loss, grads, terms, values, ngrads, tol=1e-12, named=False):
    assert abs(loss - float(g["loss"])) <= tol * abs(float(g["loss"]))
    names = [str(n) for n in g["names"]] if named and "term/0" not in g.files else list(range(len(terms)))
    for n, t in zip(names, terms):
        assert abs(t - float(g[f"term/{n}"])) <= tol * max(1.0, abs(float(g[f"term/{n}"])))
    for i in range(ngrads):
        want = g[f"g{i}"]
        assert np.max(np.abs(grads[i] - want)) <= 10 * tol * max(1.0, np.max(np.abs(want))), i
    for n, v in zip(names, values):
        if f"value/{n}" in g.files:
            want = g[f"value/{n}"]
            assert np.max(np.abs(v - want)) <= 10 * tol * max(1.0, np.max(np.abs(want))), n


def test_generic_oracle_veltracer_vs_reference():
    import veltracer

    g = load_golden("veltracer_f64")
    nlvl = int(g["nlvl"])
    geom = og.Geometry((int(g["Nt"]), int(g["Nx"]), int(g["Ny"])), ("t", "x", "y"), 0, 1, np.float64)
    extra = argparse.Namespace(args=golden_args(g), u_init=torch.tensor(g["u_init"]), u_final=torch.tensor(g["u_final"]))
    fields = mg_fields(g, ("u", "vx", "vy"), "ncc", nlvl)
    loss, grads, terms, names, values = og.eval_loss_grad(veltracer.operator, geom, fields, extra)
    assert len(terms) == int(g["nout"])
    check(g, loss, grads, terms, values, 3 * nlvl)


def test_generic_oracle_veltracer3d_vs_reference_framework():
    import veltracer3d

    g = load_golden("veltracer3d_f64")
    nlvl, n = int(g["nlvl"]), int(g["Nx"])
    geom = og.Geometry((int(g["Nt"]), n, n, n), ("t", "x", "y", "z"), 0, 1, np.float64)
    extra = argparse.Namespace(args=golden_args(g), u_init=torch.tensor(g["u_init"]), u_final=torch.tensor(g["u_final"]))
    fields = mg_fields(g, ("u",) + veltracer3d.VEL, "nccc", nlvl)
    loss, grads, terms, names, values = og.eval_loss_grad(veltracer3d.operator, geom, fields, extra)
    check(g, loss, grads, terms, values[:2], 4 * nlvl)


@pytest.mark.parametrize("which", ["heat", "heat2d"])
def test_generic_oracle_heat_vs_reference(which):
    ex = __import__(which)
    g = load_golden(which + "_f64")
    nlvl = int(g["nlvl"])
    if which == "heat":
        geom = og.Geometry((int(g["Nt"]), int(g["Nx"])), ("t", "x"), 0, 1, np.float64)
    else:
        geom = og.Geometry((int(g["Nt"]), int(g["Nx"]), int(g["Ny"])), ("t", "x", "y"), 0, 1, np.float64)
    extra = argparse.Namespace(args=golden_args(g), init_u=torch.tensor(g["init_u"]), imp_mask=torch.tensor(g["imp_mask"]),
                               imp_u=torch.tensor(g["imp_u"]), imp_size=int(g["imp_size"]))
    fields = mg_fields(g, ("u",), "c" * geom.ndim, nlvl)
    fields["k_net"] = dict(kind="net", weights=[g[f"x{nlvl + i}"] for i in range(3)],
                           biases=[g[f"x{nlvl + 3 + i}"] for i in range(3)], activation="tanh")
    loss, grads, terms, names, values = og.eval_loss_grad(ex.operator, geom, fields, extra, tracers={"epoch": int(g["epoch"])})
    assert names == [str(n) for n in g["names"]]
    check(g, loss, grads, terms, values, nlvl + 6, named=True)


@pytest.mark.parametrize("tag,tol", [("f64", 1e-12), ("f32", 2e-6)])
def test_generic_oracle_basic_fields_vs_reference(tag, tol):
    """One multigrid field per location, four output shapes (reference examples/basic/fields.py:16-40)."""
    sys.path.insert(0, os.path.join(ROOT, "examples", "basic"))
    import fields as ex

    g = load_golden("basic_fields_" + tag)
    nlvl = int(g["nlvl"])
    dtype = np.float64 if tag == "f64" else np.float32
    geom = og.Geometry((int(g["Nx"]), int(g["Ny"])), ("x", "y"), (0, 0), (2, 1), dtype)
    fields = {key: dict(kind="mg", loc=loc, terms=[g[f"x{i * nlvl + l}"] for l in range(nlvl)])
              for i, (key, loc) in enumerate(ex.FIELDS)}
    loss, grads, terms, names, values = og.eval_loss_grad(ex.operator, geom, fields)
    assert names == [str(n) for n in g["names"]]
    check(g, loss, grads, terms, values, 4 * nlvl, tol=tol)
