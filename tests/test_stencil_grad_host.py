"""CPU check of the symbolic gradient expressions behind the generated gathers (odil_amd/stencil_grad.py,
stencil_codegen._gradient_terms): evaluated with the NumPy DAG interpreter (tests/dag_eval.py) they must equal the
gradient torch autograd gives for the same operator (oracle/odil_generic.py, itself pinned against fixtures the
reference produced) -- for the tracer-velocity operators and random stencil operators.  The HIP kernels generated
from the same expressions are held to the same fixtures on the GPU (tests/test_workloads_gpu.py)."""

import argparse
import os
import sys

import numpy as np
import pytest
import torch
from conftest import ROOT
from dag_eval import DagEval

import odil_amd as odil
from odil_amd import runtime, stencil_jit
from odil_amd.stencil_codegen import _Codegen
from oracle import odil_generic as og

for sub in ("heat", "velocity_from_tracer", "wave"):
    sys.path.insert(0, os.path.join(ROOT, "examples", sub))


@pytest.fixture()
def cpu_mod():
    saved, saved_log = runtime._mod, odil.util.g_log_file
    runtime._mod = odil.ModRocm(device="cpu")
    odil.util.set_log_file(open(os.devnull, "w"))
    yield runtime._mod
    runtime._mod = saved
    odil.util.g_log_file = saved_log


def symbolic_gradients(problem, state, arrays_np):
    """Gradient of the loss with respect to every regular field array, evaluated from the expressions the gathers
    are generated from: stored adjoints (read cotangents, affine cuts, output seeds) are produced by evaluating the
    forward DAG, then every field's expression G_F."""
    tr, outs, raw, names, G = stencil_jit.trace_outputs(problem, state)
    cg = _Codegen(tr, outs, raw, G, state)
    cg.source()  # decides cuts / slots and builds nothing we do not also rebuild below
    exprs = cg._gradient_terms()
    ev = DagEval(tr, G, dict(arrays_np), problem.tracers)
    # the stored adjoint arrays k_fwd would write
    for k, mode in enumerate(cg.out_mode):
        if mode == "jac":
            f = np.asarray(ev(outs[k]), dtype=np.float64) * np.ones(G)
            seed = (1.0 if raw[k] else 2.0 * f) / cg.out_count[k] * np.ones(G)
            if cg.out_lens[k] is not None:
                box = np.ones(G, dtype=bool)
                for d, n in enumerate(cg.out_lens[k]):
                    idx = np.arange(G[d]).reshape([-1 if e == d else 1 for e in range(len(G))])
                    box = box & (idx < n)
                seed = np.where(box, seed, 0.0)
            ev.arrays[cg.seed_key[k]] = seed
    assert not cg.cots, "legacy read cotangents need k_fwd's reverse pass: not interpreted here"
    for k, n in enumerate(cg.cut_nodes):  # affine cuts directly below an output: their adjoint is that output's seed
        (kout,) = [j for j, o in enumerate(outs) if o.idx == n.idx or (o.op in ("mul", "win") and any(a.idx == n.idx for a in o.args))]
        f = np.asarray(ev(outs[kout]), dtype=np.float64) * np.ones(G)
        seed = 2.0 * f / cg.out_count[kout]
        if outs[kout].idx != n.idx:  # output = cut * constant
            other = [a for a in outs[kout].args if a.idx != n.idx][0]
            seed = seed * float(ev(other))
        ev.arrays["@c{}".format(len(cg.cots) + k)] = seed
    return {key: np.asarray(ev(e), dtype=np.float64) * np.ones(G) for key, e in exprs.items()}, cg


@pytest.mark.parametrize("which", ["veltracer", "veltracer3d"])
def test_symbolic_gradient_of_tracer_operators_equals_autograd(cpu_mod, which):
    rng = np.random.default_rng(5)
    if which == "veltracer":
        import veltracer as ex

        args = ex.parse_args(["--Nx", "8", "--Nt", "6", "--multigrid", "0", "--double", "1"])
    else:
        import veltracer3d as ex

        args = ex.parse_args(["--Nx", "8", "--Nt", "6", "--multigrid", "0", "--double", "1"])
    problem, state = ex.make_problem(args)
    domain = problem.domain
    arrays = [torch.tensor(rng.standard_normal(tuple(a.shape)) * 0.3) for a in domain.arrays_from_state(state)]
    domain.arrays_to_state(arrays, state)
    keys = [k for k in state.fields]
    fields = og.fields_of_state(domain, state)
    extra = argparse.Namespace(**{k: (torch.as_tensor(np.asarray(v)) if hasattr(v, "shape") else v)
                                  for k, v in vars(problem.extra).items()})
    loss, grads, terms, names, values = og.eval_loss_grad(ex.operator, og.Geometry.of(domain), fields, extra)
    got, cg = symbolic_gradients(problem, state, {k: a.numpy() for k, a in zip(keys, arrays)})
    assert "jac" in cg.out_mode and "virt" in cg.out_mode and cg.ncot < len(keys) + 2
    for key, want in zip(keys, grads):
        err = np.max(np.abs(got[key] - want)) / max(np.max(np.abs(want)), 1e-300)
        assert err < 1e-12, (key, err)
