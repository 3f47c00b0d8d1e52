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
    are generated from.  The stored adjoint arrays k_fwd would write are produced by the interpreter too: output
    seeds 2 f / n ('jac' outputs), and -- for the outputs k_fwd differentiates in registers -- the adjoints that
    arrive at the stored reads / affine cuts, built with the same symbolic reverse accumulation."""
    from odil_amd import stencil_grad
    from odil_amd.stencil_trace import _B, _I, _R

    tr, outs, raw, names, G = stencil_jit.trace_outputs(problem, state)
    cg = _Codegen(tr, outs, raw, G, state)
    cg.source()  # decides cuts and slots
    exprs = cg._gradient_terms()
    ev = DagEval(tr, G, dict(arrays_np), problem.tracers)

    def seed_array(k):
        f = np.asarray(ev(outs[k]), dtype=np.float64) * np.ones(G)
        seed = (1.0 if raw[k] else 2.0 * f) / cg.out_count[k] * np.ones(G)
        if cg.out_lens[k] is not None:
            box = np.ones(G, dtype=bool)
            for d, n in enumerate(cg.out_lens[k]):
                idx = np.arange(G[d]).reshape([-1 if e == d else 1 for e in range(len(G))])
                box = box & (idx < n)
            seed = np.where(box, seed, 0.0)
        return seed

    stored = list(cg.cots) + list(cg.cut_nodes)
    stop = {n.idx for n in cg.cut_nodes}
    acc = {n.idx: np.zeros(G) for n in stored}
    for k, mode in enumerate(cg.out_mode):
        if mode == "jac":
            ev.arrays[cg.seed_key[k]] = seed_array(k)
        elif mode == "legacy" and cg.need.get(outs[k].idx, False):
            key = "@seed{}".format(k)
            ev.arrays[key] = seed_array(k)
            tr.state_locs[key] = cg.gloc
            seed = tr.node("read", attr=(key, (0,) * len(G), cg.gloc, True), shape=G, kind=_R)
            gb = stencil_grad.GradBuilder(tr, G, cg.need, stop)
            if outs[k].idx in stop:
                acc[outs[k].idx] = acc[outs[k].idx] + ev.arrays[key]
                continue
            adj = gb.adjoints(outs[k], seed, stencil_grad.subdag(outs[k], stop), record_stops=True)  # reads + cuts
            for idx, e in adj.items():
                if idx in acc:
                    acc[idx] = acc[idx] + np.asarray(ev(e), dtype=np.float64) * np.ones(G)
    for slot, n in enumerate(stored):
        ev.arrays["@{}{}".format("r" if slot < len(cg.cots) else "c", slot)] = acc[n.idx]
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


@pytest.mark.parametrize("mode", ["chosen", "all_legacy", "all_recomputed"])
@pytest.mark.parametrize("seed", range(32))
def test_symbolic_gradient_of_random_operators_equals_autograd(cpu_mod, seed, mode, monkeypatch):
    """The random operators of the GPU parity test (tests/random_ops.py: shifted reads of two fields, masks, rolls,
    rows imposed by concatenation, `Array` parameters, windows): whatever mix of stored cotangents, affine cuts, stored
    output seeds and re-evaluated outputs the generator chooses, the gathers' expressions give autograd's gradient."""
    from random_ops import random_operator

    if mode == "all_legacy":  # every output differentiated in registers: stored read cotangents and affine cuts only
        monkeypatch.setenv("ODIL_TRACE_RECOMPUTE", "0")
    elif mode == "all_recomputed":  # every output that can be is cut, whatever the traffic model says
        monkeypatch.setenv("ODIL_TRACE_RECOMPUTE_ALL", "1")
    domain = odil.Domain(cshape=(8, 8), dimnames=("t", "x"), dtype=np.float64, multigrid=False)
    state = odil.State(fields={"a": odil.Field(None, loc="cc"), "b": odil.Field(None, loc="cc"),
                               "coeff": odil.Array([0.7, -0.4, 1.3])})
    state = domain.init_state(state)
    gen = torch.Generator(device="cpu").manual_seed(100 + seed)
    rows = torch.randn((8,), generator=gen, dtype=torch.float64)
    arrays = [torch.randn(tuple(a.shape), generator=gen, dtype=torch.float64) for a in domain.arrays_from_state(state)]
    domain.arrays_to_state(arrays, state)
    operator = random_operator(seed)
    problem = odil.Problem(operator, domain, extra=rows)
    loss, grads, terms, names, values = og.eval_loss_grad(operator, og.Geometry.of(domain), og.fields_of_state(domain, state), rows)
    got, cg = symbolic_gradients(problem, state, {"a": arrays[0].numpy(), "b": arrays[1].numpy(), "coeff": arrays[2].numpy()})
    scale = max(np.max(np.abs(g)) for g in grads[:2])
    for key, want in zip(("a", "b"), grads[:2]):
        if key in got:
            assert np.max(np.abs(got[key] - want)) <= 1e-11 * scale, (key, cg.out_mode)
        else:
            assert np.max(np.abs(want)) == 0


@pytest.mark.parametrize("which", ["heat", "heat2d"])
def test_shared_network_calls_are_proved_and_true(cpu_mod, which, monkeypatch):
    """odil_amd/stencil_share.py: the conductivity at the lower face of a cell IS the one at the upper face of its
    neighbour, k_m(i) = k_p(i - e), wherever i has an interior lower neighbour.  The identity is PROVED on the DAG
    (index-range simplification of the wall masks and of the periodic wrap) for every space axis of the heat
    operators -- and evaluated numerically here: the inputs of the two network calls agree at every point with
    i_axis >= 1 and differ somewhere on the wall row (where the generated kernel's halo threads evaluate the call)."""
    from odil_amd import stencil_share

    ex = __import__(which)
    argv = ["--Nt", "8", "--Nx", "16", "--infer_k", "1", "--imposed", "stripe", "--double", "1", "--multigrid", "0"]
    if which == "heat2d":
        argv += ["--Ny", "32"]
    problem, state = ex.make_problem(ex.parse_args(argv))
    rng = np.random.default_rng(3)
    arrays = problem.domain.arrays_from_state(state)
    arrays[0] = torch.tensor(rng.random(tuple(arrays[0].shape)))
    problem.domain.arrays_to_state(arrays, state)
    tr, outs, raw, names, G = stencil_jit.trace_outputs(problem, state)
    cg = _Codegen(tr, outs, raw, G, state)
    axes = tuple(range(len(G)))[1:]
    found = stencil_share.shared_network_calls(tr, cg.order, G, axes)
    assert sorted(axis for _, _, axis in found) == list(axes)
    ev = DagEval(tr, G, {"u": arrays[0].numpy()}, problem.tracers)
    for A, B, axis in found:
        shift = tuple(1 if d == axis else 0 for d in range(len(G)))
        for a, b in zip(A.args, B.args):
            va = np.asarray(ev(a), dtype=np.float64) * np.ones(G)
            vb = np.asarray(ev(tr.roll(b, shift, virtual=True)), dtype=np.float64) * np.ones(G)
            inner = tuple(slice(1, None) if d == axis else slice(None) for d in range(len(G)))
            wall = tuple(slice(0, 1) if d == axis else slice(None) for d in range(len(G)))
            assert np.array_equal(va[inner], vb[inner])
            assert not np.array_equal(va[wall], vb[wall])
    if which == "heat2d":  # on request the generator marches the last two axes and shares both (float kernels)
        monkeypatch.setenv("ODIL_TRACE_SHARE", "march")
        problem, state = ex.make_problem(ex.parse_args([a if a != "1" or argv[i - 1] != "--double" else "0" for i, a in enumerate(argv)]))
        tr, outs, raw, names, G = stencil_jit.trace_outputs(problem, state)
        cg = _Codegen(tr, outs, raw, G, state)
        assert sorted(axis for _, _, axis in cg.share) == [1, 2] and cg.share_mode == "march"
