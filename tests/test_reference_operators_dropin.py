"""north_star: "user residual callbacks drop in unchanged" -- SHOWN, in the build container (needs /root/reference; skipped
on the GPU box, nothing of the reference is shipped): the reference's own example files are loaded by path with `import
odil` resolved to THIS package, their operator functions (unchanged source) are handed to `odil_amd.Problem`, and

  * run ONCE on symbolic values through the tracer (`stencil_jit.trace_outputs`) -- i.e. they take the generated-kernel
    route, not the autograd fallback -- and the generated HIP source cross-compiles for gfx950;
  * the traced outputs, evaluated by the NumPy DAG interpreter (tests/dag_eval.py) on random states, equal those of the
    restated operator under `examples/` (the one the GPU parity tests hold to the reference-generated fixtures) to 1e-13;
    where the two record their operations in the same order the generated source is identical, which is asserted too
    for the operators where it holds today.

Reference operators: examples/poisson/poisson.py:89-123, heat/heat.py:36-137, velocity_from_tracer/veltracer.py:34-130,
wave/wave.py:29-75, heat_tmax/heat_tmax.py:29-75, infer_constant/infer_constant.py:44-74, basic/fields.py:16-40."""

import hashlib
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import pytest
from conftest import ROOT
from dag_eval import DagEval

import odil_amd
from odil_amd import runtime, stencil_jit

REF = "/root/reference/examples"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")

# (name, reference file, operator function in it, restated example module, its argv)
CASES = [
    ("poisson", "poisson/poisson.py", "operator", "poisson", ["--ndim", "2", "--N", "16"]),
    ("heat", "heat/heat.py", "operator_odil", "heat", ["--Nt", "8", "--Nx", "16", "--infer_k", "1", "--imposed", "stripe",
                                                     "--kxreg", "0.1", "--kxregdecay", "100", "--kwreg", "0.05"]),
    ("veltracer", "velocity_from_tracer/veltracer.py", "operator_advection", "veltracer", ["--Nt", "8", "--Nx", "16"]),
    ("wave", "wave/wave.py", "operator_wave", "wave", ["--Nt", "8", "--Nx", "16"]),
    ("heat_tmax", "heat_tmax/heat_tmax.py", "operator_heat", "heat_tmax", ["--Nt", "8", "--Nx", "16"]),
    ("infer_constant", "infer_constant/infer_constant.py", "operator_adv", "infer_constant", ["--Nt", "8", "--Nx", "16"]),
    ("fields", "basic/fields.py", "operator", "fields", []),
]
SAME_SOURCE = {"fields"}  # same recording order as the restated operator: byte-identical generated source


@pytest.fixture()
def as_odil(monkeypatch):
    """`import odil` (and `odil.runtime`, `odil.core`, ...) resolve to odil_amd on a CPU `mod`; the plotting modules
    (out of scope) are empty stand-ins, and `odil.runtime.tf` only knows the `@tf.function()`
    decorator that reference files apply to plotting helpers at import time."""
    import matplotlib

    matplotlib.use("Agg")
    shim = types.ModuleType("odil")
    shim.__path__ = []
    for k, v in vars(odil_amd).items():
        if not k.startswith("__"):
            setattr(shim, k, v)
    rt = types.ModuleType("odil.runtime")
    for k, v in vars(runtime).items():
        if not k.startswith("__"):
            setattr(rt, k, v)
    rt.tf = types.SimpleNamespace(function=lambda *a, **k: (lambda f: f))
    shim.runtime = rt
    monkeypatch.setitem(sys.modules, "odil", shim)
    monkeypatch.setitem(sys.modules, "odil.runtime", rt)
    for sub in ("core", "util", "history", "optimizer", "linsolver", "backend", "io"):
        monkeypatch.setitem(sys.modules, "odil." + sub, getattr(odil_amd, sub))
    for stub in ("plotutil", "plot"):
        m = types.ModuleType("odil." + stub)
        monkeypatch.setitem(sys.modules, "odil." + stub, m)
        setattr(shim, stub, m)
    monkeypatch.setattr(runtime, "_mod", odil_amd.ModRocm(device="cpu"))
    saved = odil_amd.util.g_log_file
    odil_amd.util.set_log_file(open(os.devnull, "w"))
    for sub in ("poisson", "heat", "velocity_from_tracer", "wave", "heat_tmax", "infer_constant", "basic"):
        monkeypatch.syspath_prepend(os.path.join(ROOT, "examples", sub))
    monkeypatch.setenv("ODIL_FUSE", "0")
    yield shim
    odil_amd.util.g_log_file = saved


def load_by_path(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def traced_sources_and_values(problem, state, arrays, nets):
    """(generated source of every kernel set, names, outputs evaluated by the DAG interpreter)."""
    try:
        groups = [None]
        stencil_jit.trace_outputs(problem, state)
    except stencil_jit.TraceGroups as e:
        groups = e.groups
    sources, names, values = [], [], []
    for only in groups:
        tr, outs, raw, nm, G = stencil_jit.trace_outputs(problem, state, only=only)
        ev = DagEval(tr, G, {k: v for k, v in arrays.items() if v.shape == tuple(G) or v.ndim == 1}, problem.tracers, nets)
        for n, o in zip(nm, outs):
            names.append(n)
            values.append(np.asarray(ev(o), dtype=np.float64) * np.ones(o.shape if o.win is None else G))
        sources.append(stencil_jit.TracedOperator(problem, state, only=only).source)
    return sources, names, values


@pytest.mark.parametrize("name,ref_file,opname,mine_name,argv", CASES, ids=[c[0] for c in CASES])
def test_reference_operator_source_runs_on_this_api_and_traces_like_the_restated_one(as_odil, name, ref_file, opname, mine_name, argv):
    from odil_amd.core import Array, Field, MultigridField, NeuralNet

    mine = importlib.import_module(mine_name)
    problem, state = mine.make_problem(mine.parse_args(argv))
    domain = problem.domain
    ref_operator = getattr(load_by_path(os.path.join(REF, ref_file), "reference_example_" + name), opname)
    assert ref_operator.__code__.co_filename.startswith("/root/reference/")
    ref_problem = odil_amd.Problem(ref_operator, domain, problem.extra, tracers=dict(problem.tracers or {}))
    # a random state: regular arrays of the grid fields (the DAG reads them), parameter vectors, network weights
    rng = np.random.default_rng(5)
    arrays, nets = dict(), dict()
    for key, f in state.fields.items():
        if isinstance(f, (Field, MultigridField)):
            arrays[key] = rng.standard_normal(tuple(domain.get_field_shape(f.loc))) * 0.3
        elif isinstance(f, Array):
            arrays[key] = rng.standard_normal(tuple(f.array.shape)) * 0.3 + 1.0
        elif isinstance(f, NeuralNet):
            nets[key] = ([rng.uniform(-1, 1, tuple(w.shape)) for w in f.weights], [rng.uniform(-0.5, 0.5, tuple(b.shape)) for b in f.biases])
    if problem.tracers is not None and "epoch" in (problem.tracers or {}):
        problem.tracers["epoch"] = ref_problem.tracers["epoch"] = 7
    src_mine, names_mine, val_mine = traced_sources_and_values(problem, state, arrays, nets)
    src_ref, names_ref, val_ref = traced_sources_and_values(ref_problem, state, arrays, nets)
    assert len(val_ref) == len(val_mine) and len(val_ref) >= 1
    assert [n or "" for n in names_ref] == [n or "" for n in names_mine]
    for k, (a, b) in enumerate(zip(val_ref, val_mine)):
        assert a.shape == b.shape, (name, k)
        scale = max(np.abs(b).max(), 1e-300)
        assert np.abs(a - b).max() <= 1e-13 * scale, (name, k, np.abs(a - b).max() / scale)
    same = [hashlib.sha1(a.encode()).hexdigest() == hashlib.sha1(b.encode()).hexdigest() for a, b in zip(src_ref, src_mine)]
    if name in SAME_SOURCE:
        assert all(same), name
