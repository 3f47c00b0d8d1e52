"""Host side of the operator tracer (odil_amd/stencil_jit.py): the example operators trace on CPU
tensors, the generated HIP source cross-compiles for gfx950 and exports its launchers; what the
tracer cannot express is refused (and the problem then keeps the generic path).  No launches."""

import argparse
import os
import sys

import numpy as np
import pytest
from conftest import ROOT

import odil_amd as odil
from odil_amd import stencil_jit

for sub in ("poisson", "heat", "velocity_from_tracer"):
    sys.path.insert(0, os.path.join(ROOT, "examples", sub))


@pytest.fixture()
def cpu_mod(monkeypatch):
    mod = odil.ModRocm(device="cpu")
    monkeypatch.setattr(odil.runtime, "_mod", mod)
    return mod


def test_examples_trace_and_compile(cpu_mod):
    import heat
    import veltracer

    problem, state = veltracer.make_problem(veltracer.parse_args(["--Nx", "16", "--Nt", "8"]))
    tro = stencil_jit.TracedOperator(problem, state)
    assert sorted(tro.cg.gathers) == ["u", "vx", "vy"]
    # 18 live reads.  Cotangents are cut at the OUTPUTS: the transport residual and the two Laplacian regularisers
    # store their adjoint once (3 arrays), the imposed-values term and the time differences are re-evaluated by the
    # gathers (nothing stored) -- instead of one cotangent array per stencil point
    assert tro.cg.out_mode == ["jac", "virt", "legacy", "legacy", "virt", "virt"] and tro.cg.ncot == 3
    assert not tro.cg.cots and len(tro.cg.cut_nodes) == 2 and tro.cg.vw_fwd == 4 and tro.cg.vw_gat == 4
    # (the regularisers are linear: as affine cuts of the in-register reverse pass they cost the same one array each)
    assert tro.cg.traffic_words["chosen"] < 0.7 * tro.cg.traffic_words["legacy"]
    assert hasattr(tro.lib, "jit_fwd") and hasattr(tro.lib, "jit_gather")
    assert "k_gat_2" in tro.source and "k_final" in tro.source
    # the legacy form (one cotangent per live read, affine sub-expressions cut): ODIL_TRACE_RECOMPUTE=0
    os.environ["ODIL_TRACE_RECOMPUTE"] = "0"
    try:
        old = stencil_jit.TracedOperator(problem, state)
    finally:
        del os.environ["ODIL_TRACE_RECOMPUTE"]
    assert len(old.cg.cut_nodes) == 2 and old.cg.ncot == 12

    args = heat.parse_args(["--Nx", "16", "--Nt", "8", "--infer_k", "1", "--imposed", "stripe", "--kxreg", "0.1",
                            "--kxregdecay", "100"])
    problem, state = heat.make_problem(args)
    tro = stencil_jit.TracedOperator(problem, state)
    assert tro.cg.nets == [("k_net", (1, 5, 5, 1))] and len(tro.cg.pg_decl) == 46
    assert len(tro.cg.hs) == 1  # the annealed weight: a host scalar recomputed from the epoch tracer
    problem.tracers["epoch"] = 200
    assert tro._host_value(tro.cg.hs[0], dict()) == pytest.approx(0.1 * 0.5 ** 2)


def test_slices_rolls_and_array_unknowns_trace(cpu_mod):
    """examples/heat_tmax and examples/infer_constant: rows imposed by concatenate, mod.roll of the
    unknown, an `Array` of constants inside the stencil, a residual without its first row and a
    single-point output -- slices become stencil offsets of the reads plus a window of the grid."""
    sys.path.insert(0, os.path.join(ROOT, "examples", "heat_tmax"))
    sys.path.insert(0, os.path.join(ROOT, "examples", "infer_constant"))
    import heat_tmax
    import infer_constant

    problem, state = infer_constant.make_problem(infer_constant.parse_args(["--Nt", "8", "--Nx", "16"]))
    tro = stencil_jit.TracedOperator(problem, state)
    assert tro.cg.out_lens == [(8, 16)] and tro.cg.out_count == [128]  # fu[1:] of the (9, 16) grid
    assert tro.cg.arrays == [("coeff", 3)] and len(tro.cg.pg_decl) == 3
    shifts = sorted(n.attr[1] for n in tro.cg.cots)
    assert shifts == [(0, -1), (0, 0), (0, 1), (1, -1), (1, 0), (1, 1)]  # roll by -(s) then [1:]: offsets s + (1, 0)
    problem, state = heat_tmax.make_problem(heat_tmax.parse_args(["--Nt", "8", "--Nx", "16"]))
    tro = stencil_jit.TracedOperator(problem, state)
    assert tro.cg.out_lens == [None, (1, 1)] and tro.cg.out_count == [9 * 16, 1]
    assert "inbox1" in tro.source and "AP(0, 0)" in tro.source


@pytest.mark.parametrize("shape,vw,mb", [((9, 16, 16, 16), 4, 0.004), ((9, 16, 16, 16), 1, 0.004), ((5, 12, 8), 4, 0.0002),
                                         ((9, 16, 16, 16), 4, 5), ((33, 36, 8), 1, 0.001)])
def test_chunked_traversal_is_a_permutation(cpu_mod, shape, vw, mb, monkeypatch):
    """_Codegen._chunk_remap: the generated index arithmetic (chunks of axis 1 outermost, then axis 0, then axis 1 within
    the chunk) visits every point once, walks axis 0 inside a chunk, and is the identity when a level fits the limit."""
    import re

    import veltracer

    from odil_amd import stencil_codegen

    monkeypatch.setenv("ODIL_TRACE_CHUNK_MB", str(mb))
    problem, state = veltracer.make_problem(veltracer.parse_args(["--Nx", "16", "--Nt", "8"]))
    cg = stencil_jit.TracedOperator(problem, state).cg  # (float32: 4-byte elements)
    S = []
    cg._chunk_remap(S, list(shape), vw, "raw", "flat")
    total = int(np.prod(shape)) // vw
    raw = np.arange(total)
    env = dict(raw=raw)
    for line in S:  # "const int a = e, b = e;": C's integer division of non-negative values is Python's //
        for name, expr in re.findall(r"(\w+) = ([^,;]+)", line.replace("const int ", "")):
            env[name] = eval(expr.replace("/", "//"), dict(), env)
    flat = env["flat"]
    assert sorted(flat.tolist()) == list(range(total))
    row = int(np.prod(shape[2:]))
    c1 = max([c for c in range(1, shape[1] + 1) if shape[1] % c == 0 and c * row * 4 <= mb * (1 << 20)], default=0)
    if c1 == 0 or c1 >= shape[1]:
        assert np.array_equal(flat, raw)
    else:
        per = c1 * row // vw  # threads of one chunk of one level
        assert np.array_equal(flat[:per], raw[:per])  # the first chunk of level 0 ...
        assert flat[per] == shape[1] * row // vw      # ... is followed by the first chunk of level 1


def test_outputs_in_parameter_space_are_taped(cpu_mod):
    """heat --kwreg (reference examples/heat/heat.py:131-136): `(stop_gradient(ww) - ww) * k` on the concatenated
    network weights is an output in PARAMETER space.  The operator stays on the traced path: the torch operations on the
    parameter arrays are taped during tracing (odil_amd/param_tape.py) and replayed on the current arrays at every
    evaluation, with the host scalars of the current epoch."""
    import torch

    import heat

    args = heat.parse_args(["--Nx", "16", "--Nt", "8", "--infer_k", "1", "--kwreg", "0.3", "--kwregdecay", "100"])
    problem, state = heat.make_problem(args)
    problem.tracers["epoch"] = 50
    tro = stencil_jit.TracedOperator(problem, state)
    assert tro.names == ["fu", "wreg"] and [k for k, _, _ in tro.offgrid] == [1] and len(tro.raw) == 1
    one = torch.tensor(1.0, dtype=torch.float64)
    tro.gflat.zero_()
    loss, terms, norms = tro._eval_offgrid(state, one, [one], [one])
    assert float(loss) == 1.0 and [float(t) for t in terms] == [1.0, 0.0] and float(norms[1]) == 0.0
    assert float(tro.gflat.abs().max()) == 0.0  # value and gradient of this regulariser vanish identically

    # a weight decay: value k w, gradient of its mean square 2 k^2 w / n, k a function of the epoch tracer
    base = problem.operator

    def operator(ctx):
        ww = ctx.domain.arrays_from_field(ctx.state.fields["k_net"])
        flat = ctx.mod.concatenate([ctx.mod.flatten(w) for w in ww], axis=0)
        return base(ctx) + [("wdecay", flat * (0.5 * 0.5 ** (ctx.tracers["epoch"] / 50)))]

    problem2 = odil.Problem(operator, problem.domain, problem.extra)
    problem2.tracers["epoch"] = 50
    tro = stencil_jit.TracedOperator(problem2, state)
    assert tro.names == ["fu", "wreg", "wdecay"] and [k for k, _, _ in tro.offgrid] == [1, 2]
    arrays = problem.domain.arrays_from_state(state)
    idx = [i for key, kind, pos, n in tro.layout if kind == "net" for i in range(pos, pos + n)]
    w = torch.cat([arrays[i].reshape(-1) for i in idx]).double()
    tro.gflat.zero_()
    loss, terms, norms = tro._eval_offgrid(state, one, [one], [one])
    k = 0.25
    assert abs(float(terms[2]) - float((k * w).square().mean())) < 1e-6 * float(terms[2])
    got = torch.cat([tro.gviews[i].reshape(-1) for i in idx]).double()
    assert float((got - 2 * k * k * w / w.numel()).abs().max()) < 1e-6 * float(got.abs().max())
    problem2.tracers["epoch"] = 100  # the host scalar follows the tracer without re-tracing
    tro.gflat.zero_()
    _, terms, _ = tro._eval_offgrid(state, one, [one], [one])
    assert abs(float(terms[2]) - float((0.125 * w).square().mean())) < 1e-6 * float(terms[2])


def test_parameter_space_outputs_have_an_elementwise_form(cpu_mod):
    """odil_amd/param_expr.py turns the taped torch operations of a parameter-space output into segments of elementwise
    expressions over the parameter arrays (what the generated `k_par` evaluates): interpreted here with NumPy, the
    segments reproduce the values the torch replay gives; operations without such a form are refused."""
    import torch

    import heat
    from odil_amd import param_expr

    args = heat.parse_args(["--Nx", "16", "--Nt", "8", "--infer_k", "1", "--kwreg", "0.3", "--kwregdecay", "100", "--double", "1"])
    problem, state = heat.make_problem(args)
    base = problem.operator

    def operator(ctx):
        m = ctx.mod
        ww = ctx.domain.arrays_from_field(ctx.state.fields["k_net"])
        flat = m.concatenate([m.flatten(w) for w in ww], axis=0)
        k = 0.5 * 0.5 ** (ctx.tracers["epoch"] / 50)
        return base(ctx) + [("wdecay", flat * k), ("bias0", (ww[-1] - 0.25) / (1 + k) + 2 * ww[-1])]

    problem2 = odil.Problem(operator, problem.domain, problem.extra)
    problem2.tracers["epoch"] = 50
    tro = stencil_jit.TracedOperator(problem2, state)
    assert tro.par_outputs is not None and [k for k, _ in tro.par_outputs] == [1, 2, 3]
    assert tro.cg.par_index and "k_par" in tro.source and "jit_par" in tro.source
    arrays = [a.detach().double().numpy() for a in problem.domain.arrays_from_state(state)]
    memo = dict()

    def ev(e, j):
        if e[0] == "atom":
            return arrays[e[1]].reshape(-1)[e[2] + j]
        if e[0] == "const":
            return e[1]
        if e[0] == "hs":
            return float(tro._host_value(e[1], memo))
        if e[0] == "neg":
            return -ev(e[1], j)
        a, b = ev(e[1], j), ev(e[2], j)
        return a + b if e[0] == "add" else a - b if e[0] == "sub" else a * b if e[0] == "mul" else a / b

    one = torch.tensor(1.0, dtype=torch.float64)
    tro.gflat.zero_()
    _, terms, _ = tro._eval_offgrid(state, one, [one], [one])  # the torch replay
    for q, (k, segs) in enumerate(tro.par_outputs):
        values = np.concatenate([np.array([ev(e, j) for j in range(n)]) for n, e in segs])
        assert abs(float(np.mean(values**2)) - float(terms[k])) <= 1e-12 * max(float(terms[k]), 1e-30), (k, float(terms[k]))
    # a reduction of a parameter array has no elementwise form: the torch replay keeps such outputs
    def operator3(ctx):
        ww = ctx.domain.arrays_from_field(ctx.state.fields["k_net"])
        return base(ctx) + [("wsum", ww[0].sum().reshape(1) * 0.1)]

    tro3 = stencil_jit.TracedOperator(odil.Problem(operator3, problem.domain, problem.extra, tracers={"epoch": 0}), state)
    assert tro3.offgrid and tro3.par_outputs is None


def test_parameter_space_expressions_that_broadcast_or_interleave_keep_the_torch_replay(cpu_mod):
    """The segment form of param_expr.py only says how elements lie end to end, so it must refuse what torch evaluates in
    another order: an outer product by broadcasting (n^2 elements, not n), a concatenation of 2-D pieces along the last
    axis (rows interleave), a tensor operand that is not part of the tape.  Such outputs stay with the torch replay
    (`_eval_offgrid`), which is exact -- and whose value for the accepted forms equals the segments' (cross-check)."""
    import torch

    from odil_amd import param_expr

    domain = odil.Domain(cshape=(8, 8), dtype=np.float64)
    state = odil.State(fields={"u": odil.Field(None, loc="cc"), "p": odil.Array(np.array([1.0, -2.0, 0.5, 3.0])),
                               "q": odil.Array(np.arange(6.0).reshape(2, 3) + 1)})
    state = domain.init_state(state)
    par = lambda ctx, key: ctx.domain.arrays_from_field(ctx.state.fields[key])[0]  # noqa: E731
    cases = {
        # name: (operator tail, converts to a kernel?)
        "outer product": (lambda ctx: par(ctx, "p").reshape(-1, 1) * par(ctx, "p").reshape(1, -1), False),
        "row-interleaving cat": (lambda ctx: ctx.mod.concatenate([par(ctx, "q"), 2 * par(ctx, "q")], axis=-1), False),
        "column plus row": (lambda ctx: par(ctx, "q") + par(ctx, "q")[0], None),  # indexing: not on the supported list
        "tensor operand": (lambda ctx: par(ctx, "p") * torch.tensor(2.0, dtype=torch.float64), False),
        "flatten + cat(axis=0)": (lambda ctx: ctx.mod.concatenate([ctx.mod.flatten(par(ctx, "q")), par(ctx, "p")], axis=0) * 0.5, True),
        "cat of 2-D pieces along axis 0": (lambda ctx: ctx.mod.concatenate([par(ctx, "q"), 2 * par(ctx, "q")], axis=0), True),
        "same shapes": (lambda ctx: par(ctx, "q") * par(ctx, "q") - 1.0, True),
    }
    one = torch.tensor(1.0, dtype=torch.float64)
    for name, (tail, converts) in cases.items():
        problem = odil.Problem(lambda ctx, tail=tail: [("fu", ctx.field("u") - 1.0), ("par", tail(ctx))], domain)
        tro = stencil_jit.TracedOperator(problem, state)
        assert tro.offgrid, name
        if converts is not None:
            assert (tro.par_outputs is not None) == converts, name
        # the value the evaluation will use is torch's in every case
        arrays = {"p": state.fields["p"].array.double().numpy(), "q": state.fields["q"].array.double().numpy()}
        want = {"outer product": np.outer(arrays["p"], arrays["p"]),
                "row-interleaving cat": np.concatenate([arrays["q"], 2 * arrays["q"]], axis=-1),
                "column plus row": arrays["q"] + arrays["q"][0], "tensor operand": 2 * arrays["p"],
                "flatten + cat(axis=0)": 0.5 * np.concatenate([arrays["q"].reshape(-1), arrays["p"]]),
                "cat of 2-D pieces along axis 0": np.concatenate([arrays["q"], 2 * arrays["q"]], axis=0),
                "same shapes": arrays["q"] ** 2 - 1.0}[name]
        tro.gflat.zero_()
        _, terms, _ = tro._eval_offgrid(state, one, [one], [one])
        assert abs(float(terms[1]) - float(np.mean(want**2))) <= 1e-13 * float(np.mean(want**2)), name
        if tro.par_outputs is not None:  # ... and the generated kernel's segments give the same elements in the same order
            flat = [a.detach().double().numpy().reshape(-1) for a in domain.arrays_from_state(state)]

            def ev(e, j):
                if e[0] == "atom":
                    return flat[e[1]][e[2] + j]
                if e[0] == "const":
                    return e[1]
                if e[0] == "neg":
                    return -ev(e[1], j)
                a, b = ev(e[1], j), ev(e[2], j)
                return a + b if e[0] == "add" else a - b if e[0] == "sub" else a * b if e[0] == "mul" else a / b

            (k, segs), = tro.par_outputs
            values = np.concatenate([np.array([ev(e, j) for j in range(n)]) for n, e in segs])
            assert values.shape == want.reshape(-1).shape and np.allclose(values, want.reshape(-1), rtol=1e-15, atol=0), name


def test_offgrid_only_parameters_do_not_accumulate(cpu_mod):
    """An `Array` that appears ONLY in a parameter-space output (a prior) gets its gradient from the tape replay alone:
    no grid kernel rewrites its slot of the packed gradient, so repeated evaluations must SET it, not add to it."""
    import torch

    domain = odil.Domain(cshape=(8, 8), dtype=np.float64)
    state = odil.State(fields={"u": odil.Field(None, loc="cc"), "p": odil.Array(np.array([1.0, -2.0, 0.5]))})
    state = domain.init_state(state)

    def operator(ctx):
        p = ctx.domain.arrays_from_field(ctx.state.fields["p"])[0]
        return [("fu", ctx.field("u") - 1.0), ("prior", p * 2.0)]

    problem = odil.Problem(operator, domain)
    tro = stencil_jit.TracedOperator(problem, state)
    assert [k for k, _, _ in tro.offgrid] == [1] and "p" not in tro.cg.pgrads
    pidx = [pos for key, kind, pos, n in tro.layout if key == "p"][0]
    one = torch.tensor(1.0, dtype=torch.float64)
    want = 2 * 2.0 * 2.0 * np.array([1.0, -2.0, 0.5]) / 3  # d mean((2 p)^2) / d p
    for _ in range(3):
        tro._eval_offgrid(state, one, [one], [one])
        assert np.allclose(tro.gviews[pidx].numpy(), want, rtol=1e-14), tro.gviews[pidx]


def test_untraceable_operators_are_refused(cpu_mod):
    domain = odil.Domain(cshape=(8, 8), dtype=np.float64)
    state = odil.State()
    state.fields["u"] = odil.Field(None, loc="cc")
    state.fields["p"] = odil.Array(np.zeros(3))
    state = domain.init_state(state)
    cases = {
        "strided slice": lambda ctx: [ctx.field("u")[::2]],
        "mixed windows": lambda ctx: [ctx.field("u")[1:] + ctx.field("u")],
        "reduction": lambda ctx: [ctx.field("u") - ctx.mod.mean(ctx.field("u"))],
        "control flow": lambda ctx: [ctx.field("u") if ctx.field("u") > 0 else ctx.field("u", 1, 0)],
        "whole Array arithmetic": lambda ctx: [ctx.field("u") * ctx.field("p")],
        "state bypass": lambda ctx: [ctx.field("u") * 0 + ctx.state.fields["u"].array],
        "scalar output": lambda ctx: [ctx.field("u"), ctx.tracers["epoch"] * 2.0],
    }
    for name, op in cases.items():
        with pytest.raises(stencil_jit.TraceUnsupported):
            stencil_jit.TracedOperator(odil.Problem(op, domain), state)
        assert stencil_jit.trace(odil.Problem(op, domain), state) is None, name


def test_marching_kernel_structure(cpu_mod, monkeypatch):
    """The code generator's round-4 machinery on heat with two space dimensions (float): index predicates that fold in an
    interior copy of the body and their exceptional index values, the three copies of the marching kernel's body behind
    scalar branches, the in-kernel sums of the read cotangents (one array per leading shift + edge arrays) and the final
    gather that adds the edges.  (Numerics: tests/test_workloads_gpu.py, test_fullsize_traced_gpu.py.)"""
    import heat2d

    monkeypatch.setenv("ODIL_TRACE_SHARE", "march")
    problem, state = heat2d.make_problem(heat2d.parse_args(["--Nt", "8", "--Nx", "16", "--Ny", "128", "--infer_k", "1",
                                                              "--imposed", "stripe", "--multigrid", "0"]))
    tro = stencil_jit.TracedOperator(problem, state)
    cg = tro.cg
    assert cg.share_mode == "march" and sorted(axis for _, _, axis in cg.share) == [1, 2]
    fold, exc, _ = cg._fold_plan(cg.order, 1, windows=True)
    assert exc == {0: [0], 1: [0, 15], 2: [0, 127]} and fold and all(isinstance(v, bool) for v in fold.values())
    _, exc_w, _ = cg._fold_plan(cg.order, 4, windows=True)  # four points per thread / wall strips: the last axis stays
    assert exc_w == {0: [0], 1: [0, 15]}
    assert cg._interior_cond({1: [0, 15], 2: [0, 64, 127]}) == "i1 >= 1 && i1 <= 14 && i2 >= 1 && i2 <= 126 && i2 != 64"
    live = cg._live_under(fold)
    assert live < {n.idx for n in cg.order}  # the wall / initial-row branches are dead in the interior
    src = tro.source
    k_fwd = src[src.index("void k_fwd("):src.index("void k_final(")]
    assert k_fwd.count("if (interior_ && ") == 1 and k_fwd.count("} else if (lead_ok_ && ") == 1  # three copies of the body
    assert "if (!(lead_ok_ && " in k_fwd  # the initial-row arrays only the general copy reads: fetched when the next row takes it
    assert "odil_readlane(kay0" in k_fwd and "odil_lane_prev(" in k_fwd and "odil_lane_next(" in k_fwd
    assert cg.ncot == 2 and cg.edge_numel > 0 and "a.edge[" in k_fwd  # u(t) and u(t - 1): one array each + edges
    gat = src[src.index("void k_gat_0("):]
    assert "a.edge + " in gat and "c % 64 == 63" in gat and "adam_apply4" in gat
    # a forward kernel WITHOUT a network gets the interior copy of its body; with one it does not (registers)
    import veltracer

    tro2 = stencil_jit.TracedOperator(*veltracer.make_problem(veltracer.parse_args(["--Nt", "8", "--Nx", "16", "--Ny", "16"])))
    assert "__all((int)(" in tro2.source
    monkeypatch.setenv("ODIL_TRACE_SHARE", "0")
    tro3 = stencil_jit.TracedOperator(*heat2d.make_problem(heat2d.parse_args(["--Nt", "8", "--Nx", "16", "--Ny", "32", "--infer_k", "1"])))
    k3 = tro3.source[tro3.source.index("void k_fwd("):tro3.source.index("void k_final(")]
    assert "__all((int)(" not in k3


def test_outputs_on_grids_of_different_shapes_become_one_kernel_set_per_shape(cpu_mod):
    """A field per location (reference examples/basic/fields.py:16-40) plus a second output on one of the grids: the
    outputs are grouped by shape in order of first appearance, every group is traced over the whole state."""
    domain = odil.Domain(cshape=(8, 4), dtype=np.float64, multigrid=True)
    state = odil.State()
    for key, loc in (("a", "cc"), ("b", "nn"), ("c", "nc")):
        state.fields[key] = odil.Field(None, loc=loc)
    state = domain.init_state(state)

    def op(ctx):
        x, y = ctx.points(loc="nn")
        a, b, c = ctx.field("a"), ctx.field("b"), ctx.field("c")
        return [("fa", a - ctx.field("a", frozen=True) * 0.5), ("fb", b - x), ("fc", c * c), ("fa2", (ctx.field("a", 1, 0) - a) * 2.0),
                ("fb2", b * y)]

    with pytest.raises(stencil_jit.TraceGroups) as e:
        stencil_jit.TracedOperator(odil.Problem(op, domain), state)
    assert e.value.groups == [[0, 3], [1, 4], [2]]
    traced = stencil_jit.trace(odil.Problem(op, domain), state)
    assert isinstance(traced, stencil_jit.TracedGroups) and not traced.graph_ok
    assert traced.names == ["fa", "fb", "fc", "fa2", "fb2"]
    assert [tuple(p.G) for p in traced.parts] == [(8, 4), (9, 5), (9, 4)]
    assert [p.names for p in traced.parts] == [["fa", "fa2"], ["fb", "fb2"], ["fc"]]
    # an output that mixes two grids stays untraceable
    bad = lambda ctx: [ctx.field("a"), ctx.field("b")[1:, 1:] + ctx.field("a")]
    assert stencil_jit.trace(odil.Problem(bad, domain), state) is None
