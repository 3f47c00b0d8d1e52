"""Operator tracer and HIP code generator: one fused residual + cotangent kernel per user operator.

The reference hands the user's `operator(ctx)` to XLA / TF-function, which fuse its pointwise
arithmetic with the shifted reads (reference src/odil/core.py:1038-1111).  The counterpart here
(SURVEY §8 F1): the operator runs ONCE on symbolic values -- it is straight-line `mod` code over
`ctx.field(key, *shift)` reads, index masks, arrays from `extra`, scalars and tracers -- and the
recorded expression DAG is emitted as HIP source for gfx950:

  k_fwd      one thread per grid point: loads every distinct (key, shift, loc) read once
             (periodic wrap, 'c'<->'n' pad / trim as in `Context.field`, core.py:910-975),
             evaluates all outputs, accumulates sum f_k^2 per output (deterministic two-stage
             reduction), then runs reverse-mode differentiation of the DAG in registers, seeded
             with 2 f_k / n_k, and stores one cotangent array per live read; parameter gradients
             of pointwise neural nets (core.py:807-862) are reduced per workgroup;
  k_gat_<f>  per unknown field: g[j] = sum over its reads of cot_r[j - shift_r] (the transpose of
             the gather, in gather form: no atomics, fixed summation order);
  k_final    sums the per-workgroup partials in fixed order: terms, loss, norms, net gradients.

The multigrid synthesis before and P^T (+ Adam) after are the hand-written kernels of
libodil_hip.so.  Everything the tracer cannot express (reductions, slicing of symbolic values,
host control flow on device data, `Array` unknowns) raises TraceUnsupported and the problem keeps
using the generic autograd path.  The source is compiled with hipcc into an in-tree cache
(odil_amd/_jit_cache, keyed by the source hash) and loaded with ctypes.
"""

import ctypes
import os

import torch

from . import ops
from .stencil_codegen import _Codegen, _compile
from .stencil_trace import (  # noqa: F401  (re-exported: the public names of the tracer)
    _HOST_BINARY,
    _HOST_UNARY,
    _R,
    ModTrace,
    ParamArray,
    Sym,
    TraceContext,
    Tracer,
    TraceUnsupported,
)


# ======================================================================================
# Traced evaluator
# ======================================================================================
class TraceGroups(TraceUnsupported):
    """The operator's outputs live on grids of DIFFERENT shapes (a field per location: cell centres, nodes, faces --
    reference examples/basic/fields.py:16-40): no single kernel set covers them, one per shape does (`TracedGroups`).
    `groups`: positions of the outputs, grouped by shape, in order of first appearance."""

    def __init__(self, groups):
        super().__init__("outputs on {} grids of different shapes".format(len(groups)))
        self.groups = groups


def trace_outputs(problem, state, only=None):
    """Runs `problem.operator` once on symbolic values: (tracer, output nodes, raw flags, names, grid shape).
    Only the STRUCTURE of `state` matters (array shapes; 'meta' tensors do).  `only`: positions of the outputs to keep
    (one shape group of an operator whose outputs have several, see TraceGroups)."""
    from .core import Context, Problem

    domain = problem.domain
    tr = Tracer(domain)
    # unknowns reached around ctx.field / ctx.neural_net would lose their gradient: trace on
    # differentiable leaves so that `lift` can refuse them
    leaves = [a.detach().requires_grad_(True) for a in domain.arrays_from_state(state)]
    # ... except whole PARAMETER arrays (network weights, `Array`s): operations on them are taped, and an output built
    # from them alone is evaluated by replaying the tape (param_tape.py)
    from .core import Array, NeuralNet
    from .param_tape import OffGrid, ParamTape, ParamTensor

    tape, pos = ParamTape(), 0
    for field in state.fields.values():
        n = len(domain.arrays_from_field(field))
        if isinstance(field, (NeuralNet, Array)):
            for k in range(pos, pos + n):
                if not leaves[k].is_meta:
                    leaves[k] = tape.leaf(leaves[k], k)
        pos += n
    ctx = TraceContext(tr, problem._shadow_state(state, leaves), problem.extra, problem.tracers)
    try:
        with torch.enable_grad():
            res = problem.operator(ctx)
    except TraceUnsupported:
        raise
    except Exception as e:
        # code that is not written against `ctx.mod` (torch / NumPy calls on the symbols, helper
        # kernels of this package): the eager path runs it, and reports genuine errors
        raise TraceUnsupported("{} under tracing: {}".format(type(e).__name__, str(e).splitlines()[0] if str(e) else ""))
    names, values = Problem._split_outputs(res)
    # outputs in parameter space: [(position among the outputs, expression)]
    tr.offgrid = [(k, OffGrid.of(v)) for k, v in enumerate(values) if isinstance(v, (OffGrid, ParamTensor))]
    tr.param_tape = tape
    values = [v for v in values if not isinstance(v, (OffGrid, ParamTensor))]
    if not values:
        raise TraceUnsupported("operator has no output on the grid")
    raw = [isinstance(v, Context.Raw) for v in values]
    outs = [tr.lift(v.value if r else v) for v, r in zip(values, raw)]
    if not any(n.op == "read" for n in tr.nodes):
        raise TraceUnsupported("operator reads no field")
    G = tuple(tr.grid_shape())
    if only is not None:
        assert not tr.offgrid
        names, raw, outs = [names[k] for k in only], [raw[k] for k in only], [outs[k] for k in only]
        G = tuple(outs[0].shape)
    elif not tr.offgrid and all(not o.host and o.win is None for o in outs) and len({tuple(o.shape) for o in outs}) > 1:
        shapes = []
        for o in outs:
            if tuple(o.shape) not in shapes:
                shapes.append(tuple(o.shape))
        raise TraceGroups([[k for k, o in enumerate(outs) if tuple(o.shape) == shape] for shape in shapes])
    for o in outs:  # every output lives on the grid of the reads, all of it or a window of it
        if o.host or (o.win is None and tuple(o.shape) != G):
            raise TraceUnsupported("output of shape {} on grid {}".format(tuple(o.shape), G))
    outs = [o if o.kind == _R else tr.unary("cast", o) for o in outs]
    return tr, outs, raw, names, G


class TracedOperator:
    """loss / gradient of one user operator through its generated kernels."""

    def __init__(self, problem, state, only=None, jac=False):
        """jac: also generate `k_jac`, the Jacobian coefficient arrays of `Problem.eval_operator_grad` (a library of its own:
        the Newton driver asks for it, the gradient optimizers never pay for it)."""
        from .core import Field, MultigridField

        domain = problem.domain
        self.problem, self.domain = problem, domain
        tr, outs, raw, self.names, G = trace_outputs(problem, state, only)
        self.G, self.raw = G, raw
        # outputs in parameter space (param_tape.py): [(position, expression, slice of the tape it needs)]
        self.offgrid = [(k, e, tr.param_tape.slice_for(e.param_ids())) for k, e in tr.offgrid]
        self.param_tape = tr.param_tape
        cg = _Codegen(tr, outs, raw, G, state)
        cg.want_jac = bool(jac)
        # ... as ONE generated kernel when the taped operations have an elementwise form (param_expr.py); else the torch
        # replay of _eval_offgrid
        self.par_outputs = None
        if self.offgrid:
            from . import param_expr
            from .core import Array, NeuralNet

            arrays0 = domain.arrays_from_state(state)
            try:
                self.par_outputs = param_expr.convert(tr.param_tape, self.offgrid, {i: int(a.numel()) for i, a in enumerate(arrays0)})
            except param_expr.Unsupported as e:
                from .util import printlog

                printlog("odil_amd: parameter-space output evaluated by torch ({})".format(e))
            if self.par_outputs is not None:
                cg.par_outputs, cg.par_numel, cg.par_keys, pos = self.par_outputs, dict(), dict(), 0
                for key, field in state.fields.items():
                    n = len(domain.arrays_from_field(field))
                    for i in range(pos, pos + n):
                        cg.par_numel[i] = int(arrays0[i].numel())
                        if isinstance(field, (NeuralNet, Array)):
                            cg.par_keys[i] = key
                    pos += n
        self.source = cg.source()
        self.lib, self.lib_path = _compile(self.source, cg.flags)
        self.cg, self.tr = cg, tr
        self.tracer_keys = [n.attr for n in tr.nodes if n.op == "tracer"]
        dev, dt = domain.mod.device, tr.torch_dtype
        self.total = cg.total
        # Every block ends with one block reduction per output and per network / array parameter: operators
        # that differentiate through parameters (dozens of reductions) want few, long blocks -- heat with two
        # space dimensions (46 parameters, 67 M points): 6.6 ms / epoch at 65536 blocks, 5.2 at 4096; plain
        # stencils prefer many (tracer 4-D: 62.0 ms at 65536, 64.5 at 4096).
        cap = cg.max_blocks or (4096 if len(cg.pg_decl) > 8 else 65536)
        self.nblocks = min((self.total // cg.vw_fwd + 255) // 256, cap)
        nout = len(outs)
        self.cot = [torch.empty(G, dtype=dt, device=dev) for _ in range(cg.ncot)]
        self.part = torch.empty(max(1, nout * self.nblocks), dtype=dt, device=dev)
        self.ppart = torch.empty(max(1, len(cg.pg_decl) * self.nblocks), dtype=dt, device=dev)
        self.out = torch.zeros(1 + 2 * nout, dtype=dt, device=dev)
        self.pgrad = torch.zeros(max(1, len(cg.pg_decl)), dtype=dt, device=dev)
        par_arrays = cg.par_arrays

        class Args(ctypes.Structure):
            _fields_ = [
                ("src", ctypes.c_void_p * max(1, len(cg.src_keys))),
                ("ten", ctypes.c_void_p * max(1, len(tr.tensors))),
                ("cot", ctypes.c_void_p * max(1, cg.ncot)),
                ("par", ctypes.c_void_p * max(1, par_arrays)),
                ("hs", ctypes.c_void_p), ("hsv", ctypes.c_double * max(1, len(cg.hs))),
                ("part", ctypes.c_void_p), ("ppart", ctypes.c_void_p), ("part2", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("pgrad", ctypes.c_void_p), ("edge", ctypes.c_void_p), ("nblocks", ctypes.c_int),
            ]

        self.args = Args()
        for i, t in enumerate(tr.tensors):
            self.args.ten[i] = t.data_ptr()
        for i, t in enumerate(self.cot):
            self.args.cot[i] = t.data_ptr()
        self.part2 = torch.zeros(16 * (nout + len(cg.pg_decl)), dtype=dt, device=dev)
        self.args.part, self.args.ppart = self.part.data_ptr(), self.ppart.data_ptr()
        self.args.part2 = self.part2.data_ptr()
        self.args.out, self.args.pgrad = self.out.data_ptr(), self.pgrad.data_ptr()
        # marching kernels: what their in-kernel sums of read cotangents hand across segments of rows / strips of columns
        self.edge = torch.zeros(max(1, getattr(cg, "edge_numel", 0)), dtype=dt, device=dev)
        self.args.edge = self.edge.data_ptr()
        self.args.nblocks = self.nblocks
        self.args.hs = None
        self._hs_rows = None  # graph replay: (pinned table, device table, device row, device row index)
        self.lib.jit_fwd.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        if self.par_outputs is not None:
            npar = max(1, len(cg.par_index))

            class ParArgs(ctypes.Structure):
                _fields_ = [("val", ctypes.c_void_p * npar), ("grad", ctypes.c_void_p * npar), ("pout", ctypes.c_void_p)]

            self.par_args = ParArgs()
            self.pout = torch.zeros(2 * len(self.par_outputs), dtype=dt, device=dev)
            self.par_args.pout = self.pout.data_ptr()
            self.lib.jit_par.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self.lib.jit_gather.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self.lib.jit_gather_adam.argtypes = [ctypes.c_int, ctypes.c_void_p] + [ctypes.c_void_p] * 4 + [ctypes.c_double] * 4 + [
            ctypes.c_void_p, ctypes.c_void_p]
        if getattr(cg, "jac_items", None):
            self.lib.jit_jac.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]
        if cg.merged:
            self.lib.jit_gather_all.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_void_p)] * 4 + [
                ctypes.c_double] * 4 + [ctypes.c_void_p, ctypes.c_void_p]
        # structure of the state: which arrays belong to which field
        self.layout = []
        pos = 0
        for key, field in state.fields.items():
            n = len(domain.arrays_from_field(field))
            kind = ("field" if isinstance(field, Field) else "mg" if isinstance(field, MultigridField)
                    else "net")  # NeuralNet or Array: a few parameters, gradients reduced over the grid
            self.layout.append((key, kind, pos, n))
            pos += n
        self.signature = self._signature(state)
        # gradients live in ONE packed buffer in `arrays_from_state` order (what the optimizers
        # want: no per-array copies); kernels write straight into its views
        from .optimizer import pack_like

        arrays = domain.arrays_from_state(state)
        self.gflat, self.gviews = pack_like(arrays)
        self.gflat.zero_()
        self.gtmp = dict()  # regular-array gradients that cannot alias level 0 (scaled multigrid terms)
        self.mg_meta = dict()
        for key, kind, pos, n in self.layout:
            field = state.fields[key]
            alias = kind == "field"
            if kind == "mg":
                factors = field.factors or domain.mg_factors or [1] * n
                trivial = all(float(f) == 1.0 for f in factors)
                self.mg_meta[key] = (None if trivial else tuple(float(f) for f in factors), domain._mg_loc(field),
                                     [tuple(a.shape) for a in arrays[pos:pos + n]])
                alias = trivial
            if kind not in ("field", "mg"):
                continue
            if key in cg.direct:
                if alias:
                    self.cot[cg.direct[key]] = self.gviews[pos]
                    self.args.cot[cg.direct[key]] = self.gviews[pos].data_ptr()
            elif key in cg.gathers and not alias:
                self.gtmp[key] = torch.empty(cg._field_shape(key), dtype=dt, device=dev)
        nets = [(key, pos) for key, kind, pos, n in self.layout if kind == "net" and key in cg.pgrads]
        self.pgrad_direct = len(nets) == 1 and len(cg.pgrads) == 1
        if self.pgrad_direct:
            self.args.pgrad = self.gviews[nets[0][1]].data_ptr()

    def _signature(self, state):
        return tuple((k, type(f).__name__, tuple(tuple(a.shape) for a in self.domain.arrays_from_field(f)))
                     for k, f in state.fields.items())

    def matches(self, state):
        return self._signature(state) == self.signature

    # ---- host scalars -----------------------------------------------------------------------
    def _host_value(self, n, memo):
        if n.idx in memo:
            return memo[n.idx]
        if n.op == "const":
            v = n.attr
        elif n.op == "tracer":
            v = self.problem.tracers[n.attr]
        elif n.op == "where":
            c, a, b = (self._host_value(x, memo) for x in n.args)
            v = a if c else b
        elif len(n.args) == 1:
            v = _HOST_UNARY[n.op](self._host_value(n.args[0], memo))
        else:
            v = _HOST_BINARY[n.op](self._host_value(n.args[0], memo), self._host_value(n.args[1], memo))
        memo[n.idx] = v
        return v

    def host_scalars(self):
        """Host scalars of the trace: functions of `problem.tracers`, evaluated in Python double as the
        operator itself would."""
        memo = dict()
        return [float(self._host_value(n, memo)) for n in self.cg.hs]

    def refresh_host_scalars(self):
        """Current host scalars -> the argument struct.  Every eager launch copies the struct, so epochs
        queued behind each other keep their own values however far the host runs ahead."""
        for i, v in enumerate(self.host_scalars()):
            self.args.hsv[i] = v

    # Epochs replayed as a hipGraph (optimizer._EpochGraph) cannot take new kernel arguments: the captured
    # launch reads row k of a device table, k a device counter advanced by the graph itself.  The host fills
    # row k of a PINNED table of the same shape before replay k and queues its copy; no row is ever rewritten,
    # so replays queued far ahead of the GPU cannot see each other's values (the single staging buffer this
    # replaces could).
    def graph_begin(self, nrows):
        nhs = len(self.cg.hs)
        if not nhs:
            return
        dev = self.out.device
        pinned = torch.zeros((nrows, nhs), dtype=torch.float64)
        if dev.type == "cuda":
            pinned = pinned.pin_memory()
        self._hs_rows = dict(pinned=pinned, table=torch.zeros((nrows, nhs), dtype=torch.float64, device=dev),
                             row=torch.zeros((1, nhs), dtype=torch.float64, device=dev),
                             index=torch.zeros(1, dtype=torch.int64, device=dev), next=0)

    def graph_upload(self):
        """Before replay k: this epoch's host scalars -> row k (asynchronous copy on the replay's stream)."""
        rows = self._hs_rows
        if rows is None:
            return
        k = rows["next"]
        if k >= rows["pinned"].shape[0]:
            raise RuntimeError("more graph replays than rows of host scalars")
        rows["pinned"][k] = torch.tensor(self.host_scalars(), dtype=torch.float64)
        rows["table"][k].copy_(rows["pinned"][k], non_blocking=True)
        rows["next"] = k + 1

    def graph_end(self):
        self._hs_rows = None

    def _side_streams(self, nfields):
        """Streams for per-field chains; none for a single field, for fields beyond 64 MB (their kernels
        fill the GPU on their own and concurrent streams only fight for HBM and allocator pools:
        veltracer3d 83 -> 134 ms).  Measured gain where it applies: 5 %."""
        esize = 8 if self.tr.torch_dtype == torch.float64 else 4
        if nfields < 2 or self.total * esize > (64 << 20) or not torch.cuda.is_available():
            return []
        pool = self.__dict__.setdefault("_streams", [])
        while len(pool) < min(nfields, 4):
            pool.append(torch.cuda.Stream())
        return pool[: min(nfields, 4)]

    # ---- evaluation ---------------------------------------------------------------------------
    def _launch(self, state):
        keep = self._bind(state)
        rc = self.lib.jit_fwd(ctypes.byref(self.args), ops.stream_ptr())
        if rc != 0:
            raise RuntimeError("traced operator launch failed: hip error {}".format(rc))
        return keep

    def eval_operator_grad(self, state):
        """values, grads, names of `Problem.eval_operator_grad` (reference core.py:1313-1361) from the generated `k_jac`:
        per output its value array and {(key, shift, loc): d output / d read}, one launch, no autograd graph."""
        cg = self.cg
        if not getattr(cg, "jac_items", None):
            raise RuntimeError("this operator was traced without its Jacobian kernel")
        keep = self._bind(state)
        dev, dt = self.out.device, self.tr.torch_dtype
        n = len(cg.jac_items)
        # (one buffer, the arrays its leading slices in the kernel's order: a multigrid solver that wants the coefficient
        # arrays back to back -- gmg.recognise_stencil -- takes a view instead of seven copies)
        buf = torch.empty((n,) + tuple(self.G), dtype=dt, device=dev)
        arrays = [buf[j] for j in range(n)]
        ptrs = (ctypes.c_void_p * n)(*[a.data_ptr() for a in arrays])
        rc = self.lib.jit_jac(ctypes.byref(self.args), ptrs, ops.stream_ptr())
        if rc != 0:
            raise RuntimeError("traced Jacobian launch failed: hip error {}".format(rc))
        del keep
        nout = len(self.raw)
        values, grads = [None] * nout, [dict() for _ in range(nout)]
        for (k, attr), a in zip(cg.jac_items, arrays):
            if attr is None:
                values[k] = a
            else:
                key, shift, loc, _ = attr
                grads[k][(key, tuple(int(v) for v in shift), loc)] = a
        return values, grads, list(self.names)

    def _bind(self, state):
        """The argument block filled for `state`: regular arrays of the fields (multigrid syntheses launched), parameter
        pointers, host scalars.  Returns the tensors that must stay alive until the launch that follows is queued."""
        from .core import MultigridField

        domain, cg = self.domain, self.cg
        if self._signature(state) != self.signature:
            raise RuntimeError("state structure changed since the operator was traced")
        from ._lib import ptr

        ptr(self.out)  # fails loudly (OdilHipError) when the problem lives on the CPU: there is no CPU path
        rows = self._hs_rows
        if rows is not None and torch.cuda.is_current_stream_capturing():
            torch.index_select(rows["table"], 0, rows["index"], out=rows["row"])
            rows["index"].add_(1)
            self.args.hs = rows["row"].data_ptr()
        else:
            self.args.hs = None
            self.refresh_host_scalars()
        keep = []
        # the multigrid syntheses of different fields are independent chains of mostly small launches:
        # each runs on its own stream, the forward kernel waits for all of them
        cur = torch.cuda.current_stream()
        side = self._side_streams(len(cg.src_keys))
        with torch.no_grad():
            for i, key in enumerate(cg.src_keys):
                field = state.fields[key]
                if side and isinstance(field, MultigridField):
                    s_ = side[i % len(side)]
                    s_.wait_stream(cur)
                    with torch.cuda.stream(s_):
                        u = domain.get_regular_array(field).contiguous()
                    u.record_stream(cur)
                else:
                    u = domain.get_regular_array(field).contiguous()
                keep.append(u)
                self.args.src[i] = u.data_ptr()
        for s_ in side:
            cur.wait_stream(s_)
        i = 0
        for key, layers in cg.nets:
            net = state.fields[key]
            for arr in list(net.weights) + list(net.biases):
                if not arr.is_contiguous():
                    raise RuntimeError("neural net arrays must be contiguous")
                self.args.par[i] = arr.data_ptr()
                i += 1
        for key, _ in cg.arrays:
            arr = state.fields[key].array
            if not arr.is_contiguous() or arr.dtype != self.tr.torch_dtype:
                raise RuntimeError("Array unknown '{}' must be a contiguous {} tensor".format(key, self.tr.torch_dtype))
            self.args.par[i] = arr.data_ptr()
            i += 1
        return keep

    def eval_loss_grad_adam(self, state, m, v, alpha, omb1, omb2, eps):
        """eval_loss_grad with the Adam update (reference optimizer.py:316-318) of the leading grid fields applied
        inside the generated gather for the finest level (the lane that sums g[l] owns x[l], m[l], v[l]: the
        optimizer's pass over 2^d / (2^d - 1) of the unknowns and its re-read of that gradient disappear); the
        coarser levels are updated right after their transposes.  m, v: the optimizer's moment arrays in
        `arrays_from_state` order.  Returns (..., done) with `done` = how many leading arrays were updated (fields
        that follow a network / Array in the state are left to the optimizer), or None when nothing can fuse."""
        cg = self.cg
        done, plan = 0, dict()
        # small and medium grids are better off with ONE optimizer launch over the packed vector (heat 256 x 512: 0.19
        # ms per epoch against 0.22 fused; tracer 128 x 256^2, three fields on side streams: 0.66 against 0.71); the
        # fusion pays where the update is a long pass over HBM (heat 256 x 512^2: 3.63 -> 3.52, tracer 32 x 256^3: -12 %)
        # (ODIL_FUSE_ADAM_SMALL=1: the tests force the fused update on their small grids)
        if self.total < (1 << 25) and not int(os.environ.get("ODIL_FUSE_ADAM_SMALL", 0)):
            return None
        # gathers that re-evaluate local derivatives read the fields' own regular arrays: for a plain `Field` that array
        # IS the unknown, which no launch may update while another gather of this epoch still reads it
        reread = {k for keys in cg.gather_reads_sources.values() for k in keys}
        for key, kind, pos, n in self.layout:
            fusable = (kind in ("field", "mg") and key in cg.gathers and key not in self.gtmp
                       and not (kind == "field" and key in reread)
                       and int(os.environ.get("ODIL_FUSE_ADAM0", 1)))
            if not fusable or pos != done:
                break
            plan[key] = (pos, n)
            done = pos + n
        if not plan:
            return None
        arrays = self.domain.arrays_from_state(state)
        for key, (pos, n) in plan.items():
            for k in range(pos, pos + n):
                for t in (arrays[k], m[k], v[k]):
                    if not t.is_contiguous() or t.dtype != self.tr.torch_dtype:
                        return None
        res = self.eval_loss_grad(state, adam=(plan, arrays, m, v, alpha, omb1, omb2, eps))
        return res + (done,)

    def eval_loss_grad(self, state, adam=None):
        """loss, grads (views of one packed buffer, overwritten by the next call), terms, names, norms."""
        cg = self.cg
        keep = self._launch(state)
        cur = torch.cuda.current_stream()
        chains = [item for item in self.layout if item[1] in ("field", "mg") and (item[0] in cg.gathers or item[0] in cg.direct)]
        merged = set()
        if cg.merged:
            # the gradients of all these fields in ONE launch (what their expressions read is read once)
            by_key = {key: (pos, n) for key, kind, pos, n in self.layout}
            nk = len(cg.merged)
            arr = lambda: (ctypes.c_void_p * nk)()
            gp, xp, mp, vp = arr(), arr(), arr(), arr()
            alpha, omb1, omb2, eps, adev = 0.0, 0.0, 0.0, 0.0, None
            for k, key in enumerate(cg.merged):
                pos = by_key[key][0]
                gp[k] = self.gtmp.get(key, self.gviews[pos]).data_ptr()
                if adam is not None and key in adam[0]:
                    _, arrays, mm, vv, alpha, omb1, omb2, eps = adam
                    xp[k], mp[k], vp[k] = arrays[pos].data_ptr(), mm[pos].data_ptr(), vv[pos].data_ptr()
            if isinstance(alpha, torch.Tensor):
                adev, alpha = alpha.data_ptr(), 0.0
            rc = self.lib.jit_gather_all(ctypes.byref(self.args), gp, xp, mp, vp, float(alpha), float(omb1), float(omb2),
                                         float(eps), adev, ops.stream_ptr())
            if rc != 0:
                raise RuntimeError("traced gather launch failed: hip error {}".format(rc))
            merged = set(cg.merged)
        side = self._side_streams(len(chains))
        for i, (key, kind, pos, n) in enumerate(chains):
            s_ = side[i % len(side)] if side else cur
            if side:
                s_.wait_stream(cur)
            with torch.cuda.stream(s_):
                fuse = adam is not None and key in adam[0]
                if key in merged:
                    g = self.gtmp.get(key, self.gviews[pos])
                elif key in cg.gathers:
                    g = self.gtmp.get(key, self.gviews[pos])
                    if fuse:
                        _, arrays, mm, vv, alpha, omb1, omb2, eps = adam
                        adev = alpha.data_ptr() if isinstance(alpha, torch.Tensor) else None
                        rc = self.lib.jit_gather_adam(
                            cg.gathers.index(key), ctypes.byref(self.args), g.data_ptr(), arrays[pos].data_ptr(),
                            mm[pos].data_ptr(), vv[pos].data_ptr(), 0.0 if adev else float(alpha), float(omb1), float(omb2),
                            float(eps), adev, ops.stream_ptr())
                    else:
                        rc = self.lib.jit_gather(cg.gathers.index(key), ctypes.byref(self.args), g.data_ptr(), ops.stream_ptr())
                    if rc != 0:
                        raise RuntimeError("traced gather launch failed: hip error {}".format(rc))
                else:
                    g = self.cot[cg.direct[key]]
                if kind == "mg":
                    factors, loc, shapes = self.mg_meta[key]
                    ops.mg_synth_adj(g, shapes, loc, factors=factors, grads=self.gviews[pos:pos + n])
                    if fuse:
                        # the coarser levels (1 / 2^d of the unknowns and less) keep the fastest transposes (the `*_adam`
                        # chain has no two-step route for 'nccc': 8.2 instead of 5.2 ms at 32 x 256^3) and get their
                        # update from the plain kernel, level by level
                        _, arrays, mm, vv, alpha, omb1, omb2, eps = adam
                        flat = [_flat_range(lst[pos + 1:pos + n]) for lst in (arrays, mm, vv, self.gviews)]
                        if n > 1 and all(f is not None for f in flat):  # one launch: the levels are adjacent in the packed vectors
                            ops.adam_step(*flat, alpha, omb1, omb2, eps)
                        else:
                            for k in range(pos + 1, pos + n):
                                ops.adam_step(arrays[k].view(-1), mm[k].view(-1), vv[k].view(-1), self.gviews[k].view(-1),
                                              alpha, omb1, omb2, eps)
        for s_ in side:
            cur.wait_stream(s_)
        for key, kind, pos, n in self.layout:
            if kind == "net" and key in cg.pgrads and not self.pgrad_direct:
                pofs = cg.pg_offset[key]
                for j, group in enumerate(cg.pgrads[key]):
                    self.gviews[pos + j].copy_(self.pgrad[pofs:pofs + len(group)].view(self.gviews[pos + j].shape))
                    pofs += len(group)
        if self.par_outputs is not None:
            # the parameter-space outputs: one launch behind k_loss -- terms, norms, the loss and the parameters' gradients
            arrays = self.domain.arrays_from_state(state)
            for s_, index in enumerate(cg.par_index):
                if not arrays[index].is_contiguous() or arrays[index].dtype != self.tr.torch_dtype:
                    raise RuntimeError("parameter arrays must be contiguous {} tensors".format(self.tr.torch_dtype))
                self.par_args.val[s_] = arrays[index].data_ptr()
                self.par_args.grad[s_] = self.gviews[index].data_ptr()
            rc = self.lib.jit_par(ctypes.byref(self.args), ctypes.byref(self.par_args), ops.stream_ptr())
            if rc != 0:
                raise RuntimeError("parameter-space kernel launch failed: hip error {}".format(rc))
        out = self.out.clone()
        nout = len(self.raw)
        loss = out[0]
        terms = [out[1 + k] for k in range(nout)]
        norms = [out[1 + nout + k] for k in range(nout)]
        del keep
        if self.par_outputs is not None:
            pout = self.pout.clone()
            for q, (k, _) in enumerate(self.par_outputs):
                terms.insert(k, pout[2 * q])
                norms.insert(k, pout[2 * q + 1])
        elif self.offgrid:
            loss, terms, norms = self._eval_offgrid(state, loss, terms, norms)
        return loss, list(self.gviews), terms, self.names, norms

    def _eval_offgrid(self, state, loss, terms, norms):
        """The outputs that live in parameter space: the tape of their torch operations replayed on the current
        parameter arrays, mean squares added to the loss, gradients to the parameters' gradients (autograd on a few small
        tensors; reference core.py:1076-1100 treats them like any other output)."""
        arrays = self.domain.arrays_from_state(state)
        memo = dict()
        host = lambda n: float(self._host_value(n, memo))
        terms, norms = list(terms), list(norms)
        with torch.enable_grad():
            leaves = {i: arrays[i].detach().requires_grad_(True) for i in set(self.param_tape.leaves.values())}
            total = None
            extra = []
            for k, expr, ops in self.offgrid:
                env = self.param_tape.replay(ops, leaves)
                f = expr.evaluate(env, host)
                if not isinstance(f, torch.Tensor):
                    f = torch.as_tensor(f, dtype=self.tr.torch_dtype, device=self.out.device)
                term = torch.mean(torch.square(f))
                extra.append((k, term))
                total = term if total is None else total + term
            used = [i for i in sorted(leaves)]
            grads = torch.autograd.grad(total, [leaves[i] for i in used], allow_unused=True) if total.requires_grad else []
        # parameter arrays the grid kernels write no gradient for (an Array that appears only in a prior, a network frozen
        # on the grid but regularised): nothing overwrites their slots of the packed gradient, so they are SET here --
        # adding would accumulate over evaluations
        fresh = set()
        for key, kind, pos, n in self.layout:
            if kind == "net" and key not in self.cg.pgrads:
                fresh.update(range(pos, pos + n))
        for i, g in zip(used, grads):
            if g is None:
                if i in fresh:
                    self.gviews[i].zero_()
            elif i in fresh:
                self.gviews[i].copy_(g.to(self.gviews[i].dtype))
            else:
                self.gviews[i].add_(g.to(self.gviews[i].dtype))
        for k, term in extra:
            term = term.detach().to(loss.dtype)
            terms.insert(k, term)
            norms.insert(k, torch.sqrt(term))
            loss = loss + term
        return loss, terms, norms


def _flat_range(tensors):
    """One flat view over `tensors` when they lie back to back in one storage (level arrays of a packed vector),
    else None."""
    if not tensors:
        return None
    first = tensors[0]
    off = first.storage_offset()
    for t in tensors:
        if (not t.is_contiguous() or t.dtype != first.dtype or t.storage_offset() != off
                or t.untyped_storage().data_ptr() != first.untyped_storage().data_ptr()):
            return None
        off += t.numel()
    return torch.empty(0, dtype=first.dtype, device=first.device).set_(
        first.untyped_storage(), first.storage_offset(), (off - first.storage_offset(),), (1,))


class TracedGroups:
    """The evaluator of an operator whose outputs live on grids of different shapes: one `TracedOperator` (one generated
    kernel set) per shape, every one over the whole state; losses and gradients are summed, terms and norms returned in
    the operator's order.  (The optimizer's update is applied by the plain kernel: a field may receive gradient from
    several groups; epochs are not replayed as a graph.)"""

    graph_ok = False
    offgrid, par_outputs = (), None

    def __init__(self, problem, state, groups):
        self.problem, self.domain, self.groups = problem, problem.domain, groups
        self.parts = [TracedOperator(problem, state, only=positions) for positions in groups]
        for part in self.parts:
            if part.offgrid:
                raise TraceUnsupported("outputs in parameter space beside outputs on several grids")
        self.names = [None] * sum(len(g) for g in groups)
        for positions, part in zip(groups, self.parts):
            for k, name in zip(positions, part.names):
                self.names[k] = name
        from .optimizer import pack_like

        self.gflat, self.gviews = pack_like(self.domain.arrays_from_state(state))
        self.lib_path = [part.lib_path for part in self.parts]
        self.cgs = [part.cg for part in self.parts]  # one code generator per shape group (there is no single `cg`)

    def matches(self, state):
        return all(part.matches(state) for part in self.parts)

    def graph_begin(self, nrows):
        raise RuntimeError("epochs of an operator traced as several kernel sets are not replayed as a graph")

    graph_upload = graph_end = graph_begin

    def eval_loss_grad(self, state, adam=None):
        """loss, grads (views of one packed buffer, overwritten by the next call), terms, names, norms."""
        n = len(self.names)
        terms, norms, loss = [None] * n, [None] * n, None
        for j, (positions, part) in enumerate(zip(self.groups, self.parts)):
            l, _, t, _, r = part.eval_loss_grad(state)
            loss = l if loss is None else loss + l
            if j == 0:
                self.gflat.copy_(part.gflat)
            else:
                self.gflat.add_(part.gflat)
            for k, tk, rk in zip(positions, t, r):
                terms[k], norms[k] = tk, rk
        return loss, list(self.gviews), terms, self.names, norms


def trace_jacobian(problem, state):
    """A TracedOperator with the Jacobian kernel for `Problem.eval_operator_grad`, or None (reason logged) when the operator
    has no pointwise form or its Jacobian needs dense columns / windows (the autograd route then stays)."""
    from .core import Field
    from .util import printlog

    if not all(isinstance(f, Field) for f in state.fields.values()):
        return None  # (`linearize` takes plain fields; parameter arrays mean dense Jacobian columns)
    try:
        return TracedOperator(problem, state, jac=True)
    except TraceUnsupported as e:
        printlog("odil_amd: Jacobian not generated ({}); eval_operator_grad uses autograd".format(e))
    except FileNotFoundError as e:
        # no hipcc / no prebuilt library for THIS kernel (it lives in a library of its own: a cache filled for the
        # gradient configurations does not hold it).  The Newton iterate does not depend on which evaluation forms the
        # coefficient arrays, so the autograd evaluation takes over -- loudly, once per problem.
        printlog("odil_amd: WARNING: the Jacobian kernel cannot be built here ({}); eval_operator_grad uses the slower "
                 "autograd evaluation (tools/prebuild_jit.py builds it ahead of time)".format(e))
    return None


def trace(problem, state):
    """A TracedOperator for `problem` (a TracedGroups when its outputs live on grids of several shapes), or None (with
    the reason logged) when the operator cannot be expressed as pointwise stencil kernels."""
    from .util import printlog

    try:
        try:
            return TracedOperator(problem, state)
        except TraceGroups as e:
            return TracedGroups(problem, state, e.groups)
    except TraceUnsupported as e:
        printlog("odil_amd: operator not traced ({}); using the generic autograd path".format(e))
    except FileNotFoundError as e:
        # no hipcc / no writable cache: the operator COULD run on generated kernels but this installation cannot build
        # them.  Falling back silently would leave a ~20x slower evaluation behind a log line, so this is an error
        # unless the fallback is asked for (ODIL_TRACE_FALLBACK=1).
        if not int(os.environ.get("ODIL_TRACE_FALLBACK", 0)):
            raise RuntimeError(
                "odil_amd: the operator traces to generated HIP kernels but they cannot be built here ({}). Install hipcc / "
                "prebuild the cache (tools/prebuild_jit.py, ODIL_JIT_CACHE), or set ODIL_TRACE_FALLBACK=1 to accept the "
                "generic autograd path (about 20x slower), or ODIL_TRACE=0 to choose it explicitly.".format(e)) from e
        printlog("odil_amd: no hipcc for traced operators ({}); ODIL_TRACE_FALLBACK=1: using the generic autograd path".format(e))
    return None
