"""Slab decomposition of TRACED operators over the GPUs of one node: any `operator(ctx)` the tracer can
express (odil_amd/stencil_jit.py) -- the tracer-velocity workload BASELINE.json names for 8 GPUs -- with an
Adam loop per rank.  No reference counterpart (the reference is single-device).

Layout.  The grid is cut along ONE domain axis `a` on which every grid field is cell-centred (for the
(t, x, y[, z]) workloads: x; the node-centred time axis stays whole).  Rank r owns n = N_a / P cells of every
multigrid level (every level keeps >= 2 owned cells per rank).  Level arrays are stored ghost-extended along
`a` exactly as in the Poisson slab path (odil_amd/slab.py): G = 2 ghost cells at every interior interface,
none at the two ends of the decomposition, where the prolongation has its wall rule.  All unknowns of a rank
(level arrays of all fields, then replicated network / Array parameters) live in one packed vector.

Per epoch and rank:
  1. ("halo") the first / last OWNED plane of every level array of every field -> the neighbours' inner ghost
     planes, one packed message per neighbour;
  2. u = w_0 + P(w_1 + P(...)) per field with the unmodified single-GPU transfer kernels run on the extended
     arrays: a coarse array with one valid ghost plane gives a fine array whose inner ghost plane is exact
     (the wall formulas only ever reach the outer one, which nobody reads);
  3. ("wrap") `Context.field` rolls PERIODICALLY (reference core.py:962-963): what owned cells of the first
     rank read below the grid is the last rank's last plane of u and vice versa -- one plane per field between
     those two ranks, kept in wrap buffers the generated kernel selects at the ends;
  4. the generated forward / cotangent kernel on the owned cells (global indices for masks, windows and
     constant arrays), then the generated gathers, which write the rank's ghost-extended gradient: ghost
     planes receive what this rank's cells contribute to the NEIGHBOUR's cells, wrap buffers what they
     contribute across the ends;
  5. ("wrap") those wrap contributions are added into the owned end planes of the first / last rank;
  6. P^T down the levels WITHOUT exchanges.  The transpose is linear, so every rank pushes its own partial
     gradient -- owned planes plus the inner ghost plane of contributions -- through the unmodified whole-
     array P^T kernel: with the outer ghost planes zero, the wall weights of that kernel multiply zeros, and
     its result on the coarse owned planes and inner ghost planes is exactly this rank's share;
  7. ("halo") ONE packed message per neighbour carries the inner ghost planes of the gradients of all
     levels and fields; the receiver adds them to its owned boundary planes (transpose of step 1).  Levels too
     coarse to leave every rank 2 planes are AGGLOMERATED: every rank holds the whole array, prolongates from
     its window of it, pushes its share of the gradient into a zeroed copy, and one ("sum") all-reduce of
     those few small arrays replaces their halo-add; all ranks then apply the same update;
  8. ("sum") parameter gradients of networks / Arrays, when there are any; Adam on the packed vector.
The loss needs one more ("sum") of a few scalars, issued only when a value is read.

The epoch is a generator that yields at exchanges (`(kind, send_lo, send_hi)`), so the same code runs one rank
per GPU over RCCL (slab.TorchDistComm), over gloo with CPU doubles of the kernels (tests/test_slab_traced_cpu.py)
and with all ranks emulated in one process on one GPU (slab.run_lockstep, tests/test_slab_gpu.py).
"""

import ctypes
import contextlib
import math

import numpy as np
import torch

from . import ops as hip_ops
from .stencil_codegen import _Codegen, _compile
from .stencil_jit import TracedOperator, trace_outputs
from .stencil_trace import TraceUnsupported

G = 2  # ghost planes per interior interface (the generated gathers assume 2)


class _Level:
    """One level array of one grid field on one rank."""

    def __init__(self, gshape, axis, rank, world, replicate=False):
        self.gshape, self.axis = tuple(gshape), axis
        self.rank, self.world = rank, world
        # A level with fewer than 2 planes per rank is AGGLOMERATED: every rank keeps the whole (small) array, computes
        # it redundantly and the ranks' gradient contributions are summed by one all-reduce.
        self.replicated = replicate or gshape[axis] % world != 0 or gshape[axis] // world < 2
        if self.replicated:
            self.n, self.g_lo, self.g_hi, self.off = gshape[axis], 0, 0, 0
        else:
            self.n = gshape[axis] // world
            self.g_lo = 0 if rank == 0 else G
            self.g_hi = 0 if rank == world - 1 else G
            self.off = rank * self.n
        self.shape = tuple(self.g_lo + self.n + self.g_hi if d == axis else s for d, s in enumerate(gshape))
        self.size = math.prod(self.shape)
        self.plane = self.size // self.shape[axis]

    def window_of(self, fine):
        """(first plane, count) of THIS (replicated, coarse) level that prolongates to the ghost-extended planes
        of the sharded level `fine` of this rank: half its owned range plus one plane per interior interface."""
        e_lo, e_hi = (1 if fine.g_lo else 0), (1 if fine.g_hi else 0)
        return fine.off // 2 - e_lo, fine.n // 2 + e_lo + e_hi

    def planes(self, a, first, count=1):
        """`count` planes from owned-relative position `first` along the sharded axis."""
        return a.narrow(self.axis, self.g_lo + first, count)

    def owned(self, a):
        return a.narrow(self.axis, self.g_lo, self.n)

    def inner(self, a):
        """View with ONE ghost plane per interior interface: the coarse operand of P, the result of P^T."""
        lo = self.g_lo - 1 if self.g_lo else 0
        hi = a.shape[self.axis] - (self.g_hi - 1 if self.g_hi else 0)
        return a.narrow(self.axis, lo, hi - lo)

    def plane_desc(self, start, pos):
        """(base, outer, ostride, inner) of the plane at owned-relative position `pos` inside the packed vector whose
        entry `start` is this array's first element: `outer` runs of `inner` contiguous elements (ops.PlaneList)."""
        outer = math.prod(self.shape[:self.axis])
        inner = self.plane // outer
        return (start + (self.g_lo + pos) * inner, outer, self.shape[self.axis] * inner, inner)


class HipSlabKernels:
    """The generated kernels of one rank (stencil_codegen in slab mode) and their buffers."""

    def __init__(self, problem, state, axis, n, device):
        tr, outs, raw, self.names, Gshape = trace_outputs(problem, state)
        self.problem, self.tr, self.raw = problem, tr, raw
        cg = _Codegen(tr, outs, raw, Gshape, state, slab=(axis, n))
        # outputs in parameter space (a weight regulariser): every rank evaluates them, redundantly, with the generated
        # kernel of param_expr.py AFTER the parameter gradients were summed over the ranks
        self.par_outputs = None
        if tr.offgrid:
            from . import param_expr
            from .core import Array, NeuralNet

            domain = problem.domain
            arrays0 = domain.arrays_from_state(state)
            offgrid = [(k, e, tr.param_tape.slice_for(e.param_ids())) for k, e in tr.offgrid]
            try:
                self.par_outputs = param_expr.convert(tr.param_tape, offgrid, {i: int(a.numel()) for i, a in enumerate(arrays0)})
            except param_expr.Unsupported as e:
                raise TraceUnsupported("outputs in parameter space under the slab decomposition ({})".format(e))
            cg.par_outputs, cg.par_numel, cg.par_keys, pos = self.par_outputs, dict(), dict(), 0
            self.par_where = dict()  # global array index -> (field key, position among the field's arrays)
            for key, field in state.fields.items():
                cnt = len(domain.arrays_from_field(field))
                for i in range(pos, pos + cnt):
                    cg.par_numel[i] = int(arrays0[i].numel())
                    if isinstance(field, (NeuralNet, Array)):
                        cg.par_keys[i] = key
                        self.par_where[i] = (key, i - pos)
                pos += cnt
        self.source = cg.source()
        self.lib, self.lib_path = _compile(self.source, cg.flags)
        self.cg = cg
        self.halo = cg.halo
        dt = tr.torch_dtype
        self.dtype = dt
        self.total = cg.total
        cap = cg.max_blocks or (4096 if len(cg.pg_decl) > 8 else 65536)
        self.nblocks = min((self.total // cg.vw_fwd + 255) // 256, cap)
        nout = len(outs)
        self.nout = nout
        self.cot = [torch.empty(cg.GL, dtype=dt, device=device) for _ in range(cg.ncot)]
        self.part = torch.empty(max(1, nout * self.nblocks), dtype=dt, device=device)
        self.ppart = torch.empty(max(1, len(cg.pg_decl) * self.nblocks), dtype=dt, device=device)
        self.part2 = torch.zeros(16 * (nout + len(cg.pg_decl)), dtype=dt, device=device)
        self.out = torch.zeros(1 + 2 * nout, dtype=dt, device=device)
        self.pgrad = torch.zeros(max(1, len(cg.pg_decl)), dtype=dt, device=device)
        nsrc = max(1, len(cg.src_keys))

        class Args(ctypes.Structure):
            _fields_ = [
                ("src", ctypes.c_void_p * nsrc),
                ("ten", ctypes.c_void_p * max(1, len(tr.tensors))),
                ("cot", ctypes.c_void_p * max(1, cg.ncot)),
                ("par", ctypes.c_void_p * max(1, cg.par_arrays)),
                ("hs", ctypes.c_void_p), ("hsv", ctypes.c_double * max(1, len(cg.hs))),
                ("part", ctypes.c_void_p), ("ppart", ctypes.c_void_p), ("part2", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("pgrad", ctypes.c_void_p), ("edge", ctypes.c_void_p), ("nblocks", ctypes.c_int),
                ("off", ctypes.c_int), ("lo", ctypes.c_int), ("ea", ctypes.c_int), ("hw", ctypes.c_int),
                ("wlo", ctypes.c_void_p * nsrc), ("whi", ctypes.c_void_p * nsrc),
                ("gwlo", ctypes.c_void_p * nsrc), ("gwhi", ctypes.c_void_p * nsrc),
                ("alo", ctypes.c_int), ("ahi", ctypes.c_int),
            ]

        a = self.args = Args()
        for i, t in enumerate(tr.tensors):
            a.ten[i] = t.data_ptr()
        for i, t in enumerate(self.cot):
            a.cot[i] = t.data_ptr()
        a.part, a.ppart, a.part2 = self.part.data_ptr(), self.ppart.data_ptr(), self.part2.data_ptr()
        a.out, a.pgrad, a.nblocks, a.hs = self.out.data_ptr(), self.pgrad.data_ptr(), self.nblocks, None
        self.lib.jit_fwd.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.lib.jit_gather.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self.lib.jit_gather_adam.argtypes = [ctypes.c_int, ctypes.c_void_p] + [ctypes.c_void_p] * 4 + [ctypes.c_double] * 4 + [
            ctypes.c_void_p, ctypes.c_void_p]
        a.alo = a.ahi = 0
        self.src_keys, self.gather_keys = list(cg.src_keys), list(cg.gathers)
        # fields whose own arrays the gathers read again (local derivatives re-evaluated there, stencil_grad.py)
        self.reread = {k for keys in cg.gather_reads_sources.values() for k in keys}
        self.merged = list(cg.merged)
        if self.merged:
            self.lib.jit_gather_all.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_void_p)] * 4 + [
                ctypes.c_double] * 4 + [ctypes.c_void_p, ctypes.c_void_p]
        self.param_groups = {key: (cg.pg_offset[key], [len(g) for g in groups]) for key, groups in cg.pgrads.items()}

    _host_value = TracedOperator._host_value  # host scalars (functions of problem.tracers): the single-GPU evaluator

    def set_geometry(self, off, lo, ea):
        self.args.off, self.args.lo, self.args.ea, self.args.hw = off, lo, ea, self.halo

    def set_params(self, state_arrays_of):
        """Pointers of network / Array parameters (`state_arrays_of(key)` -> this rank's replicated arrays)."""
        i = 0
        for key, _ in self.cg.nets:
            for arr in state_arrays_of(key):
                self.args.par[i] = arr.data_ptr()
                i += 1
        for key, _ in self.cg.arrays:
            self.args.par[i] = state_arrays_of(key)[0].data_ptr()
            i += 1

    def forward(self, srcs, wlo, whi):
        """Forward + cotangents on the owned cells.  srcs / wlo / whi: src key -> array."""
        from ._lib import ptr

        ptr(self.out)  # fails loudly on a CPU tensor: there is no CPU path
        memo = dict()
        for i, node in enumerate(self.cg.hs):
            self.args.hsv[i] = float(self._host_value(node, memo))
        for i, key in enumerate(self.src_keys):
            self.args.src[i] = srcs[key].data_ptr()
            self.args.wlo[i], self.args.whi[i] = wlo[key].data_ptr(), whi[key].data_ptr()
        rc = self.lib.jit_fwd(ctypes.byref(self.args), hip_ops.stream_ptr())
        if rc != 0:
            raise RuntimeError("traced slab kernel launch failed: hip error {}".format(rc))

    fused_adam = True  # gather() can apply the optimizer's update to the planes whose gradient it completes

    def gather(self, key, g, gwlo, gwhi, adam=None):
        """Ghost-extended gradient of field `key` (level 0) into g, wrap contributions into gwlo / gwhi.
        adam = (x, m, v, alpha, one_minus_b1, one_minus_b2, eps, lo, hi): Adam (reference optimizer.py:316-318) of the
        array's entries on the owned planes lo <= plane < hi of the sharded axis, by the lane that forms their gradient
        (x, m, v laid out as g)."""
        i = self.src_keys.index(key)
        self.args.gwlo[i], self.args.gwhi[i] = gwlo.data_ptr(), gwhi.data_ptr()
        which = self.gather_keys.index(key)
        if adam is None:
            self.args.alo = self.args.ahi = 0
            rc = self.lib.jit_gather(which, ctypes.byref(self.args), g.data_ptr(), hip_ops.stream_ptr())
        else:
            x, m, v, alpha, omb1, omb2, eps, lo, hi = adam
            assert x.shape == g.shape and x.is_contiguous() and m.is_contiguous() and v.is_contiguous()
            self.args.alo, self.args.ahi = int(lo), int(hi)
            rc = self.lib.jit_gather_adam(which, ctypes.byref(self.args), g.data_ptr(), x.data_ptr(), m.data_ptr(),
                                          v.data_ptr(), float(alpha), float(omb1), float(omb2), float(eps), None,
                                          hip_ops.stream_ptr())
        if rc != 0:
            raise RuntimeError("traced slab gather launch failed: hip error {}".format(rc))

    def gather_all(self, items, hyper=None, planes=(0, 0)):
        """Every merged field's ghost-extended gradient in ONE launch (see `gather`): items = [(key, g, gwlo, gwhi,
        (x, m, v) or None)] for exactly the keys of `self.merged`; hyper = (alpha, one_minus_b1, one_minus_b2, eps)."""
        nk = len(self.merged)
        arr = lambda: (ctypes.c_void_p * nk)()
        gp, xp, mp, vp = arr(), arr(), arr(), arr()
        by_key = {it[0]: it for it in items}
        for k, key in enumerate(self.merged):
            _, g, gwlo, gwhi, xmv = by_key[key]
            i = self.src_keys.index(key)
            self.args.gwlo[i], self.args.gwhi[i] = gwlo.data_ptr(), gwhi.data_ptr()
            gp[k] = g.data_ptr()
            if xmv is not None:
                x, m, v = xmv
                assert x.shape == g.shape and x.is_contiguous() and m.is_contiguous() and v.is_contiguous()
                xp[k], mp[k], vp[k] = x.data_ptr(), m.data_ptr(), v.data_ptr()
        self.args.alo, self.args.ahi = int(planes[0]), int(planes[1])
        alpha, omb1, omb2, eps = hyper if hyper is not None else (0.0, 0.0, 0.0, 0.0)
        rc = self.lib.jit_gather_all(ctypes.byref(self.args), gp, xp, mp, vp, float(alpha), float(omb1), float(omb2),
                                     float(eps), None, hip_ops.stream_ptr())
        if rc != 0:
            raise RuntimeError("traced slab gather launch failed: hip error {}".format(rc))

    def partial_terms(self):
        """This rank's share of every loss term (sum over owned cells / GLOBAL count)."""
        return self.out[1:1 + self.nout]

    def launch_par(self, values_of, grads_of):
        """The parameter-space outputs: terms into self.pout, gradients added to (set in) the parameters' gradient views.
        values_of / grads_of: field key -> list of this rank's (replicated) arrays."""
        if self.par_outputs is None:
            return
        if not hasattr(self, "par_args"):
            npar = max(1, len(self.cg.par_index))

            class ParArgs(ctypes.Structure):
                _fields_ = [("val", ctypes.c_void_p * npar), ("grad", ctypes.c_void_p * npar), ("pout", ctypes.c_void_p)]

            self.par_args = ParArgs()
            self.pout = torch.zeros(2 * len(self.par_outputs), dtype=self.dtype, device=self.out.device)
            self.par_args.pout = self.pout.data_ptr()
            self.lib.jit_par.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        for s_, index in enumerate(self.cg.par_index):
            key, j = self.par_where[index]
            self.par_args.val[s_] = values_of(key)[j].data_ptr()
            self.par_args.grad[s_] = grads_of(key)[j].data_ptr()
        rc = self.lib.jit_par(ctypes.byref(self.args), ctypes.byref(self.par_args), hip_ops.stream_ptr())
        if rc != 0:
            raise RuntimeError("parameter-space kernel launch failed: hip error {}".format(rc))


class SlabTracedAdam:
    """One rank of the slab-decomposed Adam loop of a traced operator.

    problem: the GLOBAL problem (global Domain, operator, extra -- constant arrays in `extra` are global, they
    do not carry the sharded unknowns' time axis); state: the global state's STRUCTURE; its arrays give the
    initial values when they are real tensors and zeros when they live on the 'meta' device."""

    def __init__(self, problem, state, rank, world, axis=None, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7,
                 device=None, kernels=None):
        from .core import Array, Field, MultigridField, NeuralNet

        domain = problem.domain
        self.problem, self.domain, self.rank, self.world = problem, domain, rank, world
        self.device = device if device is not None else domain.mod.device
        dtype = torch.float64 if np.dtype(domain.dtype) == np.float64 else torch.float32
        self.dtype = dtype
        self.npdt = np.float64 if dtype == torch.float64 else np.float32
        grid = {k: f for k, f in state.fields.items() if isinstance(f, (Field, MultigridField))}
        if axis is None:
            ok = [d for d in range(domain.ndim) if all(f.loc[d] == "c" for f in grid.values())]
            if not ok:
                raise ValueError("no axis on which every field is cell-centred")
            axis = ok[0]
        self.axis = axis
        N = domain.cshape[axis]
        if N % world:
            raise ValueError("{} cells on axis {} over {} ranks".format(N, axis, world))
        self.n = N // world
        # ---- layout of the packed vector ---------------------------------------------------------
        self.entries = []  # (key, kind, [levels] or [shapes])
        sizes = []
        for key, f in state.fields.items():
            if isinstance(f, MultigridField):
                factors = [float(x) for x in (f.factors or domain.mg_factors or [1] * len(f.terms))]
                mgloc = domain._mg_loc(f)
                if mgloc[axis] != "c":
                    raise ValueError("field '{}' is not refined / cell-centred on the sharded axis".format(key))
                levels = [_Level(tuple(t.array.shape), axis, rank, world) for t in f.terms]
                for l in range(len(levels) - 2, -1, -1):  # a rank's planes must be whole coarse cells at the transition
                    if levels[l + 1].replicated and not levels[l].replicated and levels[l].n % 2:
                        levels[l] = _Level(tuple(f.terms[l].array.shape), axis, rank, world, replicate=True)
                if levels[0].replicated or any(a.replicated and not b.replicated for a, b in zip(levels, levels[1:])):
                    raise ValueError("field '{}': {} cells on the sharded axis over {} ranks".format(key, N, world))
                init = [t.array for t in f.terms]
                # u = f_0 w_0 + P(f_1 w_1 + P(...)) (reference core.py:245-263); None: every factor is 1
                if len(levels) == 1 and factors[0] != 1.0:
                    raise NotImplementedError("slab decomposition of a one-level multigrid field with a factor")
                self.entries.append(dict(key=key, kind="mg", levels=levels, loc=mgloc, init=init,
                                         factors=None if all(x == 1.0 for x in factors) else factors))
                sizes += [lv.size for lv in levels]
            elif isinstance(f, Field):
                if f.loc[axis] != "c":
                    raise ValueError("field '{}' is not cell-centred on the sharded axis".format(key))
                levels = [_Level(tuple(f.array.shape), axis, rank, world)]
                if levels[0].replicated:
                    raise ValueError("field '{}': {} cells on the sharded axis over {} ranks".format(key, N, world))
                self.entries.append(dict(key=key, kind="field", levels=levels, loc=f.loc, init=[f.array]))
                sizes.append(levels[0].size)
            elif isinstance(f, (NeuralNet, Array)):
                arrays = domain.arrays_from_field(f)
                self.entries.append(dict(key=key, kind="par", shapes=[tuple(a.shape) for a in arrays], init=arrays))
                sizes += [int(a.numel()) for a in arrays]
            else:
                raise TypeError(type(f).__name__)
        total = sum(sizes)
        mk = lambda: torch.zeros(total, dtype=dtype, device=self.device)
        self.x, self.m, self.v, self.g = mk(), mk(), mk(), mk()
        # fields with multigrid factors: the UNSCALED cotangents h_l = (P^T)^l g_u travel down the levels, the gradient
        # of level l is g_l = f_l h_l -- a second packed vector with the layout of g
        scaled = any(e.get("factors") for e in self.entries)
        self.hvec = mk() if scaled else None
        pos = 0
        self._rep = []  # (start, size) of the replicated level arrays in the packed vectors
        # REDUNDANT GHOST UPDATES: every rank applies the optimizer's update to its inner
        # ghost planes as well as to its own planes -- the gradient message carries, besides the sender's contribution to
        # the receiver's boundary plane, the sender's own boundary plane of the partial gradient, so both ranks hold
        # the same complete gradient of the shared planes (a + b on one side, b + a on the other: the same bits) and
        # make the same update with the same library kernel; the exchange of the updated unknowns at the start of every
        # epoch disappears (one initial synchronisation of the ghost planes remains).  Same bytes, one exchange fewer.
        self.redundant = True
        self._x_synced = False
        send_own = dict(lo=[], hi=[])  # plane descriptors (ops.PlaneList): owned boundary planes / inner ghost planes
        recv_ghost = dict(lo=[], hi=[])
        # the same planes split by level for the gradients' halo-add: the finest levels (97 % of the bytes) travel WHILE
        # the transposes run down the levels, the coarser ones after them
        own0, own1, ghost0, ghost1 = (dict(lo=[], hi=[]) for _ in range(4))
        for e in self.entries:
            shapes = [lv.shape for lv in e["levels"]] if "levels" in e else e["shapes"]
            views = dict(x=[], m=[], v=[], g=[], h=[])
            e["start"] = pos
            for k, shape in enumerate(shapes):
                cnt = math.prod(shape)
                for name, buf in (("x", self.x), ("m", self.m), ("v", self.v), ("g", self.g)):
                    views[name].append(buf[pos:pos + cnt].view(shape))
                # where the gathers / transposes write: g itself, or the unscaled twin when the field has factors
                views["h"].append((self.hvec if e.get("factors") else self.g)[pos:pos + cnt].view(shape))
                src = e["init"][k]
                if src is not None and src.device.type != "meta":
                    if "levels" in e:
                        lv = e["levels"][k]
                        src = src.narrow(axis, lv.off - lv.g_lo, lv.shape[axis])
                    views["x"][k].copy_(src.to(device=self.device, dtype=dtype))
                if "levels" in e and e["levels"][k].replicated:
                    self._rep.append((pos, cnt))
                elif "levels" in e:
                    lv = e["levels"][k]
                    own, ghost = (own0, ghost0) if k == 0 else (own1, ghost1)
                    # `ghost`: what the gradient message to that side carries; `own`: where the message FROM that side is
                    # added.  With redundant ghost updates the message also carries the sender's own boundary plane (its
                    # partial gradient), which completes the receiver's copy of that plane -- its inner ghost plane
                    if rank > 0:
                        send_own["lo"].append(lv.plane_desc(pos, 0))
                        recv_ghost["lo"].append(lv.plane_desc(pos, -1))
                        own["lo"].append(lv.plane_desc(pos, 0))
                        ghost["lo"].append(lv.plane_desc(pos, -1))
                        if self.redundant:
                            ghost["lo"].append(lv.plane_desc(pos, 0))
                            own["lo"].append(lv.plane_desc(pos, -1))
                    if rank < world - 1:
                        send_own["hi"].append(lv.plane_desc(pos, lv.n - 1))
                        recv_ghost["hi"].append(lv.plane_desc(pos, lv.n))
                        own["hi"].append(lv.plane_desc(pos, lv.n - 1))
                        ghost["hi"].append(lv.plane_desc(pos, lv.n))
                        if self.redundant:
                            ghost["hi"].append(lv.plane_desc(pos, lv.n - 1))
                            own["hi"].append(lv.plane_desc(pos, lv.n))
                pos += cnt
            e.update(views)
            del e["init"]
        # one launch packs / unpacks / adds every plane of every level and field of an exchange (odil_planes_copy); the
        # send buffers of the two exchange points are allocated once
        mk_list = lambda planes: hip_ops.PlaneList(planes, self.device) if planes else None
        self._own = {s: mk_list(send_own[s]) for s in ("lo", "hi")}
        self._ghost = {s: mk_list(recv_ghost[s]) for s in ("lo", "hi")}
        mk_buf = lambda pl: None if pl is None else torch.empty(pl.count, dtype=dtype, device=self.device)
        self._send_x = {s: mk_buf(self._own[s]) for s in ("lo", "hi")}
        self._g_split = [({s: mk_list(o[s]) for s in ("lo", "hi")}, {s: mk_list(g_[s]) for s in ("lo", "hi")})
                         for o, g_ in ((own0, ghost0), (own1, ghost1))]
        self._send_g = [{s: mk_buf(ghost[s]) for s in ("lo", "hi")} for _, ghost in self._g_split]
        self.by_key = {e["key"]: e for e in self.entries}
        self.n_unknowns_local = sum(lv.n * lv.plane for e in self.entries if "levels" in e for lv in e["levels"]
                                    if not lv.replicated) + sum(
            math.prod(s) for e in self.entries if "shapes" in e for s in e["shapes"])
        self.local_cells = math.prod(domain.cshape) // world
        self.global_cells = math.prod(domain.cshape)
        # ---- kernels and their buffers -------------------------------------------------------------
        self.kern = kernels(problem, state, axis, self.n, self.device) if kernels is not None else HipSlabKernels(
            problem, state, axis, self.n, self.device)
        h = self.kern.halo
        if h > 1:
            raise TraceUnsupported("reads {} cells away along the sharded axis (the exchange keeps 1 plane)".format(h))
        self.h = h
        self.u, self.work, self.wrap = dict(), dict(), dict()
        # the periodic closure's planes (`h` planes per field at each end): the four sets -- u received from across
        # the low / high end, gradient contributions the gathers write for across them -- are each ONE buffer with
        # a view per field, so a set travels as one message without a pack (or is filled by one copy)
        hw = max(h, 1)
        wsizes = {key: math.prod(hw if d == axis else s_ for d, s_ in enumerate(self.by_key[key]["levels"][0].shape))
                  for key in self.kern.src_keys}
        self._wrapbuf = {name: torch.zeros(sum(wsizes.values()), dtype=dtype, device=self.device)
                         for name in ("lo", "hi", "glo", "ghi")}
        woff = 0
        for key in self.kern.src_keys:
            e = self.by_key[key]
            lv0 = e["levels"][0]
            wshape = tuple(hw if d == axis else s_ for d, s_ in enumerate(lv0.shape))
            self.wrap[key] = {name: buf[woff:woff + wsizes[key]].view(wshape) for name, buf in self._wrapbuf.items()}
            woff += wsizes[key]
            if e["kind"] == "mg" and len(e["levels"]) > 1:
                self.u[key] = None  # a view of ONE buffer for all synthesised arrays, below
                self.work[key] = [None] + [torch.zeros(lv.shape, dtype=dtype, device=self.device)
                                           for lv in e["levels"][1:-1]] + [None]
            else:
                self.u[key] = e["x"][0]
        synth = [key for key in self.kern.src_keys if self.u[key] is None]
        self._uflat = torch.zeros(sum(self.by_key[key]["levels"][0].size for key in synth), dtype=dtype, device=self.device)
        uoff, ustart = 0, dict()
        for key in synth:
            lv0 = self.by_key[key]["levels"][0]
            self.u[key] = self._uflat[uoff:uoff + lv0.size].view(lv0.shape)
            ustart[key] = uoff
            uoff += lv0.size
        # the end planes of the synthesised arrays that close the periodic direction, as ONE message per side packed by
        # one launch (odil_planes_copy) -- when every source field is synthesised; else slices + torch.cat
        self._wrap_pack = None
        if h and len(synth) == len(self.kern.src_keys):
            self._wrap_pack = dict()
            for side in ("lo", "hi"):
                descs = []
                for key in self.kern.src_keys:
                    lv0 = self.by_key[key]["levels"][0]
                    descs.append(lv0.plane_desc(ustart[key], 0 if side == "lo" else lv0.n - h))
                self._wrap_pack[side] = hip_ops.PlaneList(descs, self.device)
        # where the wrap contributions of the gradients land: the first / last `h` owned planes of every gathered
        # field's finest level, in message order (odil_planes_copy, mode add)
        self._wrap_add = dict()
        if h:
            for side, pos in (("lo", 0), ("hi", None)):
                off, lists = 0, []
                for key in self.kern.src_keys:
                    e = self.by_key[key]
                    lv = e["levels"][0]
                    if key in self.kern.gather_keys:
                        first_plane = 0 if pos == 0 else lv.n - h
                        if hasattr(hip_ops.PlaneList, "_run"):
                            base, outer, ostride, inner = lv.plane_desc(e["start"], first_plane)
                            lists.append((hip_ops.PlaneList([(base, outer, ostride, inner * h)], self.device, start=off),
                                          bool(e.get("factors"))))
                        else:  # (CPU double of the plane kernels: tests over gloo)
                            lists.append((key, lv, first_plane))
                            assert len(lists[-1]) == 3
                    off += wsizes[key]
                self._wrap_add[side] = lists
        lv0 = self.by_key[self.kern.src_keys[0]]["levels"][0]
        self.kern.set_geometry(lv0.off, lv0.g_lo, lv0.shape[axis])
        self.kern.set_params(lambda key: self.by_key[key]["x"])
        self.lr, self.b1, self.b2, self.eps = self.npdt(lr), self.npdt(beta_1), self.npdt(beta_2), epsilon
        self.t = 0
        self.has_params = any(e["kind"] == "par" for e in self.entries) and bool(self.kern.param_groups)
        # ---- the optimizer's update inside the gathers ------------------------------------------------
        # A gather completes the gradient of the owned planes 1 .. n - 2 of its array (plane 0 / n - 1 wait for the
        # neighbour's share or the periodic closure): the kernels that can, update those entries where they form the
        # gradient, and the launches at the end of the epoch cover the rest of the packed vector -- the two end pieces of
        # every such array (strided when the sharded axis is not the leading one) and the ranges between the arrays
        # (coarser levels, fields without a gather, parameters).
        self._fused, self._post_flat, self._post_pieces = dict(), [], []
        if getattr(self.kern, "fused_adam", False):
            spans = []
            for key in self.kern.gather_keys:
                e = self.by_key[key]
                lv = e["levels"][0]
                if lv.n < 4:
                    continue
                if self.u[key] is e["x"][0] and key in getattr(self.kern, "reread", ()):
                    continue  # the array a gather reads IS the unknown: no launch of this epoch may update it
                if e.get("factors"):
                    continue  # the gather forms h_0, the gradient is f_0 h_0
                self._fused[key] = (1, lv.n - 1)
                spans.append((e["start"], lv.size))
                outer = math.prod(lv.shape[:axis])
                inner = lv.plane // outer
                stride = lv.shape[axis] * inner
                first, last = lv.g_lo + 1, lv.g_lo + lv.n - 1  # planes [0, first) and [last, extent) of the array
                self._post_pieces.append((e["start"], lv.size, outer, stride, 0, first * inner))
                self._post_pieces.append((e["start"], lv.size, outer, stride, last * inner, (lv.shape[axis] - last) * inner))
            at = 0
            for start, size in sorted(spans):
                if start > at:
                    self._post_flat.append((at, start))
                at = start + size
            if at < total:
                self._post_flat.append((at, total))

    # ---- pieces of the epoch -------------------------------------------------------------------------
    def _field_streams(self, keys):
        """One side stream per field for the TRANSPOSE chains: the small, latency-bound
        launches at the coarse end of one field's chain overlap the long launches of another's -- config 5 as one rank:
        P^T chains 1.80 -> 1.69 ms; the prolongation chains gain nothing that way (2.68 -> 2.71) and stay on one stream.
        Returns [(key, stream or None)] and a function that joins the streams back into the current one."""
        if len(keys) < 2 or self.device.type != "cuda":
            return [(key, None) for key in keys], lambda: None
        cur = torch.cuda.current_stream()
        pool = self.__dict__.setdefault("_streams", [torch.cuda.Stream() for _ in range(4)])
        used = []
        for i, key in enumerate(keys):
            s_ = pool[i % len(pool)]
            if s_ not in [u for _, u in used]:
                s_.wait_stream(cur)
            used.append((key, s_))

        def join():
            for s_ in {u for _, u in used}:
                cur.wait_stream(s_)

        return used, join

    def _synthesise(self):
        for key in self.kern.src_keys:
            e = self.by_key[key]
            L = len(e["levels"])
            if e["kind"] != "mg" or L == 1:
                continue
            coarse = e["x"][L - 1]
            fac = e.get("factors") or [1.0] * L
            cscale = fac[L - 1]
            for l in range(L - 2, -1, -1):
                out = self.u[key] if l == 0 else self.work[key][l]
                fine, lvc = e["levels"][l], e["levels"][l + 1]
                if lvc.replicated and not fine.replicated:  # this rank's window of the replicated coarse field
                    operand = coarse.narrow(self.axis, *lvc.window_of(fine))
                else:
                    operand = lvc.inner(coarse)
                # (a view of the level array without its outer ghost planes: read in place by the marching kernels)
                hip_ops.interp_add(operand, e["loc"], add=e["x"][l], coarse_scale=cscale, add_scale=fac[l], out=out)
                coarse, cscale = out, 1.0

    def _end_planes(self, arrays, side):
        """The h owned planes at the low ('lo') / high end of each source field's array, packed."""
        parts = []
        for key in self.kern.src_keys:
            lv = self.by_key[key]["levels"][0]
            a = arrays(key)
            parts.append(lv.planes(a, 0 if side == "lo" else lv.n - self.h, self.h).reshape(-1))
        return torch.cat(parts)

    def _split_planes(self, buf):
        out, off = dict(), 0
        for key in self.kern.src_keys:
            shape = self.wrap[key]["lo"].shape
            cnt = math.prod(shape)
            out[key] = buf[off:off + cnt].view(shape)
            off += cnt
        return out

    def _transpose_chain(self):
        plan, join = self._field_streams(list(self.kern.gather_keys))
        for key, s_ in plan:
            with (torch.cuda.stream(s_) if s_ is not None else contextlib.nullcontext()):
                self._transpose_field(key)
        join()

    def _transpose_field(self, key):
        e = self.by_key[key]
        fac = e.get("factors")
        for l in range(1, len(e["levels"])):
            lv, fine = e["levels"][l], e["levels"][l - 1]
            if lv.replicated and not fine.replicated:
                # this rank's share of the replicated level: zero but for the window its planes reach
                e["h"][l].zero_()
                if fac:
                    e["g"][l].zero_()
                view = lambda a: a.narrow(self.axis, *lv.window_of(fine))
            else:
                view = lv.inner
            dst = view(e["h"][l])
            best = getattr(hip_ops, "interp_adj_best", hip_ops.interp_adj)
            best(e["h"][l - 1], e["loc"], tuple(dst.shape), out=dst)  # (in place, also into a strided view)
            if fac:
                torch.mul(dst, fac[l], out=view(e["g"][l]))

    # ---- one epoch -------------------------------------------------------------------------------------
    def epoch_gen(self, timers=None, update=True):
        """One Adam epoch as a generator that yields at its exchanges.  update=False: loss and gradient only (the
        quasi-Newton driver of slab_solvers.py) -- the ghost planes of the unknowns are refreshed first (somebody else
        changed the unknowns), no launch applies an update, and self.g is left complete on the owned planes, on the
        replicated levels and on the parameters."""
        rank, world, h = self.rank, self.world, self.h
        first, last = rank == 0, rank == world - 1
        fused = self._fused if update else dict()

        def tic(name):
            if timers is None:
                return None
            a, b = timers.section(name)
            a.record()
            return b

        def toc(b):
            if b is not None:
                b.record()

        if not (update and self.redundant and self._x_synced):
            # the neighbours' boundary planes of the unknowns -> inner ghost planes: every epoch, or once at the start
            # when the ghost planes are updated redundantly afterwards
            b = tic("halo")
            lo, hi = self._own["lo"], self._own["hi"]
            recv_lo, recv_hi = yield ("halo", None if lo is None else lo.pack(self.x, self._send_x["lo"]),
                                      None if hi is None else hi.pack(self.x, self._send_x["hi"]))
            if recv_lo is not None:
                self._ghost["lo"].unpack(self.x, recv_lo)
            if recv_hi is not None:
                self._ghost["hi"].unpack(self.x, recv_hi)
            self._x_synced = True
            toc(b)
        b = tic("mg_synth")
        self._synthesise()
        toc(b)
        if h:
            b = tic("halo")
            ufield = lambda key: self.u[key]
            if self._wrap_pack is not None and world == 1:
                # one rank: its high planes ARE what lies across the low end -- packed straight into place
                self._wrap_pack["hi"].pack(self._uflat, self._wrapbuf["lo"])
                self._wrap_pack["lo"].pack(self._uflat, self._wrapbuf["hi"])
                recv_lo = recv_hi = None
            elif self._wrap_pack is not None:
                recv_lo, recv_hi = yield ("wrap", self._wrap_pack["lo"].pack(self._uflat) if first else None,
                                          self._wrap_pack["hi"].pack(self._uflat) if last else None)
            else:
                recv_lo, recv_hi = yield ("wrap", self._end_planes(ufield, "lo") if first else None,
                                          self._end_planes(ufield, "hi") if last else None)
            if recv_lo is not None:
                self._wrapbuf["lo"].copy_(recv_lo.reshape(-1))
            if recv_hi is not None:
                self._wrapbuf["hi"].copy_(recv_hi.reshape(-1))
            toc(b)
        b = tic("forward")
        self.kern.forward(self.u, {k: w["lo"] for k, w in self.wrap.items()}, {k: w["hi"] for k, w in self.wrap.items()})
        toc(b)
        t = self.npdt(self.t + 1)
        alpha = self.lr * np.sqrt(1 - self.b2**t) / (1 - self.b1**t)
        hyper = (alpha, 1 - self.b1, 1 - self.b2, self.eps)
        b = tic("gather")
        merged = list(getattr(self.kern, "merged", []))
        spans = {fused[key] for key in merged if key in fused}
        if merged and len(spans) <= 1:  # one launch for all of them (the fused planes are the same for every field)
            items = []
            for key in merged:
                w, e = self.wrap[key], self.by_key[key]
                items.append((key, e["h"][0], w["glo"], w["ghi"],
                              (e["x"][0], e["m"][0], e["v"][0]) if key in fused else None))
            self.kern.gather_all(items, hyper if update else None, spans.pop() if spans else (0, 0))
        else:
            merged = []
        for key in self.kern.gather_keys:
            if key in merged:
                continue
            w, e = self.wrap[key], self.by_key[key]
            if key in fused:
                self.kern.gather(key, e["h"][0], w["glo"], w["ghi"],
                                 adam=(e["x"][0], e["m"][0], e["v"][0]) + hyper + fused[key])
            else:
                self.kern.gather(key, e["h"][0], w["glo"], w["ghi"])
        toc(b)
        if h and self.kern.gather_keys:
            b = tic("halo")
            recv_lo, recv_hi = yield ("wrap", self._wrapbuf["glo"] if first else None, self._wrapbuf["ghi"] if last else None)
            # what arrives from across the low end belongs to this rank's FIRST owned planes, and vice versa
            for recv, side in ((recv_lo, "lo"), (recv_hi, "hi")):
                if recv is None:
                    continue
                parts = None
                for item in self._wrap_add[side]:
                    if len(item) == 3:
                        parts = parts or self._split_planes(recv)
                        key, lv, first_plane = item
                        lv.planes(self.by_key[key]["h"][0], first_plane, h).add_(parts[key])
                    else:
                        plist, scaled = item
                        plist.unpack_add(self.hvec if scaled else self.g, recv.reshape(-1))
            toc(b)
        for key in self.kern.gather_keys:  # g_0 = f_0 h_0 (ghost planes included: this rank's share of the neighbour's)
            e = self.by_key[key]
            if e.get("factors"):
                torch.mul(e["h"][0], e["factors"][0], out=e["g"][0])
        # the gradients' halo-add in two messages: the finest levels' ghost planes are final now -- they are POSTED and
        # travel while the transposes run down the levels (which read this rank's partial arrays only); the coarser
        # levels' planes follow after the chain
        b = tic("halo")
        (own0, ghost0), (own1, ghost1) = self._g_split
        pack = lambda lists, bufs, side: None if lists[side] is None else lists[side].pack(self.g, bufs[side])
        token = yield ("post", pack(ghost0, self._send_g[0], "lo"), pack(ghost0, self._send_g[0], "hi"))
        toc(b)
        b = tic("mg_synth_adj")
        self._transpose_chain()
        toc(b)
        b = tic("halo")
        recv_lo, recv_hi = yield ("wait", token, None)
        if recv_lo is not None:
            own0["lo"].unpack_add(self.g, recv_lo)
        if recv_hi is not None:
            own0["hi"].unpack_add(self.g, recv_hi)
        recv_lo, recv_hi = yield ("halo", pack(ghost1, self._send_g[1], "lo"), pack(ghost1, self._send_g[1], "hi"))
        if recv_lo is not None:
            own1["lo"].unpack_add(self.g, recv_lo)
        if recv_hi is not None:
            own1["hi"].unpack_add(self.g, recv_hi)
        if self._rep and world > 1:  # agglomerated levels: the ranks' shares summed (every rank then updates alike)
            total = yield ("sum", torch.cat([self.g[a:a + c] for a, c in self._rep]), None)
            off = 0
            for a, c in self._rep:
                self.g[a:a + c].copy_(total[off:off + c])
                off += c
        toc(b)
        if self.has_params:
            total = yield ("sum", self.kern.pgrad.clone(), None)
            for key, (ofs, lens) in self.kern.param_groups.items():
                for view, cnt in zip(self.by_key[key]["g"], lens):
                    view.copy_(total[ofs:ofs + cnt].view(view.shape))
                    ofs += cnt
        if getattr(self.kern, "par_outputs", None) is not None:
            self.kern.launch_par(lambda key: self.by_key[key]["x"], lambda key: self.by_key[key]["g"])
        if not update:
            return
        self.t += 1
        b = tic("adam")
        if not self._fused:
            hip_ops.adam_step(self.x, self.m, self.v, self.g, *hyper)
        else:
            for lo, hi in self._post_flat:
                hip_ops.adam_step(self.x[lo:hi], self.m[lo:hi], self.v[lo:hi], self.g[lo:hi], *hyper)
            for start, size, pieces, stride, offset, count in self._post_pieces:
                span = slice(start, start + size)
                hip_ops.adam_step_pieces(self.x[span], self.m[span], self.v[span], self.g[span], pieces, stride, offset,
                                         count, *hyper)
        toc(b)

    def epoch(self, comm, timers=None):
        gen = self.epoch_gen(timers)
        try:
            msg = next(gen)
            while True:
                msg = gen.send(comm.exchange(*msg))
        except StopIteration:
            pass

    def last_terms(self, comm=None):
        """Global loss terms of the last evaluation (one small all-reduce)."""
        part = self.kern.partial_terms().clone()
        if comm is not None:
            part = comm.exchange("sum", part, None)
        terms = [float(v) for v in part]
        if getattr(self.kern, "par_outputs", None) is not None and hasattr(self.kern, "pout"):
            # parameter-space outputs are evaluated by every rank alike: counted ONCE (emulated ranks sum their
            # last_loss(): rank 0 carries them)
            for q, (k, _) in enumerate(self.kern.par_outputs):
                terms.insert(k, float(self.kern.pout[2 * q]) if self.rank == 0 or comm is not None else 0.0)
        return terms

    def last_loss(self, comm=None):
        return float(sum(self.last_terms(comm)))

    def owned_arrays(self):
        """This rank's part of every unknown array, in `Domain.arrays_from_state` order."""
        res = []
        for e in self.entries:
            if "levels" in e:
                res += [lv.owned(a) for lv, a in zip(e["levels"], e["x"])]
            else:
                res += list(e["x"])
        return res


def optimize_slab(args, problem, state, callback=None, axis=None):
    """`odil.util.optimize(args, "adam", ...)` for a run started with one process per GPU (e.g. `python -m
    torch.distributed.run --nproc-per-node 8 examples/velocity_from_tracer/veltracer3d.py --slab 1`): every rank
    builds the same GLOBAL problem, owns a slab of it (this module) and runs `args.epochs` Adam epochs (or, with
    `--optimizer lbfgsb`, that many L-BFGS-B iterations: slab_solvers.SlabTracedLbfgs); the loss
    terms are all-reduced and logged by rank 0 every `args.report_every` epochs (the quantity the reference's
    callback reports, src/odil/util.py:337-467).  `callback(run, epoch, terms)` is called on every rank at those
    epochs.  Returns the rank's SlabTracedAdam (its `owned_arrays()` are the rank's part of the solution; dump
    them with `odil.write_raw_slab`)."""
    from .slab import init_distributed
    from .util import printlog

    rank, world, comm = init_distributed()
    kw = {name: getattr(args, "adam_" + name) for name in ("beta_1", "beta_2", "epsilon")
          if getattr(args, "adam_" + name, None) is not None}
    run = SlabTracedAdam(problem, state, rank, world, axis=axis, lr=args.lr, **kw)
    every = getattr(args, "report_every", 0) or 0
    start = getattr(args, "epoch_start", 0)

    def report(epoch):
        terms = run.last_terms(comm)  # of the last evaluation
        if rank == 0:
            printlog("epoch={:05d} ranks={} loss={:.8g} terms={}".format(
                epoch, world, sum(terms), " ".join("{:.6g}".format(t) for t in terms)))
        if callback is not None:
            callback(run, epoch, terms)

    optname = getattr(args, "optimizer", "adam") or "adam"
    if optname == "lbfgsb":
        # L-BFGS-B on the slabs (slab_solvers.SlabTracedLbfgs): an "epoch" is an iteration, as in the undivided driver
        from .slab_solvers import SlabTracedLbfgs

        opts = {dst: getattr(args, src) for src, dst in (("bfgs_m", "m"), ("bfgs_pgtol", "pgtol"), ("bfgs_maxls", "maxls"))
                if getattr(args, src, None) is not None}
        count = [start]

        def each(_x):
            count[0] += 1
            if every and (count[0] % every == 0 or count[0] == args.epochs):
                report(count[0])

        SlabTracedLbfgs(run).minimize(comm, args.epochs - start, callback=each, **opts)
        return run
    if optname != "adam":
        raise NotImplementedError("optimizer '{}' on the slab decomposition (adam, lbfgsb)".format(optname))
    for epoch in range(start + 1, args.epochs + 1):
        run.epoch(comm)
        if every and (epoch % every == 0 or epoch == args.epochs):
            report(epoch)  # (the evaluation at the start of this epoch)
    return run


def shape_state(domain, state):
    """The structure `Domain.init_state` would give for `state`, with arrays on the 'meta' device (shapes only):
    what a rank needs of the GLOBAL state when the global arrays would not fit, or need not exist."""
    from .core import Array, Field, MultigridField, NeuralNet, State
    from .backend import torch_dtype

    dt = torch_dtype(domain.dtype)
    meta = lambda shape: torch.empty(tuple(shape), dtype=dt, device="meta")
    fields = dict()
    for key, f in state.fields.items():
        if f is None or isinstance(f, Field):
            loc = (f.loc if f is not None else None) or "c" * domain.ndim
            cshape = (f.cshape if f is not None else None) or domain.cshape
            if domain.multigrid and domain.mg_convert_all:
                terms = [Field(meta(domain._get_field_shape(cs, loc)), loc=loc, cshape=cs) for cs in domain.mg_cshapes]
                fields[key] = MultigridField(terms=terms, loc=loc, factors=domain.mg_factors, method=domain.mg_interp)
            else:
                fields[key] = Field(meta(domain._get_field_shape(cshape, loc)), loc=loc, cshape=cshape)
        elif isinstance(f, (NeuralNet, Array, MultigridField)):
            fields[key] = domain.init_field(f)
        else:
            raise TypeError(type(f).__name__)
    return State(fields=fields, initialized=True)
