"""HIP source generation for traced operators (see odil_amd/stencil_jit.py for the overview): the
forward + register-level reverse-mode kernel `k_fwd`, the per-field cotangent gathers `k_gat_*`, the
deterministic final reductions, and the hipcc / cache plumbing that turns the source into a loadable
shared object.
"""

import ctypes
import hashlib
import math
import os
import subprocess
import tempfile

import numpy as np
import torch

from . import stencil_grad
from .stencil_trace import _B, _CMP, _I, _R, TraceUnsupported, _promote

_CACHE_DIR = os.environ.get("ODIL_JIT_CACHE", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_jit_cache"))
_HIPCC_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "--offload-arch=gfx950"]
# float kernels (tolerance 2e-5, stated in the tests): contraction to FMA allowed -- the 5 x 5 layers of a pointwise
# network and their parameter-gradient accumulations become (packed) FMAs instead of multiply + add pairs
_HIPCC_FLAGS_F32 = [f if f != "-ffp-contract=off" else "-ffp-contract=fast" for f in _HIPCC_FLAGS]
_FAST_F32 = True


# ======================================================================================
# Code generation
# ======================================================================================
_PRELUDE = r"""
#include <hip/hip_runtime.h>
#include <stdint.h>
#define NB 256
typedef @T@ T;
#define FN(name) @FN@

__device__ inline T block_sum(T v, T* sm) {
  for (int off = 32; off > 0; off >>= 1) v = v + __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  T r = sm[0];
  for (int w = 1; w < NB / 64; ++w) r = r + sm[w];
  return r;
}
template <int NW>
__device__ inline T block_sum_w(T v, T* sm) {  // NW waves per workgroup
  for (int off = 32; off > 0; off >>= 1) v = v + __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  T r = sm[0];
  for (int w = 1; w < NW; ++w) r = r + sm[w];
  return r;
}
__device__ inline void wg_barrier() {  // (one per wave, wherever it stands in its code: the tiled forward kernel)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ inline int wrap(int j, int n) { return j < 0 ? j + n : (j >= n ? j - n : j); }
// (float kernels: the quotient and the square root of the update through v_rcp_f32 / v_sqrt_f32,
// ~1 ulp each, as every other division of those kernels; the IEEE forms are ~10 instructions each, sixteen of each per
// thread in a merged gather)
#ifdef ODIL_FAST_F32
#define ADAM_QUOT(a, b) ((a) * __builtin_amdgcn_rcpf(b))
#define ADAM_SQRT(a) __builtin_amdgcn_sqrtf(a)
#else
#define ADAM_QUOT(a, b) ((a) / (b))
#define ADAM_SQRT(a) FN(sqrt)(a)
#endif
// Adam of the array a gather forms the gradient of, applied by the lane that holds g[l] (reference
// src/odil/optimizer.py:316-318; x == NULL: gradient only).  alpha_dev != NULL: the step size is read from
// device memory (epochs replayed as a hipGraph).
struct AdamP { T* x; T* m; T* v; T alpha, omb1, omb2, eps; const T* alpha_dev; };
__device__ inline void adam_apply(const AdamP& ad, int l, T g) {
  if (!ad.x) return;
  T m = ad.m[l], v = ad.v[l], x = ad.x[l];
  m = m + (g - m) * ad.omb1;
  v = v + (g * g - v) * ad.omb2;
  const T alpha = ad.alpha_dev ? *ad.alpha_dev : ad.alpha;
  x = x - ADAM_QUOT(m * alpha, ADAM_SQRT(v) + ad.eps);
  ad.m[l] = m, ad.v[l] = v, ad.x[l] = x;
}
// Four consecutive points of the last axis per thread (VW == 4): 16-byte accesses (two for doubles).  The type is
// aligned like its element: rows and arrays packed back to back in one buffer need not start on 16 bytes, and
// gfx950 takes unaligned dwordx4 accesses.
typedef T T4a __attribute__((ext_vector_type(4)));
typedef T4a T4 __attribute__((aligned(sizeof(T))));
__device__ inline void adam_apply4(const AdamP& ad, int l, const T* g) {
  if (!ad.x) return;
#ifdef ODIL_NT_STREAMS  // the optimizer state streams through once per epoch: kept out of the way of the rows the kernel re-reads
  T4 m = __builtin_nontemporal_load((const T4*)(ad.m + l)), v = __builtin_nontemporal_load((const T4*)(ad.v + l));
  T4 x = __builtin_nontemporal_load((const T4*)(ad.x + l));
#else
  T4 m = *(const T4*)(ad.m + l), v = *(const T4*)(ad.v + l), x = *(const T4*)(ad.x + l);
#endif
  const T alpha = ad.alpha_dev ? *ad.alpha_dev : ad.alpha;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    m[p] = m[p] + (g[p] - m[p]) * ad.omb1;
    v[p] = v[p] + (g[p] * g[p] - v[p]) * ad.omb2;
    x[p] = x[p] - ADAM_QUOT(m[p] * alpha, ADAM_SQRT(v[p]) + ad.eps);
  }
#ifdef ODIL_NT_STREAMS
  __builtin_nontemporal_store(m, (T4*)(ad.m + l));
  __builtin_nontemporal_store(v, (T4*)(ad.v + l));
  __builtin_nontemporal_store(x, (T4*)(ad.x + l));
#else
  *(T4*)(ad.m + l) = m, *(T4*)(ad.v + l) = v, *(T4*)(ad.x + l) = x;
#endif
}
// Float transcendentals of the float kernels.  The library tanhf is ~30 instructions with two divergent branches;
// a traced operator with a pointwise network evaluates it 40 times per grid point (heat with two space dimensions:
// 1660 VALU instructions per point, 1200 of them tanh -- profiles/r02_v0_heat2d_pmc.txt).  Here tanh(x) =
// 1 - 2 / (e^2x + 1) through the hardware exp2 / rcp: 5 instructions, ABSOLUTE error <= ~2e-7 (a float rounding of
// the O(1) sums the activations enter; the relative error grows as |x| -> 0, where a series branch would cost as
// much again -- the float kernels are held to 2e-5 against the float64 fixtures, stated in the tests).  exp and
// division likewise through v_exp_f32 / v_rcp_f32 (~2 ulp).
__device__ inline float odil_fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ inline float odil_fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ inline float odil_fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.88539008177792681472f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
// Value held by the previous / next lane of the wavefront (wave-wide DPP shift, no LDS); lanes without a source get 0.
__device__ inline float odil_lane_prev(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, false));  // wave_shr:1
}
__device__ inline float odil_lane_next(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, false));  // wave_shl:1
}
typedef float T2 __attribute__((ext_vector_type(2)));
__device__ inline T2 odil_fast_tanh2(T2 x) {  // two lanes: the arithmetic packs, exp2 / rcp are per lane
  const T2 s = x * 2.88539008177792681472f;
  T2 e = {__builtin_amdgcn_exp2f(s.x), __builtin_amdgcn_exp2f(s.y)};
  e = e + 1.0f;
  const T2 r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
  return 1.0f - 2.0f * r;
}
// tanh of USER expressions in float kernels: the same exp2 / rcp form away from zero, the odd series below |x| = 0.1
// (1 - 2 / (e^2x + 1) cancels there: absolute error 1e-7 is a relative error of 1e-3 at |x| = 1e-4; the series'
// next term, 17 x^6 / 315, is 5e-8 relative at 0.1).  The hidden-layer activations of pointwise networks keep the
// bare form above: their arguments are O(1) sums and their values enter O(1) sums.
__device__ inline float odil_tanh_f32(float x) {
  const float ax = __builtin_fabsf(x), x2 = x * x;
  const float e = __builtin_amdgcn_exp2f(ax * 2.88539008177792681472f);
  const float far = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  const float near = ax * (1.0f + x2 * (-0.33333333333333333f + x2 * 0.13333333333333333f));
  return __builtin_copysignf(ax < 0.1f ? near : far, x);
}
__device__ inline T2 odil_step2(T2 h) { T2 r = {h.x > 0.0f ? 1.0f : 0.0f, h.y > 0.0f ? 1.0f : 0.0f}; return r; }
"""


def _lit(value, kind):
    if kind == _B:
        return "true" if value else "false"
    if kind == _I:
        return "{}L".format(int(value))
    v = float(value)
    if math.isnan(v):
        return "((T)NAN)"
    if math.isinf(v):
        return "((T)INFINITY)" if v > 0 else "((T)-INFINITY)"
    return "((T){!r})".format(v)


class _Codegen:
    def __init__(self, tr, outputs, raw, shape, state, slab=None):
        """slab = (axis, n): the kernels of ONE RANK of a slab decomposition along grid axis `axis` (n owned
        cells of it per rank; odil_amd/slab_traced.py).  Threads then cover the owned cells only; the index of
        that axis seen by index leaves, constant arrays and windows is the GLOBAL one (i + a.off); sources are
        the rank's ghost-extended arrays (a.lo ghost cells below the owned ones, a.ea cells in all along the
        axis), reads that leave the global grid come from the wrap planes a.wlo / a.whi (periodic roll of
        `Context.field`, reference core.py:962-963, across the ends of the decomposition); the gathers write
        the ghost-extended gradient, ghost cells receiving what belongs to the neighbour."""
        self.tr, self.outputs, self.raw, self.G, self.state = tr, outputs, raw, tuple(shape), state
        self.ndim = len(shape)
        self.fast = _FAST_F32 and tr.torch_dtype == torch.float32
        self.flags = _HIPCC_FLAGS_F32 if self.fast else _HIPCC_FLAGS
        self.slab = slab
        self.GL = tuple(shape)  # the grid the threads cover
        if slab is not None:
            ax, n = slab
            if shape[ax] % n:
                raise TraceUnsupported("slab of {} cells on an axis of {}".format(n, shape[ax]))
            self.GL = tuple(n if d == ax else g for d, g in enumerate(shape))
            self.halo = 0
        self.total = int(np.prod(self.GL))
        # element offsets are 32-bit in the generated kernels: the grid AND the largest array on it (a node-centred axis
        # adds a layer) must stay below 2^31 elements; larger problems keep the autograd path
        largest = self.total
        if slab is not None:  # (the state carries the GLOBAL shapes there: the rank's arrays are its slab plus ghost planes)
            largest = int(np.prod([(g + 5) if d == slab[0] else (g + 1) for d, g in enumerate(self.GL)]))
        else:
            for f in (state.fields.values() if state is not None else ()):
                for t in ([getattr(f, "array", None)] + [getattr(x, "array", None) for x in getattr(f, "terms", [])]):
                    if torch.is_tensor(t):
                        largest = max(largest, int(t.numel()))
        if largest >= 2**31 - 1024:
            raise TraceUnsupported("grid too large for 32-bit indexing")
        self.lines = []
        self.max_blocks = 0  # 0: chosen by the operator (stencil_jit)
        # reachable nodes
        live = set()
        stack = list(outputs)
        while stack:
            n = stack.pop()
            if n.idx in live:
                continue
            live.add(n.idx)
            stack.extend(n.args)
        self.order = [n for n in tr.nodes if n.idx in live]
        for n in self.order:
            if n.op in ("read", "index") and tuple(n.shape) != self.G:
                raise TraceUnsupported("{} of shape {} on grid {}".format(n.op, n.shape, self.G))
        # host scalars consumed by device nodes
        self.hs = []
        hs_slot = dict()
        for n in self.order:
            if n.host:
                continue
            for a in n.args:
                if a.host and a.op != "const" and a.idx not in hs_slot:
                    hs_slot[a.idx] = len(self.hs)
                    self.hs.append(a)
        for o in outputs:
            if o.host:
                raise TraceUnsupported("scalar output")
        self.hs_slot = hs_slot
        # sources (regular arrays of fields) and load slots
        self.src_keys = []
        self.loads = dict()  # (key, shift, loc) -> variable
        self.cots = []  # live read nodes that receive a cotangent
        self.cut_nodes = []  # affine sub-expressions whose adjoint is stored instead of their reads' cotangents
        self.nets = []  # (key, layers) with parameter pointers
        self.net_slot = dict()
        self.arrays = []  # (key, numel) of `Array` unknowns read through a[k]
        self.array_slot = dict()
        self.need = self._needs_grad()
        self.partner, self.pair_first = dict(), set()
        self.share, self.shared_A, self.shared_B = [], set(), set()
        self._choose_shared_calls()
        if self.fast:
            self._pair_mlps()
        # per output: None (the whole grid) or the lens of its window; the mean runs over that many points
        self.out_lens = [None if o.win is None else tuple(o.win[0]) for o in outputs]
        self.out_count = [int(np.prod(l)) if l is not None else int(np.prod(self.G)) for l in self.out_lens]  # GLOBAL counts
        # four points of the last axis per thread, 16-byte accesses
        last = self.ndim - 1
        can_vec = self.GL[last] % 4 == 0 and self.GL[last] >= 8 and (slab is None or slab[0] != last)
        vec = "auto"  # (frozen: "0" / "1" force one / four points per thread in tests of the generator)
        # a forward kernel with a pointwise network is bound by its arithmetic, not by its loads: four points per thread
        # only cost it registers (heat with two space dimensions: 2.0 -> 2.5 ms)
        has_net = any(n.op == "mlp" for n in self.order)
        vec_fwd = vec
        self.vw_gat = 4 if can_vec and vec != "0" else 1
        self.vw_fwd = 4 if can_vec and (vec_fwd == "1" or (vec_fwd == "auto" and not has_net)) else 1
        self.vw = self.vw_fwd
        self.gloc = next(n.attr[2] for n in self.order if n.op == "read")
        self.in_gather = False  # emitting a gather (slab mode: threads cover the ghost planes too)
        self.fold = None  # emitting the interior copy of a body: {predicate node idx: constant value} (_fold_plan)
        self.wregs = "mem"  # where the marching kernel keeps network parameters (source())
        self.march_pref = None  # marching kernel: {read descriptor: (slot, load expression)} of the reads requested a step ahead
        self.march_live, self.march_used = None, set()  # ... nodes the variant being emitted really evaluates; slots it reads
        self.mlp_out_seen = dict()  # network call idx -> {output index: mlp_out node}
        self.pseudo_slot = dict()  # "@..." pseudo-field of a stored adjoint array -> its slot in a.cot
        self.out_mode = self._choose_output_cuts()
        self.cut_set = self._choose_cuts()

    def _needs_grad(self):
        need = dict()
        for n in self.order:
            if n.op == "read":
                need[n.idx] = not n.attr[3]
            elif n.op == "aparam":
                need[n.idx] = not n.attr[2]
            elif n.op == "stopgrad" or n.kind != _R or n.host:
                need[n.idx] = False
            elif n.op == "mlp":
                need[n.idx] = (not n.attr[1]) or any(need[a.idx] for a in n.args)
            else:
                need[n.idx] = any(need[a.idx] for a in n.args)
        return need

    # ---- cotangents cut at linear sub-expressions ---------------------------------------------
    def _affine(self, n, memo):
        """{read idx: coefficient expression} when `n` is a linear combination of live reads with
        coefficients that are scalars known to every kernel (literals, host scalars); {} for values
        that do not depend on live reads; None otherwise."""
        if n.idx in memo:
            return memo[n.idx]
        res = None
        if not self.need.get(n.idx, False):
            res = dict()
        elif n.op == "read":
            res = {n.idx: "(T)1"}
        elif n.op in ("win", "cast"):
            res = self._affine(n.args[0], memo)
        elif n.op == "neg":
            a = self._affine(n.args[0], memo)
            res = None if a is None else {k: "-({})".format(c) for k, c in a.items()}
        elif n.op in ("add", "sub"):
            a, b = self._affine(n.args[0], memo), self._affine(n.args[1], memo)
            if a is not None and b is not None:
                res = dict(a)
                for k, c in b.items():
                    c = c if n.op == "add" else "-({})".format(c)
                    res[k] = "({} + {})".format(res[k], c) if k in res else c
        elif n.op == "mul":
            for x, y in (n.args, n.args[::-1]):
                a = self._affine(y, memo)
                if x.host and x.kind != _B and a is not None:
                    res = {k: "({} * {})".format(c, self.r(x)) for k, c in a.items()}
                    break
        elif n.op == "div":
            a = self._affine(n.args[0], memo)
            if n.args[1].host and a is not None:
                res = {k: "({} / {})".format(c, self.r(n.args[1])) for k, c in a.items()}
        memo[n.idx] = res
        return res

    def _choose_cuts(self, modes=None):
        """Nodes whose adjoint is stored as ONE array for all the reads below them: a regulariser
        `lap = (q+ - 2 q + q-) / h^2 + ...` needs one array, not one per stencil point.  A node is cut when
        it is linear in at least two live reads and somebody non-linear (or the loss) consumes it."""
        modes = self.out_mode if modes is None else modes
        memo = self.__dict__.setdefault("_affine_memo", dict())
        consumers = self.__dict__.get("_consumers")
        if consumers is None:
            consumers = self._consumers = dict()
            for n in self.order:
                for a in n.args:
                    consumers.setdefault(a.idx, []).append(n)
        outputs = {o.idx for o in self.outputs}
        # nodes only the recomputed outputs reach are never visited by k_fwd's reverse pass
        legacy_live, stack = set(), [o for k, o in enumerate(self.outputs) if modes[k] == "legacy"]
        while stack:
            n = stack.pop()
            if n.idx not in legacy_live:
                legacy_live.add(n.idx)
                stack.extend(n.args)
        cuts = dict()
        for n in self.order:
            form = self._affine(n, memo)
            if n.op == "read" or not form or len(form) < 2 or n.kind != _R or n.idx not in legacy_live:
                continue
            outside = n.idx in outputs or any(self._affine(c, memo) is None for c in consumers.get(n.idx, []))
            if outside:
                cuts[n.idx] = form
        # a cut only pays when it makes read cotangents disappear: drop cuts greedily while that lowers
        # the number of stored arrays (ties: fewer cuts)
        best = len(self._stored_nodes(cuts, modes))
        improved = True
        while cuts and improved:
            improved = False
            for idx in list(cuts):
                trial = {k: v for k, v in cuts.items() if k != idx}
                count = len(self._stored_nodes(trial, modes))
                if count <= best:
                    cuts, best, improved = trial, count, True
                    break
        return cuts

    # ---- cotangents cut at the OUTPUTS, local Jacobians re-evaluated by the gathers (stencil_grad.py) ---------
    def _regular(self, n):
        """A read the symbolic gathers can shift: same location and shape as its field, on the grid itself."""
        key, _, loc, _ = n.attr
        return loc == self.state.fields[key].loc and tuple(self._field_shape(key)) == self.G

    def _choose_output_cuts(self):
        """Per output: 'legacy' (k_fwd differentiates it down to its reads / affine cuts and stores those cotangents),
        'jac' (its adjoint -- the seed 2 f / n -- is stored ONCE; every gather re-evaluates d f / d read at the
        neighbouring points from the source fields) or 'virt' (nothing stored: the gathers re-evaluate the output
        itself too; for outputs that read a few values -- imposed values, a time difference)."""
        nout = len(self.outputs)
        mode = ["legacy"] * nout
        self.seed_key = [None] * nout
        self.out_adj = [None] * nout  # jac / virt: {read idx: adjoint expression at the point of evaluation}
        # pad / trim reads ('c' fields read at 'n' and back) keep the legacy gathers: one such read anywhere and
        # the whole operator stays with them
        self.all_regular = all(self._regular(n) for n in self.order if n.op == "read")
        if not int(os.environ.get("ODIL_TRACE_RECOMPUTE", 1)) or not self.all_regular:
            return mode
        virt_max = 12  # (64 -- the tracer's Laplacian regularisers re-evaluated by the gathers, 4 -> 1 stored arrays -- was measured in round 5: k_fwd 2.71 -> 2.19 ms, merged gather 7.26 -> 13.96 ms)
        used = {a.idx for n in self.order for a in n.args}
        self.tr.state_locs = dict(getattr(self.tr, "state_locs", dict()))
        out_ids = [o.idx for o in self.outputs]
        for k, o in enumerate(self.outputs):
            if not self.need.get(o.idx, False) or o.idx in used or out_ids.count(o.idx) != 1:
                continue
            nodes = stencil_grad.subdag(o)
            live = [n for n in nodes if n.op == "read" and self.need.get(n.idx, False)]
            if not live or not stencil_grad.differentiable(nodes, self.need) or not all(self._regular(n) for n in live):
                continue
            gb = stencil_grad.GradBuilder(self.tr, self.G, self.need, stop=())
            loads, heavy = stencil_grad.cost([n for n in nodes if not n.host])
            shifts = {(n.attr[0], n.attr[1]) for n in live}
            virt = heavy == 0 and loads * len(shifts) <= virt_max
            scale = (1.0 if self.raw[k] else 2.0) / self.out_count[k]
            try:
                if virt:
                    seed = gb.const(scale) if self.raw[k] else gb.mul(gb.real(o), gb.const(scale))
                else:
                    key = "@o{}".format(k)
                    self.tr.state_locs[key] = self.gloc
                    seed = self.tr.node("read", attr=(key, (0,) * self.ndim, self.gloc, True), shape=self.G, kind=_R)
                if virt and self.out_lens[k] is not None:
                    inbox = None
                    for d in range(self.ndim):
                        if self.out_lens[k][d] < self.G[d]:
                            idx = self.tr.node("index", attr=(d, self.gloc), shape=self.G, kind=_I)
                            c = gb.cmp("lt", idx, self.tr.const(int(self.out_lens[k][d])))
                            inbox = c if inbox is None else gb._node("and", (inbox, c), kind=_B)
                    if inbox is not None:
                        seed = gb.where(inbox, seed, None)
                adj = gb.adjoints(o, seed, nodes)
            except TraceUnsupported:
                continue
            if not virt:
                # a Jacobian that needs transcendentals at every neighbour costs more than the arrays it saves
                extra = set()
                for e in adj.values():
                    extra.update(n.idx for n in stencil_grad.subdag(e) if not n.host)
                jheavy = stencil_grad.cost([self.tr.nodes[i] for i in extra])[1]
                if jheavy > 0:
                    continue
            mode[k] = "virt" if virt else "jac"
            self.seed_key[k] = None if virt else key
            self.out_adj[k] = adj
        # ---- which of the candidates pay: words per grid point written by k_fwd and read by the gathers ---------
        cand = [k for k in range(nout) if mode[k] != "legacy"]
        if not cand:
            return mode
        total = float(np.prod(self.G))

        def arrays_of(expr):
            res = dict()
            for n in stencil_grad.subdag(expr):
                if n.op == "read":
                    res[("arr", n.attr[0])] = 1.0
                elif n.op in ("tensor", "rtensor"):
                    slot = n.attr if n.op == "tensor" else n.attr[0]
                    res[("ten", slot)] = self.tr.tensors[slot].numel() / total
            return res

        per_out = dict()  # candidate -> {field: {array: words}}
        for k in cand:
            per_out[k] = dict()
            for ridx, expr in self.out_adj[k].items():
                per_out[k].setdefault(self.tr.nodes[ridx].attr[0], dict()).update(arrays_of(expr))

        def traffic(on):
            modes = [mode[k] if k in on else "legacy" for k in range(nout)]
            cuts = self._choose_cuts(modes)
            stored = self._stored_nodes(cuts, modes)
            words = len(stored) + sum(1 for k in on if mode[k] == "jac")  # written once by k_fwd
            reads = dict()
            for n in stored:
                keys = {n.attr[0]} if n.op == "read" else {self.tr.nodes[r].attr[0] for r in cuts[n.idx]}
                for key in keys:
                    reads.setdefault(key, dict())[("arr", "@{}".format(n.idx))] = 1.0
            for k in on:
                for key, arrs in per_out[k].items():
                    reads.setdefault(key, dict()).update(arrs)
            return words + sum(sum(arrs.values()) for arrs in reads.values())

        def descend(on):
            best = traffic(on)
            while True:
                trials = [(traffic(on ^ {k}), k) for k in cand]
                cost, k = min(trials)
                if cost >= best - 1e-9:
                    return best, on
                best, on = cost, on ^ {k}

        if int(os.environ.get("ODIL_TRACE_RECOMPUTE_ALL", 0)):  # (tests: every output that can be cut is, whatever the model says)
            chosen = set(cand)
        else:
            (c0, on0), (c1, on1) = descend(frozenset()), descend(frozenset(cand))
            chosen = on1 if c1 < c0 - 1e-9 else on0
        self.traffic_words = dict(legacy=traffic(frozenset()), chosen=traffic(frozenset(chosen)))
        for k in cand:
            if k not in chosen:
                mode[k], self.seed_key[k], self.out_adj[k] = "legacy", None, None
        return mode

    def _stored_nodes(self, cuts, modes=None):
        """The nodes (live reads, cut nodes) whose adjoint k_fwd's reverse pass stores for a given set of cut nodes."""
        modes = self.out_mode if modes is None else modes
        reached = {o.idx for k, o in enumerate(self.outputs) if self.need.get(o.idx, False) and modes[k] == "legacy"}
        stored = []
        for n in reversed(self.order):
            if n.idx not in reached:
                continue
            if n.idx in cuts or n.op == "read":
                stored.append(n)
                continue
            if n.op in ("stopgrad", "floor", "tensor", "index", "aparam"):
                continue
            for a in n.args:
                if self.need.get(a.idx, False):
                    reached.add(a.idx)
        return stored

    # ---- index predicates decided once per wave: an interior copy of the body -------------------------------------
    # Wall masks (`idx == 0`, `idx == n - 1`, `it == 0`), the wrap of a shifted index and windows are functions of ONE
    # grid index.  Evaluated over that axis they are true (or false) on a handful of values only; everywhere else --
    # > 98 % of the points of the BASELINE grids -- every `where` they steer takes the same branch, and the ghost
    # extrapolations, initial-row selects and their reverse pass are dead code.  The body of a kernel is therefore
    # emitted twice: as it is, and with those predicates replaced by their majority value (the compiler folds the
    # selects and drops what only the other branch needed, loads included); a wave takes the second copy when none of
    # its points has an exceptional index (`__all`: one scalar branch).  Predicates of the last axis are folded only by
    # kernels with one point per thread (a wave of a four-point kernel spans a row of 256: always at a wall).
    _FOLD_MAX = 8

    def _index_values(self, n, memo):
        """(axis or None, values along that axis) when `n` depends on at most one grid index and literals, else None."""
        if n.idx in memo:
            return memo[n.idx]
        res = None
        op, A = n.op, n.args
        if op == "const":
            res = (None, np.asarray(n.attr))
        elif op == "index":
            res = (n.attr[0], np.arange(self.G[n.attr[0]], dtype=np.int64))
        elif not n.host and op in ("cast", "win", "neg", "not", "abs") and len(A) == 1:
            a = self._index_values(A[0], memo)
            if a is not None:
                f = {"cast": lambda x: np.asarray(x, np.float64), "win": lambda x: x, "neg": np.negative,
                     "not": np.logical_not, "abs": np.abs}[op]
                res = (a[0], f(a[1]))
        elif not n.host and (op in ("add", "sub", "mul", "min", "max", "and", "or", "where") or op in _CMP):
            parts = [self._index_values(a, memo) for a in A]
            axes = {p[0] for p in parts if p is not None and p[0] is not None}
            if all(p is not None for p in parts) and len(axes) <= 1:
                f = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "min": np.minimum, "max": np.maximum,
                     "and": np.logical_and, "or": np.logical_or, "where": np.where, "lt": np.less, "le": np.less_equal,
                     "gt": np.greater, "ge": np.greater_equal, "eq": np.equal, "ne": np.not_equal}[op]
                res = (axes.pop() if axes else None, f(*[p[1] for p in parts]))
        memo[n.idx] = res
        return res

    def _fold_plan(self, nodes, vw, windows=False):
        """(fold, exc, inbox) or None: fold = {predicate node idx: its value away from the exceptional indices},
        exc = {axis: sorted exceptional index values}, inbox = {(output, axis)} window tests that hold in the interior."""
        last, memo = self.ndim - 1, dict()
        fold, exc, inbox = dict(), dict(), set()
        for n in nodes:
            if n.kind != _B or n.host or n.op == "const":
                continue
            r = self._index_values(n, memo)
            if r is None or r[0] is None or (r[0] == last and vw != 1):
                continue
            d = r[0]
            vals = np.broadcast_to(np.asarray(r[1], dtype=bool), (self.G[d],))
            common = bool(2 * int(vals.sum()) > vals.size)
            minority = np.nonzero(vals != common)[0]
            if minority.size <= self._FOLD_MAX and 4 * minority.size <= self.G[d]:
                fold[n.idx] = common
                exc.setdefault(d, set()).update(int(i) for i in minority)
        if windows:
            for k, lens in enumerate(self.out_lens):
                for d in range(self.ndim if lens is not None else 0):
                    cut = self.G[d] - lens[d]
                    if 0 < cut <= self._FOLD_MAX and 4 * cut <= self.G[d] and (d != last or vw == 1):
                        inbox.add((k, d))
                        exc.setdefault(d, set()).update(range(lens[d], self.G[d]))
        if not any(exc.values()):
            return None
        return fold, {d: sorted(v) for d, v in exc.items() if v}, inbox

    def _live_under(self, fold):
        """Indices of the nodes the outputs depend on once the predicates of `fold` have their constant values: a `where`
        with a decided condition reaches only the branch it takes."""
        live, stack = set(), list(self.outputs)
        while stack:
            n = stack.pop()
            if n.idx in live:
                continue
            live.add(n.idx)
            if n.op == "where" and n.args[0].idx in fold:
                stack.append(n.args[1] if fold[n.args[0].idx] else n.args[2])
            elif n.idx not in fold:
                stack.extend(n.args)
        return live

    def _interior_cond(self, exc):
        """C expression: none of the thread's indices is exceptional."""
        conds = []
        for d, values in sorted(exc.items()):
            i, n, values = self.gi(d), self.G[d], list(values)
            lo = 0
            while values and values[0] == lo:
                values.pop(0)
                lo += 1
            hi = n - 1
            while values and values[-1] == hi:
                values.pop()
                hi -= 1
            if lo > 0:
                conds.append("{} >= {}".format(i, lo))
            if hi < n - 1:
                conds.append("{} <= {}".format(i, hi))
            conds.extend("{} != {}".format(i, e) for e in values)
        return " && ".join(conds)

    def _inbox_lines(self, inbox=()):
        out = []
        for k, lens in enumerate(self.out_lens):
            if lens is not None:
                conds = ["{} < {}".format(self.gi(d), lens[d]) for d in range(self.ndim)
                         if lens[d] < self.G[d] and (k, d) not in inbox]
                out.append("  const bool inbox{} = {};".format(k, " && ".join(conds) or "true"))
        return out

    def _interior_copy(self, nodes, vw, reverse=False):
        """Second emission of the body whose first emission just ended (self.pre / self.groups hold its row loads), with
        the index predicates folded: ((forward lines, reverse lines), plan) or None when nothing folds."""
        plan = self._fold_plan(nodes, vw, windows=reverse)
        if plan is None:
            return None
        keep = ("cots", "cut_nodes", "jac_store", "pg_decl", "pg_offset", "pgrads", "pg2_used")
        saved = {k: getattr(self, k) for k in keep if hasattr(self, k)}
        if reverse:
            self.cots, self.cut_nodes, self.jac_store, self.pg_decl, self.pg_offset, self.pg2_used = [], [], [], [], dict(), set()
        self.fold, self.lines, self.loads = plan[0], [], dict()
        self.forward()
        fwd, self.lines, rev = self.lines, [], []
        if reverse:
            self.reverse()
            rev, self.lines = self.lines, []
            same = ([n.idx for n in self.cots] == [n.idx for n in saved["cots"]] and self.pg_decl == saved["pg_decl"]
                    and [n.idx for n in self.cut_nodes] == [n.idx for n in saved["cut_nodes"]] and self.jac_store == saved["jac_store"])
            if not same:
                raise RuntimeError("interior copy of the traced kernel stores other adjoints than the general one")
        self.fold = None
        for k, v in saved.items():
            setattr(self, k, v)
        self.pre = list(dict.fromkeys(self.pre))  # (row loads are requested again by the second emission)
        return (fwd, rev), plan

    # ---- expressions ----------------------------------------------------------------------
    def ex(self, n):
        if n.op == "const":
            return _lit(n.attr, n.kind)
        if n.host:
            if n.idx not in self.hs_slot:  # (host scalars of the gradient expressions: met after the forward pass)
                self.hs_slot[n.idx] = len(self.hs)
                self.hs.append(n)
            e = "HS({})".format(self.hs_slot[n.idx])
            return {"r": "((T){})", "i": "((long){})", "b": "({} != 0.0)"}[n.kind].format(e)
        return "v{}".format(n.idx)

    def r(self, n):
        return self.ex(n) if n.kind == _R else "((T){})".format(self.ex(n))

    def i(self, n):
        return self.ex(n) if n.kind == _I else "((long){})".format(self.ex(n))

    def over(self, num, den):
        """`num / den` -- as a multiplication when `den` is a constant power of two (grid steps of 2^k cells:
        bit-identical, and a float division is ~10 instructions, twenty of them per point in the tracer
        operator)."""
        if den.op == "const" and den.kind == _R:
            value = float(den.attr)
            if value != 0.0 and math.isfinite(value):
                mant, exp = math.frexp(abs(value))
                if mant == 0.5 and -100 < exp < 100:
                    return "{} * {}".format(num, _lit(1.0 / value, _R))
                if self.fast:
                    return "{} * {}".format(num, _lit(1.0 / value, _R))
        if self.fast:
            return "odil_fast_div({}, {})".format(num, self.r(den))
        return "{} / {}".format(num, self.r(den))

    def b(self, n):
        return self.ex(n) if n.kind == _B else "({} != 0)".format(self.ex(n))

    def typed(self, n, kind):
        return {"r": self.r, "i": self.i, "b": self.b}[kind](n)

    def emit(self, s):
        self.lines.append("  " + s)

    def _src_slot(self, key):
        if key not in self.src_keys:
            self.src_keys.append(key)
        return self.src_keys.index(key)

    def _field_shape(self, key):
        return self.tr.domain.get_field_shape(self.state.fields[key].loc)

    def _offset(self, idx_exprs, shape):
        e = idx_exprs[0]
        for d in range(1, len(shape)):
            e = "({} * {} + {})".format(e, shape[d], idx_exprs[d])
        return e

    # ---- forward ----------------------------------------------------------------------------
    def _source_of(self, key):
        """(pointer expression, array shape, location) of a field's regular array, or of a stored adjoint array
        ('@...' pseudo-fields of the gradient expressions: on the grid itself)."""
        if key.startswith("@"):
            return "a.cot[{}]".format(self.pseudo_slot[key]), self.GL, self.gloc
        return "a.src[{}]".format(self._src_slot(key)), self._field_shape(key), self.state.fields[key].loc

    def _begin(self):
        """Fresh emission state of one kernel body."""
        self.lines, self.pre, self.loads, self.groups = [], [], dict(), dict()

    def _emit_read(self, n):
        key, shift, loc, _ = n.attr
        desc = (key, shift, loc)
        if desc in self.loads:
            self.emit("const T v{} = {};".format(n.idx, self.loads[desc]))
            return
        ptr, fshape, floc = self._source_of(key)
        last = self.ndim - 1
        ax = self.slab[0] if self.slab is not None else None
        idx, zero, pre_lines = [], [], []
        fast = self.vw == 4 and floc[last] == loc[last] and fshape[last] == self.G[last]
        s_last, sa = 0, 0
        for d in range(self.ndim):
            ns = fshape[d]
            ext = max(ns, self.G[d])  # extent of the padded / untrimmed array the roll acts on
            s = shift[d] % ext
            if s > ext // 2:
                s -= ext
            if d == ax:
                # the sharded axis of a slab rank: no wrap inside the rank -- cell i reads cell i + s of the
                # ghost-extended array, or a wrap plane where i + s falls off the GLOBAL grid (_address)
                if floc[d] != "c" or loc[d] != "c":
                    raise TraceUnsupported("slab axis {} must be cell-centred for field '{}'".format(ax, key))
                sa = s
                if not self.in_gather:
                    self.halo = max(self.halo, abs(s))
                idx.append(None)
                continue
            if d == last and fast and abs(s) <= 1:
                s_last = s
                idx.append("0")
                continue
            if d == last:
                fast = False
            j = "i{}".format(d) if s == 0 else "wrap(i{} + ({}), {})".format(d, s, ext)
            if floc[d] == "c" and loc[d] == "n":  # zero padded at the low end
                name = "p{}_{}".format(n.idx, d)
                pre_lines.append("const int {} = {};".format(name, j))
                zero.append("{} == 0".format(name))
                j = "({} == 0 ? 0 : {} - 1)".format(name, name)
            idx.append(j)
        if not fast:
            e = "*({})".format(self._address(key, ptr, fshape, idx, sa))
            if self.march_pref is not None and not zero and not pre_lines and (self.march_live is None or n.idx in self.march_live):
                # marching kernel: the value was requested one step ahead (_march_kernel)
                slot = self.march_pref.setdefault(desc, (len(self.march_pref), e))[0]
                self.march_used.add(slot)
                self.emit("const T v{} = ld_{};".format(n.idx, slot))
                self.loads[desc] = "v{}".format(n.idx)
                return
            for line in pre_lines:
                self.emit(line)
            if zero:
                e = "(({}) ? (T)0 : {})".format(" || ".join(zero), e)
            self.emit("const T v{} = {};".format(n.idx, e))
            self.loads[desc] = "v{}".format(n.idx)
            return
        # four points per thread: the row of this (field, shift on the other axes) is loaded once as a 16-byte pack
        # plus the element left / right of it where a read of the row is shifted along the last axis
        gkey = (key, tuple(shift[:last]), loc)
        grp = self.groups.get(gkey)
        if grp is None:
            gid = len(self.groups)
            rename = lambda text: text.replace("p{}_".format(n.idx), "pg{}_".format(gid))
            grp = self.groups[gkey] = dict(id=gid, left=False, right=False, zero=rename(" || ".join(zero)))
            for line in pre_lines:
                self.pre.append("  " + rename(line))
            self.pre.append("  const T* const R{} = {};".format(gid, rename(self._address(key, ptr, fshape, idx, sa))))
            self.pre.append("  const T4 R{0}v = *(const T4*)(R{0} + ib);".format(gid))
        gid, nl = grp["id"], self.G[last]
        if s_last < 0 and not grp["left"]:
            grp["left"] = True
            self.pre.append("  const T R{0}l = R{0}[ib == 0 ? {1} : ib - 1];".format(gid, nl - 1))
        if s_last > 0 and not grp["right"]:
            grp["right"] = True
            self.pre.append("  const T R{0}r = R{0}[ib + 4 == {1} ? 0 : ib + 4];".format(gid, nl))
        e = "R{}a[p + {}]".format(gid, 1 + s_last)
        if grp["zero"]:
            e = "(({}) ? (T)0 : {})".format(grp["zero"], e)
        self.emit("const T v{} = {};".format(n.idx, e))
        self.loads[desc] = "v{}".format(n.idx)

    def _address(self, key, ptr, fshape, idx, sa):
        """Pointer expression of the element at `idx` (None on the sharded axis of a slab rank, where the position is
        the thread's cell + sa)."""
        if self.slab is None:
            return "{} + {}".format(ptr, self._offset(idx, fshape))
        ax, nloc = self.slab

        def at(base, along, extent):
            full = [along if d == ax else idx[d] for d in range(self.ndim)]
            shape = [extent if d == ax else fshape[d] for d in range(self.ndim)]
            return "{} + {}".format(base, self._offset(full, shape))

        if key.startswith("@"):  # a stored adjoint array: owned cells only, no ghosts
            if not self.in_gather:
                raise TraceUnsupported("stored adjoint read by the forward kernel")
            return at(ptr, "min(max(jo + ({}), 0), {})".format(sa, nloc - 1), nloc)
        slot = self._src_slot(key)
        if not self.in_gather:
            # owned cell i (0 <= i < n): position i + lo + sa of the ghost-extended array is always there (|sa| <= the
            # ghost depth at interfaces); off the GLOBAL grid -- first / last rank only -- a wrap plane
            main = at(ptr, "(i{} + a.lo + ({}))".format(ax, sa), "a.ea")
            if sa == 0:
                return main
            if sa < 0:
                cond = "i{}g + ({}) < 0".format(ax, sa)
                alt = at("a.wlo[{}]".format(slot), "(i{} + ({}) + a.hw)".format(ax, sa), "a.hw")
            else:
                cond = "i{}g + ({}) >= {}".format(ax, sa, self.G[ax])
                alt = at("a.whi[{}]".format(slot), "(i{} + ({}) - {})".format(ax, sa, nloc), "a.hw")
            return "(({}) ? ({}) : ({}))".format(cond, alt, main)
        # gather thread at owned-relative plane jo (-2 <= jo < n + 2): a term is kept only where the cell it belongs to
        # is owned, and then the position is in reach as above; everything else is masked AFTER the (clamped) load
        gpos = "(jo + a.off + ({}))".format(sa)
        main = at(ptr, "min(max(jo + a.lo + ({}), 0), a.ea - 1)".format(sa), "a.ea")
        lo = at("a.wlo[{}]".format(slot), "min(max({} + a.hw, 0), max(a.hw - 1, 0))".format(gpos), "a.hw")
        hi = at("a.whi[{}]".format(slot), "min(max({} - {}, 0), max(a.hw - 1, 0))".format(gpos, self.G[ax]), "a.hw")
        return "({0} < 0 ? ({1}) : ({0} >= {2} ? ({3}) : ({4})))".format(gpos, lo, self.G[ax], hi, main)

    def _group_arrays(self):
        """Declarations of the per-row register windows [left, 4 values, right] the fast reads index."""
        out = []
        for grp in self.groups.values():
            g = grp["id"]
            out.append("  const T R{0}a[6] = {{{1}, R{0}v.x, R{0}v.y, R{0}v.z, R{0}v.w, {2}}};".format(
                g, "R{}l".format(g) if grp["left"] else "(T)0", "R{}r".format(g) if grp["right"] else "(T)0"))
        return out

    def gi(self, d):
        """Index expression of grid axis d as user code sees it (global on the sharded axis)."""
        return "i{}g".format(d) if self.slab is not None and d == self.slab[0] else "i{}".format(d)

    def _emit_tensor(self, n):
        slot, roll = (n.attr, None) if n.op == "tensor" else n.attr
        t = self.tr.tensors[slot]
        shape = (1,) * (self.ndim - t.dim()) + tuple(t.shape)
        if len(shape) != self.ndim or any(s > g for s, g in zip(shape, self.G)):
            raise TraceUnsupported("tensor of shape {} on grid {}".format(tuple(t.shape), self.G))
        last = self.ndim - 1
        terms, stride = [], 1
        for d in reversed(range(self.ndim)):
            if shape[d] != 1:
                # shorter than the grid: an operand of a windowed value; clamped outside its window
                i = self.gi(d) if shape[d] == self.G[d] else "min({}, {})".format(self.gi(d), shape[d] - 1)
                if roll is not None and roll[d]:  # numpy.roll by roll[d]: the value at i comes from i - roll[d]
                    i = "({0} >= {1} ? {0} - {1} : {0} + {2})".format(self.gi(d), roll[d], self.G[d] - roll[d])
                terms.append("{} * {}".format(i, stride) if stride != 1 else i)
                stride *= shape[d]
        ctype = {torch.float32: "float", torch.float64: "double", torch.int32: "int", torch.int64: "long",
                 torch.bool: "unsigned char"}[t.dtype]
        cast = {"r": "(T)", "i": "(long)", "b": "0 != "}[n.kind]
        ktype = {"r": "T", "i": "long", "b": "bool"}[n.kind]
        line = "const {} v{} = {}((const {}*)a.ten[{}])[{}];".format(
            ktype, n.idx, cast, ctype, slot, " + ".join(terms) or "0")
        tdt = "float" if self.tr.torch_dtype == torch.float32 else "double"
        if (self.march_pref is not None and n.kind == _R and self.vw == 1
                and (self.march_live is None or n.idx in self.march_live)):
            e = "{}((const {}*)a.ten[{}])[{}]".format(cast, ctype, slot, " + ".join(terms) or "0")
            k = self.march_pref.setdefault(("@ten", n.op, n.attr), (len(self.march_pref), e))[0]
            self.march_used.add(k)
            self.emit("const T v{} = ld_{};".format(n.idx, k))
            return
        if self.vw == 4 and shape[last] == 1:
            self.pre.append("  " + line)  # the same value for the thread's four points: loaded once
        elif (self.vw == 4 and shape[last] == self.G[last] and ctype == tdt and (roll is None or not roll[last])
              and n.kind == _R):
            # a full row of a constant array of the kernel's own type: one 16-byte pack for the four points
            # (terms[0] is the last axis' own index)
            self.pre.append("  const T4 v{}q = *(const T4*)((const T*)a.ten[{}] + {} + ib);".format(
                n.idx, slot, " + ".join(terms[1:]) or "0"))
            self.emit("const T v{0} = v{0}q[p];".format(n.idx))
        else:
            self.emit(line)

    # ---- pointwise networks: two evaluations of the same network per pair of float lanes ---------------------
    def _choose_shared_calls(self):
        """Network evaluations another thread's evaluation equals (stencil_share.py: k_m(i) = k_p(i - e) away from the
        wall row) are not repeated: the forward kernel becomes the MARCHING kernel (_march_kernel: values carried along the
        second-to-last axis in registers, exchanged along the last axis by lane shifts).  Needs frozen inputs (the reverse
        pass of a shared evaluation runs where it was evaluated), one network, float arithmetic (packed evaluations) and
        such pairs on both of the last two axes.  ODIL_TRACE_SHARE = auto (default: where a grid is large enough for it to
        pay) | march (always, tests) | 0.  (An LDS-tiled variant -- interior + halo threads of a 7 x 32 tile, values and
        adjoints through LDS, two barriers per tile -- was built in round 3, measured slower than the plain kernel, 4.29
        against 3.55 ms at 256 x 512^2, and removed in round 4; docs/rounds/kernel_log_r01-r03.md.)"""
        from . import stencil_share

        mlps = [n for n in self.order if n.op == "mlp"]
        mode = os.environ.get("ODIL_TRACE_SHARE", "auto")
        self.share_mode = None
        if not mlps or self.slab is not None or self.ndim < 3 or mode == "0" or self.GL != self.G:
            return
        a1, a2 = self.ndim - 2, self.ndim - 1
        if len({n.attr for n in mlps}) != 1 or not self.fast:
            return  # (one network; packed float evaluations)
        if mode == "auto" and (self.G[a2] < 128 or self.G[a1] < 16):
            return
        if any(self.need.get(a.idx, False) for n in mlps for a in n.args):
            return
        try:
            found = stencil_share.shared_network_calls(self.tr, self.order, self.G, (a1, a2))
        except TraceUnsupported:
            return
        per_axis = dict()
        for A, B, axis in found:
            per_axis.setdefault(axis, (A, B, axis))
        pairs = list(per_axis.values())
        nodes = [x for A, B, _ in pairs for x in (A, B)]
        if not pairs or len({x.idx for x in nodes}) != len(nodes):
            return
        # the inputs of every call involved must be computable before any shared value is known
        outs_of = {x.idx for x in nodes}
        for x in nodes:
            for arg in x.args:
                if any(m.op == "mlp_out" and m.args[0].idx in outs_of for m in stencil_grad.subdag(arg)):
                    return
        if sorted(per_axis) != [a1, a2] or len(pairs[0][0].attr[2]) < 2:
            return  # (the marching kernel pairs the upper faces of the last two axes in ONE packed evaluation)
        self.share = pairs
        self.share_mode = "march"
        self.shared_A = {A.idx for A, _, _ in pairs}
        self.shared_B = {B.idx for _, B, _ in pairs}

    def _pair_mlps(self):
        """Float kernels: evaluations of the same network (heat: the conductivity at the four faces of a cell) are
        emitted two at a time on 2-vectors, so that every multiply-add of the layers, of the reverse pass and of the
        parameter-gradient sums is ONE packed instruction (v_pk_fma_f32) for two evaluations, with the weights
        broadcast.  The pair is evaluated where the later of the two would be: the node order is re-sorted
        (topologically) with the pair as one unit; pairs that would close a cycle stay single."""
        groups = dict()
        for n in self.order:
            if n.op == "mlp" and n.idx not in self.shared_A and n.idx not in self.shared_B:  # (shared calls: _march_kernel)
                groups.setdefault(n.attr, []).append(n)
        pairs = [(nodes[k], nodes[k + 1]) for nodes in groups.values() for k in range(0, len(nodes) - 1, 2)]
        while pairs:
            order = self._sorted_with_pairs(pairs)
            if order is not None:
                self.order = order
                for a, b in pairs:
                    self.partner[a.idx], self.partner[b.idx] = b, a
                    self.pair_first.add(a.idx)
                return
            pairs.pop()  # a cycle through some pair: try with fewer

    def _sorted_with_pairs(self, pairs):
        import heapq

        deps = {n.idx: {a.idx for a in n.args} for n in self.order}
        second = dict()
        for a, b in pairs:
            deps[a.idx] |= {x.idx for x in b.args}
            deps[b.idx] = deps[b.idx] | {a.idx}
            second[a.idx] = b
        by_idx = {n.idx: n for n in self.order}
        users = {i: [] for i in deps}
        for i, ds in deps.items():
            for d in ds:
                if d in users:
                    users[d].append(i)
        left = {i: len([d for d in ds if d in deps]) for i, ds in deps.items()}
        ready = [i for i, c in left.items() if c == 0]
        heapq.heapify(ready)
        out, held = [], set(b.idx for _, b in pairs)

        def emit(i):
            out.append(by_idx[i])
            for u in users[i]:
                left[u] -= 1
                if left[u] == 0 and u not in held:
                    heapq.heappush(ready, u)

        while ready:
            i = heapq.heappop(ready)
            emit(i)
            if i in second:  # its partner follows immediately (its dependencies are a subset of this node's)
                j = second[i].idx
                if left[j] != 0:
                    return None
                emit(j)
        return out if len(out) == len(self.order) else None

    def _act(self, kind, x, width=1):
        if width == 2:
            return {"tanh": "odil_fast_tanh2({})", "relu": "__builtin_elementwise_max({}, (T2)(0.0f))", "none": "{}"}[kind].format(x)
        tanh = "odil_fast_tanh({})" if self.fast else "FN(tanh)({})"  # (network activations: see the prelude)
        return {"tanh": tanh, "relu": "({0} > (T)0 ? {0} : (T)0)", "none": "{}"}[kind].format(x)

    def _net_base(self, attr):
        key, frozen, layers, act = attr
        if key not in self.net_slot:
            self.net_slot[key] = len(self.nets)
            self.nets.append((key, layers))
        return self.net_slot[key]

    def _launder_params(self, base, nl):
        if getattr(self, "wregs", "mem") != "const":
            return
        for l in range(nl):
            for c, ofs in (("w", "WOFS"), ("b", "BOFS")):
                self.emit('CP {0}p_{1}_{2} = (CP)a.par[{3}_{1}_{2}]; asm volatile("" : "+s"({0}p_{1}_{2}));'.format(c, base, l, ofs))

    def _mlp_forward(self, p, width, attr, inputs):
        """Layers of one (width 1) or two packed (width 2) evaluations of a pointwise network under the name prefix p;
        inputs: per network input the value expression (width 2: a pair)."""
        V = "T2" if width == 2 else "T"
        key, frozen, layers, act = attr
        base = self._net_base(attr)
        nl = len(layers) - 1
        self._launder_params(base, nl)
        for i, vals in enumerate(inputs):
            self.emit("const {} {}_h0_{} = {};".format(V, p, i, vals if width == 1 else "{{{}, {}}}".format(*vals)))
        for l in range(1, nl + 1):
            ni, no = layers[l - 1], layers[l]
            for j in range(no):
                terms = " + ".join("W({},{},{}) * {}_h{}_{}".format(base, l - 1, j * ni + i, p, l - 1, i) for i in range(ni))
                self.emit("const {} {}_z{}_{} = ({}) + Bv({},{},{});".format(V, p, l, j, terms, base, l - 1, j))
                if l < nl:
                    self.emit("const {0} {1}_h{2}_{3} = {4};".format(V, p, l, j, self._act(act, "{}_z{}_{}".format(p, l, j), width)))

    def _emit_mlp(self, n):
        if n.idx in self.shared_A or n.idx in self.shared_B:
            return  # shared evaluations: the marching kernel's unified packed evaluation (_march_kernel)
        if n.idx in self.partner and n.idx not in self.pair_first:
            return  # emitted with its partner
        group = [n, self.partner[n.idx]] if n.idx in self.pair_first else [n]
        width = len(group)
        nl = len(n.attr[2]) - 1
        p = ("mm{}" if width == 2 else "m{}").format(n.idx)
        inputs = []
        for i in range(len(n.args)):
            vals = [self.r(m.args[i]) for m in group]
            inputs.append(vals[0] if width == 1 else vals)
        self._mlp_forward(p, width, n.attr, inputs)
        if width == 2:  # the outputs under the names the single form gives them (what mlp_out nodes read)
            for lane, m in zip("xy", group):
                for j in range(n.attr[2][nl]):
                    self.emit("const T m{}_z{}_{} = {}_z{}_{}.{};".format(m.idx, nl, j, p, nl, j, lane))

    def forward(self, only=None):
        for n in self.order:
            if n.host or (only is not None and n.idx not in only):
                continue
            op, A = n.op, n.args
            kt = {"r": "T", "i": "long", "b": "bool"}[n.kind]
            v = "const {} v{} = ".format(kt, n.idx)
            if self.fold is not None and n.idx in self.fold:
                # interior copy of the body: an index predicate with the value it has away from the walls (_fold_plan)
                self.emit(v + ("true;" if self.fold[n.idx] else "false;"))
            elif op == "read":
                self._emit_read(n)
            elif op in ("tensor", "rtensor"):
                self._emit_tensor(n)
            elif op == "index":
                self.emit(v + "(long){};".format(self.gi(n.attr[0])))
            elif op == "lindex":
                self.emit(v + "(long){};".format("jo" if self.in_gather else "i{}".format(n.attr[0])))
            elif op == "win":
                self.emit(v + "{};".format(self.typed(A[0], n.kind)))
            elif op == "aparam":
                key, k, _ = n.attr
                if key not in self.array_slot:
                    self.array_slot[key] = len(self.arrays)
                    self.arrays.append((key, int(np.prod(self.state.fields[key].array.shape))))
                self.emit(v + "AP({}, {});".format(self.array_slot[key], k))
            elif op == "mlp":
                self._emit_mlp(n)
            elif op == "mlp_out":
                self.emit(v + "m{}_z{}_{};".format(A[0].idx, len(A[0].attr[2]) - 1, n.attr))
                self.mlp_out_seen.setdefault(A[0].idx, dict())[n.attr] = n
            elif op in ("add", "sub", "mul"):
                sym = {"add": "+", "sub": "-", "mul": "*"}[op]
                self.emit(v + "{} {} {};".format(self.typed(A[0], n.kind), sym, self.typed(A[1], n.kind)))
            elif op == "div":
                self.emit(v + self.over(self.r(A[0]), A[1]) + ";")
            elif op == "pow":
                if A[1].op == "const" and float(A[1].attr) == 2.0:
                    self.emit(v + "{0} * {0};".format(self.r(A[0])))
                elif A[1].op == "const" and float(A[1].attr) == 1.0:
                    self.emit(v + "{};".format(self.r(A[0])))
                else:
                    self.emit(v + "FN(pow)({}, {});".format(self.r(A[0]), self.r(A[1])))
            elif op in ("min", "max"):
                k = n.kind
                c = "<" if op == "min" else ">"
                self.emit(v + "({0} {2} {1} ? {0} : {1});".format(self.typed(A[0], k), self.typed(A[1], k), c))
            elif op == "atan2":
                self.emit(v + "FN(atan2)({}, {});".format(self.r(A[0]), self.r(A[1])))
            elif op in _CMP:
                k = _promote(A[0].kind, A[1].kind)
                self.emit(v + "{} {} {};".format(self.typed(A[0], k), _CMP[op], self.typed(A[1], k)))
            elif op in ("and", "or"):
                self.emit(v + "{} {} {};".format(self.b(A[0]), "&&" if op == "and" else "||", self.b(A[1])))
            elif op == "not":
                self.emit(v + "!{};".format(self.b(A[0])))
            elif op == "where":
                self.emit(v + "{} ? {} : {};".format(self.b(A[0]), self.typed(A[1], n.kind), self.typed(A[2], n.kind)))
            elif op == "neg":
                self.emit(v + "-{};".format(self.typed(A[0], n.kind)))
            elif op == "abs":
                x = self.typed(A[0], n.kind)
                self.emit(v + ("FN(fabs)({});".format(x) if n.kind == _R else "({0} < 0 ? -{0} : {0});".format(x)))
            elif op == "relu":
                x = self.typed(A[0], n.kind)
                self.emit(v + "({0} > 0 ? {0} : 0);".format(x))
            elif op in ("cos", "sin", "exp", "log", "tanh", "sqrt", "floor"):
                self.emit(v + "FN({})({});".format(op, self.r(A[0])))
            elif op in ("cast", "stopgrad"):
                self.emit(v + "{};".format(self.typed(A[0], n.kind)))
            else:
                raise TraceUnsupported("op " + op)

    # ---- reverse ----------------------------------------------------------------------------
    def reverse(self):
        defined = set()

        def acc(arg, expr):
            if not self.need.get(arg.idx, False):
                return
            if arg.idx in defined:
                self.emit("g{0} = g{0} + {1};".format(arg.idx, expr))
            else:
                self.emit("T g{} = {};".format(arg.idx, expr))
                defined.add(arg.idx)

        # seeds: d loss / d output = 2 f / n (or 1 / n for a Raw output) inside the output's window
        for k, (o, raw) in enumerate(zip(self.outputs, self.raw)):
            seed = "((T){!r})".format(1.0 / self.out_count[k]) if raw else "{} * ((T){!r})".format(
                self.r(o), 2.0 / self.out_count[k])
            if self.out_lens[k] is not None:
                seed = "(inbox{} ? {} : (T)0)".format(k, seed)
            if self.out_mode[k] == "virt":
                continue  # the gathers re-evaluate the output
            if self.out_mode[k] == "jac":
                # the seed IS the adjoint of this output (nothing else consumes it): stored once, expanded onto the
                # reads by the gathers with the local derivatives re-evaluated there
                self.emit("const T gs{} = {};".format(k, seed))
                self.jac_store.append((k, "gs{}".format(k)))
                continue
            acc(o, seed)
        self.pgrads = dict()  # net key -> list of per-array lists of accumulator names
        for n in reversed(self.order):
            op, A = n.op, n.args
            if op == "mlp":
                self._reverse_mlp(n, defined, acc)
                continue
            if n.idx not in defined:
                continue
            g, v = "g{}".format(n.idx), "v{}".format(n.idx)
            if n.idx in self.cut_set:
                self.cut_nodes.append(n)  # its adjoint is stored; the gathers expand it onto the reads
                continue
            if op == "read":
                self.cots.append(n)
            elif op == "win":
                acc(A[0], g)
            elif op == "aparam":
                key, k, _ = n.attr
                if key not in self.pgrads:
                    numel = dict(self.arrays)[key]
                    names = ["pa_{}_{}".format(self.array_slot[key], i) for i in range(numel)]
                    self.pgrads[key] = [names]
                    self.pg_offset[key] = len(self.pg_decl)
                    self.pg_decl.extend(names)
                name = self.pgrads[key][0][k]
                self.emit("{0} = {0} + {1};".format(name, g))
            elif op == "add":
                acc(A[0], g)
                acc(A[1], g)
            elif op == "sub":
                acc(A[0], g)
                acc(A[1], "-" + g)
            elif op == "mul":
                acc(A[0], "{} * {}".format(g, self.r(A[1])))
                acc(A[1], "{} * {}".format(g, self.r(A[0])))
            elif op == "div":
                acc(A[0], self.over(g, A[1]))
                acc(A[1], self.over("-({} * {})".format(g, v), A[1]))
            elif op == "pow":
                x, p = self.r(A[0]), self.r(A[1])
                if A[1].op == "const" and float(A[1].attr) == 2.0:
                    acc(A[0], "{} * ((T)2 * {})".format(g, x))
                elif A[1].op == "const" and float(A[1].attr) == 1.0:
                    acc(A[0], g)
                else:
                    acc(A[0], "{} * ({} * FN(pow)({}, {} - (T)1))".format(g, p, x, p))
                    acc(A[1], "{} * ({} * FN(log)({}))".format(g, v, x))
            elif op in ("min", "max"):
                c = "<" if op == "min" else ">"
                x, y = self.r(A[0]), self.r(A[1])
                acc(A[0], "({0} {2} {1} ? {3} : ({0} == {1} ? {3} * (T)0.5 : (T)0))".format(x, y, c, g))
                acc(A[1], "({1} {2} {0} ? {3} : ({0} == {1} ? {3} * (T)0.5 : (T)0))".format(x, y, c, g))
            elif op == "atan2":
                y, x = self.r(A[0]), self.r(A[1])
                acc(A[0], "{0} * {2} / ({1} * {1} + {2} * {2})".format(g, y, x))
                acc(A[1], "-{0} * {1} / ({1} * {1} + {2} * {2})".format(g, y, x))
            elif op == "where":
                acc(A[1], "({} ? {} : (T)0)".format(self.b(A[0]), g))
                acc(A[2], "({} ? (T)0 : {})".format(self.b(A[0]), g))
            elif op == "neg":
                acc(A[0], "-" + g)
            elif op == "abs":
                x = self.r(A[0])
                acc(A[0], "({0} > (T)0 ? {1} : ({0} < (T)0 ? -{1} : (T)0))".format(x, g))
            elif op == "relu":
                acc(A[0], "({} > (T)0 ? {} : (T)0)".format(self.r(A[0]), g))
            elif op == "cos":
                acc(A[0], "-({} * FN(sin)({}))".format(g, self.r(A[0])))
            elif op == "sin":
                acc(A[0], "{} * FN(cos)({})".format(g, self.r(A[0])))
            elif op == "exp":
                acc(A[0], "{} * {}".format(g, v))
            elif op == "log":
                acc(A[0], "{} / {}".format(g, self.r(A[0])))
            elif op == "tanh":
                acc(A[0], "{0} * ((T)1 - {1} * {1})".format(g, v))
            elif op == "sqrt":
                acc(A[0], "{} / ((T)2 * {})".format(g, v))
            elif op == "cast":
                acc(A[0], g)
            elif op in ("mlp_out",):
                pass  # collected by the mlp node
            elif op in ("floor", "stopgrad", "tensor", "index"):
                pass
            else:
                raise TraceUnsupported("derivative of " + op)
        self.cots.reverse()
        self.cut_nodes.reverse()

    def _reverse_mlp(self, n, defined, acc):
        if n.idx in self.shared_A or n.idx in self.shared_B:
            return  # the reverse pass of shared evaluations runs one step late (_march_kernel)
        if n.idx in self.partner and n.idx not in self.pair_first:
            return  # handled when the traversal reaches its partner (the earlier node of the pair)
        group = [n, self.partner[n.idx]] if n.idx in self.pair_first else [n]
        if not any(self.need[m.idx] for m in group):
            return
        width = len(group)
        layers = n.attr[2]
        outs = [{m.attr: m for m in self.order if m.op == "mlp_out" and m.args[0] is g and m.idx in defined} for g in group]
        if not any(outs):
            return
        nl = len(layers) - 1
        p = ("mm{}" if width == 2 else "m{}").format(n.idx)
        dvals = []
        for j in range(layers[nl]):
            vals = ["g{}".format(o[j].idx) if j in o else "(T)0" for o in outs]
            dvals.append(vals[0] if width == 1 else vals)
        inputs_need = any(self.need[a.idx] for m in group for a in m.args)
        self._mlp_backward(p, width, n.attr, dvals, inputs_need)
        if inputs_need:
            for lane, m in zip("xy", group):
                for i, a in enumerate(m.args):
                    acc(a, "{}_d0_{}{}".format(p, i, "" if width == 1 else "." + lane))

    def _mlp_backward(self, p, width, attr, dvals, inputs_need):
        """Reverse pass of the evaluation(s) emitted under prefix p given the adjoints of the outputs: parameter
        gradients into the accumulators, the adjoints of the inputs (`{p}_d0_{i}`) when somebody needs them."""
        V = "T2" if width == 2 else "T"
        key, frozen, layers, act = attr
        base = self.net_slot[key]
        nl = len(layers) - 1
        sfx = "2" if width == 2 else ""  # packed sums of two evaluations live in their own accumulators
        self._launder_params(base, nl)
        for j, vals in enumerate(dvals):
            self.emit("const {} {}_d{}_{} = {};".format(V, p, nl, j, vals if width == 1 else "{{{}, {}}}".format(*vals)))
        # (one accumulator per parameter for both slots of a packed evaluation was measured in round 4 and lost: removed)
        if not frozen and width == 2:
            self.pg2_used.add(key)
        if not frozen and key not in self.pgrads:
            names = []
            for l in range(nl):
                names.append(["pw_{}_{}_{}".format(base, l, k) for k in range(layers[l] * layers[l + 1])])
            for l in range(nl):
                names.append(["pb_{}_{}_{}".format(base, l, k) for k in range(layers[l + 1])])
            self.pgrads[key] = names
            self.pg_offset[key] = len(self.pg_decl)
            self.pg_decl.extend(name for group in names for name in group)
        for l in range(nl, 0, -1):
            ni, no = layers[l - 1], layers[l]
            if not frozen:
                for j in range(no):
                    for i in range(ni):
                        self.emit("pw{8}_{0}_{1}_{2} = pw{8}_{0}_{1}_{2} + {3}_d{4}_{5} * {3}_h{6}_{7};".format(
                            base, l - 1, j * ni + i, p, l, j, l - 1, i, sfx))
                    self.emit("pb{5}_{0}_{1}_{2} = pb{5}_{0}_{1}_{2} + {3}_d{4}_{2};".format(base, l - 1, j, p, l, sfx))
            if l == 1 and not inputs_need:
                break
            for i in range(ni):
                s_ = " + ".join("W({},{},{}) * {}_d{}_{}".format(base, l - 1, j * ni + i, p, l, j) for j in range(no))
                if l > 1:
                    h = "{}_h{}_{}".format(p, l - 1, i)
                    one = "(T)1" if width == 1 else "(T2)(1.0f)"
                    d = {"tanh": "({1} - {0} * {0})".format(h, one),
                         "relu": ("({} > (T)0 ? (T)1 : (T)0)" if width == 1 else "odil_step2({})").format(h),
                         "none": one}[act]
                    self.emit("const {} {}_d{}_{} = ({}) * {};".format(V, p, l - 1, i, s_, d))
                else:
                    self.emit("const {} {}_d0_{} = {};".format(V, p, i, s_))

    # ---- whole source -----------------------------------------------------------------------
    def _index_prologue(self, S, shape, names, vw, flat="l"):
        """Decomposition of the flat thread index into grid indices names[d]; with vw == 4 the thread owns four
        consecutive points of the last axis starting at `ib` (the loop over p defines names[last] = ib + p)."""
        rem = flat
        last = len(shape) - 1
        # A row of the last axis that is a whole number of wavefronts (64 lanes x vw points): the 64 lanes of a wavefront
        # hold consecutive flat indices starting at a multiple of 64 (workgroups of 256, chunk and XCD remaps move whole
        # multiples of a row), so every index but the last is the same in all of them.  Said so explicitly (the first
        # lane's value in a scalar register), the row base pointers, the wrap / slab-plane selects and the index
        # predicates are scalar arithmetic instead of one copy per lane: config 5 as one rank, merged gather 7.48 -> 7.30
        # ms.  Smaller grids (rows shorter than a wavefront) are emitted unchanged.
        uniform = (len(shape) >= 2 and (shape[last] // vw) % 64 == 0 and shape[last] % vw == 0
                   )
        for d in reversed(range(len(shape))):
            ext = shape[d] // vw if d == last else shape[d]
            var = "ib" if (vw == 4 and d == last) else names[d]
            mul = " * 4" if (vw == 4 and d == last) else ""
            if d == 0:
                S.append("  const int {} = ({}){};".format(var, rem, mul))
            else:
                S.append("  const int {} = ({} % {}){};".format(var, rem, ext, mul))
                if uniform and d == last:
                    S.append("  const int r{}_ = __builtin_amdgcn_readfirstlane({} / {});".format(d, rem, ext))
                else:
                    S.append("  const int r{}_ = {} / {};".format(d, rem, ext))
                rem = "r{}_".format(d)

    @staticmethod
    def _block_index(shape, vw):
        """Which slice of the index range a workgroup takes.  The hardware deals consecutive workgroups to the eight XCDs
        in turn, so the rows two neighbouring workgroups both read (the -+ 1 rows of the second-to-last axis at their
        common edge: a workgroup is 4 rows of 256 points) are filled into two L2s.  Here K consecutive slices go to one
        XCD, then K to the next: the interleaving across the XCDs stays fine-grained, the shared rows of K - 1 of K edges
        meet in one L2.  K = 8 at most, and such that the neighbours along the THIRD-to-last axis (prod(shape[-2:]) / vw /
        256 slices away) stay on the same XCD: config 5 as one rank `k_fwd` 2.80 -> 2.72 ms, gather 8.25 -> 7.8 - 8.0 with
        K = 8 (K = 2: 8.07; K = 16: 8.4, K = 32: 8.8 -- the x -+ 1 rows then land on another XCD; a CONTIGUOUS eighth of
        the range per XCD: 9.05); tracer 32 x 256^3 33.85 -> 32.9 ms; heat 256 x 512^2 unchanged.  ODIL_TRACE_XCD_GROUP
        overrides K (1: the hardware's order).  Needs a grid that is a multiple of 8 K (else the hardware's order)."""
        k = int(os.environ.get("ODIL_TRACE_XCD_GROUP", 0))
        if k == 0:
            dist = (int(np.prod(shape[-2:])) // vw) // 256 if len(shape) >= 3 else 0
            k = next((c for c in (8, 4, 2) if dist and dist % (8 * c) == 0), 1)
        if k > 1:
            return ("  const int bx_ = gridDim.x % {0} == 0 ? (int)((blockIdx.x / {0}) * {0} + (blockIdx.x % 8) * {1} + "
                    "(blockIdx.x % {0}) / 8) : (int)blockIdx.x;").format(8 * k, k)
        return "  const int bx_ = blockIdx.x;"

    def _chunk_remap(self, S, shape, vw, raw, flat):
        """Defines the flat index `flat` of the point(s) a thread owns from its launch index `raw`.  Plain order
        (flat = raw) walks axis 0 slowest: a stencil that reads the levels i0 - 1 / i0 + 1 (time differences of the
        space-time operators) meets each of them again one whole level later -- 67 MB per array at 256^3, times every
        array the kernel streams: far beyond the last-level cache, the neighbouring levels are read from HBM again.
        With CHUNKS of axis 1 outermost -- (chunk, i0, i1 within the chunk, rest) -- the distance shrinks to one chunk
        of a level (<= ODIL_TRACE_CHUNK_MB per array, default 5) and the neighbouring levels are cache hits, while
        every chunk is still megabytes of contiguous memory per array and level (the walk along i0 in 4 - 16 KB
        pieces tried before lost more in DRAM pages / TLB reach than it gained).  Measured:
        config 5 as one rank 17.2 -> 16.75 ms with two chunks of 4.7 MB (16.97 with three of 3.1 MB, 17.3 with 0.5
        MB); tracer 32 x 256^3 39.2 -> 38.0 with 4.2 MB chunks (38.5 with 8.4 MB, 39.4 with 16.8 MB)."""
        limit = float(os.environ.get("ODIL_TRACE_CHUNK_MB", 5)) * (1 << 20)
        esize = 8 if self.tr.torch_dtype == torch.float64 else 4
        c1 = 0
        if len(shape) >= 3 and limit > 0 and shape[0] >= 3:
            row = int(np.prod(shape[2:])) * esize
            c1 = max([c for c in range(1, shape[1] + 1) if shape[1] % c == 0 and c * row <= limit], default=0)
        if c1 == 0 or c1 >= shape[1]:
            S.append("  const int {} = {};".format(flat, raw))
            return
        rest = int(np.prod(shape[2:])) // vw
        S.append("  const int cq_ = {0} / {1}, cr_ = {0} % {1};".format(raw, shape[0] * c1 * rest))
        S.append("  const int {} = ((cr_ / {}) * {} + cq_ * {}) * {} + cr_ % {};".format(
            flat, c1 * rest, shape[1], c1, rest, c1 * rest))

    def _loop_open(self, S, vw):
        last = self.ndim - 1
        if vw == 4:
            S.append("#pragma unroll")
            S.append("  for (int p = 0; p < 4; ++p) {")
            S.append("  const int i{} = ib + p;".format(last))
            S.append("  const int l = l4 * 4 + p;")

    def _gradient_terms(self):
        """Per regular field: the gradient expression G_F(j) = sum of [A_c * dc/dr](j - shift_r) over every stored
        or re-evaluated cut c and live read r of the field (module docstring of stencil_grad.py).  Fields with pad /
        trim reads keep the legacy gather (their reads are never below an output cut)."""
        tr = self.tr
        stop = set(self.cut_set)  # (reads are terminal anyway; a read below a cut ALSO collects through that cut)
        terms = dict()  # key -> [expression at the point of evaluation, read attr]
        gb = stencil_grad.GradBuilder(tr, self.G, self.need, stop)

        def pseudo(tag, slot):
            key = "@{}{}".format(tag, slot)
            tr.state_locs[key] = self.gloc
            self.pseudo_slot[key] = slot
            return tr.node("read", attr=(key, (0,) * self.ndim, self.gloc, True), shape=self.G, kind=_R)

        for slot, n in enumerate(self.cots):
            if self._regular(n):
                terms.setdefault(n.attr[0], []).append((pseudo("r", slot), n.attr))
        for k, n in enumerate(self.cut_nodes):
            slot = len(self.cots) + k
            reads = [tr.nodes[i] for i in self.cut_set[n.idx]]
            if not all(self._regular(r) for r in reads):
                continue
            adj = gb.adjoints(n, pseudo("c", slot), stencil_grad.subdag(n, stop))
            for ridx, expr in adj.items():
                terms.setdefault(tr.nodes[ridx].attr[0], []).append((expr, tr.nodes[ridx].attr))
        for k, adj in enumerate(self.out_adj):
            if adj is None:
                continue
            for ridx, expr in adj.items():
                terms.setdefault(tr.nodes[ridx].attr[0], []).append((expr, tr.nodes[ridx].attr))
        exprs = dict()
        owned = None
        if self.slab is not None:
            # one rank of a slab decomposition: a term belongs to this rank where the cell it is evaluated at is OWNED
            # (the neighbour forms the others and the halo-add of slab_traced.py joins them)
            ax, nloc = self.slab
            li = tr.node("lindex", attr=(ax,), shape=self.G, kind=_I)
            owned = gb._node("and", (gb.cmp("ge", li, tr.const(0)), gb.cmp("lt", li, tr.const(int(nloc)))), kind=_B)
        for key, lst in terms.items():
            total = None
            for expr, attr in lst:
                if owned is not None:
                    expr = gb.where(owned, expr, None)
                shifted = tr.roll(expr, tuple(attr[1]), virtual=True)
                total = gb.add(total, shifted)
            exprs[key] = total
        return exprs

    def source(self):
        tdt = self.tr.torch_dtype
        self.pg_decl, self.pg_offset = [], dict()
        self.pg2_used = set()
        self.jac_store = []
        self.vw = self.vw_fwd
        vw, last = self.vw, self.ndim - 1
        self.tr.state_locs = dict(getattr(self.tr, "state_locs", dict()))
        self._begin()
        interior = None
        march = None
        if self.share:
            march = self._march_parts()
            fwd, rev = [], []
        else:
            self.forward()
            fwd = self.lines
            self.lines = []
            self.reverse()
            rev = self.lines
            # (not for kernels with a pointwise network: two copies of the network's forward and reverse pass cost the
            # registers of a second resident wave -- heat 256 x 512^2: 283 + 27 spilled to AGPRs, 3.29 -> 4.36 ms / epoch)
            has_net = any(n.op == "mlp" for n in self.order)
            interior = None if has_net else self._interior_copy(self.order, vw, reverse=True)
        fwd_pre = self.pre + self._group_arrays()
        nout = len(self.outputs)
        self.npar = sum(len(g) for names in self.pgrads.values() for g in names)
        par_arrays = sum(2 * (len(layers) - 1) for _, layers in self.nets) + len(self.arrays)
        self.par_arrays = par_arrays
        T = "double" if tdt == torch.float64 else "float"
        fn = "name" if T == "double" else "name##f"
        # the optimizer state and the gradient stream through a gather once per epoch: non-temporal accesses keep them
        # out of the way of the rows the kernel re-reads (config 5 as one rank: gather 7.77 -> 7.47 ms)
        # (only where an array is beyond what the caches could hand to the next launch anyway)
        self.nt_streams = self.total * (8 if T == "double" else 4) > (64 << 20)
        nt = "#define ODIL_NT_STREAMS 1\n" if self.nt_streams else ""
        HEAD = [("#define ODIL_FAST_F32 1\n" if self.fast else "") + nt + _PRELUDE.replace("@T@", T).replace("@FN@", fn)]
        if self.fast:
            HEAD.append("#define tanhf odil_tanh_f32\n#define expf odil_fast_exp")
        if march is not None:  # (value of lane `l` -- wave-uniform -- for every lane; only the marching kernel uses it)
            HEAD.append("__device__ inline float odil_readlane(float v, int l) {\n"
                        "  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));\n}")
        S = []
        # parameter access macros: W(net, layer, k), Bv(net, layer, k)
        wofs, bofs, o = dict(), dict(), 0
        for s, (key, layers) in enumerate(self.nets):
            nl = len(layers) - 1
            for l in range(nl):
                wofs[(s, l)] = o + l
                bofs[(s, l)] = o + nl + l
            o += 2 * nl
        self.par_layout = [(key, layers) for key, layers in self.nets]
        S.append("#define AP(s, k) a.par[{} + s][k]".format(o))  # Array unknowns follow the net arrays
        # the marching kernel keeps the network's parameters in registers for the whole launch (they would be re-read
        # through the vector memory pipe -- uniform addresses, but stores to the adjoint arrays in between -- in every
        # evaluation and every reverse pass of every step).  (Round 4 also measured them as vector-register and as
        # scalar-register residents: both lost to the constant-address-space form and were removed.)
        if self.wregs == "const":
            # the parameter arrays read through the CONSTANT address space: the compiler may then use scalar loads, merge
            # them across the stores of the adjoint arrays and re-load instead of spilling
            # (scalar loads at every use instead of 46 values hoisted to the top of the kernel and spilled: the pointers
            # are passed through an empty asm before every evaluation / reverse pass, _launder_params)
            S.append("typedef const __attribute__((address_space(4))) T* CP;")
            S.append("#define W(s, l, k) wp_##s##_##l[k]")
            S.append("#define Bv(s, l, k) bp_##s##_##l[k]")
        else:
            S.append("#define W(s, l, k) a.par[WOFS_##s##_##l][k]")
            S.append("#define Bv(s, l, k) a.par[BOFS_##s##_##l][k]")
        for (s, l), v in wofs.items():
            S.append("#define WOFS_{}_{} {}".format(s, l, v))
        for (s, l), v in bofs.items():
            S.append("#define BOFS_{}_{} {}".format(s, l, v))
        # ---- k_fwd ---------------------------------------------------------------------------------------------
        occ = 0  # register budget of k_fwd as waves per SIMD (0: the compiler's)
        self.fwd_threads = 256

        S.append('extern "C" __global__ __launch_bounds__({}) {}void k_fwd(const Args a) {{'.format(
            self.fwd_threads, "__attribute__((amdgpu_waves_per_eu({0}, {0}))) ".format(occ) if occ else ""))
        S.append("  __shared__ T sm[{}];".format(self.fwd_threads // 64))
        for k in range(nout):
            S.append("  T s_{} = (T)0;".format(k))
        for name in self.pg_decl:
            S.append("  T {} = (T)0;".format(name))
        pg2 = [name for key in self.pg2_used for group in self.pgrads[key] for name in group]
        for name in pg2:  # packed partial sums of paired network evaluations
            S.append("  T2 {}2{} = (T2)(0.0f);".format(name[:2], name[2:]))
        esize = 8 if tdt == torch.float64 else 4
        stored = ([(n, "g{}".format(n.idx)) for n in self.cots] + [(n, "g{}".format(n.idx)) for n in self.cut_nodes]
                  + [(self.outputs[k], name) for k, name in self.jac_store])  # cut arrays follow the read cotangents
        self.ncot = len(stored)
        self.edge_numel = 0
        mgather = march["gather"] if march is not None else None
        if mgather is not None:  # one array per (field, leading shift) + the edge arrays (_march_gather_plan)
            self.ncot = len(mgather["groups"])
            self.edge_numel = self._march_edge_offsets(self.ncot)[1]
        for slot_k, (k, _) in enumerate(self.jac_store):
            self.pseudo_slot[self.seed_key[k]] = len(self.cots) + len(self.cut_nodes) + slot_k
        stream = self.ncot * self.total * esize > (128 << 20)  # beyond what the last-level cache keeps
        if march is not None:
            self._march_kernel(S, march, stored, stream)
        threads = self.total // vw
        if march is None:
            flat = "l4" if vw == 4 else "l"
            S.append(self._block_index(self.GL, vw))
            if threads <= self.max_blocks * 256:  # one thread per point (or four points)
                S.append("  const int {}r = bx_ * NB + threadIdx.x;".format(flat))
                S.append("  if ({}r < {}) {{".format(flat, threads))
            else:
                S.append("  for (int {0}r = bx_ * NB + threadIdx.x; {0}r < {1}; {0}r += a.nblocks * NB) {{".format(flat, threads))
            self._chunk_remap(S, self.GL, vw, flat + "r", flat)
            self._index_prologue(S, self.GL, ["i{}".format(d) for d in range(self.ndim)], vw, flat)
            if self.slab is not None:
                S.append("  const int i{0}g = i{0} + a.off;".format(self.slab[0]))
            S.extend(fwd_pre)
            if vw == 4:
                for slot in range(len(stored)):
                    S.append("  T O{}[4];".format(slot))

            def point_block(fwd_, rev_, inbox_):
                B = []
                self._loop_open(B, vw)
                B.extend(self._inbox_lines(inbox_))
                B.extend(fwd_)
                B.extend(rev_)
                for slot, (n, name) in enumerate(stored):
                    if vw == 4:
                        B.append("  O{}[p] = {};".format(slot, name))
                    elif stream:
                        B.append("  __builtin_nontemporal_store({}, &a.cot[{}][l]);".format(name, slot))
                    else:
                        B.append("  a.cot[{}][l] = {};".format(slot, name))
                for k, (o_, raw) in enumerate(zip(self.outputs, self.raw)):
                    term = self.r(o_) if raw else "{0} * {0}".format(self.r(o_))
                    if self.out_lens[k] is not None:
                        term = "(inbox{} ? {} : (T)0)".format(k, term)
                    B.append("  s_{0} = s_{0} + {1};".format(k, term))
                if vw == 4:
                    B.append("  }")  # p
                return B

            if interior is None:
                S.extend(point_block(fwd, rev, ()))
            else:
                (fwd_i, rev_i), (_, exc, inbox_i) = interior
                S.append("  if (__all((int)({}))) {{".format(self._interior_cond(exc)))
                S.extend(point_block(fwd_i, rev_i, inbox_i))
                S.append("  } else {")
                S.extend(point_block(fwd, rev, ()))
                S.append("  }")
            if vw == 4:
                for slot in range(len(stored)):
                    vec = "(T4){{O{0}[0], O{0}[1], O{0}[2], O{0}[3]}}".format(slot)
                    if stream:
                        S.append("  __builtin_nontemporal_store({}, (T4*)(a.cot[{}] + l4 * 4));".format(vec, slot))
                    else:
                        S.append("  *(T4*)(a.cot[{}] + l4 * 4) = {};".format(slot, vec))
            S.append("  }")
        for name in pg2:
            S.append("  {0} = {0} + ({1}2{2}.x + {1}2{2}.y);".format(name, name[:2], name[2:]))
        bsum = "block_sum" if self.fwd_threads == 256 else "block_sum_w<{}>".format(self.fwd_threads // 64)
        for k in range(nout):
            S.append("  {{ const T s = {1}(s_{0}, sm); if (threadIdx.x == 0) a.part[{0} * a.nblocks + blockIdx.x] = s; }}".format(k, bsum))
        for k, name in enumerate(self.pg_decl):
            S.append("  {{ const T s = {}({}, sm); if (threadIdx.x == 0) a.ppart[{} * a.nblocks + blockIdx.x] = s; }}".format(bsum, name, k))
        S.append("}")
        # final reduction in two deterministic stages: k_final sums SEG segments of every row of
        # partials (one workgroup each), k_loss combines them in order: out = [loss, terms..., norms...]
        # and the parameter gradients
        nrows = nout + len(self.pg_decl)
        S.append("#define SEG 16")
        S.append('extern "C" __global__ __launch_bounds__(NB) void k_final(const Args a) {')
        S.append("  __shared__ T sm[NB / 64];")
        S.append("  const int k = blockIdx.x, seg = blockIdx.y;")
        S.append("  const T* row = k < {0} ? a.part + k * a.nblocks : a.ppart + (k - {0}) * a.nblocks;".format(nout))
        S.append("  const int len = (a.nblocks + SEG - 1) / SEG, j0 = seg * len, j1 = min(j0 + len, a.nblocks);")
        S.append("  T s = (T)0;")
        S.append("  for (int j = j0 + threadIdx.x; j < j1; j += NB) s = s + row[j];")
        S.append("  s = block_sum(s, sm);")
        S.append("  if (threadIdx.x == 0) a.part2[k * SEG + seg] = s;")
        S.append("}")
        S.append('extern "C" __global__ __launch_bounds__(64) void k_loss(const Args a) {')
        S.append("  const bool raw[{}] = {{{}}};".format(nout, ", ".join("true" if r else "false" for r in self.raw)))
        S.append("  for (int k = threadIdx.x; k < {}; k += 64) {{".format(nrows))
        S.append("    T s = (T)0;")
        S.append("    for (int seg = 0; seg < SEG; ++seg) s = s + a.part2[k * SEG + seg];")
        S.append("    if (k >= {0}) {{ a.pgrad[k - {0}] = s; continue; }}".format(nout))
        S.append("    const T count[{}] = {{{}}};".format(nout, ", ".join("(T){!r}".format(float(c)) for c in self.out_count)))
        S.append("    s = s / count[k];")
        S.append("    a.out[1 + k] = s;")
        S.append("    a.out[1 + {} + k] = raw[k] ? s : FN(sqrt)(s);".format(nout))
        S.append("  }")
        S.append("  __syncthreads();")
        S.append("  if (threadIdx.x != 0) return;")
        S.append("  T loss = (T)0;")
        S.append("  for (int k = 0; k < {}; ++k) loss = loss + a.out[1 + k];".format(nout))
        S.append("  a.out[0] = loss;")
        S.append("}")
        # ---- gathers -------------------------------------------------------------------------------------------
        if mgather is not None:
            self._march_gather_kernels(S, mgather)
        self.gathers_done = mgather is not None
        self._standard_gathers(S)
        if getattr(self, "want_jac", False):
            self._jacobian_kernel(S)
        self._launchers(S, nout, par_arrays, HEAD)
        return "\n".join(HEAD + S) + "\n"

    def _jacobian_kernel(self, S):
        """`k_jac`: what `Problem.eval_operator_grad` returns (reference core.py:1313-1361, the input of `linearize`,
        core.py:1113-1217) as ONE pointwise kernel -- the value of every output and d output / d read for every distinct
        read (key, shift, loc), i.e. the per-shift coefficient arrays of the Jacobian -- from the SYMBOLIC derivative of the
        traced DAG (stencil_grad.GradBuilder with the unit seed), instead of one autograd pass per output over a graph
        of torch elementwise kernels.  Operators whose outputs are windows of the grid, or that differentiate through
        parameter arrays (dense Jacobian columns), keep the autograd route (TraceUnsupported)."""
        tr = self.tr
        items, self.jac_items = [], []  # jac_items[j] = (output position, None for its value | the read's attr)
        for k, o in enumerate(self.outputs):
            if o.win is not None or tuple(o.shape) != self.G or self.raw[k]:
                raise TraceUnsupported("Jacobian kernel: output {} is not a plain residual on the whole grid".format(k))
            nodes = stencil_grad.subdag(o)
            if not stencil_grad.differentiable(nodes, self.need):
                raise TraceUnsupported("Jacobian kernel: parameters below output {} (dense columns)".format(k))
            gb = stencil_grad.GradBuilder(tr, self.G, self.need, stop=())
            items.append(("@jv{}".format(k), gb.real(o)))
            self.jac_items.append((k, None))
            adj = gb.adjoints(o, gb.const(1.0), nodes) if self.need.get(o.idx, False) else dict()
            for ridx in sorted(adj):
                expr = adj[ridx]
                if expr is None:
                    continue
                items.append(("@jd{}_{}".format(k, ridx), expr))
                self.jac_items.append((k, tuple(tr.nodes[ridx].attr)))
        if len(items) > 96:
            raise TraceUnsupported("Jacobian kernel: {} arrays".format(len(items)))
        saved = dict(self.gather_reads_sources)
        S.append("struct JacP {{ T* p[{}]; }};".format(len(items)))
        self.jac_blocks = self._gather_kernel(S, "k_jac", items, "const JacP jp, const AdamP ad", lambda k: "jp.p[{}]".format(k),
                                              lambda k: "ad")
        self.gather_reads_sources = saved  # (bookkeeping of the optimizer fusion: the Jacobian kernel is not a gather)

    def _standard_gathers(self, S):
        if self.gathers_done:
            return
        self.gathers = []  # keys of the fields that need a gather launch
        self.gather_reads_sources = dict()  # key -> the gather reads the fields' own arrays (not only stored adjoints)
        self.direct = dict()  # key -> cot slot that already IS the gradient
        by_key = dict()  # key -> [(slot, read attr, coefficient expression or None)]
        for slot, n in enumerate(self.cots):
            by_key.setdefault(n.attr[0], []).append((slot, n.attr, None))
        for k, n in enumerate(self.cut_nodes):
            for ridx, coeff in self.cut_set[n.idx].items():
                attr = self.tr.nodes[ridx].attr
                by_key.setdefault(attr[0], []).append((len(self.cots) + k, attr, coeff))
        symbolic = dict()
        if self.all_regular:
            symbolic = self._gradient_terms()
        self.gather_blocks = dict()
        keys = list(by_key) + [k for k in symbolic if k not in by_key]
        for key in keys:
            reads = by_key.get(key, [])
            floc = self.state.fields[key].loc
            fshape = self._field_shape(key)
            only_legacy = all(adj is None or not any(self.tr.nodes[r].attr[0] == key for r in adj) for adj in self.out_adj)
            if (self.slab is None and only_legacy and len(reads) == 1 and reads[0][2] is None and not any(reads[0][1][1])
                    and reads[0][1][2] == floc):
                self.direct[key] = reads[0][0]
                continue
            gi = len(self.gathers)
            self.gathers.append(key)
            regular = tuple(fshape) == self.G and all(attr[2] == floc for _, attr, _ in reads)
            if key in symbolic and regular and symbolic[key] is not None:
                self._gather_symbolic(S, gi, key, symbolic[key])
                continue
            if self.slab is not None:
                self._gather_slab(S, gi, key, reads, floc, fshape)
                continue
            tot = int(np.prod(fshape))
            self.gather_blocks[gi] = (tot + 255) // 256
            S.append('extern "C" __global__ __launch_bounds__(NB) void k_gat_{}(const Args a, T* __restrict__ g, const AdamP ad) {{'.format(gi))
            S.append("  const int l = blockIdx.x * NB + threadIdx.x;")
            S.append("  if (l >= {}) return;".format(tot))
            rem = "l"
            for d in reversed(range(self.ndim)):
                if d == 0:
                    S.append("  const int j0 = {};".format(rem))
                else:
                    S.append("  const int j{} = {} % {};".format(d, rem, fshape[d]))
                    S.append("  const int q{} = {} / {};".format(d, rem, fshape[d]))
                    rem = "q{}".format(d)
            S.append("  T acc = (T)0;")
            for entry, (slot, attr, coeff) in enumerate(reads):
                _, shift, loc, _ = attr
                idx, valid = [], []
                for d in range(self.ndim):
                    ns, nr = fshape[d], self.G[d]
                    ext = max(ns, nr)
                    s_ = shift[d] % ext
                    if s_ > ext // 2:
                        s_ -= ext
                    pos = "j{}".format(d) if not (floc[d] == "c" and loc[d] == "n") else "(j{} + 1)".format(d)
                    e = pos if s_ == 0 else "wrap({} - ({}), {})".format(pos, s_, ext)
                    if floc[d] == "n" and loc[d] == "c":  # trimmed: the last padded position was dropped
                        name = "t{}_{}".format(entry, d)
                        S.append("  const int {} = {};".format(name, e))
                        valid.append("{} < {}".format(name, nr))
                        e = name
                    idx.append(e)
                load = "a.cot[{}][{}]".format(slot, self._offset(idx, self.G))
                if coeff is not None:  # a cut array: the stored adjoint times d(node) / d(read)
                    load = "({}) * {}".format(coeff, load)
                if valid:
                    load = "(({}) ? {} : (T)0)".format(" && ".join(valid), load)
                S.append("  acc = acc + {};".format(load))
            S.append("  g[l] = acc;")
            S.append("  adam_apply(ad, l, acc);")
            S.append("}")
        # every symbolic gather in ONE launch: the fields' expressions share most of what they read (stored seeds, each
        # other's arrays), a merged pass reads it once (tracer with three space dimensions: 52 -> 36 words per point)
        self.merged = []
        sym_keys = [key for key in self.gathers if key in symbolic and symbolic[key] is not None
                    and tuple(self._field_shape(key)) == self.G
                    and all(attr[2] == self.state.fields[key].loc for _, attr, _ in by_key.get(key, []))]
        if len(sym_keys) >= 2:
            self.merged = sym_keys
            nk = len(sym_keys)
            S.append("struct GatAll {{ T* g[{0}]; AdamP ad[{0}]; }};".format(nk))
            nblocks_all = self._gather_kernel(S, "k_gat_all", [(key, symbolic[key]) for key in sym_keys], "const GatAll ga",
                                              lambda k: "ga.g[{}]".format(k), lambda k: "ga.ad[{}]".format(k))
            S.append('extern "C" int jit_gather_all(const Args* a, void* const* g, void* const* x, void* const* m, void* const* v,')
            S.append('                               double alpha, double omb1, double omb2, double eps, const void* alpha_dev, void* stream) {')
            S.append("  GatAll ga;")
            S.append("  for (int k = 0; k < {}; ++k) {{".format(nk))
            S.append("    ga.g[k] = (T*)g[k];")
            S.append("    ga.ad[k] = AdamP{(T*)x[k], (T*)m[k], (T*)v[k], (T)alpha, (T)omb1, (T)omb2, (T)eps, (const T*)alpha_dev};")
            S.append("  }")
            S.append("  hipLaunchKernelGGL(k_gat_all, dim3({}), dim3(NB), 0, (hipStream_t)stream, *a, ga);".format(nblocks_all))
            S.append("  return (int)hipGetLastError();")
            S.append("}")

    def _launchers(self, S, nout, par_arrays, HEAD):
        # outputs in parameter space as one generated kernel (param_expr.py), when their tape has an elementwise form
        self.par_index = []
        if getattr(self, "par_outputs", None):
            from . import param_expr

            fresh = {i for i, key in self.par_keys.items() if key not in self.pgrads}
            self.par_index = param_expr.emit(self, S, self.par_outputs, fresh)
        # launchers
        S.append('extern "C" int jit_fwd(const Args* a, void* stream) {')
        S.append("  hipLaunchKernelGGL(k_fwd, dim3(a->nblocks), dim3({}), 0, (hipStream_t)stream, *a);".format(self.fwd_threads))
        S.append("  hipLaunchKernelGGL(k_final, dim3({}, SEG), dim3(NB), 0, (hipStream_t)stream, *a);".format(nout + len(self.pg_decl)))
        S.append("  hipLaunchKernelGGL(k_loss, dim3(1), dim3(64), 0, (hipStream_t)stream, *a);")
        S.append("  return (int)hipGetLastError();")
        S.append("}")
        S.append('extern "C" int jit_gather_adam(int which, const Args* a, void* g, void* x, void* m, void* v, double alpha,')
        S.append('                                double omb1, double omb2, double eps, const void* alpha_dev, void* stream);')
        S.append('extern "C" int jit_gather(int which, const Args* a, void* g, void* stream) {')
        S.append("  return jit_gather_adam(which, a, g, nullptr, nullptr, nullptr, 0.0, 0.0, 0.0, 0.0, nullptr, stream);")
        S.append("}")
        S.append('extern "C" int jit_gather_adam(int which, const Args* a, void* g, void* x, void* m, void* v, double alpha,')
        S.append('                                double omb1, double omb2, double eps, const void* alpha_dev, void* stream) {')
        S.append("  const AdamP ad = {(T*)x, (T*)m, (T*)v, (T)alpha, (T)omb1, (T)omb2, (T)eps, (const T*)alpha_dev};")
        S.append("  switch (which) {")
        for gi, key in enumerate(self.gathers):
            tot = int(np.prod(self._field_shape(key)))
            nblk = str(self.gather_blocks.get(gi, (tot + 255) // 256))
            if self.slab is not None and gi not in self.gather_blocks:  # planes -GH .. n + GH of the sharded axis, GH = 2 (slab_traced.G)
                per = tot // self._field_shape(key)[self.slab[0]]
                nblk = "(unsigned)(((long){} * ({} + 4) + 255) / 256)".format(per, self.slab[1])
            S.append("    case {}: hipLaunchKernelGGL(k_gat_{}, dim3({}), dim3(NB), 0, (hipStream_t)stream, *a, (T*)g, ad); break;".format(
                gi, gi, nblk))
        S.append("    default: return -1;")
        S.append("  }")
        S.append("  return (int)hipGetLastError();")
        S.append("}")
        if getattr(self, "jac_items", None):
            S.append('extern "C" int jit_jac(const Args* a, void* const* ptrs, void* stream) {')
            S.append("  JacP jp;")
            S.append("  for (int k = 0; k < {}; ++k) jp.p[k] = (T*)ptrs[k];".format(len(self.jac_items)))
            S.append("  const AdamP ad = {nullptr, nullptr, nullptr, (T)0, (T)0, (T)0, (T)0, nullptr};")
            S.append("  hipLaunchKernelGGL(k_jac, dim3({}), dim3(NB), 0, (hipStream_t)stream, *a, jp, ad);".format(self.jac_blocks))
            S.append("  return (int)hipGetLastError();")
            S.append("}")
        # the argument block: sized last (the gradient expressions add host scalars of their own)
        nsrc = max(1, len(self.src_keys))
        slab_members = ""
        if self.slab is not None:
            # off: global index of the first owned cell; lo / ea: ghost cells below the owned ones / extent of the
            # local arrays along the sharded axis; hw: cells in a wrap plane buffer; wlo / whi: wrap planes of the
            # sources (read), gwlo / gwhi: of the gradients (written by the gathers where the array has no ghosts)
            slab_members = " int off, lo, ea, hw; const T* wlo[{0}]; const T* whi[{0}]; T* gwlo[{0}]; T* gwhi[{0}]; int alo, ahi;".format(nsrc)
        # host scalars (functions of `tracers`): BY VALUE in the argument struct (hsv) -- an eager launch owns its
        # copy, nothing the host rewrites later is read by a queued kernel; a launch captured into a hipGraph
        # reads them from device memory instead (hs != NULL: the row of the epoch being replayed)
        HEAD.append("struct Args {{ const T* src[{}]; const void* ten[{}]; T* cot[{}]; const T* par[{}]; const double* hs; "
                    "double hsv[{}]; T* part; T* ppart; T* part2; T* out; T* pgrad; T* edge; int nblocks;{} }};".format(
                        nsrc, max(1, len(self.tr.tensors)), max(1, self.ncot),
                        max(1, par_arrays), max(1, len(self.hs)), slab_members))
        HEAD.append("#define HS(i) (a.hs ? a.hs[i] : a.hsv[i])")

    # ---- network evaluations shared by MARCHING (float kernels with a pointwise network at the faces) ---------------------
    def _march_parts(self):
        """Line groups of the marching forward kernel (see _march_kernel): per variant (general / interior) the forward
        lines before and after the shared network values, and the reverse pass."""
        a1, a2 = self.ndim - 2, self.ndim - 1
        self.wregs = "const"
        by_axis = {axis: (A, B) for A, B, axis in self.share}
        (Ax, Bx), (Ay, By) = by_axis[a1], by_axis[a2]
        shared = {x.idx for x in (Ax, Bx, Ay, By)}
        late = set()
        for n in self.order:
            if (n.op == "mlp_out" and n.args[0].idx in shared) or any(a.idx in late for a in n.args):
                late.add(n.idx)
        early = {n.idx for n in self.order} - late
        attr = Bx.attr
        nlast = len(attr[2]) - 1
        nz, nin = attr[2][nlast], len(Bx.args)
        parts = dict(Ax=Ax, Bx=Bx, Ay=Ay, By=By, nz=nz, nin=nin, attr=attr, variants=[])
        plan = self._fold_plan(self.order, 1, windows=True)
        # a third copy for the strips that touch a wall of the LANE axis (2 of 9 at 512 columns): predicates of the other
        # axes folded, those of the lane axis kept
        plan_w = self._fold_plan(self.order, 4, windows=True) if plan is not None else None
        if plan_w is not None and (plan_w[0] == plan[0] or a2 not in plan[1]):
            plan_w = None  # (no predicate of the lane axis: the interior copy serves every strip)
        parts["plan"], parts["plan_w"] = plan, plan_w
        keep = ("cots", "cut_nodes", "jac_store", "pg_decl", "pg_offset", "pgrads", "pg2_used")
        first = None
        self.march_pref = dict()
        for fold in ([None] if plan is None else ([None, plan[0]] + ([plan_w[0]] if plan_w is not None else []))):
            self.fold, self.lines, self.loads = fold, [], dict()
            self.march_live, self.march_used = (None if fold is None else self._live_under(fold)), set()
            self.cots, self.cut_nodes, self.jac_store, self.pg_decl, self.pg_offset, self.pg2_used = [], [], [], [], dict(), set()
            self.forward(only=early)
            fwd1, self.lines = self.lines, []
            xin = [(self.r(Bx.args[k]), self.r(By.args[k]), self.r(Ay.args[k])) for k in range(nin)]
            self.forward(only=late)
            fwd2, self.lines = self.lines, []
            self.reverse()
            rev, self.lines = self.lines, []
            rev_text = "\n".join(rev)

            def adjoint_of(call, j):
                out = self.mlp_out_seen.get(call.idx, dict()).get(j)
                return "g{}".format(out.idx) if out is not None and "T g{} ".format(out.idx) in rev_text else "(T)0"

            adj = {name: [adjoint_of(call, j) for j in range(nz)] for name, call in (("ax", Ax), ("bx", Bx), ("ay", Ay), ("by", By))}
            state = {k: getattr(self, k) for k in keep}
            if first is None:
                first = state
            elif ([n.idx for n in state["cots"]] != [n.idx for n in first["cots"]] or state["pg_decl"] != first["pg_decl"]
                  or [n.idx for n in state["cut_nodes"]] != [n.idx for n in first["cut_nodes"]]):
                raise RuntimeError("interior copy of the traced kernel stores other adjoints than the general one")
            parts["variants"].append(dict(fwd1=fwd1, fwd2=fwd2, rev=rev, xin=xin, adj=adj, used=set(self.march_used)))
        self.fold, self.march_live = None, None
        parts["pref"] = sorted(self.march_pref.values()) if self.march_pref else []
        self.march_pref = None
        for k, v in first.items():
            setattr(self, k, v)
        parts["gather"] = self._march_gather_plan()
        # the pre-step of a row segment: the inputs of the LOWER face along the marching axis at the segment's first row
        saved = (self.order, self.loads, self.pre, self.groups)
        seen = dict()
        for arg in Ax.args:
            for n in stencil_grad.subdag(arg):
                seen[n.idx] = n
        self.order = [seen[i] for i in sorted(seen)]
        self.loads, self.pre, self.groups, self.lines = dict(), [], dict(), []
        self.forward()
        parts["pre_lines"], parts["pre_in"] = self.lines, [self.r(arg) for arg in Ax.args]
        seen = dict()
        for arg in Ay.args:
            for n in stencil_grad.subdag(arg):
                seen[n.idx] = n
        self.order = [seen[i] for i in sorted(seen)]
        self.loads, self.pre, self.groups, self.lines = dict(), [], dict(), []
        self.forward()
        parts["pre_lines_y"], parts["pre_in_y"] = self.lines, [self.r(arg) for arg in Ay.args]
        self.order, self.loads, self.pre, self.groups = saved
        # the packed evaluation (prefix mu) and the reverse pass of the PREVIOUS step's evaluation (prefix mp)
        self.lines = []
        self._mlp_forward("mu", 2, attr, [("ux{}_0".format(k), "ux{}_1".format(k)) for k in range(nin)])
        parts["mlp_fwd"], self.lines = self.lines, []
        self._mlp_backward("mp", 2, attr, [("ud{}_0".format(j), "ud{}_1".format(j)) for j in range(nz)], False)
        parts["mlp_bwd"], self.lines = self.lines, []
        self._mlp_backward("mu", 2, attr, [("ud{}_0".format(j), "ud{}_1".format(j)) for j in range(nz)], False)
        parts["mlp_bwd_mu"], self.lines = self.lines, []
        layers = attr[2]
        parts["acts"] = ["h{}_{}".format(l, i) for l in range(nlast) for i in range(layers[l])]  # what the reverse pass reads
        return parts

    def _march_gather_plan(self):
        """The cotangents of the reads of a marching kernel summed IN the kernel over the marching axis (a three-row delay
        line in registers) and over the lane axis (lane shifts) before they are stored: one array per (field, shift on the
        leading axes) instead of one per stencil read -- heat with two space dimensions: 2 instead of 10 (32 bytes per
        point less written by k_fwd and read again by the gather).  What crosses a segment of rows or a strip of columns
        goes to small EDGE arrays which the final gather adds.  None when the reads do not have that shape."""
        if self.cut_nodes or self.jac_store or not self.cots:
            return None
        a1, a2 = self.ndim - 2, self.ndim - 1
        groups = dict()
        for slot, n in enumerate(self.cots):
            key, shift, loc, _ = n.attr
            if not self._regular(n):
                return None
            cs = []
            for d, sh in enumerate(shift):
                ext = self.G[d]
                v = sh % ext
                cs.append(v - ext if v > ext // 2 else v)
            sx, sy = cs[a1], cs[a2]
            if abs(sx) > 1 or abs(sy) > 1 or (sx and sy):
                return None
            groups.setdefault((key, tuple(cs[:a1])), []).append((slot, sx, sy))
        if any(adj is not None for adj in self.out_adj):
            return None
        return dict(groups=list(groups.items()))

    def _march_gather_geometry(self):
        a1, a2 = self.ndim - 2, self.ndim - 1
        G1, G2 = self.G[a1], self.G[a2]
        R = max(1, min(int(os.environ.get("ODIL_TRACE_MARCH_ROWS", 64)), G1, 64))  # (<= 64: one lane of the pre-step per row)
        nseg, nstrip = (G1 + R - 1) // R, (G2 + 63) // 64
        lead = int(np.prod(self.G[:a1])) if a1 > 0 else 1
        return R, nseg, nstrip, lead

    def _march_edge_offsets(self, ngroups):
        """Element offsets into a.edge of the four edge arrays of every group: E_lo, E_hi [lead, nseg, G2] (rows that
        cross a segment), F_lo, F_hi [lead, G1, nstrip] (columns that cross a strip); total size."""
        a1, a2 = self.ndim - 2, self.ndim - 1
        R, nseg, nstrip, lead = self._march_gather_geometry()
        esz, fsz = lead * nseg * self.G[a2], lead * self.G[a1] * nstrip
        offs, o = [], 0
        for _ in range(ngroups):
            offs.append((o, o + esz, o + 2 * esz, o + 2 * esz + fsz))
            o += 2 * esz + 2 * fsz
        return offs, o

    def _march_gather_kernels(self, S, plan):
        """The final gathers of a marching kernel with the in-kernel partial sums (_march_gather_plan): per field
        g[j] = sum over its groups of (P + edge terms)[j - leading shift], four points of the last axis per thread."""
        a1, a2 = self.ndim - 2, self.ndim - 1
        G1, G2 = self.G[a1], self.G[a2]
        R, nseg, nstrip, lead = self._march_gather_geometry()
        groups = plan["groups"]
        offs, _ = self._march_edge_offsets(len(groups))
        vw = 4 if G2 % 4 == 0 else 1
        keys = []
        for (key, _), _ in groups:
            if key not in keys:
                keys.append(key)
        self.gathers, self.direct, self.merged, self.gather_blocks = list(keys), dict(), [], dict()
        self.gather_reads_sources = {key: [] for key in keys}
        for gi, key in enumerate(keys):
            threads = self.total // vw
            self.gather_blocks[gi] = (threads + 255) // 256
            S.append('extern "C" __global__ __launch_bounds__(NB) void k_gat_{}(const Args a, T* __restrict__ g, const AdamP ad) {{'.format(gi))
            S.append("  const int lr = blockIdx.x * NB + threadIdx.x;")
            S.append("  if (lr >= {}) return;".format(threads))
            names = ["i{}".format(d) for d in range(self.ndim)]
            self._index_prologue(S, self.G, names, vw, "lr")
            if vw == 1:
                S.append("  const int ib = i{};".format(a2))
            S.append("  T acc[{}];".format(vw))
            S.append("  for (int p = 0; p < {}; ++p) acc[p] = (T)0;".format(vw))
            S.append("  const int seg = i{} / {};".format(a1, R))
            for k, ((gkey, lshift), _) in enumerate(groups):
                if gkey != key:
                    continue
                e_lo, e_hi, f_lo, f_hi = offs[k]
                S.append("  {")
                # the point this group's sums were formed at: j - leading shift (periodic)
                lidx = []
                for d in range(a1):
                    lidx.append("i{}".format(d) if lshift[d] == 0 else "wrap(i{} - ({}), {})".format(d, lshift[d], self.G[d]))
                lflat = self._offset(lidx, self.G[:a1]) if a1 > 0 else "0"
                S.append("  const int lf = {};".format(lflat))
                S.append("  const T* const P = a.cot[{}] + ((long)lf * {} + i{}) * {};".format(k, G1, a1, G2))
                if vw == 4:
                    S.append("  { const T4 q = *(const T4*)(P + ib); acc[0] += q.x; acc[1] += q.y; acc[2] += q.z; acc[3] += q.w; }")
                else:
                    S.append("  acc[0] += P[ib];")
                # rows that received a contribution from the neighbouring segment
                S.append("  if (i{0} % {1} == {1} - 1 || i{0} == {2}) {{".format(a1, R, G1 - 1))
                S.append("    const T* const E = a.edge + {} + ((long)lf * {} + (seg + 1 == {} ? 0 : seg + 1)) * {};".format(e_lo, nseg, nseg, G2))
                S.append("    for (int p = 0; p < {}; ++p) acc[p] += E[ib + p];".format(vw))
                S.append("  }")
                S.append("  if (i{} % {} == 0) {{".format(a1, R))
                S.append("    const T* const E = a.edge + {} + ((long)lf * {} + (seg == 0 ? {} : seg - 1)) * {};".format(e_hi, nseg, nseg - 1, G2))
                S.append("    for (int p = 0; p < {}; ++p) acc[p] += E[ib + p];".format(vw))
                S.append("  }")
                # columns that received a contribution from the neighbouring strip
                S.append("  for (int p = 0; p < {}; ++p) {{".format(vw))
                S.append("    const int c = ib + p, st = c / 64;")
                S.append("    const T* const F = a.edge + ((long)lf * {} + i{}) * {};".format(G1, a1, nstrip))
                S.append("    if (c % 64 == 63 || c == {}) acc[p] += F[{} + (st + 1 == {} ? 0 : st + 1)];".format(G2 - 1, f_lo, nstrip))
                S.append("    if (c % 64 == 0) acc[p] += F[{} + (st == 0 ? {} : st - 1)];".format(f_hi, nstrip - 1))
                S.append("  }")
                S.append("  }")
            o = "lr * 4" if vw == 4 else "lr"
            if vw == 4:
                if self.nt_streams:
                    S.append("  __builtin_nontemporal_store((T4){{acc[0], acc[1], acc[2], acc[3]}}, (T4*)(g + {}));".format(o))
                else:
                    S.append("  *(T4*)(g + {}) = (T4){{acc[0], acc[1], acc[2], acc[3]}};".format(o))
                S.append("  adam_apply4(ad, {}, acc);".format(o))
            else:
                S.append("  g[{}] = acc[0];".format(o))
                S.append("  adam_apply(ad, {}, acc[0]);".format(o))
            S.append("}")

    def _march_kernel(self, S, parts, stored, stream):
        """Body of the MARCHING forward kernel of an operator that evaluates one pointwise network at the faces of every
        cell (heat with two space dimensions: reference examples/heat/heat.py:86-98 per axis).  The lower face of cell i is
        the upper face of cell i - e (stencil_share.py proves it on the DAG), so half of the evaluations of the plain
        kernel -- and of their reverse passes, two thirds of its instructions -- are repeats.  Here a WAVE owns a strip of
        64 columns of the last axis and marches along the second-to-last axis over a segment of rows; every lane makes ONE
        packed evaluation per point: (upper face along the marching axis, upper face along the lane axis).

        * marching axis: the value of a point's upper face is carried in registers to the next row, where it is the lower
          face; the adjoint it collects there is added to its own before the reverse pass of the evaluation, which
          therefore runs ONE STEP LATE, from the previous step's activations (carried as well);
        * lane axis: lane L takes its lower face from lane L - 1 and returns the adjoint to it by wave-wide lane shifts
          (DPP: no LDS, no barrier);
        * a segment starts with a PRE-STEP whose packed evaluation holds the lower faces nobody hands over: slot 0 the
          marching axis' at the segment's first row (every lane its column), slot 1 the lane axis' at the strip's FIRST
          column -- lane j for row r0 + j (a segment has at most 64 rows).  Lane 0 fetches its row's value with
          v_readlane as the march goes and hands the adjoint back the same way; after the march a post-step repeats the
          pre-step's forward pass from the kept inputs and runs its reverse pass with the collected adjoints.  (Until
          this form a HELPER lane per wave evaluated the first column's lower face every step: 63 columns per wave,
          nine waves per row of 512 where eight suffice.)

        Values and adjoints of lanes without a point are masked; sums of network-parameter gradients are linear in the
        adjoints, so a face shared by two waves (or two segments) simply contributes from both sides.  The body exists
        twice: as traced, and with every index predicate folded to its interior value (_fold_plan); the branch is scalar
        (row index, strip and leading indices are wave-uniform)."""
        a1, a2 = self.ndim - 2, self.ndim - 1
        G1, G2 = self.G[a1], self.G[a2]
        R, nseg, nstrip, lead = self._march_gather_geometry()
        nitems = lead * nseg * nstrip
        nz, nin, attr = parts["nz"], parts["nin"], parts["attr"]
        nl = len(attr[2]) - 1
        Ax, Bx, Ay, By = parts["Ax"], parts["Bx"], parts["Ay"], parts["By"]
        acts = parts["acts"]
        S.append("  const int lane = threadIdx.x & 63;")
        S.append("  const int wave_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);")
        S.append("  for (int item = blockIdx.x * 4 + wave_; item < {}; item += a.nblocks * 4) {{".format(nitems))
        S.append("  const int strip = item % {}, seg = (item / {}) % {};".format(nstrip, nstrip, nseg))
        rem = "(item / {})".format(nstrip * nseg)
        for d in reversed(range(a1)):
            if d == 0:
                S.append("  const int i0 = {};".format(rem))
            else:
                S.append("  const int i{} = {} % {};".format(d, rem, self.G[d]))
                S.append("  const int q{}_ = {} / {};".format(d, rem, self.G[d]))
                rem = "q{}_".format(d)
        S.append("  const int r0 = seg * {0}, r1 = min(r0 + {0}, {1});".format(R, G1))
        S.append("  const int p2 = strip * 64 + lane;")
        S.append("  const bool valid = p2 < {};".format(G2))
        S.append("  const int i{} = min(p2, {});".format(a2, G2 - 1))
        # interior test, scalar: leading indices, strip range; the row is tested per step
        plan, plan_w = parts["plan"], parts.get("plan_w")
        outer, rowc, leadc = [], [], []
        if plan is not None:
            _, exc, _ = plan
            for d, values in sorted(exc.items()):
                if d < a1:
                    outer.append(self._interior_cond({d: values}))
                    leadc.append(outer[-1])
                elif d == a1:
                    rowc.append(self._interior_cond({d: values}))
                else:  # no exceptional column among the strip's: [s0, s0 + 63]
                    values, lo, hi = list(values), 0, G2 - 1
                    while values and values[0] == lo:
                        values.pop(0)
                        lo += 1
                    while values and values[-1] == hi:
                        values.pop()
                        hi -= 1
                    if lo > 0:
                        outer.append("strip * 64 >= {}".format(lo))
                    if hi < G2 - 1:
                        outer.append("strip * 64 + 63 <= {}".format(hi))
                    outer.extend("!(strip * 64 <= {0} && {0} <= strip * 64 + 63)".format(e) for e in values)
            S.append("  const bool interior_ = {};".format(" && ".join(outer) or "true"))
            S.append("  const bool lead_ok_ = {};".format(" && ".join(leadc) or "true"))  # (the wall-strip copy's condition)
        # carried state
        mg = parts["gather"]
        if mg is not None:
            for k in range(len(mg["groups"])):
                S.append("  T ap{0} = (T)0, ac{0} = (T)0;".format(k))  # sums of rows r - 1 and r so far
        for j in range(nz):
            S.append("  T kx{0} = (T)0, gbx{0} = (T)0, gy{0} = (T)0;".format(j))
        for name in acts:
            S.append("  T2 mp_{0} = (T2)(0.0f);".format(name))
        # ---- pre-step: the LOWER faces the march cannot take from a neighbour -- slot 0: along the marching axis at row r0,
        # own column; slot 1: along the lane axis at the strip's FIRST column, lane j for row r0 + j (the strip's lane 0
        # fetches them with v_readlane as the march goes; its adjoints come back the same way and are passed through the
        # network's reverse pass after the march, from a second forward pass over the kept inputs)
        for k in range(nin):
            S.append("  T axin{0}, ayin{0};".format(k))
        S.append("  {")
        S.append("  const int i{} = r0;".format(a1))
        S.extend(parts["pre_lines"])
        for k in range(nin):
            S.append("  axin{} = {};".format(k, parts["pre_in"][k]))
        S.append("  }")
        S.append("  {")
        S.append("  const int i{} = min(r0 + lane, r1 - 1);".format(a1))
        S.append("  const int i{} = min(strip * 64, {});".format(a2, G2 - 1))
        S.extend(parts["pre_lines_y"])
        for k in range(nin):
            S.append("  ayin{} = {};".format(k, parts["pre_in_y"][k]))
        S.append("  }")
        for j in range(nz):
            S.append("  T kay{0}, gedge{0} = (T)0;".format(j))
        S.append("  {")
        for k in range(nin):
            S.append("  const T ux{0}_0 = axin{0}, ux{0}_1 = ayin{0};".format(k))
        S.extend(parts["mlp_fwd"])
        for j in range(nz):
            S.append("  kx{0} = mu_z{1}_{0}.x; kay{0} = mu_z{1}_{0}.y;".format(j, nl))
        for name in acts:
            S.append("  mp_{0} = mu_{0};".format(name))
        S.append("  }")
        # ---- the march ----------------------------------------------------------------------------------------------
        # the field values a step reads are requested during the step before (the step's arithmetic covers their
        # latency: two resident waves per SIMD cannot)
        pref = parts["pref"]
        if pref:
            S.append("  T {};".format(", ".join("ldn_{}".format(k) for k, _ in pref)))
            S.append("  {{ const int i{} = r0;".format(a1))
            for k, e in pref:
                S.append("    ldn_{} = {};".format(k, e))
            S.append("  }")
        S.append("  for (int i{0} = r0; i{0} < r1; ++i{0}) {{".format(a1))
        S.append("  const int l = {};".format(self._offset(["i{}".format(d) for d in range(self.ndim)], self.G)))
        if pref:
            for k, _ in pref:
                S.append("  const T ld_{0} = ldn_{0};".format(k))
            S.append("  {{ const int inext_ = min(i{0} + 1, r1 - 1); {{ const int i{0} = inext_;".format(a1))
            vs = parts["variants"]
            row = " && ".join(rowc) or "true"
            always = vs[1]["used"] if len(vs) > 1 else {k for k, _ in pref}
            wall = (vs[2]["used"] | always) if len(vs) > 2 else None
            for k, e in pref:
                if k in always:
                    S.append("    ldn_{} = {};".format(k, e))
            if wall is not None and len(wall) > len(always):  # what the wall-strip copy reads beyond the interior one
                S.append("    if (!(interior_ && {})) {{".format(row))
                for k, e in pref:
                    if k in wall and k not in always:
                        S.append("      ldn_{} = {};".format(k, e))
                S.append("    }")
            rest = [(k, e) for k, e in pref if k not in (wall if wall is not None else always)]
            if rest:  # what only the general copy of the body reads: when the next row takes that copy
                S.append("    if (!({} && {})) {{".format("lead_ok_" if wall is not None else "interior_", row))
                for k, e in rest:
                    S.append("      ldn_{} = {};".format(k, e))
                S.append("    }")
            S.append("  } }")
        for j in range(nz):
            S.append("  T gax{0}, gbc{0}, gay{0}, gby{0}, zx{0};".format(j))
        for name in acts:
            S.append("  T2 mc_{};".format(name))
        if mg is not None:
            for k in range(len(mg["groups"])):
                S.append("  T cm{0}, c0{0}, cp{0}, yl{0}, yr{0};".format(k))

        mg = parts["gather"]

        def body(var, inbox):
            B = []
            B.extend(self._inbox_lines(inbox))
            B.extend(var["fwd1"])
            for k, (bx, by, ay) in enumerate(var["xin"]):
                B.append("  const T ux{0}_0 = {1}, ux{0}_1 = {2};".format(k, bx, by))
            B.extend(parts["mlp_fwd"])
            for name in acts:
                B.append("  mc_{0} = mu_{0};".format(name))
            for j in range(nz):
                z = "mu_z{}_{}".format(nl, j)
                B.append("  const T m{}_z{}_{} = kx{};".format(Ax.idx, nl, j, j))
                B.append("  const T m{}_z{}_{} = {}.x;".format(Bx.idx, nl, j, z))
                B.append("  const T m{}_z{}_{} = {}.y;".format(By.idx, nl, j, z))
                # (two statements: a lane shift inside one arm of a conditional would run with lane 0 masked off, and a DPP
                # read from a disabled lane is invalid -- lane 1 would get 0)
                B.append("  const T ayp{0}_ = odil_lane_prev({1}.y), ay0{0}_ = odil_readlane(kay{0}, i{2} - r0);".format(j, z, a1))
                B.append("  const T m{0}_z{1}_{2} = lane == 0 ? ay0{2}_ : ayp{2}_;".format(Ay.idx, nl, j))
                B.append("  zx{} = {}.x;".format(j, z))
            B.extend(var["fwd2"])
            B.extend(var["rev"])
            if mg is not None:
                # partial sums of the read cotangents (_march_gather_plan): per group what this row gives to the rows
                # r - 1, r, r + 1 (cm, c0, cp; c0 with the lane neighbours' shares) and to the neighbouring strips
                for k, (_, members) in enumerate(mg["groups"]):
                    gsum = lambda sel: " + ".join("g{}".format(self.cots[slot].idx) for slot, sx, sy in members if sel(sx, sy)) or None
                    term = lambda e: "(valid ? {} : (T)0)".format(e) if e else None
                    c0, cm, cp = gsum(lambda sx, sy: sx == 0 and sy == 0), gsum(lambda sx, sy: sx == -1), gsum(lambda sx, sy: sx == 1)
                    yl, yr = gsum(lambda sx, sy: sy == -1), gsum(lambda sx, sy: sy == 1)  # to the column left / right
                    B.append("  cm{} = {};".format(k, term(cm) or "(T)0"))
                    B.append("  cp{} = {};".format(k, term(cp) or "(T)0"))
                    B.append("  yl{} = {};".format(k, term(yl) or "(T)0"))
                    B.append("  yr{} = {};".format(k, term(yr) or "(T)0"))
                    B.append("  c0{} = {};".format(k, term(c0) or "(T)0"))
            B.append("  if (valid) {")
            for slot, (n, name) in enumerate(stored if mg is None else []):
                if stream:
                    B.append("    __builtin_nontemporal_store({}, &a.cot[{}][l]);".format(name, slot))
                else:
                    B.append("    a.cot[{}][l] = {};".format(slot, name))
            for k, (o_, raw) in enumerate(zip(self.outputs, self.raw)):
                term = self.r(o_) if raw else "{0} * {0}".format(self.r(o_))
                if self.out_lens[k] is not None:
                    term = "(inbox{} ? {} : (T)0)".format(k, term)
                B.append("    s_{0} = s_{0} + {1};".format(k, term))
            B.append("  }")
            for j in range(nz):
                B.append("  gax{0} = valid ? {1} : (T)0; gbc{0} = valid ? {2} : (T)0;".format(j, var["adj"]["ax"][j], var["adj"]["bx"][j]))
                B.append("  gay{0} = valid ? {1} : (T)0; gby{0} = valid ? {2} : (T)0;".format(j, var["adj"]["ay"][j], var["adj"]["by"][j]))
            return B

        variants = parts["variants"]
        if len(variants) == 1:
            S.extend(body(variants[0], ()))
        else:
            S.append("  if (interior_ && {}) {{".format(" && ".join(rowc) or "true"))
            S.extend(body(variants[1], plan[2]))
            if len(variants) > 2:
                S.append("  }} else if (lead_ok_ && {}) {{".format(" && ".join(rowc) or "true"))
                S.extend(body(variants[2], plan_w[2]))
            S.append("  } else {")
            S.extend(body(variants[0], ()))
            S.append("  }")
        if mg is not None:
            offs, _ = self._march_edge_offsets(len(mg["groups"]))
            lead_flat = self._offset(["i{}".format(d) for d in range(a1)], self.G[:a1]) if a1 > 0 else "0"
            S.append("  const long lf_ = {};".format(lead_flat))
            S.append("  const int lastl_ = min(63, {} - strip * 64);".format(G2 - 1))  # the strip's last lane with a point
            for k in range(len(mg["groups"])):
                e_lo, e_hi, f_lo, f_hi = offs[k]
                # the row's own sum: the lane neighbours' shares arrive by lane shifts; what leaves the strip goes to F
                S.append("  const T fromr{0} = odil_lane_next(yl{0}), froml{0} = odil_lane_prev(yr{0});".format(k))
                S.append("  const T row{0} = c0{0} + (valid ? froml{0} + fromr{0} : (T)0);".format(k))
                S.append("  if (lane == 0) a.edge[{} + (lf_ * {} + i{}) * {} + strip] = yl{};".format(f_lo, G1, a1, nstrip, k))
                S.append("  if (lane == lastl_) a.edge[{} + (lf_ * {} + i{}) * {} + strip] = yr{};".format(f_hi, G1, a1, nstrip, k))
                # delay line along the marching axis: row r - 1 is complete (within the segment) once row r has given its share
                S.append("  if (i{} > r0) {{ if (valid) {}; }}".format(
                    a1, ("__builtin_nontemporal_store(ap{0} + cm{0}, &a.cot[{0}][l - {1}])" if stream else "a.cot[{0}][l - {1}] = ap{0} + cm{0}").format(k, G2)))
                S.append("  else if (valid) a.edge[{} + (lf_ * {} + seg) * {} + i{}] = cm{};".format(e_lo, nseg, G2, a2, k))
                S.append("  ap{0} = ac{0} + row{0}; ac{0} = cp{0};".format(k))
        for j in range(nz):  # what lane 0 found for the strip's first lower face of this row: back to the lane that evaluated it
            S.append("  {{ const T g0_ = odil_readlane(gay{0}, 0); gedge{0} = lane == i{1} - r0 ? g0_ : gedge{0}; }}".format(j, a1))
        # reverse pass of the PREVIOUS step's evaluation: its own adjoints + what this row found for the carried face
        S.append("  {")
        for j in range(nz):
            S.append("  const T ud{0}_0 = gbx{0} + gax{0}, ud{0}_1 = gy{0};".format(j))
        S.extend(parts["mlp_bwd"])
        S.append("  }")
        for j in range(nz):
            S.append("  kx{0} = zx{0}; gbx{0} = gbc{0}; gy{0} = gby{0} + odil_lane_next(gay{0});".format(j))
        for name in acts:
            S.append("  mp_{0} = mc_{0};".format(name))
        S.append("  }")  # rows
        if mg is not None:
            # the segment's last row (what the next segment's first row gives it arrives through E_lo), and what the last
            # row gives to the next segment's first row
            offs, _ = self._march_edge_offsets(len(mg["groups"]))
            lead_flat = self._offset(["i{}".format(d) for d in range(a1)], self.G[:a1]) if a1 > 0 else "0"
            S.append("  if (valid) {")
            S.append("    const long lf_ = {};".format(lead_flat))
            last_l = self._offset(["i{}".format(d) if d != a1 else "(r1 - 1)" for d in range(self.ndim)], self.G)
            S.append("    const int ll_ = {};".format(last_l))
            for k in range(len(mg["groups"])):
                e_lo, e_hi, f_lo, f_hi = offs[k]
                S.append("    a.cot[{0}][ll_] = ap{0};".format(k))
                S.append("    a.edge[{} + (lf_ * {} + seg) * {} + i{}] = ac{};".format(e_hi, nseg, G2, a2, k))
            S.append("  }")
        # flush: the last row's evaluation (its upper face along the marching axis belongs to the next segment too)
        S.append("  {")
        for j in range(nz):
            S.append("  const T ud{0}_0 = gbx{0}, ud{0}_1 = gy{0};".format(j))
        S.extend(parts["mlp_bwd"])
        S.append("  }")
        # post-step: the reverse pass of the pre-step's second slot (the lane axis' lower faces of the first column)
        S.append("  {")
        for k in range(nin):
            S.append("  const T ux{0}_0 = axin{0}, ux{0}_1 = ayin{0};".format(k))
        S.extend(parts["mlp_fwd"])
        for j in range(nz):
            S.append("  const T ud{0}_0 = (T)0, ud{0}_1 = lane < r1 - r0 ? gedge{0} : (T)0;".format(j))
        S.append("  {")
        S.extend(parts["mlp_bwd_mu"])
        S.append("  }")
        S.append("  }")
        S.append("  }")  # items

    def _gather_symbolic(self, S, gi, key, root):
        """The gather of ONE regular field as a pointwise kernel over its gradient expression."""
        self.gather_blocks[gi] = self._gather_kernel(S, "k_gat_{}".format(gi), [(key, root)],
                                                     "T* __restrict__ g, const AdamP ad", lambda k: "g", lambda k: "ad")

    def _gather_kernel(self, S, name, items, params, G_, AD_):
        """A pointwise kernel over the gradient expressions of `items` = [(field key, expression)] (one thread per
        point, or per four points of the last axis): common sub-expressions and loads of the fields' expressions are
        shared, every field's gradient is stored, and the optimizer's update applied, by the lane that holds it.
        Slab mode: threads cover planes -2 .. n + 2 of the sharded axis; planes that exist in the rank's ghost-extended
        gradient array are stored there (ghost planes: what this rank's cells contribute to the neighbour's), planes
        beyond an end of the decomposition that a periodic read reached go to the wrap buffers (as the legacy slab
        gather).  Returns the number of workgroups to launch."""
        self.vw = self.vw_gat
        vw, last = self.vw, self.ndim - 1
        saved = (self.order, self.lines, self.pre, self.loads, self.groups)
        seen = dict()
        for _, root in items:
            for n in stencil_grad.subdag(root):
                seen[n.idx] = n
        nodes = [seen[i] for i in sorted(seen)]
        self.order = nodes
        self._begin()
        self.in_gather = True
        self.forward()
        body = self.lines
        interior = self._interior_copy(nodes, vw)
        self.in_gather = False
        self.vw = self.vw_fwd
        pre = self.pre + self._group_arrays()
        values = [self.r(root) for _, root in items]
        sources = sorted({n.attr[0] for n in nodes if n.op == "read" and not n.attr[0].startswith("@")})
        for key, _ in items:
            self.gather_reads_sources[key] = sorted(set(self.gather_reads_sources.get(key, [])) | set(sources))
        self.order, self.lines, self.pre, self.loads, self.groups = saved
        shape = list(self.G)
        names = ["i{}".format(d) for d in range(self.ndim)]
        if self.slab is not None:
            ax, nloc = self.slab
            shape[ax] = nloc + 4
            names[ax] = "jx"
        threads = int(np.prod(shape)) // vw
        if threads >= 2**31 - 1024:
            raise TraceUnsupported("grid too large for 32-bit indexing")
        flat = "l4" if vw == 4 else "l"
        occ = 0
        S.append('extern "C" __global__ __launch_bounds__(NB) {}void {}(const Args a, {}) {{'.format(
            "__attribute__((amdgpu_waves_per_eu({0}, {0}))) ".format(occ) if occ else "", name, params))
        S.append(self._block_index(shape, vw))
        S.append("  const int {}r = bx_ * NB + threadIdx.x;".format(flat))
        S.append("  if ({}r >= {}) return;".format(flat, threads))
        self._chunk_remap(S, shape, vw, flat + "r", flat)
        self._index_prologue(S, shape, names, vw, flat)
        nblocks = (threads + 255) // 256
        if self.slab is not None:
            S.append("  const int jo = jx - 2;")  # owned-relative position on the sharded axis
            S.append("  const int i{}g = wrap(jo + a.off, {});".format(ax, self.G[ax]))
        S.extend(pre)
        for k in range(len(items)):
            S.append("  T acc{}[{}];".format(k, vw))

        def point_block(body_):
            B = []
            self._loop_open(B, vw)
            B.extend(body_)
            for k, value in enumerate(values):
                B.append("  acc{}[{}] = {};".format(k, "p" if vw == 4 else "0", value))
            if vw == 4:
                B.append("  }")
            return B

        if interior is None:
            S.extend(point_block(body))
        else:
            S.append("  if (__all((int)({}))) {{".format(self._interior_cond(interior[1][1])))
            S.extend(point_block(interior[0][0]))
            S.append("  } else {")
            S.extend(point_block(body))
            S.append("  }")
        adam = "adam_apply4({ad}, {o}, acc{k});" if vw == 4 else "adam_apply({ad}, {o}, acc{k}[0]);"
        put = "*(T4*)({dst} + {o}) = (T4){{acc{k}[0], acc{k}[1], acc{k}[2], acc{k}[3]}};" if vw == 4 else "{dst}[{o}] = acc{k}[0];"
        if vw == 4 and self.nt_streams:  # (with the optimizer state, see adam_apply4)
            put = "__builtin_nontemporal_store((T4){{acc{k}[0], acc{k}[1], acc{k}[2], acc{k}[3]}}, (T4*)({dst} + {o}));"
        if self.slab is None:
            o = "l4 * 4" if vw == 4 else "l"
            for k in range(len(items)):
                S.append("  " + put.format(dst=G_(k), o=o, k=k))
                S.append("  " + adam.format(ad=AD_(k), o=o, k=k))
            S.append("}")
            return nblocks

        def offset(along, extent):
            full = [along if d == ax else ("ib" if (vw == 4 and d == last) else "i{}".format(d)) for d in range(self.ndim)]
            ext = [extent if d == ax else self.G[d] for d in range(self.ndim)]
            return self._offset(full, ext)

        S.append("  const int jl = jo + a.lo;")
        # owned planes a.alo <= jo < a.ahi have their whole gradient here (no neighbour's cell reads them): the optimizer's
        # update is applied on the spot; the planes next to an interface wait for the halo sum (slab_traced.py)
        S.append("  if (jl >= 0 && jl < a.ea) {")
        S.append("    const int o = {};".format(offset("jl", "a.ea")))
        for k in range(len(items)):
            S.append("    " + put.format(dst=G_(k), o="o", k=k))
            S.append("    if (jo >= a.alo && jo < a.ahi) " + adam.format(ad=AD_(k), o="o", k=k))
        S.append("  }")
        S.append("  else if (jo < 0 && jo >= -a.hw) {")
        S.append("    const int o = {};".format(offset("(jo + a.hw)", "a.hw")))
        for k, (key, _) in enumerate(items):
            S.append("    " + put.format(dst="a.gwlo[{}]".format(self.src_keys.index(key)), o="o", k=k))
        S.append("  }")
        S.append("  else if (jo >= {0} && jo < {0} + a.hw) {{".format(nloc))
        S.append("    const int o = {};".format(offset("(jo - {})".format(nloc), "a.hw")))
        for k, (key, _) in enumerate(items):
            S.append("    " + put.format(dst="a.gwhi[{}]".format(self.src_keys.index(key)), o="o", k=k))
        S.append("  }")
        S.append("}")
        return nblocks


def _gather_slab(self, S, gi, key, reads, floc, fshape):
    """Gather of one field on one rank's slab.  Threads cover planes -2 .. n + 2 of the sharded axis (owned
    cells 0 .. n): g = sum_r cot_r[j - shift_r] over the OWNED cells that read j.  Planes that exist in the
    rank's ghost-extended gradient array are stored there (ghost planes: the part of the neighbour's gradient
    that this rank's cells produce, sent over and added by slab_traced.py); planes beyond a side WITHOUT ghosts
    (the ends of the decomposition) that a periodic read reached go to the wrap buffers gwlo / gwhi."""
    ax, nloc = self.slab
    slot = self.src_keys.index(key)
    per = [fshape[d] for d in range(self.ndim)]
    S.append('extern "C" __global__ __launch_bounds__(NB) void k_gat_{}(const Args a, T* __restrict__ g, const AdamP ad) {{'.format(gi))
    tot_per = int(np.prod([fshape[d] for d in range(self.ndim) if d != ax]))
    # 32-bit index arithmetic whenever the thread space fits (divisions by constants: a 64-bit one costs ~4x)
    it = "int" if tot_per * (nloc + 4) < 2**31 - 512 else "long"
    S.append("  const {0} l = ({0})blockIdx.x * NB + threadIdx.x;".format(it))
    S.append("  if (l >= ({}){} * {}) return;".format(it, tot_per, nloc + 4))
    rem = "l"
    for d in reversed(range(self.ndim)):
        ext = (nloc + 4) if d == ax else per[d]
        if d == 0:
            S.append("  const int j0 = (int){};".format(rem))
        else:
            S.append("  const int j{} = (int)({} % {});".format(d, rem, ext))
            S.append("  const {} q{} = {} / {};".format(it, d, rem, ext))
            rem = "q{}".format(d)
    S.append("  const int jo = j{} - 2;".format(ax))  # owned-relative position on the sharded axis
    S.append("  T acc = (T)0;")
    loads = []
    for entry, (cslot, attr, coeff) in enumerate(reads):
        _, shift, loc, _ = attr
        idx, valid = [], []
        for d in range(self.ndim):
            ns, nr = fshape[d], self.G[d]
            ext = max(ns, nr)
            s_ = shift[d] % ext
            if s_ > ext // 2:
                s_ -= ext
            if d == ax:
                # the load is UNCONDITIONAL on a clamped position and masked afterwards: loads behind per-entry
                # branches are issued one at a time (each waits for the previous one's branch)
                name = "c{}".format(entry)
                S.append("  const int {} = jo - ({});".format(name, s_))
                valid.append("{0} >= 0 && {0} < {1}".format(name, nloc))
                idx.append("min(max({}, 0), {})".format(name, nloc - 1))
                continue
            pos = "j{}".format(d) if not (floc[d] == "c" and loc[d] == "n") else "(j{} + 1)".format(d)
            e = pos if s_ == 0 else "wrap({} - ({}), {})".format(pos, s_, ext)
            if floc[d] == "n" and loc[d] == "c":
                name = "t{}_{}".format(entry, d)
                S.append("  const int {} = {};".format(name, e))
                valid.append("{} < {}".format(name, nr))
                e = "min({}, {})".format(name, nr - 1)  # (the masked load stays inside the array)
            idx.append(e)
        S.append("  const T w{} = a.cot[{}][{}];".format(entry, cslot, self._offset(idx, self.GL)))
        loads.append((entry, coeff, " && ".join(valid)))
    for entry, coeff, valid in loads:  # every load above is in flight before the first use
        term = "w{}".format(entry) if coeff is None else "({}) * w{}".format(coeff, entry)
        S.append("  acc = acc + (({}) ? {} : (T)0);".format(valid, term))

    def offset(along, extent):
        full = [along if d == ax else "j{}".format(d) for d in range(self.ndim)]
        shape = [extent if d == ax else per[d] for d in range(self.ndim)]
        return self._offset(full, shape)

    S.append("  const int jl = jo + a.lo;")
    # owned planes a.alo <= jo < a.ahi have their whole gradient here (no neighbour's cell reads them): the optimizer's
    # update is applied on the spot; the planes next to an interface wait for the halo sum (slab_traced.py)
    S.append("  if (jl >= 0 && jl < a.ea) {")
    S.append("    const int o = {};".format(offset("jl", "a.ea")))
    S.append("    g[o] = acc;")
    S.append("    if (jo >= a.alo && jo < a.ahi) adam_apply(ad, o, acc);")
    S.append("  }")
    S.append("  else if (jo < 0 && jo >= -a.hw) a.gwlo[{}][{}] = acc;".format(slot, offset("(jo + a.hw)", "a.hw")))
    S.append("  else if (jo >= {0} && jo < {0} + a.hw) a.gwhi[{1}][{2}] = acc;".format(nloc, slot, offset("(jo - {})".format(nloc), "a.hw")))
    S.append("}")


_Codegen._gather_slab = _gather_slab


def _cache_dirs():
    """In-tree cache first (travels with the checkout); a PRIVATE per-user directory if that is read-only."""
    yield _CACHE_DIR
    yield os.path.join(os.path.expanduser("~"), ".cache", "odil_amd_jit")


def _trusted(d):
    """A cache directory libraries may be LOADED from: the in-tree one (whoever can write there can rewrite this
    module as well), or one owned by this user and writable by nobody else (a shared temp directory with a
    predictable name could be pre-created by another user with a planted library in it)."""
    if d == _CACHE_DIR:
        return True
    try:
        st = os.stat(d)
    except OSError:
        return False
    return st.st_uid == os.getuid() and not (st.st_mode & 0o022)


def _compile(src, flags=None):
    flags = flags or _HIPCC_FLAGS
    tag = hashlib.sha256((src + " ".join(flags)).encode()).hexdigest()[:20]
    name = "odil_jit_{}.so".format(tag)
    for d in _cache_dirs():
        if os.path.exists(os.path.join(d, name)) and _trusted(d):
            try:
                os.utime(os.path.join(d, name))  # last use: lets a cache be pruned by age (tools/final_r3.sh)
            except OSError:
                pass
            return ctypes.CDLL(os.path.join(d, name)), os.path.join(d, name)
    last = None
    for d in _cache_dirs():
        try:
            os.makedirs(d, mode=0o700, exist_ok=True)
            if not _trusted(d):
                raise OSError("cache directory {} is not private to this user".format(d))
            # source and library are written under temporary names and renamed: ranks that compile the same
            # operator at the same time never read each other's half-written files
            fd, hip_tmp = tempfile.mkstemp(suffix=".hip", dir=d)
            with os.fdopen(fd, "w") as f:
                f.write(src)
            fd, tmp = tempfile.mkstemp(suffix=".so", dir=d)
            os.close(fd)
        except OSError as e:
            last = e
            continue
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        res = subprocess.run([hipcc] + flags + ["-o", tmp, hip_tmp], capture_output=True, text=True)
        hip = os.path.join(d, "odil_jit_{}.hip".format(tag))
        os.replace(hip_tmp, hip)  # kept beside the library for inspection
        if res.returncode != 0:
            os.unlink(tmp)
            raise RuntimeError("hipcc failed for the traced operator ({}):\n{}".format(hip, res.stderr[-4000:]))
        path = os.path.join(d, name)
        os.replace(tmp, path)  # atomic: concurrent ranks compiling the same source do not collide
        return ctypes.CDLL(path), path
    raise FileNotFoundError("no writable private cache directory for traced operators: {}".format(last))
