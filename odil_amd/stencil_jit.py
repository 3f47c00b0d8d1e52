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
import hashlib
import math
import os
import subprocess
import tempfile

import numpy as np
import torch

from . import ops
from .backend import numpy_dtype, torch_dtype


class TraceUnsupported(Exception):
    pass


_R, _B, _I = "r", "b", "i"
_CACHE_DIR = os.environ.get("ODIL_JIT_CACHE", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_jit_cache"))
_HIPCC_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "--offload-arch=gfx950"]


def _promote(*kinds):
    if _R in kinds:
        return _R
    if _I in kinds:
        return _I
    return _B


# ======================================================================================
# Symbolic values
# ======================================================================================
class Sym:
    """A node of the traced expression DAG; behaves like a device array in user code."""

    __array_ufunc__ = None  # NumPy operands defer to the reflected methods below
    # win: None, or (lens, squeezed) -- the value is a sub-box of the grid anchored at the origin
    # (`u[1:]`, `u[-1, k]` after the offsets were pushed into the reads); `shape` is what user code sees
    __slots__ = ("tr", "op", "args", "attr", "shape", "kind", "idx", "host", "win")

    def __init__(self, tr, op, args, attr, shape, kind, host, win=None):
        self.tr, self.op, self.args, self.attr = tr, op, args, attr
        self.shape, self.kind, self.host, self.win = tuple(shape), kind, host, win
        self.idx = len(tr.nodes)
        tr.nodes.append(self)

    # array-like surface user code touches
    @property
    def dtype(self):
        return self.tr.torch_dtype if self.kind == _R else (torch.bool if self.kind == _B else torch.int64)

    @property
    def ndim(self):
        return len(self.shape)

    def __hash__(self):
        return id(self)

    def __bool__(self):
        raise TraceUnsupported("host control flow on a device value")

    def __getitem__(self, item):
        return self.tr.getitem(self, item)

    def __len__(self):
        raise TraceUnsupported("len() of a symbolic array")

    def __iter__(self):
        raise TraceUnsupported("iterating a symbolic array")

    def __add__(self, o):
        return self.tr.binary("add", self, o)

    def __radd__(self, o):
        return self.tr.binary("add", o, self)

    def __sub__(self, o):
        return self.tr.binary("sub", self, o)

    def __rsub__(self, o):
        return self.tr.binary("sub", o, self)

    def __mul__(self, o):
        return self.tr.binary("mul", self, o)

    def __rmul__(self, o):
        return self.tr.binary("mul", o, self)

    def __truediv__(self, o):
        return self.tr.binary("div", self, o)

    def __rtruediv__(self, o):
        return self.tr.binary("div", o, self)

    def __pow__(self, o):
        return self.tr.binary("pow", self, o)

    def __rpow__(self, o):
        return self.tr.binary("pow", o, self)

    def __neg__(self):
        return self.tr.unary("neg", self)

    def __pos__(self):
        return self

    def __abs__(self):
        return self.tr.unary("abs", self)

    def __lt__(self, o):
        return self.tr.binary("lt", self, o)

    def __le__(self, o):
        return self.tr.binary("le", self, o)

    def __gt__(self, o):
        return self.tr.binary("gt", self, o)

    def __ge__(self, o):
        return self.tr.binary("ge", self, o)

    def __eq__(self, o):
        return self.tr.binary("eq", self, o)

    def __ne__(self, o):
        return self.tr.binary("ne", self, o)

    def __and__(self, o):
        return self.tr.binary("and", self, o)

    def __rand__(self, o):
        return self.tr.binary("and", o, self)

    def __or__(self, o):
        return self.tr.binary("or", self, o)

    def __ror__(self, o):
        return self.tr.binary("or", o, self)

    def __invert__(self):
        return self.tr.unary("not", self)


_HOST_UNARY = {
    "neg": lambda a: -a, "abs": abs, "cos": math.cos, "sin": math.sin, "exp": math.exp, "log": math.log,
    "tanh": math.tanh, "sqrt": math.sqrt, "floor": math.floor, "not": lambda a: not a, "cast": float,
    "stopgrad": lambda a: a, "relu": lambda a: max(a, 0),
}
_HOST_BINARY = {
    "add": lambda a, b: a + b, "sub": lambda a, b: a - b, "mul": lambda a, b: a * b, "div": lambda a, b: a / b,
    "pow": lambda a, b: a ** b, "min": min, "max": max, "lt": lambda a, b: a < b, "le": lambda a, b: a <= b,
    "gt": lambda a, b: a > b, "ge": lambda a, b: a >= b, "eq": lambda a, b: a == b, "ne": lambda a, b: a != b,
    "and": lambda a, b: bool(a) and bool(b), "or": lambda a, b: bool(a) or bool(b), "atan2": math.atan2,
}
_CMP = {"lt": "<", "le": "<=", "gt": ">", "ge": ">=", "eq": "==", "ne": "!="}


class Tracer:
    def __init__(self, domain):
        self.domain = domain
        self.real_mod = domain.mod
        self.torch_dtype = torch_dtype(domain.dtype)
        self.nodes = []
        self.cse = dict()
        self.tensors = []  # concrete device tensors referenced by 'tensor' leaves

    # ---- node construction ---------------------------------------------------------------
    def node(self, op, args=(), attr=None, shape=(), kind=_R, host=False, win=None):
        key = (op, tuple(a.idx for a in args), attr, tuple(shape), kind, win)
        try:
            hit = self.cse.get(key)
        except TypeError:
            key, hit = None, None
        if hit is not None:
            return hit
        n = Sym(self, op, tuple(args), attr, shape, kind, host, win)
        if key is not None:
            self.cse[key] = n
        return n

    def const(self, value):
        if isinstance(value, (bool, np.bool_)):
            return self.node("const", attr=bool(value), kind=_B, host=True)
        if isinstance(value, (int, np.integer)):
            return self.node("const", attr=int(value), kind=_I, host=True)
        return self.node("const", attr=float(value), kind=_R, host=True)

    def lift(self, x):
        if isinstance(x, Sym):
            if x.tr is not self:
                raise TraceUnsupported("value from another trace")
            return x
        if isinstance(x, (bool, int, float, np.bool_, np.integer, np.floating)):
            return self.const(x)
        if isinstance(x, np.ndarray):
            x = self.real_mod.array(x)
        if isinstance(x, torch.Tensor):
            if x.requires_grad:
                raise TraceUnsupported("differentiable tensor outside ctx.field / ctx.neural_net")
            if x.dim() == 0:
                return self.const(x.item())
            return self.tensor(x)
        raise TraceUnsupported("operand of type {}".format(type(x).__name__))

    def tensor(self, t):
        if t.dtype not in (torch.float32, torch.float64, torch.int32, torch.int64, torch.bool):
            raise TraceUnsupported("tensor dtype {}".format(t.dtype))
        if t.device != self.real_mod.device:
            t = t.to(self.real_mod.device)
        t = t.detach().contiguous()
        for slot, old in enumerate(self.tensors):
            if old.data_ptr() == t.data_ptr() and old.shape == t.shape and old.dtype == t.dtype:
                break
        else:
            slot = len(self.tensors)
            self.tensors.append(t)
        kind = _R if t.dtype.is_floating_point else (_B if t.dtype == torch.bool else _I)
        return self.node("tensor", attr=slot, shape=tuple(t.shape), kind=kind)

    @staticmethod
    def _bshape(*shapes):
        try:
            return tuple(np.broadcast_shapes(*shapes))
        except ValueError as e:
            raise TraceUnsupported(str(e))

    def _combine(self, nodes):
        """(args, shape, win) of an elementwise operation: windowed operands must agree, tensors that
        meet a window with squeezed axes are re-aligned to the full grid rank."""
        wins = {n.win for n in nodes if n.win is not None}
        if len(wins) > 1:
            raise TraceUnsupported("operands cover different parts of the grid")
        win = wins.pop() if wins else None
        if win is not None and any(win[1]):
            fixed = []
            for n in nodes:
                if n.win is None and n.shape != ():
                    n = self._realign(n, win)
                fixed.append(n)
            nodes = fixed
            shape = tuple(l for l, q in zip(*win) if not q)
        else:
            shape = self._bshape(*[n.shape for n in nodes])
            if win is not None and shape != tuple(win[0]):
                raise TraceUnsupported("broadcast of a sliced field value to {}".format(shape))
        return nodes, shape, win

    def _realign(self, n, win):
        """Array expression of the user-visible shape (constant arrays, possibly scaled by scalars) ->
        the same expression with unit axes inserted where the window is squeezed."""
        if n.shape == () or n.host:
            return n
        if n.op != "tensor":
            if n.op in ("read", "index", "mlp", "mlp_out", "win") or n.win is not None:
                raise TraceUnsupported("grid value of another shape combined with an indexed field value")
            args = tuple(self._realign(a, win) for a in n.args)
            return self.node(n.op, args, attr=n.attr, shape=self._bshape(*[a.shape for a in args]), kind=n.kind,
                             host=n.host)
        t = self.tensors[n.attr]
        lens, sq = win
        vis = [d for d in range(len(lens)) if not sq[d]]
        if t.dim() > len(vis):
            raise TraceUnsupported("tensor of rank {} with an indexed field value".format(t.dim()))
        full = [1] * len(lens)
        for k, size in enumerate(t.shape):
            full[vis[len(vis) - t.dim() + k]] = int(size)
        return self.tensor(t.reshape(full))

    # ---- views: slices and picks become rolls pushed into the leaves + a window ---------------
    def grid_shape(self):
        for n in self.nodes:
            if n.op == "read":
                return n.shape
        raise TraceUnsupported("indexing before any field was read")

    def getitem(self, x, item):
        if x.host or (x.win is None and x.shape != self.grid_shape()):
            raise TraceUnsupported("indexing a value that is not a grid array")
        G = self.grid_shape()
        lens, sq = x.win if x.win is not None else (tuple(G), (False,) * len(G))
        vis = [d for d in range(len(G)) if not sq[d]]
        items = list(item) if isinstance(item, tuple) else [item]
        if any(i is None for i in items):
            raise TraceUnsupported("newaxis on a symbolic array")
        if Ellipsis in items:
            k = items.index(Ellipsis)
            items = items[:k] + [slice(None)] * (len(vis) - len(items) + 1) + items[k + 1:]
        items += [slice(None)] * (len(vis) - len(items))
        if len(items) != len(vis):
            raise TraceUnsupported("too many indices")
        shifts, lens, sq = [0] * len(G), list(lens), list(sq)
        for d, it in zip(vis, items):
            n = lens[d]
            if isinstance(it, slice):
                if it.step not in (None, 1):
                    raise TraceUnsupported("strided slice")
                a, b, _ = it.indices(n)
                if b <= a:
                    raise TraceUnsupported("empty slice")
                shifts[d], lens[d] = -a, b - a
            elif isinstance(it, (int, np.integer)):
                k = int(it) + (n if it < 0 else 0)
                if not 0 <= k < n:
                    raise IndexError("index {} out of range for axis of size {}".format(int(it), n))
                shifts[d], lens[d], sq[d] = -k, 1, True
            else:
                raise TraceUnsupported("index of type {}".format(type(it).__name__))
        return self.view(self.roll(x, tuple(shifts)), tuple(lens), tuple(sq))

    def view(self, x, lens, sq):
        G = self.grid_shape()
        win = None if tuple(lens) == tuple(G) and not any(sq) else (tuple(lens), tuple(sq))
        shape = tuple(l for l, q in zip(lens, sq) if not q)
        return self.node("win", (x,), shape=shape, kind=x.kind, win=win)

    def roll(self, n, shifts):
        """The grid function i -> n(i - shifts) (periodic, numpy.roll convention), built by pushing
        the shift into the leaves: reads change their stencil offset, index leaves wrap, tensors are
        rolled once on the device; everything else is pointwise."""
        if not any(shifts) or n.host:
            return n
        memo = self.__dict__.setdefault("_roll_memo", dict())
        key = (n.idx, shifts)
        if key in memo:
            return memo[key]
        G = self.grid_shape()
        if n.op == "read":
            k, s, loc, frozen = n.attr
            if loc != self.domain_loc(k):
                raise TraceUnsupported("roll of a field read at another location")
            res = self.node("read", attr=(k, tuple(a - b for a, b in zip(s, shifts)), loc, frozen), shape=n.shape)
        elif n.op == "index":
            d = n.attr[0]
            r, size = shifts[d] % G[d], G[d]
            if r == 0:
                res = n
            else:  # (i - r) mod size
                moved = self.binary("sub", n, r)
                res = self.where(self.binary("lt", moved, 0), self.binary("add", moved, size), moved)
        elif n.op == "tensor":
            t = self.tensors[n.attr]
            dims, amounts = [], []
            for d, r in enumerate(shifts):
                td = d - (len(G) - t.dim())
                if r and td >= 0 and t.shape[td] > 1:
                    dims.append(td)
                    amounts.append(int(r))
            res = self.tensor(torch.roll(t, amounts, dims)) if dims else n
        elif n.op == "aparam":
            res = n
        else:
            args = tuple(self.roll(a, shifts) for a in n.args)
            res = self.node(n.op, args, attr=n.attr, shape=n.shape, kind=n.kind, host=n.host, win=n.win)
        memo[key] = res
        return res

    def domain_loc(self, key):
        return self.state_locs[key]

    def concatenate(self, pieces, axis):
        """numpy.concatenate of grid values and concrete arrays along one axis (rows imposed exactly:
        `concatenate([u_init[None], u[1:]])`): every piece is moved to its offset and selected by index."""
        G = self.grid_shape()
        ndim = len(G)
        pieces = [self.lift(p) if isinstance(p, Sym) else p for p in pieces]
        syms = [p for p in pieces if isinstance(p, Sym)]
        if not syms or any(p.win is not None and any(p.win[1]) for p in syms):
            raise TraceUnsupported("concatenate of indexed values")
        axis = axis % ndim
        lens0 = [list(p.win[0]) if p.win is not None else list(G) for p in syms]
        other = lens0[0][:axis] + lens0[0][axis + 1:]
        sizes, offs, total = [], [], 0
        for p in pieces:
            shape = tuple(p.shape)
            if len(shape) != ndim or list(shape[:axis] + shape[axis + 1:]) != other:
                raise TraceUnsupported("concatenate of shapes that do not match")
            sizes.append(shape[axis])
            offs.append(total)
            total += shape[axis]
        lens = list(lens0[0])
        lens[axis] = total
        if total > G[axis]:
            raise TraceUnsupported("concatenate longer than the grid")
        win = None if lens == list(G) else (tuple(lens), (False,) * ndim)
        shape = tuple(lens)
        placed = []
        for p, o, size in zip(pieces, offs, sizes):
            if isinstance(p, Sym):
                sh = [0] * ndim
                sh[axis] = o
                placed.append(self.roll(p, tuple(sh)))
            else:  # concrete: embed at its offset in an array of the full length
                t = self.real_mod.array(p) if not isinstance(p, torch.Tensor) else p
                if t.requires_grad:
                    raise TraceUnsupported("differentiable tensor outside ctx.field / ctx.neural_net")
                full = list(t.shape)
                full[axis] = total
                buf = torch.zeros(full, dtype=t.dtype, device=self.real_mod.device)
                buf.narrow(axis, o, size).copy_(t)
                placed.append(self.tensor(buf))
        idx = self.node("index", attr=(axis, None), shape=tuple(G), kind=_I)
        res = placed[-1]
        for p, o, size in reversed(list(zip(placed[:-1], offs[:-1], sizes[:-1]))):
            cond = self.node("lt", (idx, self.const(o + size)), shape=tuple(G), kind=_B)
            kind = _promote(p.kind, res.kind)
            res = self.node("where", (cond, p, res), shape=shape, kind=kind, win=win)
        if res.win != win:
            res = self.node("win", (res,), shape=shape, kind=res.kind, win=win)
        return res

    def unary(self, op, a):
        a = self.lift(a)
        if a.op == "const" and op in _HOST_UNARY:
            return self.const(_HOST_UNARY[op](a.attr))
        kind = _B if op == "not" else (a.kind if op in ("neg", "abs", "stopgrad", "relu") and a.kind != _B else _R)
        if op == "floor" and a.kind != _R:
            return a
        return self.node(op, (a,), shape=a.shape, kind=kind, host=a.host and op in _HOST_UNARY, win=a.win)

    def binary(self, op, a, b):
        a, b = self.lift(a), self.lift(b)
        if a.op == "const" and b.op == "const":
            return self.const(_HOST_BINARY[op](a.attr, b.attr))
        if op in _CMP or op in ("and", "or"):
            kind = _B
        elif op in ("div", "pow", "atan2"):
            kind = _R
        else:
            kind = _promote(a.kind, b.kind)
            if kind == _B:
                kind = _I
        (a, b), shape, win = self._combine([a, b])
        return self.node(op, (a, b), shape=shape, kind=kind, host=a.host and b.host, win=win)

    def where(self, c, a, b):
        c, a, b = self.lift(c), self.lift(a), self.lift(b)
        if c.op == "const":
            return a if c.attr else b
        kind = _promote(a.kind, b.kind)
        (c, a, b), shape, win = self._combine([c, a, b])
        return self.node("where", (c, a, b), shape=shape, kind=kind, host=c.host and a.host and b.host, win=win)


def _has_sym(x):
    if isinstance(x, Sym):
        return True
    if isinstance(x, (list, tuple)):
        return any(_has_sym(v) for v in x)
    if isinstance(x, dict):
        return any(_has_sym(v) for v in x.values())
    return False


class ModTrace:
    """The `mod` namespace seen by an operator being traced: elementwise functions build DAG nodes,
    anything applied to concrete values runs eagerly on the real backend (constants of the trace)."""

    def __init__(self, tr):
        self._tr = tr
        self._real = tr.real_mod
        self.mod = self

    def __getattr__(self, name):
        attr = getattr(self._real, name)
        if not callable(attr) or isinstance(attr, type):
            return attr

        def eager(*args, **kwargs):
            if _has_sym(args) or _has_sym(kwargs):
                raise TraceUnsupported("mod.{} of a symbolic array".format(name))
            return attr(*args, **kwargs)

        return eager

    def _u(self, op, x):
        return self._tr.unary(op, x) if isinstance(x, Sym) else getattr(self._real, op)(x)

    def abs(self, x):
        return self._u("abs", x)

    def cos(self, x):
        return self._u("cos", x)

    def sin(self, x):
        return self._u("sin", x)

    def exp(self, x):
        return self._u("exp", x)

    def log(self, x):
        return self._u("log", x)

    def tanh(self, x):
        return self._u("tanh", x)

    def sqrt(self, x):
        return self._u("sqrt", x)

    def floor(self, x):
        return self._u("floor", x)

    def relu(self, x):
        return self._u("relu", x)

    def square(self, x):
        return x * x if isinstance(x, Sym) else self._real.square(x)

    def sigmoid(self, x):
        return 1 / (1 + self._tr.unary("exp", -x)) if isinstance(x, Sym) else self._real.sigmoid(x)

    def stop_gradient(self, x):
        return self._tr.unary("stopgrad", x) if isinstance(x, Sym) else self._real.stop_gradient(x)

    def cast(self, x, dtype):
        if not isinstance(x, Sym):
            return self._real.cast(x, dtype)
        td = torch_dtype(dtype)
        if td is None or (td.is_floating_point and x.kind == _R):
            return x
        if td.is_floating_point:
            return self._tr.unary("cast", x)
        raise TraceUnsupported("cast of a symbolic array to {}".format(td))

    def array(self, x, dtype=None):
        return self.cast(x, dtype) if isinstance(x, Sym) else self._real.array(x, dtype)

    constant = array
    native = array

    def copy(self, x):
        return x if isinstance(x, Sym) else self._real.copy(x)

    def is_tensor(self, x):
        return isinstance(x, Sym) or self._real.is_tensor(x)

    def zeros_like(self, x):
        return x * 0 if isinstance(x, Sym) else self._real.zeros_like(x)

    def ones_like(self, x):
        return x * 0 + 1 if isinstance(x, Sym) else self._real.ones_like(x)

    def where(self, c, a, b):
        if _has_sym((c, a, b)):
            return self._tr.where(c, a, b)
        return self._real.where(c, a, b)

    def _b(self, op, a, b):
        if _has_sym((a, b)):
            return self._tr.binary(op, a, b)
        return getattr(self._real, {"min": "minimum", "max": "maximum", "atan2": "arctan2"}[op])(a, b)

    def minimum(self, a, b):
        return self._b("min", a, b)

    def maximum(self, a, b):
        return self._b("max", a, b)

    def arctan2(self, a, b):
        return self._b("atan2", a, b)

    def roll(self, x, shift, axis=None):
        if not isinstance(x, Sym):
            return self._real.roll(x, shift, axis)
        if x.win is not None or axis is None:
            raise TraceUnsupported("roll of a sliced or flattened symbolic array")
        ndim = len(x.shape)
        axes = [int(axis)] if isinstance(axis, (int, np.integer)) else [int(a) for a in axis]
        amounts = [int(v) for v in np.broadcast_to(np.asarray(shift), (len(axes),))]
        shifts = [0] * ndim
        for a, r in zip(axes, amounts):
            shifts[a % ndim] += r
        return self._tr.roll(x, tuple(shifts))

    def concatenate(self, xs, axis=0):
        if not _has_sym(xs):
            return self._real.concatenate(xs, axis)
        return self._tr.concatenate(list(xs), int(axis))

    def clip(self, x, a, b):
        if _has_sym((x, a, b)):
            return self._tr.binary("min", self._tr.binary("max", x, a), b)
        return self._real.clip(x, a, b)


class ParamArray:
    """An `Array` unknown (a few scalars, e.g. the constants of infer_constant) seen by a traced
    operator: indexing gives device scalars whose gradients are reduced over the grid."""

    def __init__(self, tr, key, shape, frozen):
        self.tr, self.key, self.shape, self.frozen = tr, key, tuple(shape), frozen

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, item):
        if isinstance(item, tuple) and len(item) == 1:
            item = item[0]
        if len(self.shape) != 1 or not isinstance(item, (int, np.integer)):
            raise TraceUnsupported("Array unknowns support a[k] only")
        k = int(item) + (self.shape[0] if item < 0 else 0)
        if not 0 <= k < self.shape[0]:
            raise IndexError("index {} out of range".format(int(item)))
        return self.tr.node("aparam", attr=(self.key, k, self.frozen), kind=_R)

    def __iter__(self):
        return (self[k] for k in range(self.shape[0]))

    def _no(self, *a, **k):
        raise TraceUnsupported("arithmetic on a whole Array unknown (index it: a[k])")

    __add__ = __radd__ = __sub__ = __rsub__ = __mul__ = __rmul__ = __truediv__ = __rtruediv__ = __neg__ = _no
    __array_ufunc__ = None


class TraceContext:
    """`Context` (reference core.py:865-990) whose reads return symbols."""

    class Raw:
        def __init__(self, value):
            self.value = value

    def __init__(self, tr, state, extra, tracers):
        from .core import Context

        self.Raw = Context.Raw
        self._tr = tr
        self.domain = tr.domain
        self.state = state
        self.extra = extra
        self.dtype = tr.domain.dtype
        self.mod = ModTrace(tr)
        self.distinct_shift = False
        self.step = tr.domain.step
        self.size = tr.domain.size
        self.tracer_names = []
        self._tracers = dict()
        for k, v in (tracers or dict()).items():
            if isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool):
                self._tracers[k] = tr.node("tracer", attr=k, kind=_R, host=True)
            else:
                self._tracers[k] = v
        self.tracers_accessed = False
        self.nets = dict()
        from .core import Field as _Field, MultigridField as _MgField

        tr.state_locs = {k: f.loc for k, f in state.fields.items() if isinstance(f, (_Field, _MgField))}

    @property
    def tracers(self):
        self.tracers_accessed = True
        return self._tracers

    def cast(self, value, dtype=None):
        return self.mod.cast(value, dtype or self.dtype)

    def indices(self, *dims, loc=None):
        domain = self.domain
        loc = loc or "c" * domain.ndim
        if any(c not in "cn" for c in loc) or len(loc) != domain.ndim:
            return domain.indices(*dims, loc=loc)
        shape = domain.get_field_shape(loc)
        idims = domain._names_to_indices(dims, list(domain.dimnames))
        res = tuple(self._tr.node("index", attr=(d, loc), shape=shape, kind=_I) for d in idims)
        return res[0] if len(dims) == 1 else res

    def points(self, *dims, loc=None):
        domain = self.domain
        loc = loc or "c" * domain.ndim
        if any(c not in "cn" for c in loc) or len(loc) != domain.ndim:
            return domain.points(*dims, loc=loc)
        cache = domain.__dict__.setdefault("_points_bcast", dict())
        idims = domain._names_to_indices(dims, list(domain.dimnames))
        res = []
        for d in idims:
            if (d, loc[d]) not in cache:
                shape = [1] * domain.ndim
                shape[d] = -1
                cache[(d, loc[d])] = domain.mod.array(domain._points_1d(d, loc[d])).reshape(shape)
            res.append(self._tr.tensor(cache[(d, loc[d])]))
        return res[0] if len(dims) == 1 else tuple(res)

    def field(self, key, *shift, loc=None, frozen=False):
        from .core import Array, Field, MultigridField

        domain = self.domain
        field = self.state.fields[key]
        if isinstance(field, Array):
            if len(shift):
                raise RuntimeError("Array requires an empty shift")
            return ParamArray(self._tr, key, tuple(field.array.shape), bool(frozen))
        if not isinstance(field, (Field, MultigridField)):
            raise TypeError(
                "Expected Field or MultigridField, got type {} for key='{}'".format(type(field).__name__, key))
        shift = tuple(int(s) for s in shift) or (0,) * domain.ndim
        loc = loc or field.loc
        if len(shift) != domain.ndim:
            raise RuntimeError("Expected {} shift components, got shift={}".format(domain.ndim, shift))
        if len(loc) != domain.ndim or any(c not in "cn" for c in loc + field.loc):
            raise TraceUnsupported("loc '{}'".format(loc))
        return self._tr.node("read", attr=(key, shift, loc, bool(frozen)), shape=domain.get_field_shape(loc), kind=_R)

    def neural_net(self, key, frozen=False):
        from .core import NeuralNet

        net = self.state.fields[key]
        if not isinstance(net, NeuralNet):
            raise TypeError("Expected NeuralNet, got type {} for key='{}'".format(type(net).__name__, key))
        if net.activation not in ("tanh", "relu", "none"):
            raise TraceUnsupported("activation " + str(net.activation))
        tr = self._tr
        layers = [int(net.weights[0].shape[1])] + [int(w.shape[0]) for w in net.weights]
        self.nets[key] = layers

        def res(*inputs):
            if net.func_in is not None:
                inputs = net.func_in(*inputs)
            inputs = [tr.lift(v) for v in inputs]
            if len(inputs) != layers[0]:
                raise RuntimeError("Weights and inputs do not match")
            shape = tr._bshape(*[v.shape for v in inputs])
            call = tr.node("mlp", tuple(inputs), attr=(key, bool(frozen), tuple(layers), net.activation), shape=shape)
            outputs = [tr.node("mlp_out", (call,), attr=j, shape=shape) for j in range(layers[-1])]
            if net.func_out is not None:
                outputs = net.func_out(*outputs)
            return outputs

        return res


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
__device__ inline int wrap(int j, int n) { return j < 0 ? j + n : (j >= n ? j - n : j); }
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
    def __init__(self, tr, outputs, raw, shape, state):
        self.tr, self.outputs, self.raw, self.G, self.state = tr, outputs, raw, tuple(shape), state
        self.ndim = len(shape)
        self.total = int(np.prod(shape))
        if self.total >= 2**31 - 1024:
            raise TraceUnsupported("grid too large for 32-bit indexing")
        self.lines = []
        self.max_blocks = int(os.environ.get("ODIL_JIT_NBLOCKS", 65536))
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
        self.nets = []  # (key, layers) with parameter pointers
        self.net_slot = dict()
        self.arrays = []  # (key, numel) of `Array` unknowns read through a[k]
        self.array_slot = dict()
        self.need = self._needs_grad()
        # per output: None (the whole grid) or the lens of its window; the mean runs over that many points
        self.out_lens = [None if o.win is None else tuple(o.win[0]) for o in outputs]
        self.out_count = [int(np.prod(l)) if l is not None else self.total for l in self.out_lens]

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

    # ---- expressions ----------------------------------------------------------------------
    def ex(self, n):
        if n.op == "const":
            return _lit(n.attr, n.kind)
        if n.host:
            e = "a.hs[{}]".format(self.hs_slot[n.idx])
            return {"r": "((T){})", "i": "((long){})", "b": "({} != 0.0)"}[n.kind].format(e)
        return "v{}".format(n.idx)

    def r(self, n):
        return self.ex(n) if n.kind == _R else "((T){})".format(self.ex(n))

    def i(self, n):
        return self.ex(n) if n.kind == _I else "((long){})".format(self.ex(n))

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
    def _emit_read(self, n):
        key, shift, loc, _ = n.attr
        desc = (key, shift, loc)
        if desc in self.loads:
            self.emit("const T v{} = {};".format(n.idx, self.loads[desc]))
            return
        floc = self.state.fields[key].loc
        fshape = self._field_shape(key)
        slot = self._src_slot(key)
        idx, zero = [], []
        for d in range(self.ndim):
            ns = fshape[d]
            ext = max(ns, self.G[d])  # extent of the padded / untrimmed array the roll acts on
            s = shift[d] % ext
            if s > ext // 2:
                s -= ext
            j = "i{}".format(d) if s == 0 else "wrap(i{} + ({}), {})".format(d, s, ext)
            if floc[d] == "c" and loc[d] == "n":  # zero padded at the low end
                name = "p{}_{}".format(n.idx, d)
                self.emit("const int {} = {};".format(name, j))
                zero.append("{} == 0".format(name))
                j = "({} == 0 ? 0 : {} - 1)".format(name, name)
            idx.append(j)
        off = self._offset(idx, fshape)
        e = "a.src[{}][{}]".format(slot, off)
        if zero:
            e = "(({}) ? (T)0 : {})".format(" || ".join(zero), e)
        self.emit("const T v{} = {};".format(n.idx, e))
        self.loads[desc] = "v{}".format(n.idx)

    def _emit_tensor(self, n):
        t = self.tr.tensors[n.attr]
        shape = (1,) * (self.ndim - t.dim()) + tuple(t.shape)
        if len(shape) != self.ndim or any(s > g for s, g in zip(shape, self.G)):
            raise TraceUnsupported("tensor of shape {} on grid {}".format(tuple(t.shape), self.G))
        terms, stride = [], 1
        for d in reversed(range(self.ndim)):
            if shape[d] != 1:
                # shorter than the grid: an operand of a windowed value; clamped outside its window
                i = "i{}".format(d) if shape[d] == self.G[d] else "min(i{}, {})".format(d, shape[d] - 1)
                terms.append("{} * {}".format(i, stride) if stride != 1 else i)
                stride *= shape[d]
        ctype = {torch.float32: "float", torch.float64: "double", torch.int32: "int", torch.int64: "long",
                 torch.bool: "unsigned char"}[t.dtype]
        cast = {"r": "(T)", "i": "(long)", "b": "0 != "}[n.kind]
        ktype = {"r": "T", "i": "long", "b": "bool"}[n.kind]
        self.emit("const {} v{} = {}((const {}*)a.ten[{}])[{}];".format(
            ktype, n.idx, cast, ctype, n.attr, " + ".join(terms) or "0"))

    def _act(self, kind, x):
        return {"tanh": "FN(tanh)({})", "relu": "({0} > (T)0 ? {0} : (T)0)", "none": "{}"}[kind].format(x)

    def _emit_mlp(self, n):
        key, frozen, layers, act = n.attr
        if key not in self.net_slot:
            self.net_slot[key] = len(self.nets)
            self.nets.append((key, layers))
        base = self.net_slot[key]
        nl = len(layers) - 1
        p = "m{}".format(n.idx)
        for i, a in enumerate(n.args):
            self.emit("const T {}_h0_{} = {};".format(p, i, self.r(a)))
        for l in range(1, nl + 1):
            ni, no = layers[l - 1], layers[l]
            for j in range(no):
                terms = " + ".join("W({},{},{}) * {}_h{}_{}".format(base, l - 1, j * ni + i, p, l - 1, i) for i in range(ni))
                self.emit("const T {}_z{}_{} = ({}) + Bv({},{},{});".format(p, l, j, terms, base, l - 1, j))
                if l < nl:
                    self.emit("const T {0}_h{1}_{2} = {3};".format(p, l, j, self._act(act, "{}_z{}_{}".format(p, l, j))))

    def forward(self):
        for n in self.order:
            if n.host:
                continue
            op, A = n.op, n.args
            kt = {"r": "T", "i": "long", "b": "bool"}[n.kind]
            v = "const {} v{} = ".format(kt, n.idx)
            if op == "read":
                self._emit_read(n)
            elif op == "tensor":
                self._emit_tensor(n)
            elif op == "index":
                self.emit(v + "(long)i{};".format(n.attr[0]))
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
            elif op in ("add", "sub", "mul"):
                sym = {"add": "+", "sub": "-", "mul": "*"}[op]
                self.emit(v + "{} {} {};".format(self.typed(A[0], n.kind), sym, self.typed(A[1], n.kind)))
            elif op == "div":
                self.emit(v + "{} / {};".format(self.r(A[0]), self.r(A[1])))
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
                acc(A[0], "{} / {}".format(g, self.r(A[1])))
                acc(A[1], "-({} * {}) / {}".format(g, v, self.r(A[1])))
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

    def _reverse_mlp(self, n, defined, acc):
        if not self.need[n.idx]:
            return
        key, frozen, layers, act = n.attr
        outs = [m for m in self.order if m.op == "mlp_out" and m.args[0] is n and m.idx in defined]
        if not outs:
            return
        base = self.net_slot[key]
        nl = len(layers) - 1
        p = "m{}".format(n.idx)
        by_j = {m.attr: m for m in outs}
        for j in range(layers[nl]):
            self.emit("const T {}_d{}_{} = {};".format(p, nl, j, "g{}".format(by_j[j].idx) if j in by_j else "(T)0"))
        if not frozen and key not in self.pgrads:
            names = []
            for l in range(nl):
                names.append(["pw_{}_{}_{}".format(base, l, k) for k in range(layers[l] * layers[l + 1])])
            for l in range(nl):
                names.append(["pb_{}_{}_{}".format(base, l, k) for k in range(layers[l + 1])])
            self.pgrads[key] = names
            self.pg_offset[key] = len(self.pg_decl)
            self.pg_decl.extend(name for group in names for name in group)
        inputs_need = any(self.need[a.idx] for a in n.args)
        for l in range(nl, 0, -1):
            ni, no = layers[l - 1], layers[l]
            if not frozen:
                for j in range(no):
                    for i in range(ni):
                        self.emit("pw_{0}_{1}_{2} = pw_{0}_{1}_{2} + {3}_d{4}_{5} * {3}_h{6}_{7};".format(
                            base, l - 1, j * ni + i, p, l, j, l - 1, i))
                    self.emit("pb_{0}_{1}_{2} = pb_{0}_{1}_{2} + {3}_d{4}_{2};".format(base, l - 1, j, p, l))
            if l == 1 and not inputs_need:
                break
            for i in range(ni):
                s = " + ".join("W({},{},{}) * {}_d{}_{}".format(base, l - 1, j * ni + i, p, l, j) for j in range(no))
                if l > 1:
                    h = "{}_h{}_{}".format(p, l - 1, i)
                    d = {"tanh": "((T)1 - {0} * {0})".format(h), "relu": "({} > (T)0 ? (T)1 : (T)0)".format(h),
                         "none": "(T)1"}[act]
                    self.emit("const T {}_d{}_{} = ({}) * {};".format(p, l - 1, i, s, d))
                else:
                    self.emit("const T {}_d0_{} = {};".format(p, i, s))
        if inputs_need:
            for i, a in enumerate(n.args):
                acc(a, "{}_d0_{}".format(p, i))

    # ---- whole source -----------------------------------------------------------------------
    def source(self):
        tdt = self.tr.torch_dtype
        self.pg_decl, self.pg_offset = [], dict()
        self.forward()
        fwd, self.lines = self.lines, []
        self.reverse()
        rev, self.lines = self.lines, []
        nout = len(self.outputs)
        self.npar = sum(len(g) for names in self.pgrads.values() for g in names)
        par_arrays = sum(2 * (len(layers) - 1) for _, layers in self.nets) + len(self.arrays)
        self.par_arrays = par_arrays
        T = "double" if tdt == torch.float64 else "float"
        fn = "name" if T == "double" else "name##f"
        S = [_PRELUDE.replace("@T@", T).replace("@FN@", fn)]
        S.append("struct Args {{ const T* src[{}]; const void* ten[{}]; T* cot[{}]; const T* par[{}]; const double* hs; "
                 "T* part; T* ppart; T* part2; T* out; T* pgrad; int nblocks; }};".format(
                     max(1, len(self.src_keys)), max(1, len(self.tr.tensors)), max(1, len(self.cots)),
                     max(1, par_arrays)))
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
        S.append("#define W(s, l, k) a.par[WOFS_##s##_##l][k]")
        S.append("#define Bv(s, l, k) a.par[BOFS_##s##_##l][k]")
        for (s, l), v in wofs.items():
            S.append("#define WOFS_{}_{} {}".format(s, l, v))
        for (s, l), v in bofs.items():
            S.append("#define BOFS_{}_{} {}".format(s, l, v))
        S.append('extern "C" __global__ __launch_bounds__(NB) void k_fwd(const Args a) {')
        S.append("  __shared__ T sm[NB / 64];")
        for k in range(nout):
            S.append("  T s_{} = (T)0;".format(k))
        for name in self.pg_decl:
            S.append("  T {} = (T)0;".format(name))
        if self.total <= self.max_blocks * 256:  # one grid point per thread
            S.append("  const int l = blockIdx.x * NB + threadIdx.x;")
            S.append("  if (l < {}) {{".format(self.total))
        else:
            S.append("  for (int l = blockIdx.x * NB + threadIdx.x; l < {}; l += a.nblocks * NB) {{".format(self.total))
        rem = "l"
        for d in reversed(range(self.ndim)):
            if d == 0:
                S.append("  const int i0 = {};".format(rem))
            else:
                S.append("  const int i{} = {} % {};".format(d, rem, self.G[d]))
                S.append("  const int r{} = {} / {};".format(d, rem, self.G[d]))
                rem = "r{}".format(d)
        for k, lens in enumerate(self.out_lens):
            if lens is not None:
                conds = ["i{} < {}".format(d, lens[d]) for d in range(self.ndim) if lens[d] < self.G[d]]
                S.append("  const bool inbox{} = {};".format(k, " && ".join(conds) or "true"))
        S.extend(fwd)
        S.extend(rev)
        esize = 8 if tdt == torch.float64 else 4
        stream = len(self.cots) * self.total * esize > (128 << 20)  # beyond what the last-level cache keeps
        for slot, n in enumerate(self.cots):
            if stream:
                S.append("  __builtin_nontemporal_store(g{}, &a.cot[{}][l]);".format(n.idx, slot))
            else:
                S.append("  a.cot[{}][l] = g{};".format(slot, n.idx))
        for k, (o_, raw) in enumerate(zip(self.outputs, self.raw)):
            term = self.r(o_) if raw else "{0} * {0}".format(self.r(o_))
            if self.out_lens[k] is not None:
                term = "(inbox{} ? {} : (T)0)".format(k, term)
            S.append("  s_{0} = s_{0} + {1};".format(k, term))
        S.append("  }")
        for k in range(nout):
            S.append("  {{ const T s = block_sum(s_{0}, sm); if (threadIdx.x == 0) a.part[{0} * a.nblocks + blockIdx.x] = s; }}".format(k))
        for k, name in enumerate(self.pg_decl):
            S.append("  {{ const T s = block_sum({}, sm); if (threadIdx.x == 0) a.ppart[{} * a.nblocks + blockIdx.x] = s; }}".format(name, k))
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
        # gathers
        self.gathers = []  # (key, [cot slots]) for fields that need a gather launch
        self.direct = dict()  # key -> cot slot that already IS the gradient
        by_key = dict()
        for slot, n in enumerate(self.cots):
            by_key.setdefault(n.attr[0], []).append((slot, n))
        for key, reads in by_key.items():
            floc = self.state.fields[key].loc
            fshape = self._field_shape(key)
            if len(reads) == 1 and not any(reads[0][1].attr[1]) and reads[0][1].attr[2] == floc:
                self.direct[key] = reads[0][0]
                continue
            gi = len(self.gathers)
            self.gathers.append(key)
            tot = int(np.prod(fshape))
            S.append('extern "C" __global__ __launch_bounds__(NB) void k_gat_{}(const Args a, T* __restrict__ g) {{'.format(gi))
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
            for slot, n in reads:
                _, shift, loc, _ = n.attr
                idx, valid = [], []
                for d in range(self.ndim):
                    ns, nr = fshape[d], self.G[d]
                    ext = max(ns, nr)
                    s = shift[d] % ext
                    if s > ext // 2:
                        s -= ext
                    pos = "j{}".format(d) if not (floc[d] == "c" and loc[d] == "n") else "(j{} + 1)".format(d)
                    e = pos if s == 0 else "wrap({} - ({}), {})".format(pos, s, ext)
                    if floc[d] == "n" and loc[d] == "c":  # trimmed: the last padded position was dropped
                        name = "t{}_{}".format(slot, d)
                        S.append("  const int {} = {};".format(name, e))
                        valid.append("{} < {}".format(name, nr))
                        e = name
                    idx.append(e)
                load = "a.cot[{}][{}]".format(slot, self._offset(idx, self.G))
                if valid:
                    load = "(({}) ? {} : (T)0)".format(" && ".join(valid), load)
                S.append("  acc = acc + {};".format(load))
            S.append("  g[l] = acc;")
            S.append("}")
        # launchers
        S.append('extern "C" int jit_fwd(const Args* a, void* stream) {')
        S.append("  hipLaunchKernelGGL(k_fwd, dim3(a->nblocks), dim3(NB), 0, (hipStream_t)stream, *a);")
        S.append("  hipLaunchKernelGGL(k_final, dim3({}, SEG), dim3(NB), 0, (hipStream_t)stream, *a);".format(nout + len(self.pg_decl)))
        S.append("  hipLaunchKernelGGL(k_loss, dim3(1), dim3(64), 0, (hipStream_t)stream, *a);")
        S.append("  return (int)hipGetLastError();")
        S.append("}")
        S.append('extern "C" int jit_gather(int which, const Args* a, void* g, void* stream) {')
        S.append("  switch (which) {")
        for gi, key in enumerate(self.gathers):
            tot = int(np.prod(self._field_shape(key)))
            S.append("    case {}: hipLaunchKernelGGL(k_gat_{}, dim3({}), dim3(NB), 0, (hipStream_t)stream, *a, (T*)g); break;".format(
                gi, gi, (tot + 255) // 256))
        S.append("    default: return -1;")
        S.append("  }")
        S.append("  return (int)hipGetLastError();")
        S.append("}")
        return "\n".join(S) + "\n"


def _cache_dirs():
    """In-tree cache first (travels with the checkout); a per-user temp dir if that is read-only."""
    import getpass

    yield _CACHE_DIR
    try:
        user = getpass.getuser()
    except Exception:
        user = str(os.getuid())
    yield os.path.join(tempfile.gettempdir(), "odil_jit_cache_" + user)


def _compile(src):
    tag = hashlib.sha256((src + " ".join(_HIPCC_FLAGS)).encode()).hexdigest()[:20]
    name = "odil_jit_{}.so".format(tag)
    for d in _cache_dirs():
        if os.path.exists(os.path.join(d, name)):
            return ctypes.CDLL(os.path.join(d, name)), os.path.join(d, name)
    last = None
    for d in _cache_dirs():
        try:
            os.makedirs(d, exist_ok=True)
            hip = os.path.join(d, "odil_jit_{}.hip".format(tag))
            with open(hip, "w") as f:
                f.write(src)
            fd, tmp = tempfile.mkstemp(suffix=".so", dir=d)
            os.close(fd)
        except OSError as e:
            last = e
            continue
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        res = subprocess.run([hipcc] + _HIPCC_FLAGS + ["-o", tmp, hip], capture_output=True, text=True)
        if res.returncode != 0:
            os.unlink(tmp)
            raise RuntimeError("hipcc failed for the traced operator ({}):\n{}".format(hip, res.stderr[-4000:]))
        path = os.path.join(d, name)
        os.replace(tmp, path)  # atomic: concurrent ranks compiling the same source do not collide
        return ctypes.CDLL(path), path
    raise FileNotFoundError("no writable cache directory for traced operators: {}".format(last))


# ======================================================================================
# Traced evaluator
# ======================================================================================
class TracedOperator:
    """loss / gradient of one user operator through its generated kernels."""

    def __init__(self, problem, state):
        from .core import Context, Field, MultigridField, NeuralNet, Problem

        domain = problem.domain
        self.problem, self.domain = problem, domain
        tr = Tracer(domain)
        # unknowns reached around ctx.field / ctx.neural_net would lose their gradient: trace on
        # differentiable leaves so that `lift` can refuse them
        leaves = [a.detach().requires_grad_(True) for a in domain.arrays_from_state(state)]
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
        self.names = names
        raw = [isinstance(v, Context.Raw) for v in values]
        outs = [tr.lift(v.value if r else v) for v, r in zip(values, raw)]
        if not any(n.op == "read" for n in tr.nodes):
            raise TraceUnsupported("operator reads no field")
        G = tuple(tr.grid_shape())
        for o in outs:  # every output lives on the grid of the reads, all of it or a window of it
            if o.host or (o.win is None and tuple(o.shape) != G):
                raise TraceUnsupported("output of shape {} on grid {}".format(tuple(o.shape), G))
        outs = [o if o.kind == _R else tr.unary("cast", o) for o in outs]
        self.G, self.raw = G, raw
        cg = _Codegen(tr, outs, raw, G, state)
        self.source = cg.source()
        self.lib, self.lib_path = _compile(self.source)
        self.cg, self.tr = cg, tr
        self.tracer_keys = [n.attr for n in tr.nodes if n.op == "tracer"]
        dev, dt = domain.mod.device, tr.torch_dtype
        self.total = cg.total
        self.nblocks = min((self.total + 255) // 256, cg.max_blocks)
        nout = len(outs)
        self.cot = [torch.empty(G, dtype=dt, device=dev) for _ in cg.cots]
        self.part = torch.empty(max(1, nout * self.nblocks), dtype=dt, device=dev)
        self.ppart = torch.empty(max(1, len(cg.pg_decl) * self.nblocks), dtype=dt, device=dev)
        self.out = torch.zeros(1 + 2 * nout, dtype=dt, device=dev)
        self.pgrad = torch.zeros(max(1, len(cg.pg_decl)), dtype=dt, device=dev)
        par_arrays = cg.par_arrays

        class Args(ctypes.Structure):
            _fields_ = [
                ("src", ctypes.c_void_p * max(1, len(cg.src_keys))),
                ("ten", ctypes.c_void_p * max(1, len(tr.tensors))),
                ("cot", ctypes.c_void_p * max(1, len(cg.cots))),
                ("par", ctypes.c_void_p * max(1, par_arrays)),
                ("hs", ctypes.c_void_p),
                ("part", ctypes.c_void_p), ("ppart", ctypes.c_void_p), ("part2", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("pgrad", ctypes.c_void_p), ("nblocks", ctypes.c_int),
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
        self.args.nblocks = self.nblocks
        nhs = max(1, len(cg.hs))
        self.hs_host = torch.zeros(nhs, dtype=torch.float64).pin_memory() if torch.cuda.is_available() else torch.zeros(nhs, dtype=torch.float64)
        self.hs_dev = torch.zeros(nhs, dtype=torch.float64, device=dev)
        self._hs_last = None
        self.args.hs = self.hs_dev.data_ptr()
        self.lib.jit_fwd.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.lib.jit_gather.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
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

    def refresh_host_scalars(self):
        """Host scalars of the trace (functions of `problem.tracers`, evaluated in Python double as the
        operator itself would) -> pinned buffer -> device.  The copy is issued on the current stream
        when a value changed; captured into a hipGraph it re-reads the pinned buffer at every replay,
        so a replayed epoch only needs this method's host part to be called first."""
        if not self.cg.hs:
            return
        memo = dict()
        vals = [float(self._host_value(n, memo)) for n in self.cg.hs]
        if vals != self._hs_last or torch.cuda.is_current_stream_capturing():
            self.hs_host.copy_(torch.tensor(vals, dtype=torch.float64))
            self._hs_last = vals
            self.hs_dev.copy_(self.hs_host, non_blocking=True)

    def _side_streams(self, nfields):
        """Streams for per-field chains; none for a single field, for fields beyond 64 MB (their kernels
        fill the GPU on their own and concurrent streams only fight for HBM and allocator pools:
        veltracer3d 83 -> 134 ms) or when ODIL_TRACE_STREAMS=0.  Measured gain where it applies: 5 %."""
        esize = 8 if self.tr.torch_dtype == torch.float64 else 4
        if (nfields < 2 or self.total * esize > (64 << 20) or not torch.cuda.is_available()
                or not int(os.environ.get("ODIL_TRACE_STREAMS", 1))):
            return []
        pool = self.__dict__.setdefault("_streams", [])
        while len(pool) < min(nfields, 4):
            pool.append(torch.cuda.Stream())
        return pool[: min(nfields, 4)]

    # ---- evaluation ---------------------------------------------------------------------------
    def _launch(self, state):
        from .core import MultigridField

        domain, cg = self.domain, self.cg
        if self._signature(state) != self.signature:
            raise RuntimeError("state structure changed since the operator was traced")
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
        stream = ops.stream_ptr()
        rc = self.lib.jit_fwd(ctypes.byref(self.args), stream)
        if rc != 0:
            raise RuntimeError("traced operator launch failed: hip error {}".format(rc))
        return keep

    def eval_loss_grad(self, state):
        """loss, grads (views of one packed buffer, overwritten by the next call), terms, names, norms."""
        cg = self.cg
        keep = self._launch(state)
        cur = torch.cuda.current_stream()
        chains = [item for item in self.layout if item[1] in ("field", "mg") and (item[0] in cg.gathers or item[0] in cg.direct)]
        side = self._side_streams(len(chains))
        for i, (key, kind, pos, n) in enumerate(chains):
            s_ = side[i % len(side)] if side else cur
            if side:
                s_.wait_stream(cur)
            with torch.cuda.stream(s_):
                if key in cg.gathers:
                    g = self.gtmp.get(key, self.gviews[pos])
                    rc = self.lib.jit_gather(cg.gathers.index(key), ctypes.byref(self.args), g.data_ptr(), ops.stream_ptr())
                    if rc != 0:
                        raise RuntimeError("traced gather launch failed: hip error {}".format(rc))
                else:
                    g = self.cot[cg.direct[key]]
                if kind == "mg":
                    factors, loc, shapes = self.mg_meta[key]
                    ops.mg_synth_adj(g, shapes, loc, factors=factors, grads=self.gviews[pos:pos + n])
        for s_ in side:
            cur.wait_stream(s_)
        for key, kind, pos, n in self.layout:
            if kind == "net" and key in cg.pgrads and not self.pgrad_direct:
                pofs = cg.pg_offset[key]
                for j, group in enumerate(cg.pgrads[key]):
                    self.gviews[pos + j].copy_(self.pgrad[pofs:pofs + len(group)].view(self.gviews[pos + j].shape))
                    pofs += len(group)
        out = self.out.clone()
        nout = len(self.raw)
        loss = out[0]
        terms = [out[1 + k] for k in range(nout)]
        norms = [out[1 + nout + k] for k in range(nout)]
        del keep
        return loss, list(self.gviews), terms, self.names, norms


def trace(problem, state):
    """A TracedOperator for `problem`, or None (with the reason logged) when the operator cannot be
    expressed as one pointwise stencil kernel."""
    from .util import printlog

    try:
        return TracedOperator(problem, state)
    except TraceUnsupported as e:
        printlog("odil_amd: operator not traced ({}); using the generic autograd path".format(e))
    except FileNotFoundError as e:
        printlog("odil_amd: no hipcc for traced operators ({}); using the generic autograd path".format(e))
    return None
