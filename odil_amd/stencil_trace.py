"""Symbolic execution of a user `operator(ctx)`: the expression DAG behind odil_amd/stencil_jit.py.

`Sym` nodes stand in for device arrays, `Tracer` builds and normalises the DAG (constant folding,
common subexpressions, slices / rolls / concatenation pushed into stencil offsets and grid windows),
`ModTrace` is the `mod` namespace the operator sees and `TraceContext` its `ctx` (reference
src/odil/core.py:865-990).  Code generation lives in stencil_codegen.py.
"""

import math

import numpy as np
import torch

from .param_tape import OffGrid, ParamTensor

from .backend import torch_dtype


class TraceUnsupported(Exception):
    pass


_R, _B, _I = "r", "b", "i"


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

    def roll(self, n, shifts, virtual=False):
        """The grid function i -> n(i - shifts) (periodic, numpy.roll convention), built by pushing
        the shift into the leaves: reads change their stencil offset, index leaves wrap, tensors are
        rolled once on the device (virtual: indexed with an offset instead -- the gradient expressions of
        stencil_grad.py shift whole sub-expressions to every stencil neighbour, and a rolled copy of a constant
        field per neighbour would cost its size each); everything else is pointwise."""
        if not any(shifts) or n.host:
            return n
        memo = self.__dict__.setdefault("_roll_memo", dict())
        key = (n.idx, shifts, virtual)
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
        elif n.op == "lindex":  # rank-local cell index of a slab decomposition: shifts, never wraps
            res = self.binary("sub", n, shifts[n.attr[0]]) if shifts[n.attr[0]] else n
        elif n.op in ("tensor", "rtensor"):
            slot, before = (n.attr, (0,) * len(G)) if n.op == "tensor" else n.attr
            t = self.tensors[slot]
            dims, amounts, full = [], [], True
            for d, r in enumerate(shifts):
                td = d - (len(G) - t.dim())
                if r and td >= 0 and t.shape[td] > 1:
                    dims.append(td)
                    amounts.append(int(r))
                    full = full and t.shape[td] == G[d]
            if not dims:
                res = n
            elif (virtual and full) or n.op == "rtensor":
                if not full:
                    raise TraceUnsupported("roll of a constant array shorter than the grid")
                total = [0] * len(G)
                for td, r in zip(dims, amounts):
                    total[td + len(G) - t.dim()] = r
                total = tuple((a + b) % G[d] for d, (a, b) in enumerate(zip(before, total)))
                res = self.node("rtensor", attr=(slot, total), shape=n.shape, kind=n.kind) if any(total) else self.node(
                    "tensor", attr=slot, shape=n.shape, kind=n.kind)
            else:
                res = self.tensor(torch.roll(t, amounts, dims))
        elif n.op == "aparam":
            res = n
        else:
            args = tuple(self.roll(a, shifts, virtual) for a in n.args)
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
        if isinstance(a, (ParamTensor, OffGrid)) or isinstance(b, (ParamTensor, OffGrid)):
            # a whole parameter array (or an expression of such) times / plus a host scalar of the trace: an output in
            # parameter space (param_tape.py)
            if op not in ("add", "sub", "mul", "div"):
                raise TraceUnsupported("operation '{}' between a symbol and a parameter array".format(op))
            return OffGrid(op, a, b)
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


class _TraceDomain:
    """The domain as an operator sees it WHILE IT IS TRACED: every attribute is the real Domain's, except `mod`, which is
    the tracing namespace -- reference operators take their backend from the domain (`mod = ctx.domain.mod`,
    reference examples/poisson/poisson.py:90-93), and arithmetic on symbols through the real backend would fail (and the
    operator would silently keep the autograd path)."""

    def __init__(self, domain, mod):
        object.__setattr__(self, "_domain", domain)
        object.__setattr__(self, "mod", mod)

    def __getattr__(self, name):
        return getattr(self._domain, name)

    def __setattr__(self, name, value):
        setattr(self._domain, name, value)


class TraceContext:
    """`Context` (reference core.py:865-990) whose reads return symbols."""

    class Raw:
        def __init__(self, value):
            self.value = value

    def __init__(self, tr, state, extra, tracers):
        from .core import Context

        self.Raw = Context.Raw
        self._tr = tr
        self.state = state
        self.extra = extra
        self.dtype = tr.domain.dtype
        self.mod = ModTrace(tr)
        self.domain = _TraceDomain(tr.domain, self.mod)
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
        domain = self._tr.domain
        loc = loc or "c" * domain.ndim
        if any(c not in "cn" for c in loc) or len(loc) != domain.ndim:
            return domain.indices(*dims, loc=loc)
        shape = domain.get_field_shape(loc)
        idims = domain._names_to_indices(dims, list(domain.dimnames))
        res = tuple(self._tr.node("index", attr=(d, loc), shape=shape, kind=_I) for d in idims)
        return res[0] if len(dims) == 1 else res

    def points(self, *dims, loc=None):
        domain = self._tr.domain
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

        domain = self._tr.domain
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
