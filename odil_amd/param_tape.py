"""Outputs of a traced operator that live in PARAMETER space, not on the grid.

An operator may return, next to its grid residuals, terms built from the parameter arrays alone -- the weight
regulariser of the heat examples (reference examples/heat/heat.py:131-136: `(stop_gradient(ww) - ww) * k` on the
concatenated network weights, `k` annealed with the epoch), a weight decay, a prior on an `Array` of constants.  Such a
term is a few dozen numbers: nothing for a generated kernel, but until round 3 its presence sent the WHOLE operator to
the generic autograd path.

While the operator is traced, the parameter arrays of the state are `ParamTensor`s: ordinary tensors whose every torch
operation is also written to a tape (`__torch_function__`).  User code -- `mod.flatten`, `mod.concatenate`,
`mod.stop_gradient`, arithmetic -- runs on them as on any tensor; where such a value meets a symbol of the trace (a host
scalar: a function of `ctx.tracers`) the result is an `OffGrid` expression.  An operator output that is a `ParamTensor`
or an `OffGrid` is kept out of the generated kernels; every evaluation REPLAYS the slice of the tape it needs on the
current parameter arrays (differentiable leaves), multiplies in the current host scalars, and adds the term's mean
square to the loss and its gradient to the parameters' gradients (torch autograd on the few small tensors).
"""

import torch
from torch.utils._pytree import tree_map


class _Ref:
    __slots__ = ("id",)

    def __init__(self, ident):
        self.id = ident


class ParamTape:
    def __init__(self):
        self.ops = []      # (func, args, kwargs, out ids) with ParamTensors replaced by _Ref
        self.leaves = {}   # tape id -> index of the array in `Domain.arrays_from_state` order
        self.shapes = {}   # tape id -> shape of the value when the operator was traced (param_expr.py checks broadcasts)
        self.count = 0

    def new_id(self):
        self.count += 1
        return self.count - 1

    def leaf(self, tensor, index):
        ident = self.new_id()
        self.leaves[ident] = index
        self.shapes[ident] = tuple(tensor.shape)
        return ParamTensor.wrap(tensor, self, ident)

    def slice_for(self, ids):
        """The recorded operations the values `ids` depend on, in order."""
        need, keep = set(ids), []
        for op in reversed(self.ops):
            if any(o in need for o in op[3]):
                keep.append(op)
                tree_map(lambda a: need.add(a.id) if isinstance(a, _Ref) else None, (op[1], op[2]))
        return keep[::-1]

    def replay(self, ops, arrays):
        """Re-runs `ops` with the leaves taken from `arrays`; returns {tape id: tensor}."""
        env = {ident: arrays[index] for ident, index in self.leaves.items()}
        get = lambda a: env[a.id] if isinstance(a, _Ref) else a
        for func, args, kwargs, outs in ops:
            res = func(*tree_map(get, args), **tree_map(get, kwargs))
            if isinstance(res, torch.Tensor):
                res = [res]
            for ident, r in zip(outs, res):
                env[ident] = r
        return env


_ARITH = {"__mul__": ("mul", 0), "mul": ("mul", 0), "multiply": ("mul", 0), "__rmul__": ("mul", 1),
          "__add__": ("add", 0), "add": ("add", 0), "__radd__": ("add", 1),
          "__sub__": ("sub", 0), "sub": ("sub", 0), "subtract": ("sub", 0), "__rsub__": ("sub", 1),
          "__truediv__": ("div", 0), "div": ("div", 0), "true_divide": ("div", 0), "divide": ("div", 0),
          "__rtruediv__": ("div", 1)}


class ParamTensor(torch.Tensor):
    """A parameter array of the state (or a value computed from such) during tracing: see the module docstring."""

    @staticmethod
    def wrap(tensor, tape, ident):
        t = torch.Tensor._make_subclass(ParamTensor, tensor, tensor.requires_grad)
        t._tape, t._id = tape, ident
        return t

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        from .stencil_trace import Sym

        name = getattr(func, "__name__", "")
        flat = list(args) + list(kwargs.values())
        if any(isinstance(a, (Sym, OffGrid)) for a in flat):
            # a parameter-space value meets a symbol of the trace: host scalars only (OffGrid checks)
            if name in _ARITH and len(args) == 2 and not kwargs:
                op, swapped = _ARITH[name]
                a, b = (args[1], args[0]) if swapped else args
                return OffGrid(op, a, b)
            return NotImplemented
        tape = next(a._tape for a in _flatten(flat) if isinstance(a, ParamTensor))
        plain = lambda a: a.as_subclass(torch.Tensor) if isinstance(a, ParamTensor) else a
        with torch._C.DisableTorchFunctionSubclass():
            res = func(*tree_map(plain, args), **tree_map(plain, kwargs))
        tensors = [res] if isinstance(res, torch.Tensor) else (
            list(res) if isinstance(res, (tuple, list)) and res and all(isinstance(r, torch.Tensor) for r in res) else None)
        if tensors is None:
            return res  # shapes, dtypes, Python numbers: not part of the tape
        ref = lambda a: _Ref(a._id) if isinstance(a, ParamTensor) else a
        outs = [tape.new_id() for _ in tensors]
        for ident, t in zip(outs, tensors):
            tape.shapes[ident] = tuple(t.shape)
        tape.ops.append((func, tree_map(ref, args), tree_map(ref, kwargs), outs))
        wrapped = [ParamTensor.wrap(t, tape, i) for t, i in zip(tensors, outs)]
        return wrapped[0] if isinstance(res, torch.Tensor) else type(res)(wrapped)


def _flatten(items):
    for a in items:
        if isinstance(a, (list, tuple)):
            yield from _flatten(a)
        else:
            yield a


class OffGrid:
    """Expression over parameter-space values (ParamTensor), host symbols of the trace and Python numbers."""

    def __init__(self, op, a, b=None):
        from .stencil_trace import Sym, TraceUnsupported

        for x in (a, b):
            if isinstance(x, Sym) and not x.host:
                raise TraceUnsupported("grid value combined with a whole parameter array")
            if isinstance(x, torch.Tensor) and not isinstance(x, ParamTensor) and x.requires_grad:
                raise TraceUnsupported("differentiable tensor outside ctx.field / ctx.neural_net")
        self.op, self.a, self.b = op, a, b

    @staticmethod
    def of(x):
        return x if isinstance(x, OffGrid) else OffGrid("leaf", x)

    def __mul__(self, o):
        return OffGrid("mul", self, o)

    def __rmul__(self, o):
        return OffGrid("mul", o, self)

    def __add__(self, o):
        return OffGrid("add", self, o)

    def __radd__(self, o):
        return OffGrid("add", o, self)

    def __sub__(self, o):
        return OffGrid("sub", self, o)

    def __rsub__(self, o):
        return OffGrid("sub", o, self)

    def __truediv__(self, o):
        return OffGrid("div", self, o)

    def __rtruediv__(self, o):
        return OffGrid("div", o, self)

    def __neg__(self):
        return OffGrid("neg", self)

    # ---- what the evaluation needs ---------------------------------------------------------------------------
    def tape(self):
        for x in (self.a, self.b):
            t = x._tape if isinstance(x, ParamTensor) else (x.tape() if isinstance(x, OffGrid) else None)
            if t is not None:
                return t
        return None

    def param_ids(self, acc=None):
        acc = [] if acc is None else acc
        for x in (self.a, self.b):
            if isinstance(x, ParamTensor):
                acc.append(x._id)
            elif isinstance(x, OffGrid):
                x.param_ids(acc)
        return acc

    def evaluate(self, env, host_value):
        """env: {tape id: tensor} of a replay; host_value(sym) -> float."""
        from .stencil_trace import Sym

        def val(x):
            if isinstance(x, OffGrid):
                return x.evaluate(env, host_value)
            if isinstance(x, ParamTensor):
                return env[x._id]
            if isinstance(x, Sym):
                return host_value(x)
            return x

        a = val(self.a)
        if self.op == "leaf":
            return a
        if self.op == "neg":
            return -a
        b = val(self.b)
        return {"mul": lambda: a * b, "add": lambda: a + b, "sub": lambda: a - b, "div": lambda: a / b}[self.op]()
