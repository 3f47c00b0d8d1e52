"""Outputs in PARAMETER space as one generated kernel (instead of the torch replay of odil_amd/param_tape.py).

The heat examples regularise the network's weights: `(stop_gradient(ww) - ww) * k` on the concatenated weight arrays, `k`
annealed with the epoch (reference examples/heat/heat.py:131-136); a weight decay or a prior on an `Array` has the same
shape.  Such a term is an ELEMENTWISE expression over pieces of the parameter arrays laid end to end.  This module turns
the tape of torch operations recorded while tracing (param_tape.py) into that form -- a list of segments, each an
expression tree over "atoms" (array, offset, detached?) -- and emits one single-workgroup kernel `k_par` that

  1. evaluates every such output, reduces its mean square deterministically (thread-strided sums in a fixed order,
     block_sum), writes term and norm, adds the term to the loss of the grid kernels;
  2. forms the gradient in GATHER form: one thread per parameter entry sums, in a fixed order, `2 f / n * df/d(entry)`
     over the places the entry occurs (no atomics), and adds it to (or, for arrays no grid kernel writes, sets) the
     entry's slot of the packed gradient.

No torch kernel runs per evaluation, the host scalars travel like those of the grid kernels (by value, or from the device
table of a replayed hipGraph), and the slab path can run it after its parameter all-reduce.  Tape operations outside the
supported set -- flatten / reshape / contiguous / clone / casts, concatenation of flat pieces, detach, + - * / and negation
with each other and with scalars -- raise `Unsupported`; the torch replay stays as the fallback for those.
"""


from .param_tape import OffGrid, ParamTensor, _Ref


class Unsupported(Exception):
    pass


# expression trees: ("atom", array index, offset, detached) | ("const", float) | ("hs", Sym) | ("neg", a) | (op, a, b)
_PASS = {"flatten", "reshape", "view", "contiguous", "clone", "ravel", "squeeze", "unsqueeze", "to", "float", "double",
         "type_as", "view_as", "reshape_as", "_to_copy", "alias", "requires_grad_", "type"}
_CAT = {"cat", "concat", "concatenate", "hstack"}
_BIN = {"add": "add", "__add__": "add", "__radd__": "radd", "sub": "sub", "subtract": "sub", "__sub__": "sub", "__rsub__": "rsub",
        "mul": "mul", "multiply": "mul", "__mul__": "mul", "__rmul__": "rmul", "div": "div", "true_divide": "div",
        "divide": "div", "__truediv__": "div", "__rtruediv__": "rdiv"}


def _numel(shape):
    n = 1
    for s in shape or ():
        n *= int(s)
    return n


def _shift(expr, by):
    """The expression with every atom's offset advanced by `by` elements."""
    if expr[0] == "atom":
        return ("atom", expr[1], expr[2] + by, expr[3])
    if expr[0] in ("const", "hs"):
        return expr
    return (expr[0],) + tuple(_shift(e, by) for e in expr[1:])


def _detach(expr):
    if expr[0] == "atom":
        return ("atom", expr[1], expr[2], True)
    if expr[0] in ("const", "hs"):
        return expr
    return (expr[0],) + tuple(_detach(e) for e in expr[1:])


def _combine(op, a, b):
    """Elementwise `a op b` of two segment lists (or a list and a scalar expression)."""
    if not isinstance(a, list) and not isinstance(b, list):
        return (op, a, b)
    if not isinstance(a, list):
        return [(n, (op, a, e)) for n, e in b]
    if not isinstance(b, list):
        return [(n, (op, e, b)) for n, e in a]
    if sum(n for n, _ in a) != sum(n for n, _ in b):
        raise Unsupported("broadcast between parameter-space values of different sizes")
    out, ia, ib, oa, ob = [], 0, 0, 0, 0  # segment index and offset inside the segment, per operand
    while ia < len(a) and ib < len(b):
        na, ea = a[ia]
        nb, eb = b[ib]
        n = min(na - oa, nb - ob)
        out.append((n, (op, _shift(ea, oa), _shift(eb, ob))))
        oa, ob = oa + n, ob + n
        if oa == na:
            ia, oa = ia + 1, 0
        if ob == nb:
            ib, ob = ib + 1, 0
    return out


class Symbolic:
    """Tape values as segment lists IN C ORDER of the value's shape.  `leaves`: {tape id: array index in
    arrays_from_state order}, `numel`: {index: size}.  Every value carries the shape it had when the operator was traced
    (`tape.shapes`): a segment list only says how the elements lie end to end, so an operation is accepted only when its
    torch result is that same end-to-end order -- elementwise operations between EQUAL shapes (or with a scalar),
    concatenation of 1-D pieces or along axis 0; anything that broadcasts, or interleaves rows, raises Unsupported and the
    torch replay of param_tape.py (which is exact for any tape) evaluates the output."""

    def __init__(self, tape, numel):
        self.tape, self.numel, self.env = tape, numel, dict()
        for ident, index in tape.leaves.items():
            self.env[ident] = [(numel[index], ("atom", index, 0, False))]

    def shape(self, x):
        """Shape of a tape value at trace time; None for scalars (numbers, host symbols)."""
        if isinstance(x, ParamTensor):
            x = _Ref(x._id)
        if isinstance(x, _Ref):
            if x.id not in self.tape.shapes:
                raise Unsupported("value of the tape without a recorded shape")
            return tuple(self.tape.shapes[x.id])
        return None

    def value(self, x):
        from .stencil_trace import Sym

        if isinstance(x, _Ref):
            if x.id not in self.env:
                raise Unsupported("value of the tape that was not formed by a supported operation")
            return self.env[x.id]
        if isinstance(x, ParamTensor):
            return self.value(_Ref(x._id))
        if isinstance(x, Sym):
            if not x.host:
                raise Unsupported("grid value in a parameter-space expression")
            return ("hs", x) if x.op != "const" else ("const", float(x.attr))
        if isinstance(x, (bool, int, float)):
            return ("const", float(x))
        # (a plain tensor operand -- even 0-dim -- is NOT baked in: the operator may rebuild it between epochs, and
        # reading it here would synchronise with the device; the torch replay re-reads it every evaluation)
        raise Unsupported("operand of type {}".format(type(x).__name__))

    def binary(self, op, x, y):
        """`x op y` of two operands as they appear on the tape / in an OffGrid expression."""
        a, b = self.value(x), self.value(y)
        sa, sb = self.shape(x), self.shape(y)
        if isinstance(a, list) and isinstance(b, list) and sa != sb:
            raise Unsupported("elementwise '{}' between shapes {} and {} (broadcast)".format(op, sa, sb))
        return _combine(op, a, b)

    def run(self, ops):
        for func, args, kwargs, outs in ops:
            name = getattr(func, "__name__", "")
            if len(outs) != 1:
                raise Unsupported("operation '{}' with several results".format(name))
            if name in _PASS:
                res = self.value(args[0])
                # (these keep the C order of the elements; the NEW shape is on the tape for the operations that follow)
                if isinstance(res, list) and sum(n for n, _ in res) != _numel(self.tape.shapes.get(outs[0])):
                    raise Unsupported("operation '{}' changes the number of elements".format(name))
            elif name == "detach":
                res = [(n, _detach(e)) for n, e in self.value(args[0])]
            elif name in _CAT:
                pieces = args[0]
                axis = kwargs.get("dim", kwargs.get("axis", args[1] if len(args) > 1 else 0))
                res = []
                for p in pieces:
                    v = self.value(p)
                    shape = self.shape(p)
                    if not isinstance(v, list) or shape is None:
                        raise Unsupported("scalar in a concatenation")
                    # end-to-end order of the result == pieces laid end to end: 1-D pieces, or axis 0 of C-ordered ones
                    if not (len(shape) == 1 and axis in (0, -1)) and not (len(shape) > 1 and axis in (0, -len(shape))):
                        raise Unsupported("concatenation of pieces of shape {} along axis {}".format(shape, axis))
                    res += v
            elif name in _BIN and len(args) == 2 and not {k for k in kwargs if k != "alpha"}:
                op = _BIN[name]
                if kwargs.get("alpha", 1) != 1:
                    raise Unsupported("alpha argument")
                x, y = args
                if op.startswith("r"):
                    op, x, y = op[1:], y, x
                res = self.binary(op, x, y)
                if isinstance(res, list) and sum(n for n, _ in res) != _numel(self.tape.shapes.get(outs[0])):
                    raise Unsupported("operation '{}' broadcasts".format(name))
            elif name in ("neg", "__neg__", "negative"):
                v = self.value(args[0])
                res = [(n, ("neg", e)) for n, e in v] if isinstance(v, list) else ("neg", v)
            else:
                raise Unsupported("operation '{}' on a parameter array".format(name))
            self.env[outs[0]] = res

    def offgrid(self, expr):
        """Segment list of an OffGrid expression (param_tape.OffGrid) whose tape slice has been run."""
        return self._offgrid(expr)[0]

    def _offgrid(self, expr):
        """(segments or scalar expression, shape or None)."""
        if not isinstance(expr, OffGrid):
            return self.value(expr), self.shape(expr)
        a, sa = self._offgrid(expr.a)
        if expr.op == "leaf":
            return a, sa
        if expr.op == "neg":
            return ([(n, ("neg", e)) for n, e in a] if isinstance(a, list) else ("neg", a)), sa
        b, sb = self._offgrid(expr.b)
        if isinstance(a, list) and isinstance(b, list) and sa != sb:
            raise Unsupported("elementwise '{}' between shapes {} and {} (broadcast)".format(expr.op, sa, sb))
        return _combine(expr.op, a, b), (sa if isinstance(a, list) else sb)


def convert(tape, offgrid, numel):
    """[(position among the outputs, segment list)] for the parameter-space outputs `offgrid` = [(k, expr, tape slice)];
    raises Unsupported when some operation has no elementwise form."""
    res = []
    for k, expr, ops in offgrid:
        sym = Symbolic(tape, numel)
        sym.run(ops)
        segs = sym.offgrid(expr)
        if not isinstance(segs, list):
            raise Unsupported("scalar output in parameter space")
        res.append((k, segs))
    return res


# ======================================================================================
# code generation
# ======================================================================================
def _c(expr, j, slot_of, cg):
    """C expression of `expr` at element `j` (a C expression) of its segment."""
    kind = expr[0]
    if kind == "atom":
        return "pa.val[{}][{} + {}]".format(slot_of[expr[1]], expr[2], j)
    if kind == "const":
        return "((T){!r})".format(expr[1])
    if kind == "hs":
        return cg.r(expr[1])
    if kind == "neg":
        return "(-{})".format(_c(expr[1], j, slot_of, cg))
    sym = {"add": "+", "sub": "-", "mul": "*", "div": "/"}[kind]
    return "({} {} {})".format(_c(expr[1], j, slot_of, cg), sym, _c(expr[2], j, slot_of, cg))


def _atoms(expr, path=()):
    """[(path, atom)] of the differentiable atoms of a tree."""
    if expr[0] == "atom":
        return [] if expr[3] else [(path, expr)]
    if expr[0] in ("const", "hs"):
        return []
    out = []
    for i, e in enumerate(expr[1:]):
        out += _atoms(e, path + (i,))
    return out


def _d(expr, target, j, slot_of, cg, path=()):
    """C expression of d expr / d (the atom at `target`), or None when it vanishes."""
    kind = expr[0]
    if kind == "atom":
        return "((T)1)" if path == target else None
    if kind in ("const", "hs"):
        return None
    if kind == "neg":
        d = _d(expr[1], target, j, slot_of, cg, path + (0,))
        return None if d is None else "(-{})".format(d)
    da = _d(expr[1], target, j, slot_of, cg, path + (0,))
    db = _d(expr[2], target, j, slot_of, cg, path + (1,))
    a, b = (lambda: _c(expr[1], j, slot_of, cg)), (lambda: _c(expr[2], j, slot_of, cg))
    if kind in ("add", "sub"):
        if da is None and db is None:
            return None
        if db is None:
            return da
        db = db if kind == "add" else "(-{})".format(db)
        return db if da is None else "({} + {})".format(da, db)
    if kind == "mul":
        terms = ([] if da is None else ["({} * {})".format(da, b())]) + ([] if db is None else ["({} * {})".format(a(), db)])
        return "({})".format(" + ".join(terms)) if terms else None
    # div
    terms = ([] if da is None else ["({} / {})".format(da, b())]) + (
        [] if db is None else ["(-({} * {}) / ({} * {}))".format(a(), db, b(), b())])
    return "({})".format(" + ".join(terms)) if terms else None


def emit(cg, S, outputs, fresh):
    """Appends `k_par` and its launcher to the source lines S.  outputs: [(k, segments)] of convert(); fresh: array
    indices whose gradient slot no grid kernel writes (set instead of added to).  Returns the array indices in the order
    of ParArgs.val / .grad."""
    arrays = []
    for _, segs in outputs:
        for _, e in segs:
            stack = [e]
            while stack:
                x = stack.pop()
                if x[0] == "atom":
                    if x[1] not in arrays:
                        arrays.append(x[1])
                elif x[0] not in ("const", "hs"):
                    stack.extend(x[1:])
    slot_of = {index: s for s, index in enumerate(arrays)}
    K, nq = max(1, len(arrays)), len(outputs)
    S.append("struct ParArgs {{ const T* val[{0}]; T* grad[{0}]; T* pout; }};".format(K))
    S.append('extern "C" __global__ __launch_bounds__(NB) void k_par(const Args a, const ParArgs pa) {')
    S.append("  __shared__ T sm[NB / 64];")
    # ---- terms ------------------------------------------------------------------------------------------------------
    for q, (_, segs) in enumerate(outputs):
        n = sum(m for m, _ in segs)
        S.append("  {")
        S.append("  T s = (T)0;")
        start = 0
        for m, e in segs:
            S.append("  for (int j = threadIdx.x; j < {}; j += NB) {{ const T f = {}; s = s + f * f; }}".format(m, _c(e, "j", slot_of, cg)))
            start += m
        S.append("  s = block_sum(s, sm);")
        S.append("  if (threadIdx.x == 0) {{ const T t = s / (T){!r}; pa.pout[{}] = t; pa.pout[{}] = FN(sqrt)(t); a.out[0] = a.out[0] + t; }}".format(
            float(n), 2 * q, 2 * q + 1))
        S.append("  __syncthreads();")
        S.append("  }")
    # ---- gradients, gather form ------------------------------------------------------------------------------------------
    for index in arrays:
        s_ = slot_of[index]
        S.append("  for (int p = threadIdx.x; p < {}; p += NB) {{".format(cg.par_numel[index]))
        S.append("    T acc = (T)0;")
        for q, (_, segs) in enumerate(outputs):
            n = sum(m for m, _ in segs)
            for m, e in segs:
                for path, atom in _atoms(e):
                    if atom[1] != index:
                        continue
                    d = _d(e, path, "j", slot_of, cg)
                    if d is None:
                        continue
                    S.append("    if (p >= {0} && p < {1}) {{ const int j = p - {0}; acc = acc + ((T){2!r} * {3}) * {4}; }}".format(
                        atom[2], atom[2] + m, 2.0 / n, _c(e, "j", slot_of, cg), d))
        S.append("    pa.grad[{0}][p] = {1}acc;".format(s_, "" if index in fresh else "pa.grad[{}][p] + ".format(s_)))
        S.append("  }")
    S.append("}")
    S.append('extern "C" int jit_par(const Args* a, const ParArgs* pa, void* stream) {')
    S.append("  hipLaunchKernelGGL(k_par, dim3(1), dim3(NB), 0, (hipStream_t)stream, *a, *pa);")
    S.append("  return (int)hipGetLastError();")
    S.append("}")
    return arrays
