"""Symbolic reverse mode on the traced DAG: the gradient of a traced operator as expressions of its own.

`k_fwd` of odil_amd/stencil_codegen.py differentiates the operator in registers and used to store one cotangent
array per live read (`cot_r(i) = dL / d read_r` at grid point i); the gathers then form
`g[j] = sum_r cot_r(j - shift_r)`.  For the flow-reconstruction workload that is 17 arrays written and read again
per grid point where the fields themselves are 4 (reference examples/velocity_from_tracer/veltracer.py:34-130,
src/odil/core.py:1098-1107: the reference leaves this to XLA's fusion).

Here the cotangents are CUT higher up: the adjoint `A_c(i)` of a node c (an output of the operator, typically) is
stored once -- or not at all when c is cheap enough to evaluate again -- and the gather of a field forms

    g[j] = sum over cuts c, over live reads r of the field below c:   [A_c * dc/dr] (j - shift_r)

where `dc/dr` is the LOCAL derivative of the sub-expression, re-evaluated at the neighbouring point from the source
fields (cache hits).  Both factors are built here as nodes of the same DAG: `dc/dr` by reverse accumulation node by
node, "evaluated at j - shift" by `Tracer.roll`, which pushes the shift into the leaves (reads change their stencil
offset, index leaves wrap, constant arrays are indexed with an offset).  Common sub-expressions of the shifted
copies merge in the tracer's table.  The generated gather is then a plain pointwise kernel over that expression.
"""

from .stencil_trace import _B, _R, TraceUnsupported

_TRANSCENDENTAL = {"exp", "log", "tanh", "sqrt", "sin", "cos", "pow", "atan2", "div"}


class GradBuilder:
    """Expression builder with the few algebraic identities reverse accumulation needs (x * 1, x + 0, x * 0)."""

    def __init__(self, tr, G, need, stop):
        """need: {node idx: depends on an unknown}; stop: node indices whose own storage carries their adjoint (other
        cuts): differentiation does not descend through them."""
        self.tr, self.G, self.need, self.stop = tr, tuple(G), need, stop

    # ---- node construction (windows are ignored: a gradient expression lives on the whole grid; whatever lies
    # outside an output's window is masked by its seed) ----------------------------------------------------------
    def _node(self, op, args, kind=_R, attr=None):
        host = all(a.host for a in args)
        return self.tr.node(op, tuple(args), attr=attr, shape=() if host else self.G, kind=kind, host=host)

    def const(self, v):
        return self.tr.const(float(v))

    @staticmethod
    def _is(n, value):
        return n is not None and n.op == "const" and n.kind != _B and float(n.attr) == value

    def add(self, a, b):
        if a is None or self._is(a, 0.0):
            return b
        if b is None or self._is(b, 0.0):
            return a
        if a.op == "const" and b.op == "const":
            return self.const(float(a.attr) + float(b.attr))
        return self._node("add", (a, b))

    def neg(self, a):
        if a is None:
            return None
        if a.op == "const":
            return self.const(-float(a.attr))
        if a.op == "neg":
            return a.args[0]
        return self._node("neg", (a,))

    def mul(self, a, b):
        if a is None or b is None or self._is(a, 0.0) or self._is(b, 0.0):
            return None
        if self._is(a, 1.0):
            return b
        if self._is(b, 1.0):
            return a
        if a.op == "const" and b.op == "const":
            return self.const(float(a.attr) * float(b.attr))
        return self._node("mul", (a, b))

    def div(self, a, b):
        if a is None:
            return None
        if self._is(b, 1.0):
            return a
        if a.op == "const" and b.op == "const":
            return self.const(float(a.attr) / float(b.attr))
        return self._node("div", (a, b))

    def where(self, c, a, b):
        if a is None and b is None:
            return None
        zero = self.const(0.0)
        return self._node("where", (c, a if a is not None else zero, b if b is not None else zero))

    def cmp(self, op, a, b):
        return self._node(op, (a, b), kind=_B)

    def unary(self, op, a):
        return self._node(op, (a,))

    def real(self, n):
        if n.kind == _R:
            return n
        if n.op == "const":
            return self.const(float(n.attr))
        return self._node("cast", (n,))

    # ---- reverse accumulation --------------------------------------------------------------------------------
    def adjoints(self, root, seed, order, record_stops=False):
        """{read idx: adjoint expression} of the live reads below `root` given the adjoint `seed` of root.
        `order`: topologically sorted nodes (ascending) that contain the sub-DAG.  record_stops: the adjoints that
        arrive at the `stop` nodes are returned as well (what k_fwd's reverse pass stores for them)."""
        adj = {root.idx: seed}
        out = dict()
        for n in reversed(order):
            a = adj.pop(n.idx, None)
            if a is None:
                continue
            if n.idx != root.idx and n.idx in self.stop:
                if record_stops:
                    out[n.idx] = self.add(out.get(n.idx), a)
                continue
            if n.op == "read":
                out[n.idx] = self.add(out.get(n.idx), a)
                continue
            for arg, contrib in self._pullback(n, a):
                if contrib is None or not self.need.get(arg.idx, False):
                    continue
                adj[arg.idx] = self.add(adj.get(arg.idx), contrib)
        return out

    def _pullback(self, n, g):
        op, A = n.op, n.args
        R = self.real
        if op in ("win", "cast"):
            return [(A[0], g)]
        if op == "add":
            return [(A[0], g), (A[1], g)]
        if op == "sub":
            return [(A[0], g), (A[1], self.neg(g))]
        if op == "mul":
            return [(A[0], self.mul(g, R(A[1]))), (A[1], self.mul(g, R(A[0])))]
        if op == "div":
            return [(A[0], self.div(g, R(A[1]))), (A[1], self.neg(self.div(self.mul(g, n), R(A[1]))))]
        if op == "neg":
            return [(A[0], self.neg(g))]
        if op == "where":
            return [(A[1], self.where(A[0], g, None)), (A[2], self.where(A[0], None, g))]
        if op == "pow":
            x, p = R(A[0]), R(A[1])
            if A[1].op == "const" and float(A[1].attr) == 2.0:
                return [(A[0], self.mul(g, self.mul(self.const(2.0), x)))]
            if A[1].op == "const" and float(A[1].attr) == 1.0:
                return [(A[0], g)]
            dx = self.mul(g, self.mul(p, self._node("pow", (x, self._node("sub", (p, self.const(1.0)))))))
            dp = self.mul(g, self.mul(n, self.unary("log", x)))
            return [(A[0], dx), (A[1], dp)]
        if op in ("min", "max"):
            x, y = R(A[0]), R(A[1])
            first = "lt" if op == "min" else "gt"
            half = self.mul(g, self.const(0.5))
            tie = self.cmp("eq", x, y)
            return [(A[0], self.where(self.cmp(first, x, y), g, self.where(tie, half, None))),
                    (A[1], self.where(self.cmp(first, y, x), g, self.where(tie, half, None)))]
        if op == "atan2":
            y, x = R(A[0]), R(A[1])
            den = self.add(self.mul(y, y), self.mul(x, x))
            return [(A[0], self.div(self.mul(g, x), den)), (A[1], self.neg(self.div(self.mul(g, y), den)))]
        if op == "abs":
            x = R(A[0])
            zero = self.const(0.0)
            return [(A[0], self.where(self.cmp("gt", x, zero), g, self.where(self.cmp("lt", x, zero), self.neg(g), None)))]
        if op == "relu":
            return [(A[0], self.where(self.cmp("gt", R(A[0]), self.const(0.0)), g, None))]
        if op == "cos":
            return [(A[0], self.neg(self.mul(g, self.unary("sin", R(A[0])))))]
        if op == "sin":
            return [(A[0], self.mul(g, self.unary("cos", R(A[0]))))]
        if op == "exp":
            return [(A[0], self.mul(g, n))]
        if op == "log":
            return [(A[0], self.div(g, R(A[0])))]
        if op == "tanh":
            return [(A[0], self.mul(g, self._node("sub", (self.const(1.0), self.mul(n, n)))))]
        if op == "sqrt":
            return [(A[0], self.div(g, self.mul(self.const(2.0), n)))]
        if op in ("floor", "stopgrad", "tensor", "rtensor", "index", "lindex", "const", "tracer", "aparam"):
            return []  # (parameters: their gradients are reduced by k_fwd; here they are coefficients)
        raise TraceUnsupported("symbolic derivative of " + op)


def subdag(root, stop=()):
    """Nodes reachable from `root` (inclusive) without descending through `stop` nodes, ascending."""
    seen, stack = dict(), [root]
    while stack:
        n = stack.pop()
        if n.idx in seen:
            continue
        seen[n.idx] = n
        if n.idx != root.idx and n.idx in stop:
            continue
        stack.extend(n.args)
    return [seen[i] for i in sorted(seen)]


def differentiable(nodes, need):
    """Can the sub-DAG be differentiated symbolically?  Parameters (network weights, `Array` unknowns) below a cut
    would lose their gradients (they are reduced by k_fwd's in-register reverse pass), networks have no symbolic
    derivative here."""
    for n in nodes:
        if not need.get(n.idx, False):
            continue
        if n.op in ("mlp", "mlp_out", "aparam"):
            return False
    return True


def cost(nodes):
    """(distinct leaf loads, transcendental operations) of evaluating a set of nodes once."""
    loads = sum(1 for n in nodes if n.op in ("read", "tensor", "rtensor"))
    heavy = sum(1 for n in nodes if n.op in _TRANSCENDENTAL and not n.host and not (n.op == "div" and n.args[1].host)
                and not (n.op == "pow" and n.args[1].op == "const" and float(n.args[1].attr) in (1.0, 2.0)))
    return loads, heavy
