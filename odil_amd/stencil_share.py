"""Which evaluations of a pointwise network does a traced operator repeat at NEIGHBOURING grid points?

The heat operators evaluate the conductivity network at the faces of every cell (reference
examples/heat/heat.py:86-98: `k_m`, `k_p` per space axis) -- and the lower face of cell i IS the upper face of cell
i - e: `k_m(i) = k_p(i - e)` wherever cell i has an interior lower neighbour, i.e. away from the wall row i = 0, where the
face value is extrapolated instead.  Half of the network evaluations (and of their reverse passes) of `k_fwd` are
therefore repeats.  This module PROVES such identities on the traced DAG so that the code generator may share the
evaluations between threads (stencil_codegen.py, tiled forward kernel):

    A(i) == B(i - e_axis)   for every i with i_axis >= 1

holds when the input expressions of network call A, and those of B shifted by one cell (`Tracer.roll`, which pushes the
shift into reads, index leaves and constant arrays), simplify to the SAME node once the index range i_axis in [1, n - 1]
is known: comparisons of index expressions against constants are decided over the range, `where` nodes with decided
conditions fold (the wall masks `idx == 0`, `idx == n - 1`, the periodic wrap of a shifted index), constant arrays are
compared by content.  Anything that does not fold to identity is simply not shared.
"""

import hashlib

import torch

from .stencil_trace import _B, _CMP, _I

_COMMUTATIVE = {"add", "mul", "min", "max", "and", "or", "eq", "ne"}


class RangeSimplifier:
    def __init__(self, tr, G, ranges):
        """ranges: {axis: (lo, hi)} inclusive bounds of the grid index along those axes (the others: whole axis)."""
        self.tr, self.G = tr, tuple(G)
        self.ranges = {d: ranges.get(d, (0, n - 1)) for d, n in enumerate(self.G)}
        self.memo = dict()
        self.tensor_ids = tr.__dict__.setdefault("_tensor_content_ids", dict())

    # ---- integer expressions as index + offset --------------------------------------------------------------
    def affine(self, n):
        """(axis or None, offset) when the integer node is index(axis) + offset or a constant, else None."""
        if n.op == "const" and n.kind in (_I, _B):
            return (None, int(n.attr))
        if n.op == "const" and float(n.attr) == int(n.attr):
            return (None, int(n.attr))
        if n.op == "index":
            return (n.attr[0], 0)
        if n.op in ("add", "sub") and n.kind == _I:
            a, b = self.affine(n.args[0]), self.affine(n.args[1])
            if a is None or b is None:
                return None
            sign = 1 if n.op == "add" else -1
            if b[0] is None:
                return (a[0], a[1] + sign * b[1])
            if a[0] is None and n.op == "add":
                return (b[0], a[1] + b[1])
        if n.op in ("cast", "win"):
            return self.affine(n.args[0])
        return None

    def bounds(self, n):
        a = self.affine(n)
        if a is None:
            return None
        if a[0] is None:
            return (a[1], a[1])
        lo, hi = self.ranges[a[0]]
        return (lo + a[1], hi + a[1])

    def decide(self, op, x, y):
        """True / False when the comparison holds / fails over the whole range, None when it depends on the point."""
        ax, ay = self.affine(x), self.affine(y)
        if ax is None or ay is None:
            return None
        if ax[0] is not None and ay[0] is not None:
            if ax[0] != ay[0]:
                return None
            d = ax[1] - ay[1]  # x - y is the constant d
            return {"lt": d < 0, "le": d <= 0, "gt": d > 0, "ge": d >= 0, "eq": d == 0, "ne": d != 0}[op]
        (xl, xh), (yl, yh) = self.bounds(x), self.bounds(y)
        if op in ("gt", "ge"):
            return self.decide({"gt": "lt", "ge": "le"}[op], y, x)
        if op == "lt":
            return True if xh < yl else (False if xl >= yh else None)
        if op == "le":
            return True if xh <= yl else (False if xl > yh else None)
        if op == "eq":
            return False if (xh < yl or xl > yh) else (True if xl == xh == yl == yh else None)
        if op == "ne":
            r = self.decide("eq", x, y)
            return None if r is None else not r
        return None

    # ---- constant arrays by content ---------------------------------------------------------------------------
    def tensor_leaf(self, n):
        slot, roll = (n.attr, None) if n.op == "tensor" else n.attr
        t = self.tr.tensors[slot]
        if t.numel() * t.element_size() > (64 << 20):
            return n
        if roll is not None and any(roll):
            lead = len(self.G) - t.dim()
            dims = [d - lead for d, r in enumerate(roll) if r and d >= lead and t.shape[d - lead] > 1]
            amounts = [roll[d + lead] for d in dims]
            t = torch.roll(t, amounts, dims) if dims else t
        key = (tuple(t.shape), str(t.dtype), hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest())
        canon = self.tensor_ids.get(key)
        if canon is None:
            canon = self.tensor_ids[key] = self.tr.tensor(t)
        return canon

    # ---- the rewrite ---------------------------------------------------------------------------------------------
    def __call__(self, n):
        if n.idx in self.memo:
            return self.memo[n.idx]
        res = self._simplify(n)
        self.memo[n.idx] = res
        return res

    def _simplify(self, n):
        tr = self.tr
        if n.host or n.op in ("read", "index", "lindex", "aparam", "const", "tracer"):
            return n
        if n.op in ("tensor", "rtensor"):
            return self.tensor_leaf(n)
        args = [self(a) for a in n.args]
        if n.op in _CMP:
            r = self.decide(n.op, args[0], args[1])
            if r is not None:
                return tr.const(bool(r))
        if n.op == "where":
            c = args[0]
            if c.op == "const":
                return args[1] if c.attr else args[2]
            if args[1] is args[2]:
                return args[1]
        if n.op in ("and", "or") and any(a.op == "const" for a in args):
            k = [a for a in args if a.op == "const"][0]
            other = [a for a in args if a is not k][0] if len([a for a in args if a is not k]) else k
            truth = bool(k.attr)
            if n.op == "and":
                return other if truth else tr.const(False)
            return tr.const(True) if truth else other
        if n.op == "not" and args[0].op == "const":
            return tr.const(not bool(args[0].attr))
        if n.op == "win":
            return args[0] if args[0].kind == n.kind else tr.node(n.op, tuple(args), attr=None, shape=n.shape, kind=n.kind, win=None)
        if n.op in _COMMUTATIVE and len(args) == 2 and args[0].idx > args[1].idx:
            args = [args[1], args[0]]
        return tr.node(n.op, tuple(args), attr=n.attr, shape=n.shape, kind=n.kind, host=all(a.host for a in args) and n.host,
                       win=None)


def shared_network_calls(tr, order, G, axes):
    """[(A, B, axis)]: network calls of the DAG with A(i) == B(i - e_axis) for every point with i_axis >= 1.
    `order`: the live nodes; axes: the grid axes along which threads can exchange values (the generator's tile)."""
    groups = dict()
    for n in order:
        if n.op == "mlp":
            groups.setdefault(n.attr, []).append(n)
    found, taken = [], set()
    for calls in groups.values():
        for axis in axes:
            n_ax = G[axis]
            if n_ax < 4:
                continue
            shift = tuple(1 if d == axis else 0 for d in range(len(G)))
            simp = RangeSimplifier(tr, G, {axis: (1, n_ax - 1)})
            for A in calls:
                if A.idx in taken:
                    continue
                for B in calls:
                    if B is A or B.idx in taken or any(B is b for _, b, _ in found) and False:
                        continue
                    try:
                        same = all(simp(a) is simp(tr.roll(b, shift, virtual=True)) for a, b in zip(A.args, B.args))
                    except Exception:
                        same = False
                    if same:
                        found.append((A, B, axis))
                        taken.add(A.idx)
                        break
    return found
