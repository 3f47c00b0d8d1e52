"""NumPy interpreter of the tracer's expression DAG (TEST INFRASTRUCTURE): evaluates a `Sym` node of
odil_amd/stencil_trace.py on the whole grid, so that the symbolic gradient expressions of
odil_amd/stencil_grad.py -- what the generated gathers compute point by point -- can be checked on the CPU against
autograd, without a GPU."""

import math

import numpy as np

_UN = {"neg": np.negative, "abs": np.abs, "cos": np.cos, "sin": np.sin, "exp": np.exp, "log": np.log,
       "tanh": np.tanh, "sqrt": np.sqrt, "floor": np.floor, "not": np.logical_not,
       "relu": lambda a: np.maximum(a, 0), "stopgrad": lambda a: a, "win": lambda a: a}
_BIN = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.divide, "pow": np.power,
        "min": np.minimum, "max": np.maximum, "lt": np.less, "le": np.less_equal, "gt": np.greater,
        "ge": np.greater_equal, "eq": np.equal, "ne": np.not_equal, "and": np.logical_and, "or": np.logical_or,
        "atan2": np.arctan2}


class DagEval:
    def __init__(self, tr, G, arrays, tracers=None, nets=None):
        """arrays: key -> ndarray of shape G (fields and '@...' stored adjoints); nets: key -> (weights, biases) as
        lists of ndarrays, for operators with a pointwise network (`mlp` nodes)."""
        self.tr, self.G, self.arrays, self.tracers = tr, tuple(G), arrays, tracers or dict()
        self.nets = nets or dict()
        self.memo = dict()

    def __call__(self, n):
        if n.idx in self.memo:
            return self.memo[n.idx]
        v = self._eval(n)
        self.memo[n.idx] = v
        return v

    def _eval(self, n):
        op, A, G = n.op, n.args, self.G
        nd = len(G)
        if op == "const":
            return n.attr
        if op == "tracer":
            return float(self.tracers[n.attr])
        if op == "aparam":  # Array unknown a[k]: arrays[key] holds the parameter vector
            return float(np.asarray(self.arrays[n.attr[0]]).reshape(-1)[n.attr[1]])
        if op == "read":
            key, shift, loc, _ = n.attr
            return np.roll(self.arrays[key], [-s for s in shift], axis=tuple(range(nd)))  # value at i: u[i + shift]
        if op == "index":
            shape = [1] * nd
            shape[n.attr[0]] = G[n.attr[0]]
            return np.broadcast_to(np.arange(G[n.attr[0]]).reshape(shape), G)
        if op in ("tensor", "rtensor"):
            slot, roll = (n.attr, None) if op == "tensor" else n.attr
            t = self.tr.tensors[slot].detach().cpu().numpy()
            t = t.reshape((1,) * (nd - t.ndim) + t.shape)
            if roll is not None:
                for d, r in enumerate(roll):
                    if r and t.shape[d] > 1:
                        t = np.roll(t, r, axis=d)
            return t
        if op == "cast":
            return np.asarray(self(A[0]), dtype=np.float64)
        if op == "where":
            return np.where(self(A[0]), self(A[1]), self(A[2]))
        if op == "mlp":  # all outputs of the pointwise network (reference core.py:807-862): (n_out,) + broadcast shape
            key, _, layers, activation = n.attr
            weights, biases = self.nets[key]
            act = {"tanh": np.tanh, "relu": lambda a: np.maximum(a, 0), "none": lambda a: a}[activation]
            vals = np.broadcast_arrays(*[np.asarray(self(a), dtype=np.float64) for a in A])
            h = np.stack(vals).reshape(len(vals), -1)
            for k, (w, b) in enumerate(zip(weights, biases)):
                h = np.asarray(w, dtype=np.float64) @ h + np.asarray(b, dtype=np.float64)[:, None]
                if k < len(weights) - 1:
                    h = act(h)
            return h.reshape((h.shape[0],) + vals[0].shape)
        if op == "mlp_out":
            return self(A[0])[n.attr]
        if op in _UN:
            return _UN[op](self(A[0]))
        if op in _BIN:
            return _BIN[op](self(A[0]), self(A[1]))
        raise NotImplementedError(op)
