"""CPU oracle for ARBITRARY operators (TEST INFRASTRUCTURE ONLY, like oracle/odil_np.py).

`oracle/odil_np.py` restates the Poisson workload by hand.  The other workloads of the path (velocity from
tracer, heat, wave ...) are user callbacks `operator(ctx)`: this module restates what the reference does with
such a callback -- build a `Context` whose `field()` pads / rolls / trims the synthesised arrays (reference
src/odil/core.py:865-990), evaluate, loss = sum_k mean(f_k^2) (core.py:1087-1095), gradient by reverse-mode
autodiff (core.py:1098-1107; here torch autograd on CPU in float64) pulled back through the multigrid
synthesis by the NumPy transposes of odil_np (core.py:245-263).

Pinned by tests/test_oracle_golden.py against fixtures the REFERENCE produced for the same operators
(tests/golden/veltracer_*.npz, veltracer3d_*.npz, wave_*.npz ...).  Only tests/ may import this module; the
product never does.
"""

import numpy as np
import torch

from . import odil_np as onp


class Mod:
    """The few `mod` names the example operators call, on torch CPU tensors (reference backend.py:17-110)."""

    float32, float64, int32 = np.float32, np.float64, np.int32

    @staticmethod
    def _t(x, like=None):
        if isinstance(x, torch.Tensor):
            return x
        return torch.as_tensor(np.asarray(x), dtype=None if like is None else like.dtype)

    def cast(self, x, dtype=None):
        td = {np.float32: torch.float32, np.float64: torch.float64}.get(np.dtype(dtype).type if dtype is not None else None)
        x = self._t(x)
        return x.to(td) if td is not None else x

    def where(self, c, a, b):
        ref = a if isinstance(a, torch.Tensor) else (b if isinstance(b, torch.Tensor) else None)
        dt = ref.dtype if ref is not None else torch.float64
        a = a if isinstance(a, torch.Tensor) else torch.as_tensor(a, dtype=dt)
        b = b if isinstance(b, torch.Tensor) else torch.as_tensor(b, dtype=dt)
        return torch.where(self._t(c), a, b)

    def roll(self, x, shift, axis=None):
        if axis is None:
            return torch.roll(x, shift)
        axes = [axis] if isinstance(axis, (int, np.integer)) else list(axis)
        shifts = [int(v) for v in np.broadcast_to(np.asarray(shift), (len(axes),))]
        return torch.roll(x, shifts, axes)

    def stop_gradient(self, x):
        return x.detach()

    def zeros_like(self, x):
        return torch.zeros_like(x)

    def ones_like(self, x):
        return torch.ones_like(x)

    def square(self, x):
        return x * x

    def sigmoid(self, x):
        return 1 / (1 + torch.exp(-x))

    def minimum(self, a, b):
        return torch.minimum(self._t(a, b if isinstance(b, torch.Tensor) else None), self._t(b, a if isinstance(a, torch.Tensor) else None))

    def maximum(self, a, b):
        return torch.maximum(self._t(a, b if isinstance(b, torch.Tensor) else None), self._t(b, a if isinstance(a, torch.Tensor) else None))

    def concatenate(self, xs, axis=0):
        return torch.cat([self._t(x) for x in xs], dim=axis)

    def flatten(self, x):
        return x.reshape(-1)

    def reshape(self, x, shape):
        return x.reshape(shape)

    def mean(self, x):
        return x.mean()

    def sum(self, x):
        return x.sum()

    def clip(self, x, a, b):
        return torch.clamp(x, a, b)

    def numpy(self, x):
        return x.detach().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)

    def is_tensor(self, x):
        return isinstance(x, torch.Tensor)


for _name in ("exp", "log", "sin", "cos", "tanh", "sqrt", "abs", "floor", "relu"):
    setattr(Mod, _name, staticmethod(getattr(torch, _name)))


class Geometry:
    """Grid geometry of a `Domain` (reference core.py:99-215) from its defining numbers."""

    def __init__(self, cshape, dimnames, lower, upper, dtype):
        self.cshape, self.dimnames, self.dtype = tuple(int(c) for c in cshape), list(dimnames), dtype
        self.ndim = len(self.cshape)
        self.lower = (np.ones(self.ndim, dtype=dtype) * lower).astype(dtype)
        self.upper = (np.ones(self.ndim, dtype=dtype) * upper).astype(dtype)

    @classmethod
    def of(cls, domain):
        return cls(domain.cshape, domain.dimnames, domain.lower, domain.upper, domain.dtype)

    def _dims(self, dims):
        return [self.dimnames.index(d) if isinstance(d, str) else int(d) for d in dims] if dims else list(range(self.ndim))

    def _pick(self, res, dims):
        return res[0] if len(dims) == 1 else tuple(res)

    def step(self, *dims):
        return self._pick([(self.upper[i] - self.lower[i]) / self.cshape[i] for i in self._dims(dims)], dims)

    def size(self, *dims, loc=None):
        shape = onp.field_shape(self.cshape, loc or "c" * self.ndim)
        return self._pick([shape[i] for i in self._dims(dims)], dims)

    @staticmethod
    def arrays_from_field(field):
        """core.py:361-374 for the containers an operator may look into (heat's weight regulariser)."""
        return list(field.weights) + list(field.biases) if hasattr(field, "weights") else [field.array]

    def grids(self, loc, kind):
        loc = loc or "c" * self.ndim
        if kind == "points":
            xx = [onp.points_1d(self.lower[d], self.upper[d], self.cshape[d], loc[d], self.dtype) for d in range(self.ndim)]
        else:
            xx = [np.arange(s) for s in onp.field_shape(self.cshape, loc)]
        return [torch.as_tensor(a) for a in np.meshgrid(*xx, indexing="ij")]


def field_access(u, field_loc, shift, loc):
    """`Context.field` on a torch array: zero pad 'c'->'n' at the low end, periodic roll by -shift, drop the
    last entry 'n'->'c' (reference core.py:956-969; the NumPy twin is odil_np.field_access)."""
    ndim = u.dim()
    pad = []
    for d in reversed(range(ndim)):
        pad += [1 if (field_loc[d] == "c" and loc[d] == "n") else 0, 0]
    if any(pad):
        u = torch.nn.functional.pad(u, pad)
    if any(shift):
        u = torch.roll(u, [-int(s) for s in shift], list(range(ndim)))
    for d in range(ndim):
        if field_loc[d] == "n" and loc[d] == "c":
            u = u.narrow(d, 0, u.shape[d] - 1)
    return u


def eval_neural_net(weights, biases, inputs, activation="tanh", frozen=False):
    """Pointwise MLP (reference core.py:807-862): per layer w @ x + b, activation between the layers."""
    act = {"tanh": torch.tanh, "relu": torch.relu, "none": lambda x: x, None: torch.tanh}[activation]
    if frozen:
        weights, biases = [w.detach() for w in weights], [b.detach() for b in biases]
    shape = torch.broadcast_shapes(*[x.shape for x in inputs])
    h = torch.stack([x.expand(shape) for x in inputs], dim=-1)  # (..., ni)
    for k, (w, b) in enumerate(zip(weights, biases)):
        h = torch.einsum("oi,...i->...o", w, h) + b
        if k + 1 < len(weights):
            h = act(h)
    return [h[..., j] for j in range(h.shape[-1])]


class Context:
    class Raw:
        def __init__(self, value):
            self.value = value

    def __init__(self, geom, regular, locs, params, extra, tracers):
        """regular: key -> synthesised torch array (a leaf that requires grad), locs: key -> loc string,
        params: key -> ("array", tensor) or ("net", weights, biases, activation, func_in, func_out)."""
        self.geom, self.regular, self.locs, self.params = geom, regular, locs, params
        self.extra, self.tracers, self.mod, self.dtype = extra, tracers, Mod(), geom.dtype
        self.step, self.size, self.domain = geom.step, geom.size, geom
        from types import SimpleNamespace

        held = dict()
        for key, p in params.items():
            held[key] = SimpleNamespace(array=p[1]) if p[0] == "array" else SimpleNamespace(weights=p[1], biases=p[2])
        for key, u in regular.items():
            held[key] = SimpleNamespace(array=u, loc=locs[key])
        self.state = SimpleNamespace(fields=held)

    def cast(self, value, dtype=None):
        return self.mod.cast(value, dtype or self.dtype)

    def indices(self, *dims, loc=None):
        return self.geom._pick([self.geom.grids(loc, "indices")[i] for i in self.geom._dims(dims)], dims)

    def points(self, *dims, loc=None):
        return self.geom._pick([self.geom.grids(loc, "points")[i] for i in self.geom._dims(dims)], dims)

    def field(self, key, *shift, loc=None, frozen=False):
        if key in self.params:  # an Array unknown (core.py:919-926)
            t = self.params[key][1]
            return t.detach() if frozen else t
        shift = tuple(int(s) for s in shift) or (0,) * self.geom.ndim
        u = field_access(self.regular[key], self.locs[key], shift, loc or self.locs[key])
        return u.detach() if frozen else u

    def neural_net(self, key, frozen=False):
        _, weights, biases, activation, func_in, func_out = self.params[key]

        def res(*inputs):
            if func_in is not None:
                inputs = func_in(*inputs)
            outputs = eval_neural_net(weights, biases, list(inputs), activation, frozen)
            return func_out(*outputs) if func_out is not None else outputs

        return res


def split_outputs(ff):
    """core.py:1087-1092: outputs are arrays, (name, array) tuples or Raw values."""
    names, values = [], []
    for i, f in enumerate(ff):
        if isinstance(f, tuple):
            names.append(f[0])
            values.append(f[1])
        else:
            names.append("")
            values.append(f)
    return names, values


def eval_loss_grad(operator, geom, fields, extra=None, tracers=None, mg_axes=None):
    """loss, grads, terms, values of a problem given as plain data.

    fields: dict key -> one of
      dict(kind="field", loc=..., array=ndarray)
      dict(kind="mg", loc=..., terms=[ndarray fine->coarse], factors=None)
      dict(kind="array", array=ndarray)
      dict(kind="net", weights=[...], biases=[...], activation=..., func_in=None, func_out=None)
    grads come in `Domain.arrays_from_state` order (core.py:361-374): per field its array / level arrays /
    weights then biases."""
    tdt = torch.float64 if np.dtype(geom.dtype) == np.float64 else torch.float32
    regular, locs, params, leaves = dict(), dict(), dict(), []
    for key, f in fields.items():
        if f["kind"] in ("field", "mg"):
            u = f["array"] if f["kind"] == "field" else onp.multigrid_to_regular(
                [np.asarray(t) for t in f["terms"]], f["loc"], f.get("factors"), mg_axes)
            t = torch.tensor(np.asarray(u), dtype=tdt, requires_grad=True)
            regular[key], locs[key] = t, f["loc"]
            leaves.append(t)
        elif f["kind"] == "array":
            t = torch.tensor(np.asarray(f["array"]), dtype=tdt, requires_grad=True)
            params[key] = ("array", t)
            leaves.append(t)
        else:
            ws = [torch.tensor(np.asarray(w), dtype=tdt, requires_grad=True) for w in f["weights"]]
            bs = [torch.tensor(np.asarray(b), dtype=tdt, requires_grad=True) for b in f["biases"]]
            params[key] = ("net", ws, bs, f.get("activation"), f.get("func_in"), f.get("func_out"))
            leaves += ws + bs
    ctx = Context(geom, regular, locs, params, extra, tracers if tracers is not None else dict(epoch=0))
    names, values = split_outputs(operator(ctx))
    terms = [v.value.mean() if isinstance(v, Context.Raw) else (v * v).mean() for v in values]
    loss = sum(terms)
    gl = torch.autograd.grad(loss, leaves, allow_unused=True)
    gl = [torch.zeros_like(t) if g is None else g for g, t in zip(gl, leaves)]
    grads, k = [], 0
    for key, f in fields.items():
        if f["kind"] == "field":
            grads.append(gl[k].numpy())
            k += 1
        elif f["kind"] == "mg":
            shapes = [np.asarray(t).shape for t in f["terms"]]
            grads += onp.multigrid_to_regular_adj(gl[k].numpy(), shapes, f["loc"], f.get("factors"), mg_axes)
            k += 1
        elif f["kind"] == "array":
            grads.append(gl[k].numpy())
            k += 1
        else:
            n = len(f["weights"]) + len(f["biases"])
            grads += [g.numpy() for g in gl[k:k + n]]
            k += n
    vals = [(v.value if isinstance(v, Context.Raw) else v).detach().numpy() for v in values]
    return float(loss.detach()), grads, [float(t.detach()) for t in terms], names, vals


def fields_of_state(domain, state):
    """The plain-data description `eval_loss_grad` takes, from an odil_amd Domain / State (any device)."""
    from odil_amd.core import Array, Field, MultigridField, NeuralNet

    npy = lambda t: t.detach().cpu().numpy()
    out = dict()
    for key, f in state.fields.items():
        if isinstance(f, Field):
            out[key] = dict(kind="field", loc=f.loc, array=npy(f.array))
        elif isinstance(f, MultigridField):
            out[key] = dict(kind="mg", loc=f.loc, terms=[npy(t.array) for t in f.terms],
                            factors=f.factors or domain.mg_factors)
        elif isinstance(f, Array):
            out[key] = dict(kind="array", array=npy(f.array))
        elif isinstance(f, NeuralNet):
            out[key] = dict(kind="net", weights=[npy(w) for w in f.weights], biases=[npy(b) for b in f.biases],
                            activation=f.activation, func_in=f.func_in, func_out=f.func_out)
    return out
