"""Domain / State / Context / Problem: the operator API of the reference
(reference src/odil/core.py) on top of the HIP kernels.

User `operator(ctx)` callbacks written for the reference run unchanged: they see the
same `ctx.field / ctx.neural_net / ctx.step / ctx.indices / ...` and the same `mod`
names.  What differs is who does the work:

  * multigrid synthesis `u = sum_l P^l (f_l w_l)` (core.py:245-263)  -> odil_mg_synth,
    cotangent -> odil_mg_synth_adj                                   (one launch chain)
  * `ctx.field(key, *shift, loc)` pad + roll + trim (core.py:955-969) -> odil_field_gather,
    cotangent -> odil_field_scatter
  * loss terms mean(f^2) (core.py:1093)                              -> odil_mean_reduce,
    cotangent -> odil_scale
  * interp_to_finer / restrict_to_coarser (core.py:606-755)          -> odil_interp_add / odil_restrict
    (cotangents odil_interp_adj / odil_restrict_adj)
  * affine stencil operators (Poisson) are recognised from their Jacobian coefficients and
    evaluated by the fused residual / adjoint kernels (see fused.py)
  * Jacobian rows (core.py:1144-1171) stay on the device as coefficient arrays
    (`LinearizedOperator`), applied matrix-free for the normal equations.

Reverse-mode differentiation of the user's pointwise arithmetic is torch.autograd
bookkeeping over device tensors; there is no CPU path.
"""

import math
import os
import pickle

import numpy as np
import torch
from torch.autograd.function import once_differentiable

from . import ops
from .backend import ModRocm, numpy_dtype, torch_dtype


def assert_equal(first, second, msg=""):
    if not (first == second):
        raise ValueError("Expected equal '{:}' and '{:}'{}".format(first, second, msg))


# ======================================================================================
# State containers (reference core.py:506-603): plain attribute holders.
# ======================================================================================
class Field:
    def __init__(self, array=None, loc=None, cshape=None):
        self.array = array  # data array
        self.loc = loc  # one of 'c' / 'n' per direction
        self.cshape = cshape  # grid size in cells

    def __repr__(self):
        return "odil.Field({}, loc='{}', cshape={:})".format(repr(self.array), self.loc, self.cshape)

    __str__ = __repr__


class MultigridField:
    def __init__(self, terms=None, loc=None, factors=None, axes=None, method=None):
        self.terms = terms  # list of Field, fine -> coarse
        self.loc = loc
        self.factors = factors  # factor of each term, defaults to 1
        self.axes = axes  # per axis: decompose it or not
        self.method = method  # 'stack' | 'conv': same values, one HIP kernel here


class NeuralNet:
    def __init__(self, weights=None, biases=None, func_in=None, func_out=None, activation=None):
        self.weights = weights  # list of (no, ni) matrices
        self.biases = biases  # list of (no,) vectors
        self.func_in = func_in
        self.func_out = func_out
        self.activation = activation or "tanh"


class Array:
    def __init__(self, array=None, shape=None):
        self.array = array
        self.shape = shape

    def __repr__(self):
        return "odil.Array({}, shape={:})".format(repr(self.array), self.shape)

    __str__ = __repr__


class State:
    def __init__(self, fields=None, initialized=False):
        self.fields = fields if fields is not None else dict()
        self.initialized = initialized


# ======================================================================================
# HIP kernels as differentiable building blocks
# ======================================================================================
class _MgSynthFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, loc, factors, *terms):
        ctx.loc, ctx.factors = loc, factors
        ctx.shapes = [tuple(t.shape) for t in terms]
        return ops.mg_synth([t.contiguous() for t in terms], loc, factors=factors)

    @staticmethod
    @once_differentiable  # (raw HIP launches: a second derivative through here must fail loudly, not vanish)
    def backward(ctx, gu):
        grads = ops.mg_synth_adj(gu.contiguous(), ctx.shapes, ctx.loc, factors=ctx.factors)
        return (None, None) + tuple(grads)


class _FieldAccessFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, field_loc, shift, loc):
        ctx.meta = (tuple(src.shape), field_loc, shift, loc)
        return ops.field_gather(src.contiguous(), field_loc, shift, loc)

    @staticmethod
    @once_differentiable  # (raw HIP launches: a second derivative through here must fail loudly, not vanish)
    def backward(ctx, g):
        shape, field_loc, shift, loc = ctx.meta
        return ops.field_scatter(g.contiguous(), shape, field_loc, shift, loc), None, None, None


class _MeanFn(torch.autograd.Function):
    """mean(x^2) (square=True) or mean(x): deterministic two-stage reduction."""

    @staticmethod
    def forward(ctx, x, square):
        x = x.contiguous()
        ctx.square = square
        ctx.save_for_backward(x)
        return ops.mean_reduce(x, square=square)

    @staticmethod
    @once_differentiable  # (raw HIP launches: a second derivative through here must fail loudly, not vanish)
    def backward(ctx, gout):
        (x,) = ctx.saved_tensors
        n = x.numel()
        gout = gout.contiguous().to(x.dtype)
        if ctx.square:
            return ops.scale(x, 2.0 / n, adev=gout.reshape(1)), None
        return ops.scale(torch.ones_like(x), 1.0 / n, adev=gout.reshape(1)), None


class _InterpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, loc):
        ctx.loc, ctx.shape = loc, tuple(u.shape)
        return ops.interp_add(u.contiguous(), loc)

    @staticmethod
    @once_differentiable  # (raw HIP launches: a second derivative through here must fail loudly, not vanish)
    def backward(ctx, g):
        return ops.interp_adj(g.contiguous(), ctx.loc, ctx.shape), None


class _RestrictFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, loc):
        ctx.loc, ctx.shape = loc, tuple(u.shape)
        return ops.restrict_to_coarser(u.contiguous(), loc)

    @staticmethod
    @once_differentiable  # (raw HIP launches: a second derivative through here must fail loudly, not vanish)
    def backward(ctx, g):
        return ops.restrict_adj(g.contiguous(), ctx.loc, ctx.shape), None


def _as_tensor(u, mod):
    if isinstance(u, torch.Tensor):
        return u
    mod = mod or _default_mod()
    return mod.array(u)


def _default_mod():
    from . import runtime

    return runtime.get_mod()


def interp_to_finer(u, loc=None, method=None, mod=None, depth=1):
    """Linear prolongation (reference core.py:606-700).  `method` ('conv' | 'stack') selects
    between two formulations of the SAME values in the reference; here both names run the
    one HIP kernel, which reproduces the 'stack' summation order."""
    method = method or "stack"
    if method not in ["conv", "stack"]:
        raise ValueError("Unknown method='{}'".format(method))
    u = _as_tensor(u, mod)
    loc = loc or "c" * u.dim()
    assert_equal(len(loc), u.dim())
    for l in loc:
        assert l in "cn.", "Invalid loc={}".format(loc)
    for _ in range(depth):
        u = _InterpFn.apply(u, loc)
    return u


def restrict_to_coarser(u, loc=None, method=None, mod=None, depth=1):
    """Full-weighting restriction (reference core.py:703-755)."""
    method = method or "conv"
    if method not in ["conv"]:
        raise ValueError("Unknown method='{}'".format(method))
    u = _as_tensor(u, mod)
    loc = loc or "c" * u.dim()
    assert_equal(len(loc), u.dim())
    for l in loc:
        assert l in "cn.", "Invalid loc={}".format(loc)
    for _ in range(depth):
        u = _RestrictFn.apply(u, loc)
    return u


def check_multigrid_cshapes(cshapes, axes=None):
    """Every level halves the one before on every decomposed axis; a ValueError naming the offending pair otherwise
    (the check of reference core.py:758-776)."""
    cshapes = [tuple(s) for s in cshapes]
    if not cshapes:
        return
    axes = [True] * len(cshapes[0]) if not axes else list(axes)
    assert_equal(len(axes), len(cshapes[0]))
    where = " with cshapes={:}".format(cshapes)
    for fine, coarse in zip(cshapes, cshapes[1:]):
        for halved, nf, nc in zip(axes, fine, coarse):
            if halved:
                assert_equal(nf, 2 * nc, where)


# ======================================================================================
# Neural networks (reference core.py:779-862) -- only as used inside stencils
# ======================================================================================
# uniform(-a, a) with a^2 = numerator / fan; fan counted over the inputs, or inputs + outputs for 'glorot'
# (the four schemes of reference core.py:779-803)
_NET_INIT = {"legacy": (1.0, False), "glorot": (6.0, True), "lecun": (3.0, False), "he": (6.0, False)}


def make_neural_net(layers, dtype, mod, initializer="lecun", func_in=None, func_out=None, activation=None):
    """Dense network with `layers[k]` units per layer: uniform weights by `initializer`, zero biases (the API of
    reference core.py:779-803)."""
    if initializer not in _NET_INIT:
        raise ValueError("Unknown initializer=" + initializer)
    numerator, both = _NET_INIT[initializer]
    weights, biases = [], []
    for fan_in, fan_out in zip(layers, layers[1:]):
        bound = np.sqrt(numerator / (fan_in + fan_out if both else fan_in))
        weights.append(mod.random.uniform(shape=(fan_out, fan_in), minval=-bound, maxval=bound, dtype=dtype))
        biases.append(mod.zeros(fan_out, dtype=dtype))
    return NeuralNet(weights, biases, func_in=func_in, func_out=func_out, activation=activation)


def eval_neural_net(net, inputs, mod, frozen=False):
    """The network applied pointwise to grid arrays (what reference core.py:807-862 computes): every grid point is a
    column, a layer is one batched product  h <- act(W h + b)  with the activation on all layers but the last.  This is
    the GENERIC path; a traced operator inlines the network into its generated kernel (stencil_codegen)."""
    layers = list(zip(net.weights, net.biases))
    if len(net.weights) != len(net.biases):
        assert_equal(len(net.weights), len(net.biases), "Weights and biases do not match")
    assert_equal(layers[0][0].shape[1], len(inputs), "Weights and inputs do not match")
    if net.activation not in ("tanh", "relu", "none"):
        raise KeyError(net.activation)
    act = None if net.activation == "none" else getattr(mod, net.activation)
    if net.func_in is not None:
        inputs = net.func_in(*inputs)
    wdtype = layers[0][0].dtype
    # every grid point a column vector: (grid..., units, 1); a layer is one batched product W h + b
    h = torch.stack(torch.broadcast_tensors(*[mod.cast(v, wdtype) for v in inputs]), dim=-1).unsqueeze(-1)
    for k, (w, b) in enumerate(layers):
        assert_equal(w.shape[0], b.shape[0])
        if frozen:
            w, b = mod.stop_gradient(w), mod.stop_gradient(b)
        h = torch.matmul(w, h) + b.unsqueeze(-1)
        if act is not None and k < len(layers) - 1:
            h = act(h)
    outputs = list(h.squeeze(-1).unbind(-1))
    if net.func_out is not None:
        outputs = net.func_out(*outputs)
    return outputs


# ======================================================================================
# Domain (reference core.py:11-504)
# ======================================================================================
def _grid_or_array_field(fields, key, what):
    """state.fields[key] if it can be read through `field()`; the reference's TypeError otherwise."""
    found = fields[key]
    if isinstance(found, (Field, MultigridField, Array)):
        return found
    raise TypeError("Expected Field or MultigridField, got type {} for {}'{}'".format(type(found).__name__, what, key))


def _net_field(fields, key):
    found = fields[key]
    if isinstance(found, NeuralNet):
        return found
    raise TypeError("Expected NeuralNet, got type {} for key='{}'".format(type(found).__name__, key))


def _no_shift_for_arrays(shift):
    if len(shift):
        raise RuntimeError("Array requires an empty shift")


def _plan_levels(cshape, axes, wanted):
    """(axes, number of levels, cell shapes fine -> coarse) of a multigrid decomposition (what reference core.py:57-79
    sets up): every decomposed axis halves from level to level, the depth is the smallest round(log2 n) among them --
    so a power-of-two axis ends at 2 cells -- optionally capped by `wanted`; undecomposed axes keep their extent."""
    ndim = len(cshape)
    axes = list(axes) if axes else [True] * ndim
    depths = [int(round(np.log2(n))) for n, on in zip(cshape, axes) if on]
    depth = min(depths) if depths else max(cshape)
    if wanted is not None:
        if wanted < 1:
            raise AssertionError("mg_nlvl must be at least 1")
        depth = min(depth, wanted)
    shapes = [tuple((int(n) >> level) if on else int(n) for n, on in zip(cshape, axes)) for level in range(depth)]
    check_multigrid_cshapes(shapes, axes)
    return axes, depth, shapes


class Domain:
    def __init__(self, cshape, dimnames=None, lower=0.0, upper=1.0, dtype=None, multigrid=False,
                 mg_convert_all=True, mg_nlvl=None, mg_factors=None, mg_axes=None, mg_interp=None, mod=None):
        ndim = len(cshape)
        dimnames = dimnames or ["x", "y", "z"][:ndim]
        if mod is None:
            mod = _default_mod()
        assert_equal(len(dimnames), ndim, f"with dimnames={dimnames}")
        self.ndim = ndim
        self.cshape = tuple(int(c) for c in cshape)
        self.dimnames = dimnames
        if dtype is None:
            from . import runtime

            dtype = runtime.dtype
        dtype = numpy_dtype(dtype) if isinstance(dtype, torch.dtype) else np.dtype(dtype)
        self.dtype = dtype.type
        self.lower = (np.ones(ndim, dtype=dtype) * lower).astype(dtype)
        self.upper = (np.ones(ndim, dtype=dtype) * upper).astype(dtype)
        self.mod = mod
        self.multigrid = multigrid
        if multigrid:
            plan = _plan_levels(self.cshape, mg_axes, mg_nlvl)
            self.mg_axes, self.mg_nlvl, self.mg_cshapes = plan
            self.mg_factors, self.mg_interp, self.mg_convert_all = mg_factors, mg_interp, mg_convert_all

    # ---- geometry: host-side NumPy, results handed to the device by mod.meshgrid ----
    @staticmethod
    def _names_to_indices(dims, dimnames):
        res = dims if dims is not None and len(dims) else range(len(dimnames))
        return tuple(dimnames.index(i) if isinstance(i, str) else i for i in res)

    def cast(self, value, dtype=None):
        return self.mod.cast(value, dtype or self.dtype)

    def _points_1d(self, d, loc):
        if loc == "c":
            x = np.linspace(self.lower[d], self.upper[d], self.cshape[d], endpoint=False, dtype=self.dtype)
            if len(x) > 1:
                x += (x[1] - x[0]) * 0.5
            return x
        elif loc == "n":
            return np.linspace(self.lower[d], self.upper[d], self.cshape[d] + 1, dtype=self.dtype)
        raise ValueError("Unknown loc=" + loc)

    def points_1d(self, *dims, loc=None):
        loc = loc or "c" * self.ndim
        idims = self._names_to_indices(dims, self.dimnames)
        res = [self._points_1d(idim, c) for idim, c in zip(idims, loc)]
        return res[0] if len(dims) == 1 else res

    def _cached_grid(self, kind, loc, make):
        """Coordinate / index grids are constants of the domain: built once on the host, kept on the
        device (also keeps host->device copies out of hipGraph capture)."""
        cache = self.__dict__.setdefault("_grid_cache", dict())
        key = (kind, loc)
        if key not in cache:
            cache[key] = self.mod.meshgrid(*make(), indexing="ij")
        return cache[key]

    def points(self, *dims, loc=None):
        loc = loc or "c" * self.ndim
        assert_equal(len(loc), self.ndim, f"with loc={loc}")
        dimnames = [v for v, c in zip(self.dimnames, loc) if c != "."]
        idims = self._names_to_indices(dims, dimnames)
        data = self._cached_grid(
            "points", loc, lambda: [self._points_1d(d, loc[d]) for d in range(self.ndim) if loc[d] != "."])
        res = tuple(data[i] for i in idims)
        return res[0] if len(dims) == 1 else res

    def _indices_1d(self, d, loc):
        if loc == "c":
            return np.arange(self.cshape[d], dtype=int)
        elif loc == "n":
            return np.arange(self.cshape[d] + 1, dtype=int)
        raise ValueError("Unknown loc=" + loc)

    def indices(self, *dims, loc=None):
        loc = loc or "c" * self.ndim
        dimnames = [v for v, c in zip(self.dimnames, loc) if c in "cn"]
        idims = self._names_to_indices(dims, dimnames)
        data = self._cached_grid(
            "indices", loc, lambda: [self._indices_1d(d, loc[d]) for d in range(self.ndim) if loc[d] in "cn"])
        res = tuple(data[i] for i in idims)
        return res[0] if len(dims) == 1 else res

    @staticmethod
    def _get_field_shape(cshape, loc=None):
        loc = loc or "c" * len(cshape)
        assert all(c in "cn" for c in loc)
        return tuple(int(s) + 1 if c == "n" else int(s) for s, c in zip(cshape, loc))

    def get_field_shape(self, loc=None):
        return self._get_field_shape(self.cshape, loc=loc)

    def size(self, *dims, loc=None):
        loc = loc or "c" * self.ndim
        assert_equal(len(loc), self.ndim, f"with loc={loc}")
        idims = self._names_to_indices(dims, self.dimnames)
        res = [self.cshape[i] + (1 if loc[i] == "n" else 0) for i in idims]
        for i in idims:
            if loc[i] not in "cn":
                raise ValueError("Unknown loc=" + loc[i])
        return res[0] if len(dims) == 1 else res

    def step_by_dim(self, i):
        return (self.upper[i] - self.lower[i]) / self.cshape[i]

    def step(self, *dims):
        idims = self._names_to_indices(dims, self.dimnames)
        res = tuple(self.step_by_dim(i) for i in idims)
        return res[0] if len(dims) == 1 else res

    # ---- multigrid decomposition -------------------------------------------------------
    def _mg_loc(self, mgfield):
        axes = mgfield.axes or self.mg_axes
        return "".join(l if ax else "." for l, ax in zip(mgfield.loc, axes))

    def multigrid_to_regular(self, mgfield):
        """u = sum_l P^l (f_l w_l) (reference core.py:245-263) as one HIP launch chain."""
        factors = mgfield.factors or self.mg_factors or [1] * len(mgfield.terms)
        axes = mgfield.axes or self.mg_axes
        assert_equal(len(factors), len(mgfield.terms))
        assert_equal(len(axes), len(mgfield.terms[0].cshape))
        arrays = [term.array for term in mgfield.terms]
        if all(float(f) == 1.0 for f in factors):
            factors = None
        else:
            factors = tuple(float(f) for f in factors)
        res = _MgSynthFn.apply(self._mg_loc(mgfield), factors, *arrays)
        return Field(res, loc=mgfield.loc)

    def get_regular_array(self, field):
        if isinstance(field, (Field, Array)):
            return field.array
        elif isinstance(field, MultigridField):
            return self.multigrid_to_regular(field).array
        raise TypeError("Expected Field or MultigridField got {}".format(type(field).__name__))

    def regular_to_multigrid(self, field, cshapes=None, factors=None, method=None):
        """A regular field as a multigrid one that synthesises to the same values: the finest term carries u / f_0, every
        coarser term starts at zero (reference core.py:276-297)."""
        if isinstance(field, (MultigridField, NeuralNet)):
            raise TypeError("Expected Field or ndarray, got type {}".format(type(field).__name__))
        fine = self.init_field(field)
        shapes = list(cshapes or self.mg_cshapes)
        scales = factors or self.mg_factors or [1] * len(shapes)
        assert_equal(len(shapes), len(scales))
        levels = [Field(fine.array / scales[0], loc=fine.loc, cshape=fine.cshape)]
        levels += [Field(self.mod.zeros(self._get_field_shape(cs, loc=fine.loc), dtype=self.dtype), loc=fine.loc, cshape=cs)
                   for cs in shapes[1:]]
        return MultigridField(terms=levels, loc=fine.loc, factors=scales, method=method or self.mg_interp)

    # ---- state initialisation (reference core.py:299-359) -------------------------------
    def _own(self, values, shape=None):
        """`values` (or zeros of `shape` when None) as an unknown of this domain's dtype on its device."""
        if values is None:
            values = self.mod.zeros(shape, dtype=self.dtype)
        return self.mod.variable(values, dtype=self.dtype)

    def init_field(self, field):
        """Whatever a user may put into `State.fields` -- None, a bare array, a list of numbers, a `Field` without values,
        a multigrid field, a network, an `Array` -- as an initialised object of the same kind."""
        mod = self.mod
        if field is None:
            field = Field(None, loc="c" * self.ndim, cshape=self.cshape)
        elif isinstance(field, list):
            values = mod.cast(mod.array(np.array(field)), self.dtype)
            field = Array(values, shape=tuple(values.shape))
        elif isinstance(field, np.ndarray) or mod.is_tensor(field):
            field = Field(field, loc="c" * len(field.shape), cshape=tuple(field.shape))
        if isinstance(field, Field):
            cshape = field.cshape or self.cshape
            loc = field.loc or "c" * len(cshape)
            assert_equal(len(loc), len(cshape))
            expect = self._get_field_shape(cshape, loc=loc)
            values = self._own(field.array, expect)
            assert_equal(tuple(values.shape), expect)
            return Field(values, loc=loc, cshape=cshape)
        if isinstance(field, MultigridField):
            return MultigridField([self.init_field(t) for t in field.terms], loc=field.loc, factors=field.factors,
                                  axes=field.axes, method=field.method)
        if isinstance(field, NeuralNet):
            return NeuralNet([self._own(w) for w in field.weights], [self._own(b) for b in field.biases],
                             func_in=field.func_in, func_out=field.func_out, activation=field.activation)
        if isinstance(field, Array):
            return Array(self._own(field.array, field.shape), field.shape)
        raise TypeError("Unknown field type '{}'".format(type(field).__name__))

    def init_state(self, state):
        decompose = self.multigrid and self.mg_convert_all
        ready = dict()
        for key, raw in state.fields.items():
            field = self.init_field(raw)
            if decompose and isinstance(field, Field):
                field = self.regular_to_multigrid(raw)
            ready[key] = field
        return State(fields=ready, initialized=True)

    # ---- flattening: defines the unknown-vector layout (reference core.py:361-469) -------
    # A field is a sequence of SLOTS (container, attribute or index), in the order that defines the packed vector:
    # a field / Array: its array; a multigrid field: its terms fine -> coarse; a network: weights, then biases.
    @staticmethod
    def _slots(field):
        if isinstance(field, (Field, Array)):
            return [(field, "array")]
        if isinstance(field, MultigridField):
            return [(term, "array") for term in field.terms]
        if isinstance(field, NeuralNet):
            return [(field.weights, i) for i in range(len(field.weights))] + [(field.biases, i) for i in range(len(field.biases))]
        raise TypeError("Unknown field type '{}'".format(type(field).__name__))

    def arrays_from_field(self, field):
        return [getattr(box, at) if isinstance(at, str) else box[at] for box, at in self._slots(field)]

    def arrays_from_state(self, state):
        return [a for field in state.fields.values() for a in self.arrays_from_field(field)]

    @staticmethod
    def arrays_to_field(arrays, field):
        """Puts the leading entries of `arrays` into the field's slots; returns how many it took."""
        slots = Domain._slots(field)
        for (box, at), value in zip(slots, arrays):
            if isinstance(at, str):
                setattr(box, at, value)
            else:
                box[at] = value
        return len(slots)

    @staticmethod
    def arrays_to_state(arrays, state):
        taken = 0
        for field in state.fields.values():
            taken += Domain.arrays_to_field(arrays[taken:], field)
        return taken

    def pack_field(self, field):
        mod = self.mod
        return mod.concatenate([mod.flatten(f) for f in self.arrays_from_field(field)], axis=0)

    def pack_state(self, state):
        mod = self.mod
        return mod.concatenate([mod.flatten(f) for f in self.arrays_from_state(state)], axis=0)

    def _unpack(self, packed, arrays):
        mod = self.mod
        sizes = [math.prod(a.shape) for a in arrays]
        split = mod.split_by_sizes(mod.cast(packed, self.dtype)[: sum(sizes)], sizes)
        return [mod.reshape(s, a.shape) for s, a in zip(split, arrays)], sum(sizes)

    def unpack_field(self, packed, field):
        arrays, n = self._unpack(packed, self.arrays_from_field(field))
        self.arrays_to_field(arrays, field)
        return n

    def unpack_state(self, packed, state):
        arrays, n = self._unpack(packed, self.arrays_from_state(state))
        self.arrays_to_state(arrays, state)
        return n

    def make_neural_net(self, layers, initializer="lecun", func_in=None, func_out=None, activation=None):
        return make_neural_net(layers, self.dtype, self.mod, initializer, func_in, func_out, activation)

    # ---- post-processing accessors (reference core.py:474-499) ---------------------------
    def field(self, state, key, *shift):
        """The (synthesised, optionally rolled) array of a field outside any differentiation: what callbacks plot."""
        found = _grid_or_array_field(state.fields, key, "field ")
        if isinstance(found, Array):
            _no_shift_for_arrays(shift)
            return found.array
        offsets = tuple(int(s) for s in shift) if shift else (0,) * self.ndim
        if len(offsets) != self.ndim:
            raise RuntimeError("Expected {} shift components, got shift={}".format(self.ndim, shift))
        with torch.no_grad():
            values = self.get_regular_array(found)
            if any(offsets):
                values = ops.field_gather(values.contiguous(), found.loc, offsets, found.loc)
        return values

    def neural_net(self, state, key):
        """Callable evaluating the network `key` of the state without recording gradients."""
        net = _net_field(state.fields, key)

        def evaluate(*inputs):
            with torch.no_grad():
                return eval_neural_net(net, inputs, self.mod)

        return evaluate

    def get_context(self, state, extra=None, tracers=None):
        return Context(self, state, extra=extra, tracers=tracers)


# ======================================================================================
# Context: stencil access (reference core.py:865-990)
# ======================================================================================
class Context:
    class Raw:
        def __init__(self, value):
            self.value = value

    def __init__(self, domain, state, watch_func=None, extra=None, tracers=None, distinct_shift=False):
        self.domain = domain
        self.state = state
        self.watch_func = watch_func or (lambda _: None)
        self.extra = extra
        self._tracers = tracers
        self.tracers_accessed = False
        self.dtype = domain.dtype
        self.mod = domain.mod
        self.distinct_shift = distinct_shift
        self.desc_to_array = dict()  # (key, shift, loc) -> array
        self.key_to_array_jac = dict()  # unknowns that need a dense Jacobian
        self.step = domain.step
        self.size = domain.size
        self.indices = domain.indices
        self.points = domain.points

    @property
    def tracers(self):
        self.tracers_accessed = True
        return self._tracers

    def cast(self, value, dtype=None):
        return self.mod.cast(value, dtype or self.dtype)

    def field(self, key, *shift, loc=None, frozen=False):
        domain = self.domain
        mod = domain.mod
        field = _grid_or_array_field(self.state.fields, key, "key=")
        if isinstance(field, Array):  # a few scalars: no stencil access, dense Jacobian columns (core.py:919-926)
            _no_shift_for_arrays(shift)
            self.watch_func(field.array)
            self.key_to_array_jac[(key, None, None)] = field.array
            return mod.stop_gradient(field.array) if frozen else field.array
        shift_src = (0,) * domain.ndim
        shift = tuple(int(s) for s in shift) or shift_src
        loc = loc or field.loc
        if len(shift) != domain.ndim:
            raise RuntimeError("Expected {} shift components, got shift={}".format(domain.ndim, shift))
        desc = (key, shift, loc)
        desc_src = (key, shift_src, field.loc)
        if desc in self.desc_to_array:
            array = self.desc_to_array[desc]
        else:
            if desc_src in self.desc_to_array:
                array_src = self.desc_to_array[desc_src]
            else:
                array_src = domain.get_regular_array(field)  # multigrid synthesis happens once per key
                if self.distinct_shift:
                    array_src = self._symbol(array_src)
                self.desc_to_array[desc_src] = array_src
            if desc == desc_src:
                array = array_src
            else:
                src = array_src.detach() if self.distinct_shift else array_src
                # pad ('c'->'n'), periodic roll by -shift, trim ('n'->'c'): one gather kernel
                array = _FieldAccessFn.apply(src, field.loc, shift, loc)
                if self.distinct_shift:
                    array = self._symbol(array)
                self.desc_to_array[desc] = array
        if frozen:
            array = mod.stop_gradient(array)
        return array

    @staticmethod
    def _symbol(array):
        """An independent differentiation variable holding `array` (distinct_shift mode,
        reference core.py:950-953, :970-971)."""
        return array.detach().clone().requires_grad_(True)

    def neural_net(self, key, frozen=False):
        """Callable applying the network `key` pointwise; its weights and biases are watched for the gradient
        and, when the Jacobian is assembled, registered as dense columns."""
        net = _net_field(self.state.fields, key)
        parameters = self.domain.arrays_from_field(net)
        self.watch_func(parameters)
        if self.distinct_shift:
            self.key_to_array_jac[(key, None, None)] = parameters
        return lambda *inputs: eval_neural_net(net, inputs, self.mod, frozen=frozen)


# ======================================================================================
# Linearised operator kept on the device (reference core.py:1113-1217)
# ======================================================================================
class LinearizedOperator:
    """M of `Problem.linearize`, as the pieces ODIL assembles it from: per output rows, per
    (key, shift, loc) a coefficient array (one CSR entry per row, core.py:1144-1171) and per
    Array / NeuralNet unknown a dense block (core.py:1189-1203).  Applies M and M^T with the
    HIP kernels; `to_scipy()` exports the same CSR matrix the reference returns."""

    def __init__(self, domain, state):
        self.domain = domain
        self.key_to_offset, self.key_to_size, self.key_to_field = dict(), dict(), dict()
        offset = 0
        for key, field in state.fields.items():
            size = sum(math.prod(a.shape) for a in domain.arrays_from_field(field))
            self.key_to_offset[key], self.key_to_size[key], self.key_to_field[key] = offset, size, field
            offset += size
        self.ncols = offset
        self.blocks = []  # (row_offset, nrows, kind, key, payload)
        self.nrows = 0
        self.dtype = torch_dtype(domain.dtype)
        self.device = domain.mod.device

    def add_output(self, value_shape, grad):
        nrows = math.prod(value_shape)
        row0 = self.nrows
        for (key, shift, loc), garray in grad.items():
            if garray is None:
                continue
            if isinstance(garray, list) and all(a is None for a in garray):
                continue
            field = self.key_to_field[key]
            if shift is None or len(value_shape) < len(shift):
                if isinstance(garray, list):
                    garray = torch.cat([a.reshape(nrows, -1) for a in garray], dim=1)
                dense = garray.reshape(nrows, -1).contiguous()
                if dense.shape[1] != self.key_to_size[key]:
                    raise ValueError("linearize: the gradient of an output of shape {} with respect to '{}' has {} columns "
                                     "for {} unknowns".format(tuple(value_shape), key, dense.shape[1], self.key_to_size[key]))
                self.blocks.append((row0, nrows, "dense", key, dense))
            else:
                if not isinstance(field, Field):
                    raise TypeError("Expected Field, got type {} for key='{}'".format(type(field).__name__, key))
                if garray.numel() != nrows:
                    # one matrix entry per row and (key, shift, loc), as reference core.py:1144-1171: an operator that shifts
                    # or slices the arrays ITSELF (mod.roll, fu[1:]) instead of reading ctx.field(key, *shift) has no such form
                    raise ValueError("linearize: output of shape {} is not pointwise in ctx.field('{}', shift={}, loc='{}') of "
                                     "shape {}; Newton needs operators written with ctx.field shifts (reference "
                                     "core.py:1144-1171)".format(tuple(value_shape), key, tuple(shift), loc, tuple(garray.shape)))
                self.blocks.append((row0, nrows, "stencil", key, (garray.contiguous(), tuple(shift), loc,
                                                                  tuple(value_shape))))
        self.nrows += nrows

    @property
    def shape(self):
        return (self.nrows, self.ncols)

    def promoted(self):
        """The same operator with float64 coefficient arrays (linsolver.solve: exact routes of float32 problems)."""
        import copy

        wide = copy.copy(self)
        wide.dtype, wide.source_dtype = torch.float64, self.dtype
        wide.blocks = [(row0, nrows, kind, key, payload.double() if kind == "dense" else (payload[0].double(),) + tuple(payload[1:]))
                       for row0, nrows, kind, key, payload in self.blocks]
        return wide

    def _field_view(self, x, key):
        off, size = self.key_to_offset[key], self.key_to_size[key]
        return x[off : off + size]

    def matvec(self, x):
        """y = M x (device vectors)."""
        y = torch.zeros(self.nrows, dtype=self.dtype, device=self.device)
        for row0, nrows, kind, key, payload in self.blocks:
            xk = self._field_view(x, key)
            if kind == "dense":
                ops.lincomb(y[row0 : row0 + nrows], 1.0, payload.t().contiguous(), xk.contiguous())
            else:
                coeff, shift, loc, vshape = payload
                field = self.key_to_field[key]
                gathered = ops.field_gather(xk.reshape(field.array.shape).contiguous(), field.loc, shift, loc)
                ops.addcmul(y[row0 : row0 + nrows], coeff.reshape(-1), gathered.reshape(-1))
        return y

    def rmatvec(self, y):
        """x = M^T y."""
        x = torch.zeros(self.ncols, dtype=self.dtype, device=self.device)
        for row0, nrows, kind, key, payload in self.blocks:
            yk = y[row0 : row0 + nrows]
            xk = self._field_view(x, key)
            if kind == "dense":
                ops.lincomb(xk, 1.0, payload.contiguous(), yk.contiguous())
            else:
                coeff, shift, loc, vshape = payload
                field = self.key_to_field[key]
                prod = torch.empty(nrows, dtype=self.dtype, device=self.device)
                ops.addcmul(prod, coeff.reshape(-1), yk.contiguous(), accumulate=False)
                ops.field_scatter(prod.reshape(vshape), field.array.shape, field.loc, shift, loc,
                                  out=xk.reshape(field.array.shape))
        return x

    def normal_diagonal(self):
        """diag(M^T M) = column sums of squares (Jacobi preconditioner / dampdiag)."""
        d = torch.zeros(self.ncols, dtype=self.dtype, device=self.device)
        for row0, nrows, kind, key, payload in self.blocks:
            dk = self._field_view(d, key)
            if kind == "dense":
                ones = torch.ones(nrows, dtype=self.dtype, device=self.device)
                sq = torch.empty_like(payload)
                ops.addcmul(sq.reshape(-1), payload.reshape(-1), payload.reshape(-1), accumulate=False)
                ops.lincomb(dk, 1.0, sq, ones)
            else:
                coeff, shift, loc, vshape = payload
                field = self.key_to_field[key]
                sq = torch.empty(nrows, dtype=self.dtype, device=self.device)
                ops.addcmul(sq, coeff.reshape(-1), coeff.reshape(-1), accumulate=False)
                ops.field_scatter(sq.reshape(vshape), field.array.shape, field.loc, shift, loc,
                                  out=dk.reshape(field.array.shape))
        return d

    def to_dense(self):
        """M as a dense device matrix (rows x unknowns) for the small-system direct solve: stencil
        blocks are scattered with the column map of `Context.field` (padded entries, which read the
        constant 0, are dropped -- the same matrix `matvec` applies)."""
        dense = torch.zeros((self.nrows, self.ncols), dtype=self.dtype, device=self.device)
        for row0, nrows, kind, key, payload in self.blocks:
            off = self.key_to_offset[key]
            if kind == "dense":
                dense[row0 : row0 + nrows, off : off + payload.shape[1]] += payload
            else:
                coeff, shift, loc, vshape = payload
                field = self.key_to_field[key]
                size = self.key_to_size[key]
                cols = (torch.arange(size, dtype=torch.float64, device=self.device) + 1).reshape(field.array.shape)
                cols = ops.field_gather(cols.contiguous(), field.loc, shift, loc).reshape(-1).to(torch.int64)
                rows = torch.arange(nrows, device=self.device) + row0
                keep = cols > 0
                dense.index_put_((rows[keep], cols[keep] - 1 + off), coeff.reshape(-1)[keep], accumulate=True)
        return dense

    def to_scipy(self, modsp=None):
        """The CSR matrix of reference core.py:1170-1214 (host, for inspection / parity)."""
        import scipy.sparse as sp

        modsp = modsp or sp
        npdt = numpy_dtype(self.dtype)
        matrix = modsp.csr_array((self.nrows, self.ncols), dtype=npdt)
        for row0, nrows, kind, key, payload in self.blocks:
            off = self.key_to_offset[key]
            if kind == "dense":
                m = modsp.csr_array(payload.cpu().numpy())
                block = modsp.csr_array((m.data, m.indices + off, m.indptr), shape=(nrows, self.ncols))
            else:
                coeff, shift, loc, vshape = payload
                field = self.key_to_field[key]
                size = self.key_to_size[key]
                # cols = offset + arange, then pad (constant 0) / roll / trim exactly as the
                # reference does (core.py:1147-1164): padded entries point at absolute column 0.
                cols = (torch.arange(size, dtype=torch.float64, device=self.device) + off).reshape(field.array.shape)
                cols = ops.field_gather(cols.contiguous(), field.loc, shift, loc).reshape(-1).cpu().numpy()
                rows = np.arange(nrows)
                block = modsp.csr_array(
                    (coeff.reshape(-1).cpu().numpy(), (rows, cols.astype(np.int64))), shape=(nrows, self.ncols)
                )
            full = modsp.vstack(
                [modsp.csr_array((row0, self.ncols), dtype=npdt), block,
                 modsp.csr_array((self.nrows - row0 - nrows, self.ncols), dtype=npdt)]
            ).tocsr()
            matrix = matrix + full
        return matrix.tocsr()


# ======================================================================================
# Problem (reference core.py:993-1386)
# ======================================================================================
class Problem:
    def __init__(self, operator, domain, extra=None, tracers=None, jit=None):
        """operator: callable(ctx) returning a list of arrays / (name, array) / Context.Raw."""
        self.domain = domain
        self.operator = operator
        self.extra = extra
        if tracers is None:
            tracers = dict()
        if "epoch" not in tracers:
            tracers["epoch"] = 0
        self.tracers = tracers
        if jit is None:
            from . import runtime

            jit = runtime.enable_jit
        self.jit = jit
        self._names = None
        self._fused = None  # fused evaluator (fused.py) once the operator has been recognised
        self._fused_checked = False
        self._traced = None  # generated per-operator kernels (stencil_jit.py)
        self._jac_traced = None  # ... with the Jacobian kernel of eval_operator_grad (None: not tried, False: not expressible)
        if not isinstance(domain.mod, ModRocm):
            raise NotImplementedError("Unsupported mod={:}".format(domain.mod))

    # ---- helpers -----------------------------------------------------------------------
    @staticmethod
    def _split_outputs(ff):
        assert isinstance(ff, (tuple, list)) and len(ff), "Operator must return a non-empty list"
        names = [f[0] if isinstance(f, tuple) else "" for f in ff]
        nonempty = [name for name in names if name]
        assert len(nonempty) == len(set(nonempty)), "Name of fields must be unique, got {}".format(nonempty)
        values = [f[1] if isinstance(f, tuple) else f for f in ff]
        return names, values

    def _shadow_state(self, state, arrays):
        """A state of the same structure whose arrays are `arrays` (no copies)."""
        fields = dict()
        for key, field in state.fields.items():
            if isinstance(field, Field):
                fields[key] = Field(None, loc=field.loc, cshape=field.cshape)
            elif isinstance(field, MultigridField):
                fields[key] = MultigridField(
                    [Field(None, loc=t.loc, cshape=t.cshape) for t in field.terms], loc=field.loc,
                    factors=field.factors, axes=field.axes, method=field.method,
                )
            elif isinstance(field, NeuralNet):
                fields[key] = NeuralNet(list(field.weights), list(field.biases), func_in=field.func_in,
                                        func_out=field.func_out, activation=field.activation)
            elif isinstance(field, Array):
                fields[key] = Array(None, field.shape)
            else:
                raise TypeError("Unknown field type '{}'".format(type(field).__name__))
        shadow = State(fields=fields, initialized=True)
        self.domain.arrays_to_state(arrays, shadow)
        return shadow

    def recognise(self, state):
        """Once per problem (like the reference's jit cache, core.py:1023-1025): the operator is probed for the fused
        Poisson route (fused.detect) and, failing that, traced into generated kernels (stencil_jit.trace)."""
        if not self._fused_checked:
            self._fused_checked = True
            from . import fused, runtime

            if runtime.enable_fuse:
                self._fused = fused.detect(self, state)
            if self._fused is None and runtime.enable_trace:
                from . import stencil_jit

                self._traced = stencil_jit.trace(self, state)

    # ---- loss + gradient (reference core.py:1038-1111, 1219-1241) --------------------------
    def eval_loss_grad_device(self, state):
        """Like eval_loss_grad but loss / terms / norms stay 0-d DEVICE tensors (no host sync)."""
        if not state.initialized:
            raise RuntimeError("Uninitialized state, use `state = domain.init_state(state)`")
        self.recognise(state)
        if self._fused is not None:
            return self._fused.eval_loss_grad(state)
        if self._traced is not None:
            if not self._traced.matches(state):  # the state was restructured (e.g. another multigrid depth): trace again
                from . import stencil_jit

                self._traced = stencil_jit.trace(self, state)
            if self._traced is not None:
                return self._traced.eval_loss_grad(state)
        if self.jit:
            return self._eval_loss_grad_graph(state)
        return self._eval_loss_grad_generic(state)

    # ---- ODIL_JIT=1: the generic evaluation captured once into a hipGraph and replayed --------
    def _eval_loss_grad_graph(self, state):
        """The reference's `jit` knob (runtime.py:25, core.py:1069,1107) compiles the traced
        loss+gradient function.  Here the same role is played by a HIP graph: the first
        evaluation records every launch of the generic path -- the user's pointwise device ops,
        their autograd backward, and the HIP kernels of this library on the same stream --
        and later evaluations replay it with one host call, which is what small (launch-bound)
        grids need.  Like a jit trace it bakes in `extra` and the Python control flow; `tracers`
        stay live: numeric entries are device scalars refreshed before every replay.  Any
        host synchronisation inside the operator makes capture fail; evaluation then falls back
        to the eager path for good."""
        domain = self.domain
        arrays = domain.arrays_from_state(state)
        key = tuple((tuple(a.shape), a.dtype) for a in arrays)
        cache = self.__dict__.setdefault("_graph_cache", dict())
        entry = cache.get(key)
        if entry is False:
            return self._eval_loss_grad_generic(state)
        if entry is None:
            try:
                entry = self._capture_graph(state, arrays)
            except Exception as e:  # capture is an optimisation: never a reason to fail
                from .util import printlog

                printlog("odil_amd: hipGraph capture failed ({}: {}); using the eager path".format(
                    type(e).__name__, str(e).splitlines()[0] if str(e) else ""))
                torch.cuda.synchronize()
                cache[key] = False
                return self._eval_loss_grad_generic(state)
            cache[key] = entry
        static_in, static_tr, graph, out = entry
        for s_, a in zip(static_in, arrays):
            if s_.data_ptr() != a.data_ptr():
                s_.copy_(a)
        for k, t in static_tr.items():
            t.fill_(self.tracers[k])
        graph.replay()
        loss, grads, terms, names, norms = out
        return loss, list(grads), list(terms), names, list(norms)

    def _capture_graph(self, state, arrays):
        static_in = [a.detach().clone() for a in arrays]
        static_tr = dict()
        live_tracers = dict(self.tracers)
        for k, v in self.tracers.items():
            if isinstance(v, (int, float, np.integer, np.floating)) and not isinstance(v, bool):
                static_tr[k] = torch.tensor(float(v), dtype=torch_dtype(self.domain.dtype), device=self.domain.mod.device)
                live_tracers[k] = static_tr[k]
        static_state = self._shadow_state(state, static_in)
        saved = self.tracers
        try:
            self.tracers = live_tracers
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):  # warm-up: allocator, lazy initialisation, workspace tensors
                    self._eval_loss_grad_generic(static_state)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._eval_loss_grad_generic(static_state)
        finally:
            self.tracers = saved
        return static_in, static_tr, graph, out

    def _eval_loss_grad_generic(self, state):
        domain = self.domain
        arrays = domain.arrays_from_state(state)
        leaves = [a.detach().requires_grad_(True) for a in arrays]
        shadow = self._shadow_state(state, leaves)
        with torch.enable_grad():
            ctx = Context(domain, shadow, extra=self.extra, tracers=self.tracers)
            names, values = self._split_outputs(self.operator(ctx))
            terms = [
                _MeanFn.apply(v.value, False) if isinstance(v, Context.Raw) else _MeanFn.apply(v, True)
                for v in values
            ]
            loss = terms[0]
            for t in terms[1:]:
                loss = loss + t
        grads = torch.autograd.grad(loss, leaves, allow_unused=True)
        grads = [g if g is not None else torch.zeros_like(a) for g, a in zip(grads, arrays)]
        loss = loss.detach()
        terms = [t.detach() for t in terms]
        norms = [t if isinstance(v, Context.Raw) else torch.sqrt(t) for t, v in zip(terms, values)]
        self._names = names
        return loss, grads, terms, names, norms

    def eval_loss_grad(self, state):
        """loss (np scalar), grads (list of device arrays), terms, names, norms
        (reference core.py:1219-1241; `np.array(loss)` is the per-call host sync there too)."""
        loss, grads, terms, names, norms = self.eval_loss_grad_device(state)
        to_np = lambda t: np.array(t.detach().cpu().numpy()) if isinstance(t, torch.Tensor) else np.array(t)
        return to_np(loss), grads, list(map(to_np, terms)), names, list(map(to_np, norms))

    # ---- operator values (reference core.py:1243-1311) -------------------------------------
    def eval_operator(self, state):
        if not state.initialized:
            raise RuntimeError("Uninitialized state, use `state = domain.init_state(state)`")
        with torch.no_grad():
            ctx = Context(self.domain, state, extra=self.extra, tracers=self.tracers)
            names, values = self._split_outputs(self.operator(ctx))
            values = [v.value if isinstance(v, Context.Raw) else v for v in values]
        return values, names

    # ---- per-shift gradients (reference core.py:1313-1383) ---------------------------------
    def eval_operator_grad(self, state):
        """values, grads (per output: dict (key, shift, loc) -> coefficient array, plus dense
        Jacobians for Array / NeuralNet unknowns under (key, None, None)), names."""
        if not state.initialized:
            raise RuntimeError("Uninitialized state, use `state = domain.init_state(state)`")
        domain = self.domain
        # the generated Jacobian kernel (stencil_codegen._jacobian_kernel): values and every per-shift coefficient array in
        # ONE pointwise launch from the symbolic derivative of the traced operator; operators it cannot express (dense
        # columns of parameter arrays, windows, untraceable code) take the autograd evaluation below
        from . import runtime

        if runtime.enable_trace and int(os.environ.get("ODIL_TRACE_JAC", 1)) and torch.cuda.is_available():
            if self._jac_traced is None or (self._jac_traced and not self._jac_traced.matches(state)):
                from . import stencil_jit

                self._jac_traced = stencil_jit.trace_jacobian(self, state) or False
            if self._jac_traced:
                return self._jac_traced.eval_operator_grad(state)
        arrays = domain.arrays_from_state(state)
        leaves = [a.detach().requires_grad_(True) for a in arrays]
        shadow = self._shadow_state(state, leaves)
        with torch.enable_grad():
            ctx = Context(domain, shadow, extra=self.extra, tracers=self.tracers, distinct_shift=True)
            names, values = self._split_outputs(self.operator(ctx))
            values = [v.value if isinstance(v, Context.Raw) else v for v in values]
            grads = []
            for v in values:
                g = dict()
                descs = list(ctx.desc_to_array.keys())
                symbols = [ctx.desc_to_array[d] for d in descs]
                if v.requires_grad and symbols:
                    gg = torch.autograd.grad(v.sum(), symbols, retain_graph=True, allow_unused=True)
                else:
                    gg = [None] * len(symbols)
                g.update(dict(zip(descs, gg)))
                for jkey, arr in ctx.key_to_array_jac.items():
                    g[jkey] = self._dense_jacobian(v, arr)
                grads.append(g)
        values = [v.detach() for v in values]
        return values, grads, names

    @staticmethod
    def _dense_jacobian(v, arr):
        """d v / d arr with shape v.shape + arr.shape (tf's tape.jacobian, core.py:1347-1349)."""
        arrs = arr if isinstance(arr, list) else [arr]
        if not v.requires_grad:
            return [None for _ in arrs] if isinstance(arr, list) else None
        flat = v.reshape(-1)
        nparam = sum(int(a.numel()) for a in arrs)
        jacs = None
        if nparam < flat.numel():
            # Few parameters, many rows (a network's 46 weights against 10^5 .. 10^6 grid rows): one reverse pass PER
            # PARAMETER instead of one per row.  gg(w) = J^T w is linear in the dummy cotangent w, so d gg_e / d w is
            # column e of J: a second reverse pass through the graph of the first (heat 64 x 128: 13.5 s -> 0.1 s; the
            # per-row loop at 256 x 512 would be 5e5 passes).
            try:
                w = torch.ones_like(v).requires_grad_(True)
                gg = torch.autograd.grad(v, arrs, grad_outputs=w, create_graph=True, allow_unused=True)
                jacs = []
                for g, a in zip(gg, arrs):
                    if g is None:  # structurally independent of this array
                        jacs.append(torch.zeros(tuple(v.shape) + tuple(a.shape), dtype=v.dtype, device=v.device))
                        continue
                    if not g.requires_grad:
                        # J^T w that does not depend on w: the graph of the first pass was cut on the way (a backward
                        # without a derivative of its own) -- NOT a zero Jacobian: row by row below
                        raise RuntimeError("first reverse pass is not differentiable")
                    gflat = g.reshape(-1)
                    cols = [torch.autograd.grad(gflat[e], w, retain_graph=True)[0] for e in range(gflat.numel())]
                    jacs.append(torch.stack(cols, dim=-1).reshape(tuple(v.shape) + tuple(a.shape)))
            except RuntimeError:  # (an operation on the way without a second derivative: row by row below)
                jacs = None
        if jacs is None:
            rows = [[] for _ in arrs]
            for i in range(flat.numel()):
                gg = torch.autograd.grad(flat[i], arrs, retain_graph=True, allow_unused=True)
                for k, (g, a) in enumerate(zip(gg, arrs)):
                    rows[k].append(g if g is not None else torch.zeros_like(a))
            jacs = [torch.stack(r).reshape(tuple(v.shape) + tuple(a.shape)) for r, a in zip(rows, arrs)]
        if all(float(j.abs().max()) == 0 for j in jacs):
            return [None for _ in arrs] if isinstance(arr, list) else None
        return jacs if isinstance(arr, list) else jacs[0]

    def linearize_device(self, state):
        """(vector, LinearizedOperator): operator(V) ~= M (V - V0) + vector, on the device."""
        if not state.initialized:
            raise RuntimeError("Uninitialized state, use `state = domain.init_state(state)`")
        values, grads, names = self.eval_operator_grad(state)
        op = LinearizedOperator(self.domain, state)
        for value, grad in zip(values, grads):
            op.add_output(tuple(value.shape), grad)
        vector = values[0].reshape(-1) if len(values) == 1 else torch.cat([v.reshape(-1) for v in values])
        return vector, op

    def linearize(self, state, modsp=None):
        """Reference signature (core.py:1113-1217): returns (vector, scipy-like CSR matrix)."""
        vector, op = self.linearize_device(state)
        return vector, op.to_scipy(modsp)

    def get_context(self, state):
        return self.domain.get_context(state, extra=self.extra, tracers=self.tracers)


# ======================================================================================
# Checkpoints (reference core.py:1389-1436), extrapolation helpers (core.py:1439-1457)
# ======================================================================================
def _to_numpy(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.array(a)


def checkpoint_save(domain, state, path):
    fields = dict()
    for key in state.fields:
        fields[key] = [_to_numpy(a) for a in domain.arrays_from_field(state.fields[key])]
    with open(path, "wb") as f:
        pickle.dump({"fields": fields}, f)


def checkpoint_load(domain, state, path, skip_missing=True, keys=None):
    """Fills the fields of `state` from a checkpoint (the pickle layout of reference core.py:1389-1436: {"fields": {key:
    array or list of arrays}}); a missing key is an error only with skip_missing=False."""
    with open(path, "rb") as f:
        stored = pickle.load(f).get("fields", dict())
    wanted = list(state.fields) if not keys else list(keys)
    missing = [key for key in wanted if key not in stored]
    if missing and not skip_missing:
        raise RuntimeError(f"Field {missing[0]} not found in {path}")
    for key in wanted:
        if key in stored:
            entry = stored[key]
            levels = entry if isinstance(entry, list) else [entry]
            domain.arrays_to_field([domain.mod.variable(a, dtype=domain.dtype) for a in levels], state.fields[key])


def extrap_quadh(u0, u1, u1p):
    "Quadratic extrapolation from points 0, 1, 1.5 to point 2."
    return (u0 - 6 * u1 + 8 * u1p) / 3


def extrap_quad(u0, u1, u2):
    "Quadratic extrapolation from points 0, 1, 2 to point 3."
    return u0 - 3 * u1 + 3 * u2


def extrap_linear(u0, u1):
    "Linear extrapolation from points 0, 1 to point 2."
    return 2 * u1 - u0


def struct_to_numpy(mod, d):
    """`d` with every tensor inside its dicts / lists / tuples replaced by a NumPy array (dicts are rewritten in place, as
    the reference's helper does, core.py:1460-1476)."""
    if mod.is_tensor(d):
        return _to_numpy(d)
    if isinstance(d, dict):
        d.update({key: struct_to_numpy(mod, val) for key, val in d.items()})
        return d
    if isinstance(d, (list, tuple)):
        return type(d)(struct_to_numpy(mod, item) for item in d)
    return d
