"""Throw-away `mod` shim over torch-CPU used ONLY to generate golden vectors.

The reference (cselab/odil, /root/reference) delegates all arithmetic to a
`mod` namespace backed by TensorFlow or JAX (reference src/odil/backend.py:12-317).
Neither is installed in the build container, so the fixtures under tests/golden/
are produced by running the reference's OWN `core.py`, `optimizer.py`,
`linsolver.py` and example operators, unchanged, on top of this minimal
torch-CPU namespace, with `torch.autograd` standing in for `jax.value_and_grad`
(reference core.py:1098-1101).  This file is test infrastructure: it is never
imported by the product (`odil_amd/`) and never runs on the GPU box.
"""

import importlib.abc
import importlib.machinery
import importlib.util
import sys
from argparse import Namespace

import numpy as np
import torch

REFERENCE_SRC = "/root/reference/src"


class _StubLoader(importlib.abc.Loader):
    """Loader that fills a stub module from a dict (works under the reference's LazyLoader)."""

    def __init__(self, attrs):
        self.attrs = attrs

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        module.__dict__.update(self.attrs)


def _stub_module(name, attrs):
    spec = importlib.machinery.ModuleSpec(name, _StubLoader(attrs))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    sys.modules[name] = module
    return module


def import_reference():
    """Imports the reference package with `odil.runtime` stubbed out
    (the real one calls exit(1) without TF/JAX, reference runtime.py:37-44)."""
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    # `examples/heat/heat.py:12,317` does `from odil.runtime import tf` and decorates a plot
    # helper with `@tf.function()`: give it an inert stand-in (never used for arithmetic).
    fake_tf = Namespace(function=lambda *a, **k: (lambda f: f))
    attrs = dict(
        tf=fake_tf, jax=None, enable_jit=False, backend_name="shim", dtype=np.dtype("float64"), mod=ModShim()
    )
    _stub_module("odil.runtime", attrs)
    _stub_module("odil.plot", dict())
    import odil  # noqa

    return odil


_TORCH_DTYPE = {
    np.dtype("float32"): torch.float32,
    np.dtype("float64"): torch.float64,
    np.dtype("int32"): torch.int32,
    np.dtype("int64"): torch.int64,
    np.dtype("bool"): torch.bool,
}


def tdtype(dtype):
    if isinstance(dtype, torch.dtype):
        return dtype
    return _TORCH_DTYPE[np.dtype(dtype)]


def T(x, dtype=None):
    """To tensor."""
    if isinstance(x, torch.Tensor):
        return x if dtype is None else x.to(tdtype(dtype))
    x = np.asarray(x)
    if dtype is not None:
        return torch.as_tensor(x).to(tdtype(dtype))
    return torch.as_tensor(x)


class ModShim:
    """NumPy-flavoured namespace over torch CPU tensors (reference backend.py:44-110 names)."""

    def __init__(self):
        self.jax = None
        self.tf = None
        self.modsp = None
        self.mod = self
        self.float32 = np.float32
        self.float64 = np.float64
        self.int32 = np.int32
        self.random = Namespace()
        self.random.set_seed = lambda seed: np.random.seed(seed)
        self.random.uniform = lambda shape, minval, maxval, dtype: T(
            np.random.uniform(low=minval, high=maxval, size=shape), dtype
        )
        self.random.normal = lambda shape, mean, stddev, dtype: T(
            np.random.normal(loc=mean, scale=stddev, size=shape), dtype
        )

    # creation / conversion
    def cast(self, x, dtype):
        return T(x, dtype)

    def array(self, x, dtype=None):
        return T(x, dtype)

    constant = array
    native = array

    def variable(self, x, dtype=None):
        return T(x, dtype).clone()

    def numpy(self, x):
        return x.detach().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)

    def is_tensor(self, x):
        return isinstance(x, torch.Tensor)

    def zeros(self, shape, dtype=None):
        shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(int(s) for s in shape)
        return torch.zeros(shape, dtype=tdtype(dtype or np.float64))

    def ones(self, shape, dtype=None):
        shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(int(s) for s in shape)
        return torch.ones(shape, dtype=tdtype(dtype or np.float64))

    def full(self, shape, value, dtype=None):
        return torch.full(tuple(shape), value, dtype=tdtype(dtype or np.float64))

    def zeros_like(self, x):
        return torch.zeros_like(T(x))

    def ones_like(self, x):
        return torch.ones_like(T(x))

    def copy(self, x):
        return T(x).clone()

    def arange(self, *a, **k):
        return torch.arange(*a, **k)

    def linspace(self, *a, **k):
        return T(np.linspace(*a, **k))

    def meshgrid(self, *xx, indexing="ij"):
        return [T(x) for x in np.meshgrid(*[self.numpy(x) for x in xx], indexing=indexing)]

    # shape ops
    def reshape(self, x, shape):
        return T(x).reshape(tuple(int(s) for s in shape))

    def flatten(self, x):
        return T(x).reshape(-1)

    def stack(self, xs, axis=0):
        return torch.stack([T(x) for x in xs], dim=axis)

    def concatenate(self, xs, axis=0):
        return torch.cat([T(x) for x in xs], dim=axis)

    def hstack(self, xs):
        return torch.hstack([T(x) for x in xs])

    def transpose(self, x, perm=None):
        x = T(x)
        if perm is None:
            perm = tuple(reversed(range(x.dim())))
        return x.permute(tuple(int(p) for p in perm))

    def moveaxis(self, x, s, d):
        return torch.moveaxis(T(x), s, d)

    def broadcast_to(self, x, shape):
        return torch.broadcast_to(T(x), tuple(shape))

    def split_by_sizes(self, x, sizes, axis=0):
        return list(torch.split(T(x), [int(s) for s in sizes], dim=axis))

    def roll(self, x, shift, axis=None):
        x = T(x)
        if axis is None:
            return torch.roll(x.reshape(-1), int(shift)).reshape(x.shape)
        if isinstance(axis, (int, np.integer)):
            return torch.roll(x, int(shift), int(axis))
        axis = [int(a) for a in axis]
        shift = [int(s) for s in np.broadcast_to(np.asarray(shift), (len(axis),))]
        return torch.roll(x, shift, axis)

    def pad(self, x, pad_width, mode="constant"):
        x = T(x)
        for d, (lo, hi) in enumerate(pad_width):
            if lo == 0 and hi == 0:
                continue
            idx = np.arange(x.shape[d])
            if mode == "constant":
                shp = list(x.shape)
                parts = []
                if lo:
                    shp[d] = lo
                    parts.append(torch.zeros(shp, dtype=x.dtype))
                parts.append(x)
                if hi:
                    shp[d] = hi
                    parts.append(torch.zeros(shp, dtype=x.dtype))
                x = torch.cat(parts, dim=d)
            else:
                idx = np.pad(idx, (lo, hi), mode=mode)
                x = torch.index_select(x, d, torch.as_tensor(idx))
        return x

    def gather_nd(self, u, idx):
        return u[tuple(torch.moveaxis(T(idx), -1, 0))]

    # math
    def where(self, c, a, b):
        c = T(c)
        ta, tb = isinstance(a, torch.Tensor), isinstance(b, torch.Tensor)
        if not ta and not tb:
            a = T(a)
            b = T(b, a.dtype if a.dtype.is_floating_point else None)
        elif not ta:
            a = T(a, b.dtype)
        elif not tb:
            b = T(b, a.dtype)
        return torch.where(c, a, b)

    def stop_gradient(self, x):
        return T(x).detach()

    def sum(self, x, axis=None):
        return torch.sum(T(x)) if axis is None else torch.sum(T(x), dim=axis)

    def mean(self, x, axis=None):
        return torch.mean(T(x)) if axis is None else torch.mean(T(x), dim=axis)

    def max(self, x):
        return torch.max(T(x))

    def min(self, x):
        return torch.min(T(x))

    def matmul(self, a, b):
        return torch.matmul(T(a), T(b))

    def sigmoid(self, x):
        return 1 / (1 + torch.exp(-T(x)))

    def relu(self, x):
        return torch.clamp(T(x), min=0)

    def clip(self, x, a, b):
        return torch.clamp(T(x), a, b)

    def norm(self, x):
        return torch.linalg.norm(T(x))

    # strided convolutions of the multigrid transfers (reference backend.py:112-126, 165-172: jax.lax.conv /
    # jax.lax.conv_transpose).  torch's conv{1,2,3}d is the same cross-correlation (no kernel flip) on (N, C, *spatial).
    def convolution(self, input, filters, strides, padding):
        input, filters = T(input), T(filters).to(T(input).dtype)
        dim = input.dim()
        if isinstance(strides, int):  # backend.py:118-119: an integer stride goes to EVERY axis, '.' ones included
            strides = (strides,) * dim
        assert padding == "VALID" and 1 <= dim <= 3, (padding, dim)
        conv = {1: torch.nn.functional.conv1d, 2: torch.nn.functional.conv2d, 3: torch.nn.functional.conv3d}[dim]
        res = conv(input.reshape((1, 1) + tuple(input.shape)), filters.reshape((1, 1) + tuple(filters.shape)),
                   stride=tuple(int(s) for s in strides))
        return res[0, 0]

    def conv_transpose(self, input, filters, output_shape=None, strides=None, padding=None):
        # called with lhs (1, *spatial, 1) and rhs (*kernel, 1, 1) (reference core.py:656-662); jax.lax.conv_transpose
        # with its default transpose_kernel=False correlates the stride-dilated input, padded by k - 1, with the kernel
        # as given: the scatter out[s i + k - 1 - j] += in[i] w[j], i.e. torch's conv_transpose with the kernel flipped
        input, filters = T(input), T(filters).to(T(input).dtype)
        dim = input.dim() - 2
        if isinstance(strides, int):
            strides = (strides,) * dim
        assert padding == "VALID" and 1 <= dim <= 3, (padding, dim)
        x = input.reshape((1, 1) + tuple(input.shape[1:-1]))
        w = torch.flip(filters.reshape(tuple(filters.shape[:-2])), dims=tuple(range(dim))).reshape(
            (1, 1) + tuple(filters.shape[:-2]))
        convt = {1: torch.nn.functional.conv_transpose1d, 2: torch.nn.functional.conv_transpose2d,
                 3: torch.nn.functional.conv_transpose3d}[dim]
        res = convt(x, w, stride=tuple(int(s) for s in strides))
        res = res.reshape((1,) + tuple(res.shape[2:]) + (1,))
        if output_shape is not None:
            assert tuple(res.shape) == tuple(int(v) for v in output_shape), (res.shape, output_shape)
        return res


for _name in ["abs", "cos", "sin", "exp", "square", "sqrt", "tanh", "log", "floor", "minimum", "maximum"]:

    def _make(name):
        f = getattr(torch, name)

        def g(self, *args):
            args = [T(a) for a in args]
            if len(args) == 2 and args[0].dtype != args[1].dtype:
                args[1] = args[1].to(args[0].dtype)
            return f(*args)

        return g

    setattr(ModShim, _name, _make(_name))
