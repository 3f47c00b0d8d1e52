"""`mod`: the backend namespace user operators are written against.

Mirrors the names of the reference's `ModNumpy` / `ModTensorflow`
(reference src/odil/backend.py:17-42, :50-110, :188-317) with NumPy-style signatures, on
torch-ROCm tensors (device memory, streams, autograd bookkeeping).  What user code does
with these names -- the pointwise arithmetic of an `operator(ctx)` -- runs as device
elementwise ops; everything the FRAMEWORK owns (multigrid transfers, stencil access,
loss reduction, optimizer updates, Jacobian assembly, solves) goes through the
hand-written HIP kernels of `libodil_hip.so` (see core.py / optimizer.py / linsolver.py).
"""

from argparse import Namespace

import numpy as np
import torch

_NP2T = {
    np.dtype("float32"): torch.float32,
    np.dtype("float64"): torch.float64,
    np.dtype("int32"): torch.int32,
    np.dtype("int64"): torch.int64,
    np.dtype("bool"): torch.bool,
}


def torch_dtype(dtype):
    if dtype is None or isinstance(dtype, torch.dtype):
        return dtype
    if dtype is int:
        return torch.int64
    if dtype is float:
        return torch.float64
    if dtype is bool:
        return torch.bool
    return _NP2T[np.dtype(dtype)]


def numpy_dtype(dtype):
    if isinstance(dtype, torch.dtype):
        return {v: k for k, v in _NP2T.items()}[dtype]
    return np.dtype(dtype)


class _ConvValidFn(torch.autograd.Function):
    """y = corr(x, w, s) (VALID) or its transpose, through `odil_conv_valid`; the cotangent is the other of the two with
    the same kernel.  (The framework's own transfers -- `core.restrict_to_coarser`, `core.interp_to_finer` -- use the
    dedicated packed / marching kernels, not this one.)"""

    @staticmethod
    def forward(ctx, x, w, strides, transposed, out_shape):
        from . import ops

        ctx.save_for_backward(w)
        ctx.strides, ctx.transposed, ctx.xshape = strides, transposed, tuple(x.shape)
        return ops.conv_valid(x, w, strides, transposed=transposed, out_shape=out_shape)

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        g = g.contiguous()
        if ctx.transposed:
            # y = C^T x (possibly zero-extended): dx = C g restricted to the VALID extent that maps back onto x
            gx = _ConvValidFn.apply(g, w, ctx.strides, False, None)
            gx = gx[tuple(slice(0, n) for n in ctx.xshape)]
        else:
            gx = _ConvValidFn.apply(g, w, ctx.strides, True, ctx.xshape)
        return gx, None, None, None, None


class ModRocm:
    """NumPy-flavoured namespace over torch tensors living on one HIP device."""

    def __init__(self, device=None):
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        if device is None:
            raise RuntimeError(
                "odil_amd needs a HIP device (MI355X): none is visible and there is no CPU fallback. "
                "Pass ModRocm(device='cpu') only to exercise host-side plumbing in tests."
            )
        self.device = torch.device(device)
        self.mod = self
        self.jax = None
        self.tf = None
        self.modsp = None
        self.float32 = np.float32
        self.float64 = np.float64
        self.int32 = np.int32
        self.ndarray = torch.Tensor
        self.batch_to_space = None
        self.random = Namespace()
        self._gen = torch.Generator(device="cpu")
        self.random.set_seed = self._set_seed
        self.random.uniform = self._uniform
        self.random.normal = self._normal

    # -- conversion ---------------------------------------------------------------
    def _t(self, x, dtype=None):
        dtype = torch_dtype(dtype)
        if isinstance(x, torch.Tensor):
            if x.device != self.device:
                x = x.to(self.device)
            return x if dtype is None or x.dtype == dtype else x.to(dtype)
        if isinstance(x, (bool, int, float, np.bool_, np.integer, np.floating)):
            # a device fill, not a host->device copy: legal inside hipGraph capture (ODIL_JIT=1)
            if dtype is None:
                dtype = (torch.bool if isinstance(x, (bool, np.bool_)) else torch.int64 if isinstance(x, (int, np.integer))
                         else torch_dtype(np.asarray(x).dtype))
            return torch.full((), x, dtype=dtype, device=self.device)
        return torch.as_tensor(np.asarray(x), device=self.device).to(dtype) if dtype is not None else torch.as_tensor(
            np.asarray(x), device=self.device
        )

    def cast(self, x, dtype):
        return self._t(x, dtype)

    def array(self, x, dtype=None):
        return self._t(x, dtype)

    constant = array
    native = array

    def variable(self, x, dtype=None):
        return self._t(x, dtype).detach().clone().contiguous()

    def numpy(self, x):
        if isinstance(x, torch.Tensor):
            return x.detach().cpu().numpy()
        return np.asarray(x)

    def spnative(self, x):
        return x

    def is_tensor(self, x):
        return isinstance(x, torch.Tensor)

    def copy(self, x):
        return self._t(x).clone()

    def stop_gradient(self, x):
        return self._t(x).detach()

    # -- creation -----------------------------------------------------------------
    @staticmethod
    def _shape(shape):
        if isinstance(shape, (int, np.integer)):
            return (int(shape),)
        return tuple(int(s) for s in shape)

    def zeros(self, shape, dtype=None):
        return torch.zeros(self._shape(shape), dtype=torch_dtype(dtype or np.float32), device=self.device)

    def ones(self, shape, dtype=None):
        return torch.ones(self._shape(shape), dtype=torch_dtype(dtype or np.float32), device=self.device)

    def full(self, shape, value, dtype=None):
        return torch.full(self._shape(shape), value, dtype=torch_dtype(dtype), device=self.device)

    def zeros_like(self, x):
        return torch.zeros_like(self._t(x))

    def ones_like(self, x):
        return torch.ones_like(self._t(x))

    def arange(self, *args, dtype=None):
        return torch.arange(*args, dtype=torch_dtype(dtype), device=self.device)

    def linspace(self, start, stop, num=50, endpoint=True, dtype=None):
        return self._t(np.linspace(start, stop, num, endpoint=endpoint, dtype=dtype))

    def meshgrid(self, *xx, indexing="ij"):
        return [self._t(x) for x in np.meshgrid(*[self.numpy(x) for x in xx], indexing=indexing)]

    # -- shape --------------------------------------------------------------------
    def reshape(self, x, shape):
        return self._t(x).reshape(self._shape(shape))

    def flatten(self, x):
        return self._t(x).reshape(-1)

    def stack(self, xs, axis=0):
        return torch.stack([self._t(x) for x in xs], dim=axis)

    def concatenate(self, xs, axis=0):
        return torch.cat([self._t(x) for x in xs], dim=axis)

    def hstack(self, xs):
        return torch.hstack([self._t(x) for x in xs])

    def transpose(self, x, perm=None):
        x = self._t(x)
        if perm is None:
            perm = tuple(reversed(range(x.dim())))
        return x.permute(tuple(int(p) for p in perm))

    def moveaxis(self, x, source, destination):
        return torch.moveaxis(self._t(x), source, destination)

    def broadcast_to(self, x, shape):
        return torch.broadcast_to(self._t(x), self._shape(shape))

    def split_by_sizes(self, x, sizes, axis=0):
        return list(torch.split(self._t(x), [int(s) for s in sizes], dim=axis))

    def roll(self, x, shift, axis=None):
        x = self._t(x)
        if axis is None:
            return torch.roll(x.reshape(-1), int(shift)).reshape(x.shape)
        if isinstance(axis, (int, np.integer)):
            return torch.roll(x, int(shift), int(axis))
        axis = [int(a) for a in axis]
        shift = [int(s) for s in np.broadcast_to(np.asarray(shift), (len(axis),))]
        return torch.roll(x, shift, axis)

    def pad(self, x, pad_width, mode="constant"):
        x = self._t(x)
        for d, (lo, hi) in enumerate(pad_width):
            if lo == 0 and hi == 0:
                continue
            if mode == "constant":
                shp = list(x.shape)
                parts = []
                if lo:
                    shp[d] = lo
                    parts.append(torch.zeros(shp, dtype=x.dtype, device=x.device))
                parts.append(x)
                if hi:
                    shp[d] = hi
                    parts.append(torch.zeros(shp, dtype=x.dtype, device=x.device))
                x = torch.cat(parts, dim=d)
            elif mode in ("reflect", "symmetric"):
                idx = np.pad(np.arange(x.shape[d]), (lo, hi), mode=mode)
                x = torch.index_select(x, d, torch.as_tensor(idx, device=x.device))
            else:
                raise ValueError("Unknown mode=" + mode)
        return x

    def gather_nd(self, u, idx):
        return u[tuple(torch.moveaxis(self._t(idx), -1, 0))]

    # -- math ---------------------------------------------------------------------
    def where(self, c, a, b):
        c = self._t(c)
        ta, tb = isinstance(a, torch.Tensor), isinstance(b, torch.Tensor)
        if not ta and not tb:
            a = self._t(a)
            b = self._t(b, a.dtype)
        elif not ta:
            b = self._t(b)
            a = self._t(a, b.dtype)
        else:
            a = self._t(a)
            b = self._t(b, a.dtype)
        return torch.where(c, a, b)

    def sum(self, x, axis=None):
        return torch.sum(self._t(x)) if axis is None else torch.sum(self._t(x), dim=axis)

    def mean(self, x, axis=None):
        return torch.mean(self._t(x)) if axis is None else torch.mean(self._t(x), dim=axis)

    def cumsum(self, x, axis=0):
        return torch.cumsum(self._t(x), dim=axis)

    def std(self, x):
        return torch.std(self._t(x), unbiased=False)

    def median(self, x):
        return torch.median(self._t(x))

    def min(self, x):
        return torch.min(self._t(x))

    def max(self, x):
        return torch.max(self._t(x))

    def relu(self, x):
        return torch.clamp(self._t(x), min=0)

    def sigmoid(self, x):
        return 1 / (1 + torch.exp(-self._t(x)))

    def clip(self, x, a, b):
        return torch.clamp(self._t(x), a, b)

    def arctan2(self, a, b):
        return torch.atan2(self._t(a), self._t(b))

    def norm(self, x):
        return torch.linalg.norm(self._t(x))

    def solve(self, a, b):
        return torch.linalg.solve(self._t(a), self._t(b))

    def matmul(self, a, b):
        a = self._t(a)
        return torch.matmul(a, self._t(b, a.dtype))

    def einsum(self, spec, *xs):
        return torch.einsum(spec, *[self._t(x) for x in xs])

    def jit_wrap(self, **kwargs):
        return lambda f: f

    def convolution(self, input, filters, strides, padding):
        """n-dimensional VALID cross-correlation of `input` with `filters` (same rank), integer stride or one per axis
        (reference backend.py:112-126: jax.lax.conv on (1, 1) + shape; the call of `restrict_to_coarser`,
        core.py:744-751) through the tap kernel `odil_conv_valid`; differentiable in `input`."""
        input = self._t(input)
        filters = self._t(filters, input.dtype)
        dim = input.dim()
        if isinstance(strides, (int, np.integer)):
            strides = (int(strides),) * dim
        strides = tuple(int(v) for v in strides)
        if padding != "VALID":
            raise NotImplementedError("mod.convolution: padding='{}' (the reference only uses 'VALID')".format(padding))
        if filters.dim() != dim or len(strides) != dim or dim > 4:
            raise ValueError("mod.convolution: input {}, filters {}, strides {}".format(tuple(input.shape), tuple(filters.shape), strides))
        if filters.requires_grad:
            raise NotImplementedError("mod.convolution: gradients with respect to the filters")
        return _ConvValidFn.apply(input.contiguous(), filters.contiguous(), strides, False, None)

    def conv_transpose(self, input, filters, output_shape=None, strides=None, padding=None):
        """Transposed VALID convolution in the layout the reference calls it with (backend.py:165-172 ->
        jax.lax.conv_transpose; core.py:656-662): input (1, *spatial, 1), filters (*kernel, 1, 1); the stride-dilated
        input is correlated with the kernel AS GIVEN (transpose_kernel=False), i.e. out[s i + K - 1 - j] += in[i] w[j].
        Result (1, *(n s + max(K - s, 0)), 1) -- (n - 1) s + K when K >= s --; differentiable in `input`.  Kernel extents and
        strides 1 .. 4, up to 4 axes (odil_conv_valid)."""
        input = self._t(input)
        filters = self._t(filters, input.dtype)
        dim = input.dim() - 2
        if dim < 1 or dim > 4 or input.shape[0] != 1 or input.shape[-1] != 1 or filters.dim() != dim + 2 \
                or tuple(filters.shape[-2:]) != (1, 1):
            raise NotImplementedError("mod.conv_transpose: expected input (1, *spatial, 1) and filters (*kernel, 1, 1), got {} "
                                      "and {}".format(tuple(input.shape), tuple(filters.shape)))
        if padding != "VALID":
            raise NotImplementedError("mod.conv_transpose: padding='{}' (the reference only uses 'VALID')".format(padding))
        if strides is None:
            strides = 1
        if isinstance(strides, (int, np.integer)):
            strides = (int(strides),) * dim
        strides = tuple(int(v) for v in strides)
        if filters.requires_grad:
            raise NotImplementedError("mod.conv_transpose: gradients with respect to the filters")
        x = input.reshape(tuple(input.shape[1:-1])).contiguous()
        w = torch.flip(filters.reshape(tuple(filters.shape[:-2])), dims=tuple(range(dim))).contiguous()
        # jax.lax.conv_transpose(VALID) returns n s + max(K - s, 0) entries per axis: (n - 1) s + K for K >= s (every call
        # of the reference), zero-extended when the kernel is shorter than the stride
        oshape = [n * s + max(k - s, 0) for n, s, k in zip(x.shape, strides, w.shape)]
        res = _ConvValidFn.apply(x, w, strides, True, oshape)
        res = res.reshape((1,) + tuple(res.shape) + (1,))
        if output_shape is not None and tuple(int(v) for v in output_shape) != tuple(res.shape):
            raise ValueError("mod.conv_transpose: output_shape {} but the VALID result is {}".format(tuple(output_shape), tuple(res.shape)))
        return res

    # -- random (host generator: reproducible across devices) -----------------------
    def _set_seed(self, seed):
        self._gen.manual_seed(int(seed))
        np.random.seed(int(seed) % (1 << 32))

    def _uniform(self, shape, minval, maxval, dtype):
        u = torch.rand(self._shape(shape), generator=self._gen, dtype=torch.float64)
        return self._t(minval + (maxval - minval) * u, dtype)

    def _normal(self, shape, mean=0, stddev=1, dtype=None):
        u = torch.randn(self._shape(shape), generator=self._gen, dtype=torch.float64)
        return self._t(mean + stddev * u, dtype or np.float32)


def _unary(name):
    f = getattr(torch, name)

    def g(self, x):
        return f(self._t(x))

    g.__name__ = name
    return g


def _binary(name):
    f = getattr(torch, name)

    def g(self, a, b):
        a = self._t(a)
        b = self._t(b)
        if a.dtype != b.dtype:
            b = b.to(a.dtype)
        return f(a, b)

    g.__name__ = name
    return g


for _name in ["abs", "cos", "sin", "exp", "square", "sqrt", "tanh", "log", "floor"]:
    setattr(ModRocm, _name, _unary(_name))
for _name in ["minimum", "maximum"]:
    setattr(ModRocm, _name, _binary(_name))

# The reference exports these two names (reference src/odil/__init__.py:8-12); user code
# that type-checks against them keeps working.
ModBase = ModRocm
ModNumpy = ModRocm
