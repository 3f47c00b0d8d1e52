"""`mod.convolution` / `mod.conv_transpose` (reference src/odil/backend.py:112-126, :165-172) on the HIP tap kernel
`odil_conv_valid`: a user operator that calls them directly -- as the reference's own transfers do (core.py:656-662,
:744-751) -- must get the reference's values.  The two transfer formulas are written out here ON THE `mod` NAMES (pad,
kron weights, convolution / conv_transpose, edge slices: the arithmetic of core.py:636-668 and :733-751) and held to the
fixtures the reference's own functions produced (`restrict.npz`, `interp_conv.npz`, tests/golden/make_golden.py):
values, depth 2 and the cotangents through autograd; plus random kernels / strides against torch's CPU convolutions."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def mod():
    import odil_amd as odil

    return odil.runtime.get_mod()


def rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def restrict_by_mod(u, loc, mod, depth=1):
    """reference core.py:733-751 on `mod.pad` + `mod.convolution`."""
    for _ in range(depth):
        dim = u.dim()
        pad_width = [(1, 1) if l == "n" else (0, 0) for l in loc]
        upad = 2 * mod.pad(u, pad_width=pad_width, mode="symmetric") - mod.pad(u, pad_width=pad_width, mode="reflect")
        wloc = {"n": np.array([1, 2, 1]) * 0.25, "c": np.array([1, 1]) * 0.5, ".": np.array([1.0])}
        w = wloc[loc[0]]
        for i in range(1, dim):
            w = np.kron(wloc[loc[i]], w[..., None])
        u = mod.convolution(upad, filters=mod.cast(w, u.dtype), strides=2, padding="VALID")
    return u


def interp_by_mod(u, loc, mod, depth=1):
    """reference core.py:636-668 (method='conv') on `mod.pad` + `mod.conv_transpose`."""
    for _ in range(depth):
        dim = u.dim()
        pad_width = [(1, 1) if l == "c" else (0, 0) for l in loc]
        upad = 2 * mod.pad(u, pad_width=pad_width, mode="symmetric") - mod.pad(u, pad_width=pad_width, mode="reflect")
        wloc = {"n": np.array([1, 2, 1]) * 0.5, "c": np.array([1, 3, 3, 1]) * 0.25, ".": np.array([1.0])}
        w = wloc[loc[0]]
        for i in range(1, dim):
            w = np.kron(wloc[loc[i]], w[..., None])
        w = mod.cast(mod.reshape(w, w.shape + (1, 1)), u.dtype)
        oshape = (1,) + tuple({"n": s * 2 + 1, "c": s * 2 + 2, ".": s}[l] for l, s in zip(loc, upad.shape)) + (1,)
        strides = tuple(1 if l == "." else 2 for l in loc)
        res = mod.conv_transpose(mod.reshape(upad, (1,) + tuple(upad.shape) + (1,)), filters=w, output_shape=oshape,
                                 strides=strides, padding="VALID")
        oslice = {"n": slice(1, -1), "c": slice(3, -3), ".": slice(0, None)}
        u = res[(0,) + tuple(oslice[l] for l in loc) + (0,)]
    return u


def locs_of(g):
    return sorted({k.split("/")[0] for k in g.files if k.endswith("/u")})


def test_restriction_written_on_mod_convolution_gives_the_reference_values(dev, mod):
    g = load_golden("restrict")
    for loc in locs_of(g):
        u = torch.as_tensor(g[f"{loc}/u"], device=dev).requires_grad_(True)
        coarse = restrict_by_mod(u, loc, mod)
        assert rel(coarse, g[f"{loc}/coarse"]) < 2e-15, loc
        (gu,) = torch.autograd.grad(coarse, u, torch.as_tensor(g[f"{loc}/gcoarse"], device=dev))
        assert rel(gu, g[f"{loc}/gu"]) < 2e-15, loc
        if f"{loc}/coarse2" in g.files:
            assert rel(restrict_by_mod(u.detach(), loc, mod, depth=2), g[f"{loc}/coarse2"]) < 4e-15, loc


def test_prolongation_written_on_mod_conv_transpose_gives_the_reference_values(dev, mod):
    g = load_golden("interp_conv")
    for loc in locs_of(g):
        u = torch.as_tensor(g[f"{loc}/u"], device=dev).requires_grad_(True)
        fine = interp_by_mod(u, loc, mod)
        assert rel(fine, g[f"{loc}/fine"]) < 4e-15, loc
        (gu,) = torch.autograd.grad(fine, u, torch.as_tensor(g[f"{loc}/gfine"], device=dev))
        assert rel(gu, g[f"{loc}/gu"]) < 4e-15, loc
        if f"{loc}/fine2" in g.files:
            assert rel(interp_by_mod(u.detach(), loc, mod, depth=2), g[f"{loc}/fine2"]) < 8e-15, loc


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-14), (torch.float32, 2e-6)])
@pytest.mark.parametrize("shape,kshape,strides", [
    ((37,), (3,), (2,)), ((9, 14), (2, 4), (2, 1)), ((7, 8, 9), (3, 2, 4), (1, 2, 3)), ((5, 6, 7), (1, 1, 1), (1, 1, 1)),
    ((4, 4, 4), (4, 4, 4), (2, 2, 2)), ((130, 70), (4, 4), (2, 2))])
def test_random_kernels_against_torch_cpu_convolutions(dev, mod, shape, kshape, strides, dtype, tol):
    """values and cotangents of both names; the checker is torch's conv / conv_transpose on the CPU (the same
    cross-correlation the reference-producing shim uses, tests/golden/ref_shim.py:266-297)."""
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(shape, generator=gen, dtype=dtype)
    w = torch.randn(kshape, generator=gen, dtype=dtype)
    dim = len(shape)
    conv = {1: torch.nn.functional.conv1d, 2: torch.nn.functional.conv2d, 3: torch.nn.functional.conv3d}[dim]
    convt = {1: torch.nn.functional.conv_transpose1d, 2: torch.nn.functional.conv_transpose2d,
             3: torch.nn.functional.conv_transpose3d}[dim]
    xr = x.clone().requires_grad_(True)
    want = conv(xr.reshape((1, 1) + shape), w.reshape((1, 1) + kshape), stride=strides)[0, 0]
    gy = torch.randn(want.shape, generator=gen, dtype=dtype)
    (gwant,) = torch.autograd.grad(want, xr, gy)
    xd = x.to(dev).requires_grad_(True)
    got = mod.convolution(xd, w.to(dev), strides, "VALID")
    assert tuple(got.shape) == tuple(want.shape)
    assert rel(got, want.detach().numpy()) < tol
    (ggot,) = torch.autograd.grad(got, xd, gy.to(dev))
    assert rel(ggot, gwant.numpy()) < tol
    # conv_transpose: torch's with the kernel flipped == jax.lax.conv_transpose(transpose_kernel=False)
    xr = x.clone().requires_grad_(True)
    wf = torch.flip(w, dims=tuple(range(dim)))
    want = convt(xr.reshape((1, 1) + shape), wf.reshape((1, 1) + kshape), stride=strides)[0, 0]
    gy = torch.randn(want.shape, generator=gen, dtype=dtype)
    (gwant,) = torch.autograd.grad(want, xr, gy)
    xd = x.to(dev).requires_grad_(True)
    got = mod.conv_transpose(xd.reshape((1,) + shape + (1,)), w.to(dev).reshape(kshape + (1, 1)), strides=strides, padding="VALID")
    assert tuple(got.shape) == (1,) + tuple(want.shape) + (1,)
    assert rel(got[0, ..., 0], want.detach().numpy()) < tol
    (ggot,) = torch.autograd.grad(got, xd, gy.to(dev).reshape(got.shape))
    assert rel(ggot, gwant.numpy()) < tol


def test_conv_transpose_with_a_kernel_shorter_than_the_stride_has_the_jax_extent(dev, mod):
    """jax.lax.conv_transpose(VALID), which the reference's `mod.conv_transpose` is (backend.py:165-172), returns
    n s + max(K - s, 0) entries per axis: n s when the kernel is shorter than the stride (zeros behind the last tap),
    (n - 1) s + K otherwise."""
    x = torch.arange(1.0, 6.0, dtype=torch.float64, device=dev).requires_grad_(True)
    w = torch.tensor([2.0], dtype=torch.float64, device=dev)
    got = mod.conv_transpose(x.reshape(1, 5, 1), w.reshape(1, 1, 1), strides=2, padding="VALID")
    assert tuple(got.shape) == (1, 10, 1)
    want = torch.zeros(10, dtype=torch.float64)
    want[0::2] = 2.0 * torch.arange(1.0, 6.0, dtype=torch.float64)
    assert torch.equal(got[0, :, 0].detach().cpu(), want)
    (gx,) = torch.autograd.grad(got, x, torch.ones_like(got))
    assert torch.equal(gx.cpu(), torch.full((5,), 2.0, dtype=torch.float64))
    # a 2-D case mixing both regimes: K = 1 < s = 3 on the first axis, K = 4 >= s = 2 on the second
    x2 = torch.randn((3, 4), dtype=torch.float64, generator=torch.Generator().manual_seed(2))
    w2 = torch.randn((1, 4), dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    got2 = mod.conv_transpose(x2.to(dev).reshape(1, 3, 4, 1), w2.to(dev).reshape(1, 4, 1, 1), strides=(3, 2), padding="VALID")
    assert tuple(got2.shape) == (1, 9, 10, 1)
    ref = torch.nn.functional.conv_transpose2d(x2.reshape(1, 1, 3, 4), torch.flip(w2, dims=(0, 1)).reshape(1, 1, 1, 4),
                                               stride=(3, 2))[0, 0]   # (7, 10): torch stops at the last tap
    assert rel(got2[0, :7, :, 0], ref.numpy()) < 1e-15 and float(got2[0, 7:, :, 0].abs().max()) == 0.0


def test_unsupported_forms_are_refused_not_approximated(dev, mod):
    x = torch.zeros((8, 8), dtype=torch.float64, device=dev)
    with pytest.raises(NotImplementedError):
        mod.convolution(x, torch.ones((2, 2), dtype=torch.float64, device=dev), 2, "SAME")
    with pytest.raises(Exception):
        mod.convolution(x, torch.ones((5, 5), dtype=torch.float64, device=dev), 1, "VALID")  # kernel extent > 4
