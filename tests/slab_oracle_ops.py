"""Test double of `odil_amd.ops` for the slab driver: the same call signatures, computed by
the NumPy oracle on CPU tensors.  Lets the world_size-2 gloo test exercise the slab
ALGORITHM (ghost layout, exchanges, cut flags, loss partials) without a GPU."""

import numpy as np
import torch

from oracle import odil_np as onp


def _np(t):
    return t.detach().cpu().numpy()


def interp_add(coarse, loc, add=None, coarse_scale=1.0, add_scale=1.0, out=None):
    res = onp.interp_to_finer(_np(coarse) * coarse_scale if coarse_scale != 1.0 else _np(coarse), loc)
    if add is not None:
        res = _np(add) * add_scale + res if add_scale != 1.0 else _np(add) + res
    res = torch.from_numpy(np.ascontiguousarray(res))
    if out is None:
        return res
    out.copy_(res)
    return out


def interp_adj(gfine, loc, cshape, scale=None, out=None, cut=(False, False)):
    res = torch.from_numpy(np.ascontiguousarray(onp.interp_to_finer_adj(_np(gfine), loc, tuple(cshape), cut=cut)))
    if out is None:
        return res
    out.copy_(res)
    return out


def poisson_residual(u, rhs, h2, fu=None, loss=None, want_fu=True, zrange=None, denom=None):
    dw = [np.sqrt(h) for h in h2]
    f = onp.poisson_residual(_np(u), _np(rhs), dw)
    if zrange is None:
        val = np.mean(np.square(f))
    else:
        val = np.sum(np.square(f[zrange[0] : zrange[1]])) / denom
    f = torch.from_numpy(np.ascontiguousarray(f))
    if fu is not None:
        fu.copy_(f)
    else:
        fu = f
    if loss is None:
        loss = torch.zeros((), dtype=u.dtype)
    loss.fill_(float(val))
    return fu, loss


def poisson_adjoint(fu, h2, scale, out=None):
    dw = [np.sqrt(h) for h in h2]
    g = torch.from_numpy(np.ascontiguousarray(onp.poisson_adjoint(_np(fu) * scale, dw)))
    if out is None:
        return g
    out.copy_(g)
    return out


def adam_step(x, m, v, g, alpha, one_minus_b1, one_minus_b2, eps):
    xn, mn, vn, gn = _np(x), _np(m), _np(v), _np(g)
    mn += (gn - mn) * one_minus_b1
    vn += (np.square(gn) - vn) * one_minus_b2
    xn -= (mn * alpha) / (np.sqrt(vn) + eps)


class PlaneList:
    """CPU double of odil_amd.ops.PlaneList (odil_planes_copy): same descriptors, index arithmetic in torch."""

    def __init__(self, planes, device, start=0):
        idx, self.start, self.count = [], int(start), int(start)
        for base, outer, ostride, inner in planes:
            o = torch.arange(outer, dtype=torch.int64).view(-1, 1) * ostride
            idx.append((base + o + torch.arange(inner, dtype=torch.int64).view(1, -1)).reshape(-1))
            self.count += outer * inner
        self.index = torch.cat(idx) if idx else torch.zeros(0, dtype=torch.int64)

    def pack(self, arr, out=None):
        res = arr.index_select(0, self.index)
        if out is not None:
            out[self.start: self.count].copy_(res)
            return out
        assert self.start == 0
        return res

    def unpack(self, arr, buf):
        arr.index_copy_(0, self.index, buf[self.start: self.count])

    def unpack_add(self, arr, buf):
        arr.index_add_(0, self.index, buf[self.start: self.count])


def poisson_jacobi(u, rhs, h2, omega, out):
    """out = u - omega (A u - rhs) / diag(A) (odil_poisson_jacobi): diag from the oracle's coefficient arrays."""
    dw = [np.sqrt(h) for h in h2]
    un, bn = _np(u), _np(rhs)
    diag = onp.poisson_jac_coeffs(un.shape, dw)[(0,) * un.ndim]
    res = un - omega * (onp.poisson_residual(un, bn, dw)) / diag
    out.copy_(torch.from_numpy(np.ascontiguousarray(res)))
    return out


def restrict_to_coarser(u, loc, depth=1):
    res = _np(u)
    for _ in range(depth):
        res = onp.restrict_to_coarser(res, loc)
    return torch.from_numpy(np.ascontiguousarray(res))


# ---- the variable-coefficient multigrid kernels (csrc/stencil_mg.hip), by their NumPy restatement tests/stencil_gmg_np.py ------
def _coeff_list(coeffs):
    return [np.asarray(c, dtype=np.float64) for c in _np(coeffs)]


def stencil_var_smooth(coeffs, x, b, omega, out):
    import stencil_gmg_np as sg

    out.copy_(torch.from_numpy(np.ascontiguousarray(sg.jacobi(_coeff_list(coeffs), _np(x), _np(b), omega))))
    return out


def stencil_var_residual(coeffs, x, b, out=None):
    import stencil_gmg_np as sg

    res = torch.from_numpy(np.ascontiguousarray(_np(b) - sg.apply(_coeff_list(coeffs), _np(x))))
    if out is None:
        return res
    out.copy_(res)
    return out


def stencil_var_coarsen(coeffs, halve=None):
    import stencil_gmg_np as sg

    assert halve is None or all(halve)
    return torch.from_numpy(np.ascontiguousarray(np.stack(sg.coarsen(_coeff_list(coeffs)))))
