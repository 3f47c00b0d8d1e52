"""Tensor-level wrappers of the HIP kernels (torch = device memory + streams only).

Each function allocates its output with torch, passes raw device pointers through
the C-ABI (`_lib.call`) on the current HIP stream and returns tensors.  Names and
semantics follow the reference functions cited in include/odil_hip.h.
"""

import math
from ctypes import c_double, c_int, c_int64, c_void_p

import torch

from . import _lib
from ._lib import call, host_reals, i64, ptr, ptr_array, stream_ptr

_workspace = {}


def reduce_workspace(device, nquant=1):
    """Per-device scratch for the deterministic reductions (f64 partial sums)."""
    key = (str(device), nquant)
    ws = _workspace.get(key)
    if ws is None:
        ws = torch.empty(_lib.reduce_workspace_elems() * nquant, dtype=torch.float64, device=device)
        _workspace[key] = ws
    return ws


def fine_shape(cshape, loc):
    return tuple({"c": 2 * s, "n": 2 * s - 1, ".": s}[l] for s, l in zip(cshape, loc))


def coarse_shape(fshape, loc):
    """Output shape of restrict_to_coarser (stride-2 VALID on every axis)."""
    return tuple((s - 2) // 2 + 1 if l == "c" else (s - 1) // 2 + 1 for s, l in zip(fshape, loc))


def _lead_stride(t):
    """The leading stride of a 4-D view that is contiguous but for that stride (a level array of the slab paths without
    its outer ghost planes), else None."""
    if t.dim() != 4 or t.is_contiguous() or t.shape[0] < 2:
        return None
    inner = t[0]
    return int(t.stride(0)) if inner.is_contiguous() and t.stride(0) >= inner.numel() else None


def interp_add(coarse, loc, add=None, coarse_scale=1.0, add_scale=1.0, out=None):
    """out = add_scale*add + P(coarse_scale*coarse)  (reference core.py:606-700, :258-262).  `coarse` may be a 4-D view
    with a leading stride larger than its volume: the marching kernels read it in place, other layouts get a copy."""
    fshape = fine_shape(coarse.shape, loc)
    if out is None:
        out = torch.empty(fshape, dtype=coarse.dtype, device=coarse.device)
    assert tuple(out.shape) == fshape, (out.shape, fshape)
    if add is not None:
        assert tuple(add.shape) == fshape and add.dtype == coarse.dtype
    if not coarse.is_contiguous():
        ld = _lead_stride(coarse)
        if ld is not None and call(
            "interp_add_ld", coarse.dtype, c_void_p(coarse.data_ptr()), c_int64(ld), ptr(add), ptr(out), i64(coarse.shape),
            c_int(coarse.dim()), loc.encode(), coarse_scale, add_scale, stream_ptr(), unserved_ok=True,
        ):
            return out
        coarse = coarse.contiguous()
    call(
        "interp_add", coarse.dtype, ptr(coarse), ptr(add), ptr(out), i64(coarse.shape), c_int(coarse.dim()),
        loc.encode(), coarse_scale, add_scale, stream_ptr(),
    )
    return out


def interp_to_finer(u, loc, depth=1):
    for _ in range(depth):
        u = interp_add(u, loc)
    return u


def interp_adj(gfine, loc, cshape, scale=None, out=None, cut=(False, False)):
    """P^T gfine (and optionally scale * P^T gfine).  cut=(lo, hi): axis 0 is an interior slab
    interface at that end (ghost planes), not a wall."""
    cshape = tuple(int(s) for s in cshape)
    assert tuple(gfine.shape) == fine_shape(cshape, loc), (gfine.shape, cshape, loc)
    if out is None:
        out = torch.empty(cshape, dtype=gfine.dtype, device=gfine.device)
    if not out.is_contiguous():
        # a view with a leading stride larger than its volume is written in place where the kernels can; else through
        # a contiguous result
        assert scale is None and not any(cut), "a strided result takes no scaled copy and no cut"
        ld = _lead_stride(out)
        if ld is None or not call("interp_adj_ld", gfine.dtype, ptr(gfine), c_void_p(out.data_ptr()), c_int64(ld), i64(cshape),
                                  c_int(len(cshape)), loc.encode(), stream_ptr(), unserved_ok=True):
            out.copy_(interp_adj(gfine, loc, cshape))
        return out
    scaled = None
    if scale is not None:
        scaled = torch.empty_like(out)
    call(
        "interp_adj_cut", gfine.dtype, ptr(gfine), ptr(out), ptr(scaled), i64(cshape), c_int(len(cshape)),
        loc.encode(), 1.0 if scale is None else float(scale), c_int(1 if cut[0] else 0), c_int(1 if cut[1] else 0),
        stream_ptr(),
    )
    return out if scale is None else (out, scaled)


def _step_size(alpha, dtype):
    """(host value, device pointer or None): `alpha` may be a one-element device tensor -- the step size
    is then read from memory by the kernel (an epoch captured into a hipGraph replays with new values)."""
    if isinstance(alpha, torch.Tensor):
        assert alpha.numel() == 1 and alpha.dtype == dtype
        return 0.0, ptr(alpha)
    return float(alpha), None


def interp_adj_adam(gfine, loc, cshape, out, x, m, v, alpha, one_minus_b1, one_minus_b2, eps, cut=(False, False)):
    """out = P^T gfine and the Adam step of (x, m, v) (arrays of out's shape) with that gradient."""
    cshape = tuple(int(s) for s in cshape)
    assert tuple(out.shape) == cshape == tuple(x.shape) == tuple(m.shape) == tuple(v.shape)
    a, adev = _step_size(alpha, gfine.dtype)
    call(
        "interp_adj_cut_adam", gfine.dtype, ptr(gfine), ptr(out), i64(cshape), c_int(len(cshape)), loc.encode(),
        c_int(1 if cut[0] else 0), c_int(1 if cut[1] else 0), ptr(x), ptr(m), ptr(v), a,
        float(one_minus_b1), float(one_minus_b2), float(eps), adev, stream_ptr(),
    )
    return out


def restrict_to_coarser(u, loc, depth=1):
    """Full weighting (reference core.py:703-755)."""
    for _ in range(depth):
        out = torch.empty(coarse_shape(u.shape, loc), dtype=u.dtype, device=u.device)
        call("restrict", u.dtype, ptr(u), ptr(out), i64(u.shape), c_int(u.dim()), loc.encode(), stream_ptr())
        u = out
    return u


def conv_valid(x, w, strides, transposed=False, out_shape=None):
    """Strided VALID correlation of `x` with the small dense kernel `w` (same rank), or its transpose (see
    include/odil_hip.h: odil_conv_valid).  `out_shape` only for the transpose: a longer output than (n - 1) s + K."""
    dim = x.dim()
    strides = [int(s) for s in strides]
    assert w.dim() == dim and len(strides) == dim and w.dtype == x.dtype
    if transposed:
        oshape = [(n - 1) * s + k for n, s, k in zip(x.shape, strides, w.shape)]
        if out_shape is not None:
            assert all(a >= b for a, b in zip(out_shape, oshape)), (out_shape, oshape)
            oshape = [int(v) for v in out_shape]
    else:
        oshape = [(n - k) // s + 1 for n, s, k in zip(x.shape, strides, w.shape)]
    out = torch.empty(oshape, dtype=x.dtype, device=x.device)
    call("conv_valid", x.dtype, ptr(x), ptr(w), ptr(out), i64(x.shape), i64(w.shape), i64(strides), i64(oshape), c_int(dim),
         c_int(1 if transposed else 0), stream_ptr())
    return out


def stencil_var_smooth(coeffs, x, b, omega, out):
    """out = x - omega (A x - b) / c0 for the (2 d + 1)-point operator with coefficient arrays `coeffs` [(2 d + 1), *shape]
    (include/odil_hip.h: odil_stencil_var_smooth, mode 0).  x = None: from the zero vector (x' = omega b / c0: two words)."""
    assert coeffs.shape[0] == 2 * b.dim() + 1 and tuple(coeffs.shape[1:]) == tuple(b.shape) == tuple(out.shape) and out is not x
    call("stencil_var_smooth", b.dtype, ptr(coeffs), ptr(x), ptr(b), ptr(out), i64(b.shape), c_int(b.dim()), float(omega),
         c_int(0), stream_ptr())
    return out


def smooth2_supported(shape):
    return 1 <= len(shape) <= 3 and shape[-1] % 2 == 0 and min(shape) >= 2


def stencil_var_smooth2(coeffs, x, b, omega1, omega2, out, zc_hint=0):
    """Two sweeps of `stencil_var_smooth` (weights omega1, then omega2) in ONE pass over the coefficient arrays;
    bit-identical to two calls (out is not x).  x = None: from the zero vector (x is not read)."""
    assert coeffs.shape[0] == 2 * b.dim() + 1 and tuple(coeffs.shape[1:]) == tuple(b.shape) == tuple(out.shape)
    assert x is None or (x.shape == b.shape and out.data_ptr() != x.data_ptr() and x.is_contiguous())
    assert smooth2_supported(tuple(b.shape)) and coeffs.is_contiguous() and b.is_contiguous()
    call("stencil_var_smooth2", b.dtype, ptr(coeffs), ptr(x), ptr(b), ptr(out), i64(b.shape), c_int(b.dim()), float(omega1),
         float(omega2), c_int(zc_hint), stream_ptr())
    return out


def stencil_var_residual(coeffs, x, b, out=None):
    """b - A x (mode 1 of odil_stencil_var_smooth)."""
    assert coeffs.shape[0] == 2 * x.dim() + 1 and tuple(coeffs.shape[1:]) == tuple(x.shape)
    out = torch.empty_like(x) if out is None else out
    call("stencil_var_smooth", x.dtype, ptr(coeffs), ptr(x), ptr(b), ptr(out), i64(x.shape), c_int(x.dim()), 0.0, c_int(1),
         stream_ptr())
    return out


def stencil_var_residual_restrict(coeffs, x, b, scale, out, loss, zrange=None, denom=0.0):
    """out = scale * (sum over the 2^d children of b - A x), loss <- mean((A x - b)^2), one pass.  zrange = (z0, z1): the
    slab form -- loss = sum over those planes of (A x - b)^2 / denom."""
    assert tuple(out.shape) == tuple(s // 2 for s in x.shape) and out.is_contiguous()
    if zrange is None:  # (loss = None: no reduction launch)
        call("stencil_var_residual_restrict", x.dtype, ptr(coeffs), ptr(x), ptr(b), ptr(out), i64(x.shape), c_int(x.dim()),
             float(scale), ptr(reduce_workspace(x.device)), ptr(loss), stream_ptr())
    else:
        call("stencil_var_residual_restrict_slab", x.dtype, ptr(coeffs), ptr(x), ptr(b), ptr(out), i64(x.shape),
             c_int(x.dim()), float(scale), c_int64(int(zrange[0])), c_int64(int(zrange[1])), c_double(float(denom)),
             ptr(reduce_workspace(x.device)), ptr(loss), stream_ptr())
    return out


def narrow_scale(x64, y32, a=1.0, msq=None):
    """y32 = (a / sqrt(msq)) * x64 in one pass (msq: 0-d float64 device tensor or None)."""
    assert x64.dtype == torch.float64 and y32.dtype == torch.float32 and x64.numel() == y32.numel()
    lib = _lib.load()
    status = lib.odil_narrow_scale(ptr(x64), ptr(y32), c_int64(x64.numel()), float(a), ptr(msq), stream_ptr())
    if status != 0:
        raise _lib.OdilHipError("odil_narrow_scale: {}".format(lib.odil_last_error().decode()))
    return y32


def widen_axpy(y64, x32, a=1.0, msq=None):
    """y64 += (a * sqrt(msq)) * x32 in one pass."""
    assert y64.dtype == torch.float64 and x32.dtype == torch.float32 and y64.numel() == x32.numel()
    lib = _lib.load()
    status = lib.odil_widen_axpy(ptr(y64), ptr(x32), c_int64(y64.numel()), float(a), ptr(msq), stream_ptr())
    if status != 0:
        raise _lib.OdilHipError("odil_widen_axpy: {}".format(lib.odil_last_error().decode()))
    return y64


def max_abs_diff(a, b):
    """2-element device tensor (max |a - b|, max |b|), one pass over both arrays."""
    assert a.numel() == b.numel() and a.dtype == b.dtype
    out = torch.empty(2, dtype=a.dtype, device=a.device)
    call("max_abs_diff", a.dtype, ptr(a), ptr(b), c_int64(a.numel()), ptr(reduce_workspace(a.device)), ptr(out), stream_ptr())
    return out


def max_abs_rows(a):
    """max |row| of every row of a 2-D view (<= 64 rows) as a device tensor: two launches for all rows."""
    a2 = a.reshape(a.shape[0], -1)
    assert a2.is_contiguous() and 1 <= a2.shape[0] <= 64
    out = torch.empty(a2.shape[0], dtype=a.dtype, device=a.device)
    call("max_abs_rows", a.dtype, ptr(a2), c_int(a2.shape[0]), c_int64(a2.shape[1]), ptr(reduce_workspace(a.device)), ptr(out),
         stream_ptr())
    return out


def stencil_var_coarsen(coeffs, halve=None):
    """Coefficient arrays of the coarse-grid operator [(2 d + 1), *(shape / 2)] (csrc/stencil_mg.hip).  halve: per axis,
    whether two cells are merged along it (default: every axis); the others keep their extent (semi-coarsening)."""
    import ctypes

    shape = tuple(coeffs.shape[1:])
    halve = [True] * len(shape) if halve is None else [bool(h) for h in halve]
    out = torch.empty((coeffs.shape[0],) + tuple(s // 2 if h else s for s, h in zip(shape, halve)), dtype=coeffs.dtype,
                      device=coeffs.device)
    if all(halve):
        call("stencil_var_coarsen", coeffs.dtype, ptr(coeffs), ptr(out), i64(shape), c_int(len(shape)), stream_ptr())
    else:
        mask = (ctypes.c_int * len(shape))(*[int(h) for h in halve])
        call("stencil_var_coarsen_axes", coeffs.dtype, ptr(coeffs), ptr(out), i64(shape), c_int(len(shape)),
             ctypes.cast(mask, ctypes.c_void_p), stream_ptr())
    return out


def stencil_vcycle_tail(coeffs, shapes, halve, xin, b, xout, work, inv, wpre, wpost, fmg=False):
    """The coarse tail of a multigrid cycle in one launch (include/odil_hip.h: odil_stencil_vcycle_tail).  coeffs: one
    flat tensor holding the coefficient arrays of every tail level; shapes: their extents, finest first; halve: per
    transition and axis, whether cell pairs are merged; xin None: zero start."""
    import ctypes

    nlev, ndim = len(shapes), len(shapes[0])
    flat = [int(n) for shape in shapes for n in shape]
    mask = (ctypes.c_int * max(1, (nlev - 1) * ndim))(*[int(bool(h)) for tr in halve for h in tr])
    wa, wap = host_reals(list(wpre) or [0.0], b.dtype)
    wb, wbp = host_reals(list(wpost) or [0.0], b.dtype)
    call("stencil_vcycle_tail", b.dtype, ptr(coeffs), i64(flat), ctypes.cast(mask, ctypes.c_void_p), c_int(nlev), c_int(ndim),
         ptr(xin), ptr(b), ptr(xout), ptr(work), ctypes.c_int64(work.numel()), ptr(inv), c_int(inv.shape[0]), wap,
         c_int(len(wpre)), wbp, c_int(len(wpost)), c_int(1 if fmg else 0), stream_ptr())
    return xout


def stencil_vcycle_tail_plan(coeffs, shapes, halve, work, inv, wpre, wpost):
    """`stencil_vcycle_tail` with everything that does not change between calls prepared once (a cycle visits its tail
    every 0.3 ms: building the argument arrays anew cost ~30 us of host time per visit, which the GPU waited for right
    after each cycle's read-back): -> launch(xin, b, xout, fmg)."""
    import ctypes

    nlev, ndim = len(shapes), len(shapes[0])
    shapes64 = i64([int(n) for shape in shapes for n in shape])
    mask = (ctypes.c_int * max(1, (nlev - 1) * ndim))(*[int(bool(h)) for tr in halve for h in tr])
    maskp = ctypes.cast(mask, ctypes.c_void_p)
    wa, wap = host_reals(list(wpre) or [0.0], coeffs.dtype)
    wb, wbp = host_reals(list(wpost) or [0.0], coeffs.dtype)
    fixed = (ptr(coeffs), shapes64, maskp, c_int(nlev), c_int(ndim))
    rest = (ptr(work), ctypes.c_int64(work.numel()), ptr(inv), c_int(inv.shape[0]), wap, c_int(len(wpre)), wbp, c_int(len(wpost)))
    keep = (coeffs, work, inv, mask, wa, wb, shapes64)  # (the arrays the pointers point into)
    dtype = coeffs.dtype

    def launch(xin, b, xout, fmg=False):
        call("stencil_vcycle_tail", dtype, *fixed, ptr(xin), ptr(b), ptr(xout), *rest, c_int(1 if fmg else 0), stream_ptr())
        return xout

    launch.keep = keep
    return launch


def restrict_adj(gcoarse, loc, fshape):
    """R^T gcoarse (cotangent of restrict_to_coarser) for a fine array of shape `fshape`."""
    fshape = tuple(int(s) for s in fshape)
    assert tuple(gcoarse.shape) == coarse_shape(fshape, loc), (gcoarse.shape, fshape, loc)
    out = torch.empty(fshape, dtype=gcoarse.dtype, device=gcoarse.device)
    call("restrict_adj", gcoarse.dtype, ptr(gcoarse), ptr(out), i64(fshape), c_int(len(fshape)), loc.encode(),
         stream_ptr())
    return out


def _shapes_flat(tensors):
    flat = []
    for t in tensors:
        flat += list(t.shape)
    return i64(flat)


def mg_synth(terms, loc, factors=None, work=None, out=None):
    """u = sum_l P^l (f_l w_l)  (reference core.py:245-263)."""
    nlvl = len(terms)
    dtype, device = terms[0].dtype, terms[0].device
    if out is None:
        out = torch.empty_like(terms[0])
    if work is None:
        work = [None] + [torch.empty_like(t) for t in terms[1:-1]] + [None]
        work = work[:nlvl]
    fac = host_reals(factors, dtype) if factors is not None else (None, None)
    call(
        "mg_synth", dtype, ptr_array(terms), fac[1], ptr_array(work), ptr(out), _shapes_flat(terms), c_int(nlvl),
        c_int(terms[0].dim()), loc.encode(), stream_ptr(),
    )
    return out


def _two_step_adjoint(gu, shapes, loc, nontrivial):
    return (gu.dim() == 4 and loc == "nccc" and len(shapes) >= 2 and not nontrivial
            and gu.numel() * gu.element_size() >= (64 << 20) and all(s % 2 == 0 for s in gu.shape[1:]))


def interp_adj_best(gfine, loc, cshape, out=None):
    """P^T of ONE level through the fastest route for the layout: large 'nccc' arrays as space part ('.ccc', batched
    march kernel) then time part ('n...'), as mg_synth_adj does for its first level; interp_adj otherwise."""
    cshape = tuple(int(s) for s in cshape)
    if _two_step_adjoint(gfine, [tuple(gfine.shape), cshape], loc, False):
        space = interp_adj(gfine, "." + loc[1:], (gfine.shape[0],) + cshape[1:])
        return interp_adj(space, "n...", cshape, out=out)
    return interp_adj(gfine, loc, cshape, out=out)


def mg_synth_adj(gu, shapes, loc, factors=None, grads=None):
    """[f_l (P^T)^l gu]  (cotangent of mg_synth)."""
    nlvl = len(shapes)
    dtype, device = gu.dtype, gu.device
    nontrivial = factors is not None and any(float(f) != 1.0 for f in factors)
    if _two_step_adjoint(gu, shapes, loc, nontrivial):
        # 4-D space-time layout 'nccc', large: P^T of the first (dominant) level as the product of
        # its space part (the batched 'ccc' march kernel, every fine value read ONCE) and its time
        # part (a stream over an array 8x smaller).  The one-kernel version holds three windows of
        # plane sums per thread and re-reads the odd fine volumes (0.5 TB/s in float); the joint
        # ghost rule couples cell axes only, so the transpose factorises exactly over the node axis.
        if grads is None:
            grads = [gu] + [torch.empty(tuple(s), dtype=dtype, device=device) for s in shapes[1:]]
        elif grads[0].data_ptr() != gu.data_ptr():
            grads[0].copy_(gu)
        space = interp_adj(gu, "." + loc[1:], (gu.shape[0],) + tuple(shapes[1][1:]))
        interp_adj(space, "n...", tuple(shapes[1]), out=grads[1])
        if nlvl > 2:
            mg_synth_adj(grads[1], shapes[1:], loc, grads=grads[1:])
        return grads
    if grads is None:
        first = torch.empty_like(gu) if (nontrivial and float(factors[0]) != 1.0) else gu
        grads = [first] + [torch.empty(tuple(s), dtype=dtype, device=device) for s in shapes[1:]]
    work = [None] * nlvl
    if nontrivial:
        work = [None] + [torch.empty(tuple(s), dtype=dtype, device=device) for s in shapes[1:]]
    fac = host_reals(factors, dtype) if factors is not None else (None, None)
    flat = []
    for s in shapes:
        flat += list(s)
    call(
        "mg_synth_adj", dtype, ptr(gu), ptr_array(grads), fac[1], ptr_array(work), i64(flat), c_int(nlvl),
        c_int(gu.dim()), loc.encode(), stream_ptr(),
    )
    return grads


def mg_synth_adj_adam(gu, shapes, loc, grads, x, m, v, alpha, one_minus_b1, one_minus_b2, eps):
    """P^T chain with the Adam update of levels >= 1 inside the launches that form their
    gradients (factors == 1).  x, m, v: level arrays (entry 0 is left alone)."""
    nlvl = len(shapes)
    flat = []
    for s in shapes:
        flat += list(s)
    none0 = lambda arrs: ptr_array([None] + list(arrs[1:]))
    a, adev = _step_size(alpha, gu.dtype)
    call(
        "mg_synth_adj_adam", gu.dtype, ptr(gu), ptr_array(grads), None, ptr_array([None] * nlvl), i64(flat),
        c_int(nlvl), c_int(gu.dim()), loc.encode(), none0(x), none0(m), none0(v), a, float(one_minus_b1),
        float(one_minus_b2), float(eps), adev, stream_ptr(),
    )
    return grads


def field_gather(src, field_loc, shift=None, loc=None):
    """Context.field access (reference core.py:955-969): pad, periodic roll, trim."""
    loc = loc or field_loc
    ndim = src.dim()
    shift = tuple(shift) if shift else (0,) * ndim
    oshape = tuple(
        s + (1 if (lf == "c" and l == "n") else 0) - (1 if (lf == "n" and l == "c") else 0)
        for s, lf, l in zip(src.shape, field_loc, loc)
    )
    out = torch.empty(oshape, dtype=src.dtype, device=src.device)
    call(
        "field_gather", src.dtype, ptr(src), ptr(out), i64(src.shape), c_int(ndim), field_loc.encode(), loc.encode(),
        i64(shift), stream_ptr(),
    )
    return out


def field_scatter(g, src_shape, field_loc, shift=None, loc=None, out=None):
    """Transpose of field_gather; accumulates into `out` if given."""
    loc = loc or field_loc
    ndim = len(src_shape)
    shift = tuple(shift) if shift else (0,) * ndim
    accumulate = out is not None
    if out is None:
        out = torch.empty(tuple(src_shape), dtype=g.dtype, device=g.device)
    call(
        "field_scatter", g.dtype, ptr(g), ptr(out), i64(src_shape), c_int(ndim), field_loc.encode(), loc.encode(),
        i64(shift), c_int(1 if accumulate else 0), stream_ptr(),
    )
    return out


def mean_reduce(x, square=True, out=None):
    """mean(x**2) or mean(x) as a 0-d device tensor, deterministic (reference core.py:1093)."""
    x = x.contiguous()
    if out is None:
        out = torch.empty((), dtype=x.dtype, device=x.device)
    call(
        "mean_reduce", x.dtype, ptr(x), c_int64(x.numel()), c_int(1 if square else 0),
        ptr(reduce_workspace(x.device)), ptr(out), stream_ptr(),
    )
    return out


def poisson_residual(u, rhs, h2, fu=None, loss=None, want_fu=True, zrange=None, denom=None):
    """fu = Lap(u) - rhs with zero-Dirichlet ghosts, loss = mean(fu**2)
    (reference examples/poisson/poisson.py:89-113, core.py:1093).  zrange=(z0, z1) / denom: slab
    variant, loss = sum over planes z0 <= z < z1 of fu^2 / denom."""
    assert u.shape == rhs.shape and u.dtype == rhs.dtype
    if fu is None and want_fu:
        fu = torch.empty_like(u)
    if loss is None:
        loss = torch.empty((), dtype=u.dtype, device=u.device)
    h2a, h2p = host_reals(h2, u.dtype)
    if zrange is None:
        call(
            "poisson_residual", u.dtype, ptr(u), ptr(rhs), ptr(fu), i64(u.shape), c_int(u.dim()), h2p,
            ptr(reduce_workspace(u.device)), ptr(loss), stream_ptr(),
        )
    else:
        call(
            "poisson_residual_slab", u.dtype, ptr(u), ptr(rhs), ptr(fu), i64(u.shape), c_int(u.dim()), h2p,
            c_int64(zrange[0]), c_int64(zrange[1]), c_double(float(denom)), ptr(reduce_workspace(u.device)), ptr(loss),
            stream_ptr(),
        )
    return fu, loss


def residual_restrict_supported(shape, dtype):
    pack = 2 if dtype == torch.float64 else 4
    return len(shape) == 3 and shape[0] % 2 == 0 and shape[1] % 2 == 0 and shape[2] % pack == 0 and min(shape) >= 2


def poisson_residual_restrict(u, rhs, h2, scale, out, loss, zrange=None, denom=0.0):
    """out = scale * (sum over 2x2x2 fine cells of (A u - rhs)) on the next coarser grid, loss = mean((A u -
    rhs)**2), in one pass over u and rhs (3-D; see residual_restrict_supported).  zrange = (z0, z1): the slab form --
    loss = sum over those planes of (A u - rhs)**2 / denom."""
    assert u.shape == rhs.shape and tuple(out.shape) == tuple(n // 2 for n in u.shape)
    assert residual_restrict_supported(tuple(u.shape), u.dtype) and out.is_contiguous()
    h2a, h2p = host_reals(h2, u.dtype)
    if zrange is None:  # (loss = None: no reduction launch)
        call("poisson_residual_restrict", u.dtype, ptr(u), ptr(rhs), ptr(out), i64(u.shape), c_int(3), h2p, float(scale),
             ptr(reduce_workspace(u.device)), ptr(loss) if loss is not None else None, stream_ptr())
    else:
        call("poisson_residual_restrict_slab", u.dtype, ptr(u), ptr(rhs), ptr(out), i64(u.shape), c_int(3), h2p,
             float(scale), c_int64(int(zrange[0])), c_int64(int(zrange[1])), c_double(float(denom)),
             ptr(reduce_workspace(u.device)), ptr(loss), stream_ptr())
    return out, loss


def poisson_jacobi(u, rhs, h2, omega, out):
    """out = u - omega (A u - rhs) / diag(A): one damped-Jacobi sweep of the Poisson stencil (out is not u).
    u = None: the sweep starts from the zero vector (nothing is read for it, nothing need be zeroed)."""
    assert rhs.shape == out.shape and (u is None or (u.shape == rhs.shape and out.data_ptr() != u.data_ptr()))
    h2a, h2p = host_reals(h2, rhs.dtype)
    call("poisson_jacobi", rhs.dtype, ptr(u) if u is not None else None, ptr(rhs), ptr(out), i64(rhs.shape),
         c_int(rhs.dim()), h2p, float(omega), stream_ptr())
    return out


def poisson_small_epochs(x, m, v, g, u, fu, rhs, shapes, h2, alphas, omb1, omb2, eps, losses, norms):
    """alphas.numel() whole epochs of the 1-D / 2-D multigrid Poisson problem in one launch (include/odil_hip.h:
    odil_poisson_small_epochs).  x, m, v, g, u: packed flat vectors of all levels; alphas, losses, norms: device tensors."""
    flat = [int(n) for shape in shapes for n in shape]
    h2a, h2p = host_reals(h2, x.dtype)
    call("poisson_small_epochs", x.dtype, ptr(x), ptr(m), ptr(v), ptr(g), ptr(u), ptr(fu), ptr(rhs), i64(flat),
         c_int(len(shapes)), c_int(len(shapes[0])), h2p, ptr(alphas), c_int(alphas.numel()), float(omb1), float(omb2),
         float(eps), ptr(losses), ptr(norms), ptr(reduce_workspace(x.device)), stream_ptr())
    return losses


def jacobi2_supported(shape, dtype):
    """Arrays whose rows are whole 16-byte packs (odil_poisson_jacobi2, odil_stencil_var_smooth2)."""
    pack = 2 if dtype == torch.float64 else 4
    return 1 <= len(shape) <= 3 and shape[-1] % pack == 0 and min(shape) >= 2


def poisson_jacobi2(u, rhs, h2, omega1, omega2, out, zc_hint=0):
    """Two damped-Jacobi sweeps (weights omega1, then omega2) in ONE pass: bit-identical to two calls of
    `poisson_jacobi`, 3 words per cell instead of 6 (out is not u).  u = None: from the zero vector (2 words)."""
    assert rhs.shape == out.shape and jacobi2_supported(tuple(rhs.shape), rhs.dtype)
    assert u is None or (u.shape == rhs.shape and out.data_ptr() != u.data_ptr() and u.is_contiguous())
    assert rhs.is_contiguous() and out.is_contiguous()
    h2a, h2p = host_reals(h2, rhs.dtype)
    call("poisson_jacobi2", rhs.dtype, ptr(u) if u is not None else None, ptr(rhs), ptr(out), i64(rhs.shape),
         c_int(rhs.dim()), h2p, float(omega1), float(omega2), c_int(zc_hint), stream_ptr())
    return out


def poisson_jacobi2_synth(coarse, x, rhs, h2, omega1, omega2, out, zc_hint=0):
    """Two sweeps (weights omega1, omega2) of u = x + P coarse in ONE pass: the coarse-grid correction of a V-cycle and
    its post-smoothing; bit-identical to `interp_add` + `poisson_jacobi2` (out is not x; see jacobi_synth_supported)."""
    assert coarse.dim() == 3 and tuple(x.shape) == tuple(2 * s for s in coarse.shape) and x.shape == rhs.shape == out.shape
    assert coarse.is_contiguous() and x.is_contiguous() and rhs.is_contiguous() and out.data_ptr() != x.data_ptr()
    h2a, h2p = host_reals(h2, x.dtype)
    call("poisson_jacobi2_synth", x.dtype, ptr(coarse), ptr(x), ptr(rhs), ptr(out), i64(coarse.shape), h2p, float(omega1),
         float(omega2), c_int(zc_hint), stream_ptr())
    return out


def jacobi_synth_supported(shape, dtype):
    """3-D arrays with even extents >= 4 (the coarse array they are prolongated from has extents >= 2)."""
    return len(shape) == 3 and all(s % 2 == 0 and s >= 4 for s in shape)


def poisson_jacobi_synth(coarse, x, rhs, h2, omega, out):
    """out = u - omega (A u - rhs) / diag(A) with u = x + P coarse formed in registers: the coarse-grid correction of
    a V-cycle and its first post-smoothing sweep in one pass (out is not x)."""
    assert coarse.dim() == 3 and tuple(x.shape) == tuple(2 * s for s in coarse.shape) and x.shape == rhs.shape == out.shape
    assert coarse.is_contiguous() and x.is_contiguous() and rhs.is_contiguous() and out.data_ptr() != x.data_ptr()
    h2a, h2p = host_reals(h2, x.dtype)
    call("poisson_jacobi_synth", x.dtype, ptr(coarse), ptr(x), ptr(rhs), ptr(out), i64(coarse.shape), h2p, float(omega),
         stream_ptr())
    return out


def poisson_residual_synth(coarse, w0, rhs, h2, fu=None, loss=None, zrange=None, denom=None):
    """fu = Lap(w0 + P coarse) - rhs, loss = mean(fu**2): the residual with the last prolongation of
    the multigrid synthesis fused in (u is never stored).  3-D cell-centred arrays, w0.shape == 2 * coarse.shape.
    zrange=(z0, z1) / denom: slab variant, loss = sum over fine planes z0 <= z < z1 of fu^2 / denom."""
    from ctypes import c_double

    assert coarse.dim() == 3 and tuple(w0.shape) == tuple(2 * s for s in coarse.shape) and w0.shape == rhs.shape
    assert coarse.is_contiguous() and w0.is_contiguous() and rhs.is_contiguous()
    if fu is None:
        fu = torch.empty_like(w0)
    if loss is None:
        loss = torch.empty((), dtype=w0.dtype, device=w0.device)
    h2a, h2p = host_reals(h2, w0.dtype)
    call(
        "poisson_residual_synth", w0.dtype, ptr(coarse), ptr(w0), ptr(rhs), ptr(fu), i64(coarse.shape), h2p,
        c_int64(zrange[0] if zrange else 0), c_int64(zrange[1] if zrange else -1), c_double(float(denom or 0.0)),
        ptr(reduce_workspace(w0.device)), ptr(loss), stream_ptr(),
    )
    return fu, loss


def poisson_adjoint(fu, h2, scale, out=None):
    """gu = J^T (scale * fu)."""
    if out is None:
        out = torch.empty_like(fu)
    h2a, h2p = host_reals(h2, fu.dtype)
    call("poisson_adjoint", fu.dtype, ptr(fu), ptr(out), i64(fu.shape), c_int(fu.dim()), h2p, float(scale), stream_ptr())
    return out


def poisson_adjoint_adam(fu, h2, scale, out, x, m, v, alpha, one_minus_b1, one_minus_b2, eps):
    """gu = J^T (scale * fu) written to `out`, and the Adam step of (x, m, v) with that gradient,
    in the same launch (x, m, v: arrays of fu's shape, updated in place)."""
    assert x.shape == fu.shape and m.numel() == v.numel() == fu.numel()
    h2a, h2p = host_reals(h2, fu.dtype)
    a, adev = _step_size(alpha, fu.dtype)
    call(
        # (out=None: the gradient is consumed by the update, not stored)
        "poisson_adjoint_adam", fu.dtype, ptr(fu), ptr(out), ptr(x), ptr(m), ptr(v), i64(fu.shape), c_int(fu.dim()),
        h2p, float(scale), a, float(one_minus_b1), float(one_minus_b2), float(eps), adev, stream_ptr(),
    )
    return out


def adjoint_transpose_supported(shape):
    """Sizes at which the one-pass stencil adjoint + first transposed prolongation pays (and is defined)."""
    return len(shape) == 3 and all(n % 2 == 0 for n in shape) and shape[0] >= 8 and shape[1] >= 32 and shape[2] >= 128


def poisson_adjoint_transpose(fu, h2, scale, g1, g0=None, adam0=None, adam1=None, alpha=0.0, one_minus_b1=0.0,
                              one_minus_b2=0.0, eps=0.0, cut=(False, False)):
    """g0 = J^T (scale * fu) (stored only if `g0` is given), g1 = P^T g0, and the Adam steps of the finest level
    (adam0 = (x, m, v)) and of the next one (adam1) inside the same launch: g0 never goes through memory.
    cut=(lo, hi): that end of axis 0 is a slab interface (ghost planes), not a wall."""
    assert fu.dim() == 3 and tuple(g1.shape) == tuple(n // 2 for n in fu.shape) and fu.is_contiguous()
    h2a, h2p = host_reals(h2, fu.dtype)
    a, adev = _step_size(alpha, fu.dtype)
    a0 = adam0 if adam0 is not None else (None, None, None)
    a1 = adam1 if adam1 is not None else (None, None, None)
    for t in list(a0) + list(a1):
        assert t is None or (t.is_contiguous() and t.dtype == fu.dtype)
    call(
        "poisson_adjoint_transpose_adam", fu.dtype, ptr(fu), ptr(g0), ptr(g1), i64(fu.shape), h2p, float(scale),
        ptr(a0[0]), ptr(a0[1]), ptr(a0[2]), ptr(a1[0]), ptr(a1[1]), ptr(a1[2]), a, float(one_minus_b1),
        float(one_minus_b2), float(eps), adev, c_int(1 if cut[0] else 0), c_int(1 if cut[1] else 0), stream_ptr(),
    )
    return g1


def poisson_jac_match(arrays, shape, h2):
    """[(2 d + 1), 2] device tensor: per coefficient array (centre, -e_0, +e_0, ...) max |a_k - e_k| and max |e_k| against the
    Poisson Jacobian's values, formed on the fly in ONE pass (include/odil_hip.h: odil_poisson_jac_match)."""
    from ._lib import ptr_array

    ndim = len(shape)
    assert len(arrays) == 2 * ndim + 1 and all(a.numel() == math.prod(shape) for a in arrays)
    arrays = [a.reshape(-1) if a.is_contiguous() else a.reshape(-1).contiguous() for a in arrays]
    dtype = arrays[0].dtype
    h2a, h2p = host_reals(h2, dtype)
    out = torch.empty((2 * ndim + 1, 2), dtype=dtype, device=arrays[0].device)
    call("poisson_jac_match", dtype, ptr_array(arrays), i64(shape), c_int(ndim), h2p, ptr(reduce_workspace(out.device)), ptr(out),
         stream_ptr())
    return out


def poisson_jac_coeffs(shape, h2, dtype, device):
    """(2*ndim+1, *shape) coefficient arrays [centre, -1 ax0, +1 ax0, ...] (reference core.py:1313-1361)."""
    ndim = len(shape)
    out = torch.empty((2 * ndim + 1,) + tuple(shape), dtype=dtype, device=device)
    h2a, h2p = host_reals(h2, dtype)
    call("poisson_jac_coeffs", dtype, ptr(out), i64(shape), c_int(ndim), h2p, stream_ptr())
    return out


_dense_ws = {}


def dense_xty(x, y):
    """X^T Y of two tall, skinny matrices (rows x <= 64 columns each, row-major, any row stride) on the matrix
    cores: v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, deterministic two-stage reduction.  The dense block of
    the Newton normal equations (reference core.py:1189-1203, linsolver.py:17-23)."""
    assert x.dim() == 2 and y.dim() == 2 and x.shape[0] == y.shape[0] and x.dtype == y.dtype
    assert x.stride(1) == 1 and y.stride(1) == 1
    key = (str(x.device), x.dtype)
    ws = _dense_ws.get(key)
    if ws is None:
        ws = _dense_ws[key] = torch.empty(_lib.load().odil_dense_block_workspace_bytes() // 8, dtype=torch.float64,
                                          device=x.device).view(x.dtype)
    out = torch.empty((x.shape[1], y.shape[1]), dtype=x.dtype, device=x.device)
    call("dense_block_xty", x.dtype, c_void_p(x.data_ptr()), c_void_p(y.data_ptr()), c_int64(x.shape[0]),
         c_int(x.shape[1]), c_int(y.shape[1]), c_int64(x.stride(0)), c_int64(y.stride(0)), ptr(out), ptr(ws), stream_ptr())
    return out


def adam_step(x, m, v, g, alpha, one_minus_b1, one_minus_b2, eps):
    """In-place AdamNativeOptimizer._step on flat vectors (reference optimizer.py:311-319)."""
    assert x.numel() == m.numel() == v.numel() == g.numel()
    a, adev = _step_size(alpha, x.dtype)
    call(
        "adam_step", x.dtype, ptr(x), ptr(m), ptr(v), ptr(g), c_int64(x.numel()), a, float(one_minus_b1),
        float(one_minus_b2), float(eps), adev, stream_ptr(),
    )


def adam_step_pieces(x, m, v, g, npieces, stride, offset, count, alpha, one_minus_b1, one_minus_b2, eps):
    """adam_step on the elements o * stride + offset + j (o < npieces, j < count) of four flat vectors."""
    assert x.numel() == m.numel() == v.numel() == g.numel() and x.is_contiguous()
    assert npieces == 0 or (npieces - 1) * stride + offset + count <= x.numel()
    a, adev = _step_size(alpha, x.dtype)
    call(
        "adam_step_pieces", x.dtype, ptr(x), ptr(m), ptr(v), ptr(g), c_int64(npieces), c_int64(stride), c_int64(offset),
        c_int64(count), a, float(one_minus_b1), float(one_minus_b2), float(eps), adev, stream_ptr(),
    )


def axpy(y, x, a):
    """y += a * x in place."""
    assert y.numel() == x.numel() and y.dtype == x.dtype
    call("axpy", y.dtype, ptr(y), ptr(x), c_int64(y.numel()), float(a), stream_ptr())
    return y


def scale(x, a, adev=None, out=None):
    """out = a * (adev[0] if adev is given) * x."""
    x = x.contiguous()
    if out is None:
        out = torch.empty_like(x)
    call("scale", x.dtype, ptr(x), ptr(out), c_int64(x.numel()), float(a), ptr(adev), stream_ptr())
    return out


def addcmul(y, a, b, accumulate=True):
    """y (+)= a * b elementwise, in place."""
    assert y.numel() == a.numel() == b.numel()
    call("addcmul", y.dtype, ptr(y), ptr(a), ptr(b), c_int64(y.numel()), c_int(1 if accumulate else 0), stream_ptr())
    return y


_dots_ws = {}


def dots_workspace(device, nvec):
    key = str(device)
    ws = _dots_ws.get(key)
    need = _lib.load().odil_dots_workspace_bytes(nvec) // 8
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float64, device=device)
        _dots_ws[key] = ws
    return ws


def dots(a, b, out=None):
    """out[k] = <a[k], b> for a of shape (nvec, n); deterministic, f64 accumulation."""
    if a.dim() == 1:
        a = a[None]
    nvec, n = a.shape
    assert b.numel() == n and a.stride(1) == 1
    if out is None:
        out = torch.empty(nvec, dtype=a.dtype, device=a.device)
    call(
        "dots", a.dtype, ptr_strided(a), c_int64(a.stride(0)), c_int(nvec), ptr(b), c_int64(n),
        ptr(dots_workspace(a.device, nvec)), ptr(out), stream_ptr(),
    )
    return out


def dots3(a, bs, out=None):
    """out[j, k] = <a[k], bs[j]> for up to three vectors bs, reading a once."""
    if a.dim() == 1:
        a = a[None]
    nvec, n = a.shape
    bs = list(bs) + [None] * (3 - len(bs))
    assert a.stride(1) == 1 and all(b is None or b.numel() == n for b in bs)
    if out is None:
        out = torch.empty((3, nvec), dtype=a.dtype, device=a.device)
    call(
        "dots3", a.dtype, ptr_strided(a), c_int64(a.stride(0)), c_int(nvec), ptr(bs[0]), ptr(bs[1]), ptr(bs[2]),
        c_int64(n), ptr(dots_workspace(a.device, 3 * nvec)), ptr(out), stream_ptr(),
    )
    return out


def lbfgs_probe(g, d, out):
    """out[:3] = (<g, d>, <g, g>, max |g|) in one pass over g."""
    assert g.numel() == d.numel() and g.dtype == d.dtype == out.dtype and out.numel() >= 3
    call("lbfgs_probe", g.dtype, ptr(g), ptr(d), c_int64(g.numel()), ptr(dots_workspace(g.device, 3)), ptr(out),
         stream_ptr())
    return out


def ptr_strided(a):
    import ctypes

    if not a.is_cuda:
        raise _lib.OdilHipError("tensor is on '{}': the HIP kernels need device memory".format(a.device))
    return ctypes.c_void_p(a.data_ptr())


def lincomb(y, beta, a, coef):
    """y = beta*y + sum_k coef[k] * a[k]  (coef: device tensor)."""
    if a.dim() == 1:
        a = a[None]
    nvec, n = a.shape
    assert y.numel() == n and coef.numel() == nvec and (a.stride(1) == 1 or n == 1), (tuple(y.shape), tuple(a.shape), a.stride())
    call(
        "lincomb", y.dtype, ptr(y), float(beta), ptr_strided(a), c_int64(a.stride(0)), c_int(nvec), ptr(coef),
        c_int64(n), stream_ptr(),
    )
    return y


def stencil_apply(coeffs, shifts, x, transpose=False, out=None):
    """y = M x or M^T x for the stencil matrix with per-shift coefficient arrays
    (reference core.py:1144-1171)."""
    nshift = len(shifts)
    ndim = x.dim()
    if out is None:
        out = torch.empty_like(x)
    flat = []
    for s in shifts:
        flat += list(s)
    call(
        "stencil_apply", x.dtype, ptr(coeffs), i64(flat), c_int(nshift), ptr(x), ptr(out), i64(x.shape), c_int(ndim),
        c_int(1 if transpose else 0), stream_ptr(),
    )
    return out


def stencil_march(coeffs, shifts, diag, b, axis, direction, out=None):
    """x with M x = b for a stencil matrix triangular along `axis` (see odil_stencil_march in the header)."""
    if out is None:
        out = torch.empty_like(b)
    flat = []
    for s in shifts:
        flat += list(s)
    call("stencil_march", b.dtype, ptr(coeffs), i64(flat), c_int(len(shifts)), c_int(diag), ptr(b), ptr(out),
         i64(b.shape), c_int(b.dim()), c_int(axis), c_int(direction), stream_ptr())
    return out


class PlaneList:
    """Planes of arrays inside one packed vector and their places in a contiguous message buffer (see
    odil_planes_copy in the header).  planes: [(base, outer, ostride, inner)] in message order; start: where the first
    plane begins in the message (lists over different arrays can share one message); `count`: the message length
    up to the end of this list's planes."""

    def __init__(self, planes, device, start=0):
        rows, off, self.max_count = [], int(start), 1
        for base, outer, ostride, inner in planes:
            rows.append((int(base), int(outer), int(ostride), int(inner), off))
            off += int(outer) * int(inner)
            self.max_count = max(self.max_count, int(outer) * int(inner))
        self.count, self.n = off, len(rows)
        self.vec_ok = int(all(r[3] % 4 == 0 for r in rows))
        self.descs = torch.tensor(rows or [[0] * 5], dtype=torch.int64, device=device).contiguous()

    def _run(self, arr, buf, mode):
        assert buf.numel() >= self.count and buf.dtype == arr.dtype and arr.dim() == 1
        if self.n:
            call("planes_copy", arr.dtype, ptr(arr), ptr(buf), ptr(self.descs), c_int(self.n), c_int64(self.max_count),
                 c_int(self.vec_ok), c_int(mode), stream_ptr())
        return buf

    def pack(self, arr, out=None):
        """The planes of `arr` (flat packed vector) as one contiguous message."""
        if out is None:
            out = torch.empty(self.count, dtype=arr.dtype, device=arr.device)
        return self._run(arr, out, 0)

    def unpack(self, arr, buf):
        self._run(arr, buf, 1)

    def unpack_add(self, arr, buf):
        self._run(arr, buf, 2)


def csr_assemble(coeffs, shifts, shape, col_offset=0):
    """(indptr, indices, data) of the stencil matrix (reference core.py:1144-1171)."""
    nshift = len(shifts)
    size = math.prod(shape)
    dev = coeffs.device
    indptr = torch.empty(size + 1, dtype=torch.int64, device=dev)
    indices = torch.empty(size * nshift, dtype=torch.int64, device=dev)
    data = torch.empty(size * nshift, dtype=coeffs.dtype, device=dev)
    flat = []
    for s in shifts:
        flat += list(s)
    call(
        "csr_assemble", coeffs.dtype, ptr(coeffs), i64(flat), c_int(nshift), i64(shape), c_int(len(shape)),
        c_int64(col_offset), ptr(indptr), ptr(indices), ptr(data), stream_ptr(),
    )
    return indptr, indices, data
