"""Geometric multigrid for the Newton step of Poisson-type operators on the device.

The reference solves the Newton system through the normal equations with SuperLU
(`linsolver.solve(..., "direct")`, reference src/odil/linsolver.py:17-26) or, with
`--linsolver multigrid`, pyamg's smoothed aggregation + CG (linsolver.py:61-72).  Neither
scales to the 1.3e8 unknowns of the 512^3 configuration.  When the linearised operator is the
(square, nonsingular) zero-Dirichlet Laplacian stencil -- recognised from its coefficient
arrays, as in fused.py -- `M^T M d = -M^T r` and `M d = -r` have the same solution, and the
latter is solved here by V-cycles built only from the HIP kernels of the hot path:

  smoother      damped Jacobi: x' = x - omega (A x - b) / diag    odil_poisson_jacobi (one pass, 3 words per cell)
  restriction   cell-centred full weighting (mean of 2^d cells)    odil_restrict
  prolongation  x += P(x_c), the multigrid-decomposition P         odil_interp_add
  coarse grids  the same stencil re-discretised with h_l = 2^l h   odil_poisson_jac_coeffs (diag)

Convergence is checked on the true residual with the deterministic dot products.
"""

import math

import numpy as np
import torch

from . import ops


def skewed(shape, dtype, device, k, zero=False):
    """A work array whose base address is moved k x 4 KB off its allocation.  The arrays of a 512^3 float64 level are
    exactly 2^30 bytes each; allocated back to back they start 2^30 bytes apart, and the streams of a sweep (x, b, x') then
    walk the same memory channels in step: 0.66 against 0.61 ms for a sweep, 0.81 against 0.77 for a pair
    (tools/mb_alias.py: any offset from 256 B up removes it)."""
    n = math.prod(shape)
    pad = (k % 16) * 4096 // torch.empty((), dtype=dtype).element_size()
    flat = (torch.zeros if zero else torch.empty)(n + pad, dtype=dtype, device=device)
    return flat[pad:pad + n].view(shape)


class PoissonGMG:
    nu_default = (2, 2)  # pre- / post-smoothing sweeps of a cycle

    def __init__(self, shape, h2, dtype, device, omega=None, nu1=None, nu2=None, min_size=2, lite=False):
        """lite: the finest level's operator only, no work arrays of the cycle -- the float64 half of a mixed-precision
        solve (`solve_mixed`), which needs this object for the residual alone."""
        self.ndim = len(shape)
        self.loc = "c" * self.ndim
        self.dtype, self.device = dtype, device
        npdt = np.float64 if dtype == torch.float64 else np.float32
        self.omega = omega if omega is not None else {1: 2.0 / 3.0, 2: 4.0 / 5.0, 3: 6.0 / 7.0}[self.ndim]
        self.nu1 = self.nu_default[0] if nu1 is None else nu1
        self.nu2 = self.nu_default[1] if nu2 is None else nu2
        self.shapes, self.h2s = [tuple(shape)], [[npdt(v) for v in h2]]
        # SEMI-coarsening while the cells are far from cubes: point smoothing only damps what oscillates along the strongly
        # coupled axes (the small spacings), so only those are halved -- the axes whose h^2 is within a factor 2 of the
        # smallest -- until the spacings meet; from there on (and from the start on a grid of cubes) every axis is halved.
        # 1024 x 64 on the unit square: (512, 64), (256, 64), (128, 64), (64, 64), (32, 32), ...  With full coarsening the
        # cycles lost their rate at cells 1 : 2 (24 passes with the Krylov hand-over), 1 : 4 (70 - 110) and failed at 1 : 16.
        self.locs = []  # per transition lvl -> lvl + 1: 'c' on the halved axes, '.' on the others
        while True:
            cur, h = self.shapes[-1], self.h2s[-1]
            hmin = min(float(v) for v in h)
            halve = [float(v) <= 2.0 * hmin for v in h]
            if not all(n % 2 == 0 and n // 2 >= min_size for n, on in zip(cur, halve) if on):
                break
            self.shapes.append(tuple(n // 2 if on else n for n, on in zip(cur, halve)))
            self.h2s.append([v * npdt(4) if on else v for v, on in zip(h, halve)])
            self.locs.append("".join("c" if on else "." for on in halve))
        self.nlvl = len(self.shapes)
        mk = lambda s: torch.zeros(s, dtype=dtype, device=device)
        self.loss = mk(())
        self._coarse_inv = None
        self._continuation = None
        self._r = [None] * self.nlvl                         # residuals (only where the fused restriction cannot be used)
        if lite:
            return
        self.x = [None] + [skewed(s, dtype, device, 2, zero=True) for s in self.shapes[1:]]   # coarse corrections
        self.b = [None] + [skewed(s, dtype, device, 3, zero=True) for s in self.shapes[1:]]   # coarse right-hand sides
        self.spare = [skewed(s, dtype, device, 1) for s in self.shapes]  # target of a sweep / prolongation

    residual_sign = 1.0  # `residual` returns A x - b

    def coarse_inverse(self):
        """Inverse of the coarsest-grid operator (at most a few dozen unknowns), built once from the
        residual kernel applied to unit vectors: the coarsest solve is then one product (odil_dots)
        instead of dozens of launch-bound sweeps."""
        if self._coarse_inv is None:
            shape = self.shapes[-1]
            n = math.prod(shape)
            eye = torch.eye(n, dtype=self.dtype, device=self.device)
            zero = torch.zeros(shape, dtype=self.dtype, device=self.device)
            cols = [ops.poisson_residual(eye[j].view(shape), zero, self.h2s[-1])[0].reshape(-1) for j in range(n)]
            amat = torch.stack(cols, dim=1).cpu().numpy().astype(np.float64)  # column j = A e_j
            inv = np.linalg.inv(amat)
            self._coarse_inv = torch.as_tensor(inv, dtype=self.dtype).to(self.device).contiguous()
        return self._coarse_inv

    def r(self, lvl):
        if self._r[lvl] is None:
            self._r[lvl] = torch.empty(self.shapes[lvl], dtype=self.dtype, device=self.device)
        return self._r[lvl]

    def residual(self, lvl, x, b, out):
        """out = A x - b."""
        ops.poisson_residual(x, b, self.h2s[lvl], fu=out, loss=self.loss)
        return out

    def smooth(self, lvl, x, b, n, chebyshev=True, zero=False):
        """n Jacobi sweeps, each ONE kernel (odil_poisson_jacobi: x' = x - omega_k (A x - b) / diag); the
        iterate ping-pongs between `x` and the level's spare buffer.  Returns the tensor holding it.
        The weights omega_k are those of the degree-n Chebyshev polynomial on [1/d, 2], the part of the
        spectrum of D^-1 A that the coarse grid cannot see (eigenvalues (1/d) sum_i (1 - cos theta_i) with
        some |theta_i| >= pi/2): two sweeps damp it by 0.34 (d = 3) where omega = 6/7 gives 0.51, three by
        0.15 instead of 0.36.  zero: the iterate is the zero vector and `x` only a buffer -- its content is not read."""
        return self.sweeps(lvl, x, b, self.weights(n, chebyshev), zero=zero)

    def weights(self, n, chebyshev=True):
        if not chebyshev:
            return [self.omega] * n
        lo, hi = 1.0 / self.ndim, 2.0
        mid, half = 0.5 * (hi + lo), 0.5 * (hi - lo)
        return [1.0 / (mid - half * math.cos(math.pi * (2 * k + 1) / (2 * n))) for k in range(n)]

    def sweeps(self, lvl, x, b, weights, zero=False):
        """Sweeps with the given weights, in PAIRS through the one-pass kernel (odil_poisson_jacobi2: the intermediate
        iterate stays on the CU, 3 words per cell and pair instead of 6; bit-identical to two single sweeps) on levels
        large enough to be bandwidth-bound.  zero: the iterate is the zero vector (every coarse level of a cycle starts
        there): the first launch does not read `x` -- the same bits as from an array of zeros, which nobody has to write."""
        weights = list(weights)
        if zero and (not weights or not self.zero_start):
            x.zero_()
            zero = False
        # (float64 only: with four floats per lane the pair is bound by the vector ALU, 0.85 against 0.70 ms at 512^3)
        pair = (self.dtype == torch.float64 and ops.jacobi2_supported(self.shapes[lvl], self.dtype)
                and math.prod(self.shapes[lvl]) >= self.pair_min_cells)
        while weights:
            y = self.spare[lvl]
            src = None if zero else x
            zero = False
            if pair and len(weights) >= 2:
                ops.poisson_jacobi2(src, b, self.h2s[lvl], weights[0], weights[1], out=y)
                weights = weights[2:]
            else:
                ops.poisson_jacobi(src, b, self.h2s[lvl], weights[0], out=y)
                weights = weights[1:]
            self.spare[lvl] = x
            x = y
        return x

    post_pair = True
    zero_start = True  # (False: the zero iterate of a coarse level is written and read back -- the tests compare both, bit for bit)
    pair_min_cells = 128**3  # below: the levels are launch-bound and the single-sweep kernels' smaller workgroups fill the chip better (measured: 128^3 / 64^3 / 32^3 as the threshold -> 25.0 / 25.7 / 26.6 ms for the 256^3 diffusion step)

    # ---- the coarse tail in one launch --------------------------------------------------------------------------------
    tail_max_cells = 8192  # levels of at most this many cells form the tail (odil_stencil_vcycle_tail); 0: off

    def tail_coeffs(self, lvl):
        """Coefficient arrays [(2 d + 1), *shape] of level `lvl` (what the one-launch tail works on)."""
        return ops.poisson_jac_coeffs(self.shapes[lvl], self.h2s[lvl], self.dtype, self.device)

    def tail(self):
        """(first tail level, flat coefficient tensor, work tensor) when the hierarchy ends in levels small enough to be
        walked by ONE workgroup -- every level from the first with <= tail_max_cells cells (never the finest) down to a
        coarsest grid of <= 512 unknowns with its dense inverse --, else None."""
        key = (self.tail_max_cells, self.nu1, self.nu2)
        cached = self.__dict__.get("_tail")
        if cached is not None and cached[0] == key:
            return cached[1]
        res = None
        first = next((l for l in range(1, self.nlvl) if math.prod(self.shapes[l]) <= self.tail_max_cells), None)
        if first is not None:
            first = max(first, self.nlvl - 8)  # (at most eight levels per launch: long 1-D hierarchies enter it lower down)
        if (self.tail_max_cells and first is not None and math.prod(self.shapes[-1]) <= 512
                and self.nu1 <= 4 and self.nu2 <= 4 and self.tail_transfers_ok(first)):
            flat = torch.cat([self.tail_coeffs(l).reshape(-1) for l in range(first, self.nlvl)])
            work = torch.empty(3 * sum(math.prod(s) for s in self.shapes[first:]), dtype=self.dtype, device=self.device)
            outs = [torch.empty(self.shapes[first], dtype=self.dtype, device=self.device) for _ in range(2)]
            res = (first, flat, work, outs)
        self.__dict__["_tail"] = (key, res)
        return res

    def tail_transfers_ok(self, first):
        """(the rediscretised hierarchy restricts with P^T / 2^k on semi-coarsened transitions, the tail with the mean of the
        children: the same only where every axis is halved)"""
        return all(loc == self.loc for loc in self.locs[first:])

    def tail_cycle(self, lvl, x, b, fmg=False):
        """One V-cycle (fmg: the nested-iteration start) on level `lvl` = the first tail level, in one launch."""
        first, flat, work, outs = self.tail()
        # (two result buffers in turn: a result may be the start iterate of the next call -- the second cycle on the first
        # coarse level of the variable-coefficient hierarchy -- and is otherwise consumed before the call after the next)
        out = outs[0] if x is None or x.data_ptr() != outs[0].data_ptr() else outs[1]
        launch = self.__dict__.get("_tail_launch")
        if launch is None or launch.keep[0] is not flat:  # (the launch's constant arguments, prepared once)
            launch = self._tail_launch = ops.stencil_vcycle_tail_plan(
                flat, self.shapes[first:], [[c == "c" for c in loc] for loc in self.locs[first:]], work, self.coarse_inverse(),
                self.weights(self.nu1), self.weights(self.nu2))
        return launch(x, b, out, fmg)

    def coarse_rhs(self, lvl, x, b):
        """b_{lvl+1} = R (b - A x), and mean((A x - b)^2) in self.loss.  One fused pass in 3-D (the fine
        residual is never stored); residual, restriction and sign as three launches otherwise."""
        bc = self.b[lvl + 1]
        if self.locs[lvl] == self.loc and ops.residual_restrict_supported(self.shapes[lvl], self.dtype):
            # (the norm is the SOLVE's measure on the finest level; a coarser level's is nobody's: no reduction launch)
            ops.poisson_residual_restrict(x, b, self.h2s[lvl], -1.0 / 2**self.ndim, bc, self.loss if lvl == 0 else None)
        else:
            r = self.residual(lvl, x, b, self.r(lvl))
            self.restrict(lvl, r, -1.0, out=bc)
        return bc

    def restrict(self, lvl, r, sign=1.0, out=None):
        """sign * R r onto level lvl + 1.  Every axis halved: the mean of the children (`odil_restrict`).  Some axes only:
        the transposed interpolation scaled to unit row sums, P^T / 2^k (`odil_restrict` SUBSAMPLES a '.' axis, as the
        reference's strided convolution does, core.py:745-751: not a restriction along it)."""
        loc = self.locs[lvl]
        if loc == self.loc:
            rc = ops.restrict_to_coarser(r, loc)
            return rc if sign == 1.0 and out is None else ops.scale(rc, sign, out=out)
        return ops.scale(ops.interp_adj(r, loc, self.shapes[lvl + 1]), sign / 2 ** loc.count("c"), out=out)

    def finish_cycle(self, lvl, x, b, post=True):
        """Second half of a V(nu1, nu2) cycle: `x` is pre-smoothed and b_{lvl+1} holds its restricted
        residual.  Coarse-grid correction and post-smoothing (post=False: none -- the caller smooths next anyway);
        returns the tensor holding the new iterate."""
        xc_new = self.coarse_correction(lvl)
        out = self.spare[lvl]
        weights = self.weights(self.nu2) if post else []
        pair = (self.post_pair and len(weights) >= 2 and self.dtype == torch.float64
                and ops.jacobi2_supported(self.shapes[lvl], self.dtype) and math.prod(self.shapes[lvl]) >= self.pair_min_cells)
        if pair:
            # x + P x_c as a pass of its own (2 1/8 words), then the post-smoothing PAIR in one pass (3 words): 5 1/8
            # against 3 1/8 + 3 for the prolongation fused into the first of two single sweeps
            ops.interp_add(xc_new, self.locs[lvl], add=x, out=out)
        elif weights and self.locs[lvl] == self.loc and ops.jacobi_synth_supported(self.shapes[lvl], self.dtype):
            # x + P x_c is formed in registers by the first post-smoothing sweep (3 1/8 words per cell instead of 5 1/8)
            ops.poisson_jacobi_synth(xc_new, x, b, self.h2s[lvl], weights[0], out=out)
            weights = weights[1:]
        else:
            ops.interp_add(xc_new, self.locs[lvl], add=x, out=out)  # x + P x_c
        self.spare[lvl] = x
        return self.sweeps(lvl, out, b, weights)

    def coarse_correction(self, lvl):
        """x_c with A_c x_c ~= b_{lvl+1}: one cycle on level lvl + 1 from the zero iterate."""
        xc = self.x[lvl + 1]
        t = self.tail()
        if t is not None and lvl + 1 == t[0]:
            return self.tail_cycle(lvl + 1, None, self.b[lvl + 1])  # (the zero start is not even stored)
        # (not zeroed: a cycle from the zero start does not read its iterate's buffer)
        xc_new = self.vcycle(lvl + 1, xc, self.b[lvl + 1], zero=True)
        if xc_new is not xc:  # keep the per-cycle buffer distinct from the level's spare
            self.x[lvl + 1] = xc_new
        return xc_new

    def last_level_coeffs(self):
        """The last level's operator as coefficient arrays [(2 d + 1), *shape] (what `continuation` pads)."""
        return ops.poisson_jac_coeffs(self.shapes[-1], self.h2s[-1], self.dtype, self.device)

    def continuation(self):
        """The hierarchy BELOW a last level that is too large for the dense inverse and cannot be halved because an extent
        is odd (N = 100: 100, 50, 25; N = 1000 in 2-D: ..., 125 x 125 = 15625 unknowns, where sweeps alone solve nothing and
        the cycles above stall): the same operator on the grid padded to even extents -- the added cells carry a small
        diagonal entry and no coupling in either direction, their unknowns and right-hand sides are zero -- as the finest
        level of a `StencilGMG` of its own (which continues the same way at its next odd level).  None when the level is
        small, even (extents below `min_size`), or couples across the padded end (a periodic axis)."""
        if self._continuation is None:
            shape = self.shapes[-1]
            child = False
            if math.prod(shape) > 512 and any(n % 2 for n in shape) and all(n >= 3 for n in shape):
                c = self.last_level_coeffs()
                inside = tuple(slice(0, n) for n in shape)
                closed = True
                for ax, n in enumerate(shape):
                    if n % 2:  # nothing may reach across the end that moves: -e at the first cell, +e at the last
                        lo = c[1 + 2 * ax].select(ax, 0)
                        hi = c[2 + 2 * ax].select(ax, n - 1)
                        closed = closed and float(lo.abs().max()) == 0.0 and float(hi.abs().max()) == 0.0
                if closed:
                    padded = tuple(n + n % 2 for n in shape)
                    cp = torch.zeros((c.shape[0],) + padded, dtype=c.dtype, device=c.device)
                    cp[0].fill_(1e-6 * (float(c[0].abs().mean()) or 1.0))
                    cp[(slice(None),) + inside] = c
                    child = (StencilGMG(cp, nu1=self.nu1, nu2=self.nu2), inside, padded)
            self._continuation = child
        return self._continuation or None

    def continued_cycle(self, x, b):
        """Two cycles of the padded hierarchy in place of the last level's solve."""
        child, inside, padded = self.continuation()
        lvl = self.nlvl - 1
        xp = torch.zeros(padded, dtype=self.dtype, device=self.device)
        bp = torch.zeros(padded, dtype=self.dtype, device=self.device)
        xp[inside] = x
        bp[inside] = b
        for _ in range(2):  # (one: the padded transition loses accuracy, 0.35 per cycle instead of 0.17 at 250^2 and 100^3)
            xp = child.vcycle(0, xp, bp)
        out = self.spare[lvl]
        out.copy_(xp[inside])
        self.spare[lvl] = x
        return out

    def vcycle(self, lvl, x, b, zero=False, post=True):
        """One V(nu1, nu2) cycle on A x = b; returns the tensor holding the new iterate.  zero: `x` is the zero vector."""
        t = self.tail()
        if t is not None and lvl == t[0]:
            return self.tail_cycle(lvl, None if zero else x, b)
        if lvl == self.nlvl - 1:
            if math.prod(self.shapes[lvl]) <= 512:  # coarsest grid: x = A^-1 b
                out = self.spare[lvl]
                ops.dots(self.coarse_inverse(), b.reshape(-1), out=out.view(-1))
                self.spare[lvl] = x
                return out
            if self.continuation() is not None:  # an odd extent: the cycle goes on below on the padded grid
                return self.continued_cycle(x.zero_() if zero else x, b)
            return self.smooth(lvl, x, b, 40, chebyshev=False, zero=zero)  # cannot coarsen further: by iteration
        x = self.smooth(lvl, x, b, self.nu1, zero=zero)
        self.coarse_rhs(lvl, x, b)
        return self.finish_cycle(lvl, x, b, post=post)

    def full_multigrid(self, b, post=True):
        """First iterate by nested iteration: the right-hand side is restricted to every level, the coarsest problem is
        solved, and each finer level starts one V-cycle from the prolongated solution of the level below.  Costs about
        one V-cycle of the finest level plus 1/7 and leaves the error near the discretisation level instead of O(1):
        the cycles that follow only have to cover the remaining distance to the tolerance."""
        t = self.tail()
        bottom = self.nlvl - 1 if t is None else t[0]  # (the tail restricts and nests on its own levels in one launch)
        fb = [b]
        for lvl in range(bottom):
            fb.append(self.restrict(lvl, fb[-1]))
        x = None
        for lvl in range(bottom, -1, -1):
            if x is None and t is not None:
                x = self.tail_cycle(lvl, None, fb[lvl], fmg=True)
                continue
            if x is None:
                start = skewed(self.shapes[lvl], self.dtype, self.device, 5, zero=True)
            else:
                # (a fresh tensor: the cycle's work buffers rotate underneath)
                start = ops.interp_add(x, self.locs[lvl], out=skewed(self.shapes[lvl], self.dtype, self.device, 5))
            # (the tensor returned is never this level's coarse-correction buffer self.x[lvl], which the next finer
            # cycle zeroes: a cycle rotates its argument with self.spare[lvl] only)
            # (post=False: the finest level's cycle ends at its coarse-grid correction -- `solve` pre-smooths right after,
            # and four sweeps in a row buy less than the pass they cost)
            x = self.vcycle(lvl, start, fb[lvl], post=post or lvl > 0)
        return x

    def solve_krylov(self, b, x, tol, maxiter, status=None, m=6):
        """A x = b by GCR(m) with one V-cycle (from the zero start) as the preconditioner, continuing from the iterate `x`:
        what the stationary cycles hand over to when they stop contracting well (cells far from cubes, convection-dominated
        rows).  Per pass: one cycle, one operator application, the Gram-Schmidt products of the new direction with the last
        m (one `odil_dots` + two `odil_lincomb` passes over the stored directions, coefficients on the device) and two
        updates; one read-back (the residual norm).  A need not be symmetric.  Returns (x, residual norm, passes)."""
        n = b.numel()
        dev, dt = b.device, b.dtype
        sign = self.residual_sign
        r = torch.empty_like(b)
        self.residual(0, x, b, r)                   # sign * (A x - b)
        ops.scale(r, -sign, out=r)                  # r = b - A x
        x = x.contiguous()
        Z = torch.empty((m, n), dtype=dt, device=dev)   # directions z_j and their images q_j = A z_j, q_j orthonormal
        Q = torch.empty((m, n), dtype=dt, device=dev)
        zero_b = torch.zeros_like(b)
        bn = float(ops.dots(b.view(1, -1), b.view(-1))[0]) ** 0.5
        res = float(ops.dots(r.view(1, -1), r.view(-1))[0]) ** 0.5
        it, have, e = 0, 0, None
        while it < maxiter and res > tol * max(bn, 1e-300) and res == res:
            if have == m:  # restart: the stored directions are dropped
                have = 0
            e = self.vcycle(0, torch.empty_like(b) if e is None else e, r, zero=True)   # M^-1 r (from the zero start)
            z, q = Z[have].view(b.shape), Q[have].view(b.shape)
            z.copy_(e)
            self.residual(0, z, zero_b, q)          # sign * A z
            if sign != 1.0:
                ops.scale(q, sign, out=q)
            if have:  # Gram-Schmidt against the stored images: ONE product launch, two combination launches
                nbeta = ops.scale(ops.dots(Q[:have], q.view(-1)), -1.0)
                ops.lincomb(q.view(-1), 1.0, Q[:have], nbeta)
                ops.lincomb(z.view(-1), 1.0, Z[:have], nbeta)
            inv = torch.rsqrt(ops.dots(q.view(1, -1), q.view(-1)))          # 1 / |q|, a one-element device tensor
            ops.scale(q, 1.0, adev=inv, out=q)
            ops.scale(z, 1.0, adev=inv, out=z)
            a = ops.dots(q.view(1, -1), r.view(-1))
            ops.lincomb(x.view(-1), 1.0, z.view(1, -1), a)
            ops.lincomb(r.view(-1), 1.0, q.view(1, -1), ops.scale(a, -1.0))
            have += 1
            it += 1
            res = float(ops.dots(r.view(1, -1), r.view(-1))[0]) ** 0.5
        return x, res, it

    def solve(self, b, tol=1e-12, maxiter=60, status=None, x0=None, copy=True, fmg=True, krylov="auto", b_meansq=None):
        """Solves A x = b to ||A x - b|| <= tol * ||b||.  The residual that is tested is the one every cycle
        forms anyway (after its pre-smoothing sweeps, on its way to the coarse grid): the iterate returned
        is that pre-smoothed one, so convergence costs no pass of its own.  b_meansq: 0-d device tensor holding mean(b^2)
        when the caller has it (the evaluation that produced b reduced it already): no pass over b for its norm."""
        n = b.numel()
        # a relative residual below ~50 ulp of the working precision cannot be reached: asking float32 for 1e-12 would
        # burn every cycle of `maxiter` and report nothing
        tol = max(tol, 50 * float(torch.finfo(self.dtype).eps))
        if x0 is not None:
            x = x0.clone()
        elif fmg and self.nlvl > 2:
            x = self.full_multigrid(b, post=False)
        else:
            x = torch.zeros_like(b)
        if b_meansq is not None:
            bn = math.sqrt(max(float(b_meansq), 0.0) * n)
        else:
            bn = float(ops.dots(b.view(1, -1), b.view(-1))[0]) ** 0.5
        res, it = bn, 0
        method = "gmg-vcycle"
        stagnated = False
        if self.nlvl == 1:
            # the finest level itself cannot be halved (an odd extent): its 'cycle' is the dense inverse (small), the padded
            # continuation, or sweeps -- as the preconditioner of GCR, which needs no contraction from it (the continuation
            # at the FINEST level contracts by 0.3 - 0.5 only, at 125^3 one mode even grows: the wall sits half a coarse
            # cell off on every padded level)
            x, res, it = self.solve_krylov(b, x, tol, maxiter, m=6)
            method = "GCR(6), preconditioner: " + ("padded multigrid cycles" if self.continuation() is not None else
                                                   "coarse inverse" if n <= 512 else "sweeps")
        while self.nlvl > 1:
            x = self.smooth(0, x, b, self.nu1)
            self.coarse_rhs(0, x, b)
            prev, res = res, math.sqrt(max(float(self.loss), 0.0) * n)
            if res <= tol * max(bn, 1e-300) or it >= maxiter:
                break
            slow = it >= 2 and (res >= 0.5 * prev or res != res)
            if slow and krylov != "never" and res > 1e3 * tol * max(bn, 1e-300) and res == res:
                # the cycles contract by less than 0.5 (or not at all) well above the tolerance: the same cycle becomes the
                # preconditioner of GCR (`solve_krylov`), which needs no contraction, only a useful direction per pass
                x = x.clone()  # (a buffer of its own: the cycle's work buffers rotate underneath)
                x, res, extra = self.solve_krylov(b, x, tol, maxiter - it, m=6)
                it += extra
                method = "gmg-vcycle + GCR(6)"
                break
            if it >= 3 and res >= 0.98 * prev:
                stagnated = res == res and res < bn  # at the rounding floor of the working precision (not: diverging)
                break
            x = self.finish_cycle(0, x, b)
            it += 1
        converged = res <= tol * max(bn, 1e-300)
        if not converged:
            from .util import printlog

            printlog("odil_amd: multigrid stopped at relative residual {:.2e} after {} cycles (tolerance {:.1e})".format(
                res / max(bn, 1e-300), it, tol))
        if status is not None:
            status["residual"] = res
            status["bnorm"] = bn  # (|b|: the caller that judges the residual against it need not reduce b again)
            status["niter"] = it
            status["method"] = method
            status["converged"] = converged
            # cycles that stopped gaining at the floor of the working precision (float32: cond * 6e-8 relative): nothing that
            # works in this precision gets further, least of all the normal equations
            status["stagnated"] = stagnated
        # the iterate may live in one of this object's work buffers: copy=False only for a caller that
        # consumes it before the next solve
        return x.clone() if copy else x


def solve_mixed(high, low, b, tol=1e-12, maxiter=60, status=None, fmg=True):
    """A x = b in float64 by ITERATIVE REFINEMENT around float32 V-cycles: every pass forms the float64 residual
    r = A x - b with `high` (one pass over the fine level, its norm on the way), hands -r / rms(r) in float32 to one cycle of
    `low` (the first pass: its nested-iteration start) and adds the correction back in float64 -- the cycle is a
    preconditioner, so its precision limits the contraction per pass (~0.17, far above float32 rounding), not the accuracy
    of the answer, which is the float64 residual's.  The cycle's traffic -- everything but one residual per pass -- is
    halved.  Conversions: `odil_narrow_scale` / `odil_widen_axpy`, the scale read from the device scalar the residual
    kernel wrote.  (No reference counterpart: the reference's direct solve is double throughout, linsolver.py:17-26; the
    Newton iterate it defines is reached to the same tolerance.)"""
    n = b.numel()
    tol = max(tol, 50 * float(torch.finfo(b.dtype).eps))
    x = torch.zeros_like(b)
    r = high.r(0)
    rl = torch.empty(b.shape, dtype=torch.float32, device=b.device)
    bn = float(ops.dots(b.view(1, -1), b.view(-1))[0]) ** 0.5
    res, it, e = bn, 0, None
    while True:
        high.residual(0, x, b, r)  # r = A x - b, high.loss = its mean square
        prev, res = res, math.sqrt(max(float(high.loss), 0.0) * n)
        if res <= tol * max(bn, 1e-300) or it >= maxiter or (it >= 3 and res >= 0.98 * prev):
            break
        ops.narrow_scale(r, rl, a=-high.residual_sign, msq=high.loss)  # right-hand side of the error equation A e = b - A x
        if it == 0 and fmg and low.nlvl > 2:
            e = low.full_multigrid(rl)
        else:
            e = low.vcycle(0, torch.empty_like(rl) if e is None else e, rl, zero=True)
        ops.widen_axpy(x, e, a=1.0, msq=high.loss)
        it += 1
    converged = res <= tol * max(bn, 1e-300)
    if not converged:
        from .util import printlog

        printlog("odil_amd: mixed-precision multigrid stopped at relative residual {:.2e} after {} passes (tolerance {:.1e})".format(
            res / max(bn, 1e-300), it, tol))
    if status is not None:
        status.update(residual=res, niter=it, method="gmg-vcycle (float32 cycles, float64 residual)", converged=converged)
    return x


class StencilGMG(PoissonGMG):
    """V-cycles for ANY (2 d + 1)-point operator with variable coefficients on a cell-centred grid (d <= 3): the Newton
    system M delta = -r of a single-field operator as `Problem.linearize` delivers it (reference core.py:1113-1217) --
    variable-coefficient diffusion, reaction and convection terms, any wall closure, periodic axes.  Same cycle as
    PoissonGMG (Chebyshev-weighted Jacobi sweeps, full-weighting restriction, the multigrid decomposition's P for the
    corrections, nested-iteration start); what differs is where the operators come from:

      fine level     the Jacobian's own coefficient arrays                    `coeffs` [(2 d + 1), *shape]
      coarse levels  odil_stencil_var_coarsen (csrc/stencil_mg.hip): aggregates of 2^d cells, piecewise-constant Galerkin
                     products of the second-order part (x 1/2), the antisymmetric part and the row sums -- formed once
                     per solve (the coefficients change with the state), 8/7 of one pass over the fine arrays
      sweeps         odil_stencil_var_smooth            (2 d + 1) + 3 words per cell (Poisson: 3)
      coarse rhs     odil_stencil_var_residual_restrict (2 d + 1) + 2 + 1 / 2^d words, the residual norm on the way

    M need not be symmetric; the cycle is a stationary iteration on M delta = -r itself (for a square nonsingular M the
    solution of the normal equations the reference forms, linsolver.py:17-23).  `solve` reports `converged`; the caller
    (linsolver.solve) falls back to the normal-equation routes when the cycles do not contract."""

    def __init__(self, coeffs, nu1=None, nu2=None, min_size=2, lite=False, store=None):
        """lite: the finest level only (the float64 half of `solve_mixed`); store: dtype the hierarchy is KEPT in (the
        coarse operators are formed in the precision of `coeffs` and cast level by level: the float32 half)."""
        shape = tuple(coeffs.shape[1:])
        self.ndim = len(shape)
        assert coeffs.shape[0] == 2 * self.ndim + 1 and self.ndim <= 3 and coeffs.is_contiguous()
        self.loc = "c" * self.ndim
        self.dtype, self.device = store or coeffs.dtype, coeffs.device
        self.omega = {1: 2.0 / 3.0, 2: 4.0 / 5.0, 3: 6.0 / 7.0}[self.ndim]
        self.nu1 = self.nu_default[0] if nu1 is None else nu1
        self.nu2 = self.nu_default[1] if nu2 is None else nu2
        self.coeffs, self.shapes = [coeffs], [shape]
        self.locs = []  # per transition: 'c' on the merged axes, '.' on the others
        cur = coeffs
        while not lite:
            # SEMI-coarsening while the couplings are of different sizes (cells far from cubes, anisotropic conductivities):
            # only the axes whose largest coupling is within a factor 2 of the largest of all are merged -- point smoothing
            # damps nothing else -- until they meet (one read-back per level, at set-up)
            if self.ndim == 1:
                halve = [True]
            else:
                size = ops.max_abs_rows(cur[1:]).cpu().numpy()  # (2 d arrays: two launches, one read-back)
                # (the smaller of the two directions: an upwind convection term inflates ONE of them, and it grows
                # relative to the diffusion on every coarser level -- that is not an anisotropy of the smoothing problem)
                strength = [min(float(size[2 * a]), float(size[2 * a + 1])) for a in range(self.ndim)]
                halve = [v >= 0.5 * max(strength) for v in strength]
            if not all(n % 2 == 0 and n // 2 >= min_size for n, on in zip(self.shapes[-1], halve) if on):
                break
            cur = ops.stencil_var_coarsen(cur, halve)
            self.coeffs.append(cur)
            self.shapes.append(tuple(n // 2 if on else n for n, on in zip(self.shapes[-1], halve)))
            self.locs.append("".join("c" if on else "." for on in halve))
        if store is not None and store != coeffs.dtype:
            assert coeffs.dtype == torch.float64 and store == torch.float32
            self.coeffs = [ops.narrow_scale(c.reshape(-1), torch.empty(c.numel(), dtype=store, device=c.device)).view(c.shape)
                           for c in self.coeffs]
        self.nlvl = len(self.shapes)
        mk = lambda s: torch.zeros(s, dtype=self.dtype, device=self.device)
        self.loss = mk(())
        self._coarse_inv = None
        self._continuation = None
        self._r = [None] * self.nlvl
        if lite:
            return
        self.x = [None] + [skewed(s, self.dtype, self.device, 2, zero=True) for s in self.shapes[1:]]
        self.b = [None] + [skewed(s, self.dtype, self.device, 3, zero=True) for s in self.shapes[1:]]
        self.spare = [skewed(s, self.dtype, self.device, 1) for s in self.shapes]

    def coarse_inverse(self):
        """(Pseudo-)inverse of the coarsest operator, from the residual kernel applied to unit vectors; a singular
        coarsest operator (all-periodic or all-Neumann problems: constants in the null space) gets the minimum-norm
        solution."""
        if self._coarse_inv is None:
            shape = self.shapes[-1]
            n = math.prod(shape)
            eye = torch.eye(n, dtype=self.dtype, device=self.device)
            zero = torch.zeros(shape, dtype=self.dtype, device=self.device)
            cols = [-ops.stencil_var_residual(self.coeffs[-1], eye[j].view(shape).contiguous(), zero).reshape(-1) for j in range(n)]
            amat = torch.stack(cols, dim=1).cpu().numpy().astype(np.float64)  # column j = A e_j
            inv = np.linalg.pinv(amat, rcond=1e-12)
            self._coarse_inv = torch.as_tensor(inv, dtype=self.dtype).to(self.device).contiguous()
        return self._coarse_inv

    residual_sign = -1.0  # `residual` returns b - A x (PoissonGMG: A x - b)

    def tail_coeffs(self, lvl):
        return self.coeffs[lvl]

    def tail_transfers_ok(self, first):
        return True  # (mean of the merged children on every transition: what the tail does)

    def last_level_coeffs(self):
        return self.coeffs[-1]

    def residual(self, lvl, x, b, out):
        """out = b - A x, its mean square in self.loss."""
        ops.stencil_var_residual(self.coeffs[lvl], x, b, out=out)
        ops.mean_reduce(out.reshape(-1), square=True, out=self.loss)
        return out

    def sweeps(self, lvl, x, b, weights, zero=False):
        """Sweeps in PAIRS through the one-pass kernel (odil_stencil_var_smooth2: the coefficient arrays -- 7 of a sweep's
        10 words in 3-D -- read once for both sweeps; bit-identical to two single sweeps) on the bandwidth-bound levels.
        zero: the iterate is the zero vector and `x` only a buffer: the first launch does not read it (x = NULL)."""
        weights = list(weights)
        if zero and (not weights or not self.zero_start):
            x.zero_()
            zero = False
        pair = ops.smooth2_supported(self.shapes[lvl]) and math.prod(self.shapes[lvl]) >= self.pair_min_cells
        while weights:
            y = self.spare[lvl]
            src = None if zero else x
            zero = False
            if pair and len(weights) >= 2:
                ops.stencil_var_smooth2(self.coeffs[lvl], src, b, weights[0], weights[1], out=y)
                weights = weights[2:]
            else:
                ops.stencil_var_smooth(self.coeffs[lvl], src, b, weights[0], out=y)
                weights = weights[1:]
            self.spare[lvl] = x
            x = y
        return x

    def coarse_rhs(self, lvl, x, b):
        bc = self.b[lvl + 1]
        if self.locs[lvl] == self.loc:
            ops.stencil_var_residual_restrict(self.coeffs[lvl], x, b, 1.0 / 2**self.ndim, bc, self.loss if lvl == 0 else None)
        else:  # some axes only: the residual (its norm on the way), then the mean of the children
            self.restrict(lvl, self.residual(lvl, x, b, self.r(lvl)), 1.0, out=bc)
        return bc

    def restrict(self, lvl, r, sign=1.0, out=None):
        """sign * (mean of the children): the R of the aggregation-built coarse operators.  (P^T / 2^k, which the
        rediscretised Poisson hierarchy uses on its semi-coarsened transitions, is NOT consistent with them: across a
        1 : 1000 jump of the conductivity the cycle diverged.)  Merged axes only: a strided mean (torch), no kernel of
        this library averages along a subset of the axes."""
        loc = self.locs[lvl]
        if loc == self.loc:
            return PoissonGMG.restrict(self, lvl, r, sign, out)
        pairs, dims = [], []
        for n, l in zip(r.shape, loc):
            if l == "c":
                dims.append(len(pairs) + 1)
                pairs += [n // 2, 2]
            else:
                pairs.append(n)
        rc = r.reshape(pairs).mean(dim=dims)
        return ops.scale(rc, sign, out=out) if (sign != 1.0 or out is not None) else rc

    def finish_cycle(self, lvl, x, b, post=True):
        xc_new = self.coarse_correction(lvl)
        # TWO cycles on the first coarse level in 3-D (a W-cycle's top, V below): the aggregation-built coarse operators
        # are slightly less accurate than a rediscretisation at the walls, and a more exact level-1 solve takes the
        # contraction from 0.24 to 0.14 per cycle (17 -> 13 cycles to 1e-10, tests/test_stencil_gmg_host.py) for 1/7 more
        # work; a full W-cycle gains one more cycle and pays it back in ~700 launch-bound coarse launches
        if lvl == 0 and self.ndim == 3 and self.nlvl > 2:
            xc_new = self.vcycle(lvl + 1, xc_new, self.b[lvl + 1])
            self.x[lvl + 1] = xc_new
        out = self.spare[lvl]
        ops.interp_add(xc_new, self.locs[lvl], add=x, out=out)  # x + P x_c
        self.spare[lvl] = x
        return self.sweeps(lvl, out, b, self.weights(self.nu2) if post else [])


def recognise_stencil(op):
    """The coefficient tensor [(2 d + 1), *shape] (order 0, -e_0, +e_0, -e_1, ...) when the LinearizedOperator is square,
    acts on ONE cell-centred `Field` (d <= 3, extents >= 4) and every stencil block's shift is 0 or a unit vector --
    the structure StencilGMG needs; None otherwise.  Shifts that do not occur are zero arrays, duplicates are summed."""
    from .core import Field

    if len(op.key_to_field) != 1 or op.nrows != op.ncols:
        return None
    (key, field), = op.key_to_field.items()
    if not isinstance(field, Field):
        return None
    shape = tuple(field.array.shape)
    ndim = len(shape)
    if ndim > 3 or field.loc != "c" * ndim or any(s < 4 for s in shape):
        return None
    want = [(0,) * ndim]
    for i in range(ndim):
        want += [tuple(-1 if j == i else 0 for j in range(ndim)), tuple(1 if j == i else 0 for j in range(ndim))]
    # the arrays already lie back to back in the wanted order (the generated Jacobian kernel writes slices of one buffer
    # in the order the operator reads them): a view, no copies
    by_shift = dict()
    for row0, nrows, kind, k, payload in op.blocks:
        if kind == "stencil" and row0 == 0 and nrows == op.ncols and payload[2] == field.loc and tuple(payload[3]) == shape:
            norm = tuple(((s + n // 2) % n) - n // 2 for s, n in zip(payload[1], shape))
            if norm in by_shift:
                by_shift = None
                break
            by_shift[norm] = payload[0]
    if by_shift is not None and len(by_shift) == len(op.blocks) and sorted(by_shift) == sorted(want):
        arrs = [by_shift[sft] for sft in want]
        first, size = arrs[0], arrs[0].numel()
        if all(a.is_contiguous() and a.dtype == first.dtype and a.untyped_storage().data_ptr() == first.untyped_storage().data_ptr()
               and a.storage_offset() == first.storage_offset() + j * size for j, a in enumerate(arrs)):
            return torch.empty(0, dtype=first.dtype, device=first.device).set_(
                first.untyped_storage(), first.storage_offset(), (len(want),) + shape)
    coeffs = torch.zeros((len(want),) + shape, dtype=op.dtype, device=op.device)
    seen = set()
    for row0, nrows, kind, k, payload in op.blocks:
        if kind != "stencil" or row0 != 0 or nrows != op.ncols:
            return None
        coeff, shift, loc, vshape = payload
        norm = tuple(((s + n // 2) % n) - n // 2 for s, n in zip(shift, shape))  # periodic roll: |shift| <= n / 2
        if loc != field.loc or tuple(vshape) != shape or norm not in want:
            return None
        slot = want.index(norm)
        if slot in seen:
            ops.axpy(coeffs[slot].view(-1), coeff.reshape(-1).contiguous(), 1.0)
        else:
            coeffs[slot].view(-1).copy_(coeff.reshape(-1))
            seen.add(slot)
    if 0 not in seen:
        return None
    return coeffs


def recognise_poisson(op):
    """(shape, h2) if the LinearizedOperator is exactly the zero-Dirichlet Laplacian stencil of one
    cell-centred field, else None."""
    from .core import Field

    if len(op.key_to_field) != 1 or op.nrows != op.ncols:
        return None
    (key, field), = op.key_to_field.items()
    if not isinstance(field, Field):
        return None
    shape = tuple(field.array.shape)
    ndim = len(shape)
    if ndim > 3 or field.loc != "c" * ndim or any(s < 2 for s in shape):
        return None
    blocks = {}
    for row0, nrows, kind, k, payload in op.blocks:
        if kind != "stencil" or row0 != 0 or nrows != op.ncols:
            return None
        coeff, shift, loc, vshape = payload
        if loc != field.loc or tuple(vshape) != shape or shift in blocks:
            return None
        blocks[shift] = coeff
    want = [(0,) * ndim]
    for i in range(ndim):
        want += [tuple(-1 if j == i else 0 for j in range(ndim)), tuple(1 if j == i else 0 for j in range(ndim))]
    if sorted(blocks) != sorted(want):
        return None
    npdt = np.float64 if op.dtype == torch.float64 else np.float32
    h2 = [npdt(op.domain.step_by_dim(i)) ** 2 for i in range(ndim)]
    rtol = 1e-11 if op.dtype == torch.float64 else 1e-4
    # ONE pass over the coefficient arrays against the values the Poisson Jacobian has there (formed on the fly: no
    # reference arrays), one read-back
    pairs = ops.poisson_jac_match([blocks[shift] for shift in want], shape, h2).cpu().numpy()
    if not np.all(pairs[:, 0] <= rtol * pairs[:, 1]):  # (NaN compares false)
        return None
    return shape, h2
