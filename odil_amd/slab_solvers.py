"""Quasi-Newton and Newton drivers of the slab-decomposed Poisson path (no reference
counterpart: the reference is single-device).  The optimizers are the single-GPU ones -- `optimizer.lbfgsb_minimize`
(the restatement of what the reference calls, reference src/odil/optimizer.py:54-117) and the conjugate-gradient solve of
the Newton step's normal equations (reference src/odil/linsolver.py:17-23, util.py:152-187) -- on vectors that hold ONE
RANK's share of the unknowns:

  * evaluations run the slab kernels of `slab.SlabPoissonAdam` (halo exchanges with the two neighbours, RCCL send / receive);
  * every reduction is formed from the ranks' partial results: the probes of an L-BFGS iteration (loss, <g, d>, <d, d>,
    max |g|) travel as ONE all-gather of eight numbers, the products of the whole (s, y) history with the new vectors
    (the rows of S^T Y, S^T S, Y^T Y of the compact representation and S^T g, Y^T g of the next direction) as ONE
    all-gather of 3 x 2m numbers -- two collectives per iteration instead of 2m sequential dot products; CG: two
    scalar reductions per iteration.  The combined values are the same bits on every rank (sums over the gathered rows
    in rank order), so the ranks take the same branches of the line search without further agreement.

`ThreadComm` runs several ranks as threads of one process (one GPU, or none with the CPU doubles of the tests): the
drivers here are ordinary loops with blocking collectives, not generators.
"""

import threading

import numpy as np
import torch

from . import ops as hip_ops
from . import slab
from .optimizer import LbfgsVectors, lbfgsb_minimize


def drive(gen, comm):
    """Runs a generator of the slab path (yields (kind, send_lo, send_hi) at its exchanges) to its end over `comm`."""
    try:
        msg = next(gen)
        while True:
            msg = gen.send(comm.exchange(*msg))
    except StopIteration:
        pass


class ThreadComm:
    """The exchanges of `slab.TorchDistComm` between ranks that are THREADS of one process (tests; one GPU).  A rank
    computes only while it holds the shared token, so the launches of two ranks never interleave (the kernels' shared
    reduction workspaces assume one caller at a time); it hands the token over while it waits at an exchange."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.token = threading.Lock()
            self.slots = [None] * world

    def __init__(self, rank, shared):
        self.rank, self.world, self.sh = rank, shared.world, shared

    def __enter__(self):
        self.sh.token.acquire()
        return self

    def __exit__(self, *exc):
        self.sh.token.release()
        if exc[0] is not None:
            self.sh.barrier.abort()  # the other ranks must not wait for one that died
        return False

    def _wait(self):
        """The barrier, with the token handed over meanwhile (and held again afterwards even when another rank died and
        the barrier broke: __exit__ releases it)."""
        self.sh.token.release()
        try:
            self.sh.barrier.wait()
        finally:
            self.sh.token.acquire()

    def exchange(self, kind, a, b):
        sh, r, P = self.sh, self.rank, self.world
        if kind == "wait":
            return a
        sh.slots[r] = (a, b)
        self._wait()
        try:
            clone = lambda t: None if t is None else t.clone()
            if kind == "sum":
                total = sh.slots[0][0].clone()
                for i in range(1, P):
                    total = total + sh.slots[i][0]
                return total
            if kind == "gather":
                return torch.stack([sh.slots[i][0] for i in range(P)])
            if kind == "wrap":
                lo = clone(sh.slots[P - 1][1]) if r == 0 else None
                hi = clone(sh.slots[0][0]) if r == P - 1 else None
                return lo, hi
            # "halo" / "post": what the lower neighbour sent up, what the upper one sent down
            return (clone(sh.slots[r - 1][1]) if r > 0 else None, clone(sh.slots[r + 1][0]) if r + 1 < P else None)
        finally:
            self._wait()  # nobody overwrites its slot before everybody has read


def run_threads(world, body):
    """body(rank, comm) -> result on `world` threads joined by a ThreadComm; returns the results in rank order."""
    shared = ThreadComm.Shared(world)
    results, errors = [None] * world, []

    def work(rank):
        try:
            with ThreadComm(rank, shared) as comm:
                results[rank] = body(rank, comm)
        except BaseException as e:  # noqa: BLE001 (re-raised below, on the caller's thread)
            errors.append(e)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        first = [e for e in errors if not isinstance(e, threading.BrokenBarrierError)] or errors
        raise first[0]
    return results


class SlabLbfgsVectors(LbfgsVectors):
    """`optimizer.LbfgsVectors` over one rank's share of the unknowns: the reductions are combined over the ranks before
    the host reads them (one all-gather each; the sum / max over its rows is formed in rank order on every rank)."""

    def __init__(self, n, m, device, comm):
        super().__init__(n, m, device)
        self.comm = comm

    def reduce_probes(self, scal):
        rows = self.comm.exchange("gather", scal, None)  # (world, 8): dtd, gd_old, -, gd, gg, gmax, f, -
        out = rows.sum(dim=0)
        out[5] = rows[:, 5].max()
        return out

    def reduce_sums(self, t):
        return self.comm.exchange("gather", t, None).sum(dim=0)


class SlabPoissonLbfgs(slab.SlabPoissonAdam):
    """One rank of the slab-decomposed Poisson multigrid problem under L-BFGS-B (3-D, all cell-centred): the unknown
    vector is the undivided one restricted to this rank's planes, level after level."""

    def __init__(self, N, rank, world, dtype=torch.float64, device=None, rhs_global=None):
        super().__init__(N, rank, world, dtype=dtype, device=device, rhs_global=rhs_global, moments=False)
        self.nfev = 0

    def minimize(self, comm, maxiter, m=50, maxls=50, pgtol=1e-16, factr=0.0, callback=None, vectors=None):
        """-> the dict of `lbfgsb_minimize` (task, nit, funcalls, f = GLOBAL loss); the owned planes of self.w hold the
        result.  `vectors`: the vector backend (tests hand in a CPU double)."""
        n = self.n_unknowns_local
        vec = vectors or SlabLbfgsVectors(n, m, self.device, comm)
        x = self.pack_owned(self.w)
        gflat = torch.empty(n, dtype=torch.float64, device=self.device)

        def fg(xflat):
            self.unpack_owned(xflat)
            drive(self.loss_grad_gen(), comm)
            self.nfev += 1
            return self.loss_part.to(torch.float64), self.pack_owned(self.gw, out=gflat)

        res = lbfgsb_minimize(x, fg, vec, maxiter, m=m, maxls=maxls, pgtol=pgtol, factr=factr, callback=callback)
        self.unpack_owned(x)
        return res


class SlabPoissonNewtonCG:
    """One rank of a matrix-free Newton step of the slab-decomposed Poisson problem WITHOUT the multigrid decomposition
    (the reference's `linearize` refuses multigrid unknowns, reference core.py:1208-1209): the step solves the normal
    equations (M^T M + damp^2) dx = -M^T f of the linearisation (reference linsolver.py:17-23) by conjugate gradients;
    M is the Laplacian stencil with the wall rows (its coefficients are the reference's `eval_operator_grad` arrays, SURVEY
    A13), applied by the residual kernel with a zero right-hand side, M^T by the stencil-adjoint kernel.  Per CG iteration:
    one plane of p and one plane of M p to each neighbour (the same halo pattern as an epoch), two scalar reductions."""

    def __init__(self, N, rank, world, dtype=torch.float64, device=None, rhs_global=None, nz=None):
        """The box (world * nz, N, N) with spacing 1 / N, nz (default N) planes per rank."""
        self.ops = hip_ops
        self.N, self.rank, self.world, self.dtype, self.device = N, rank, world, dtype, device
        npdt = np.float64 if dtype == torch.float64 else np.float32
        nz = nz or N
        self.lv = slab.SlabLevel(nz, N, N, rank, world)
        self.h2 = [npdt(1.0 / N) ** 2] * 3
        self.global_cells = world * nz * N * N
        mk = lambda: torch.zeros(self.lv.shape, dtype=dtype, device=device)
        self.u, self.f, self.p, self.q, self.r, self.z = mk(), mk(), mk(), mk(), mk(), mk()
        self.zero = mk()
        self.part = torch.zeros((), dtype=dtype, device=device)
        lv = self.lv
        if rhs_global is not None:
            lo = rank * nz - lv.g_lo
            self.rhs = rhs_global[lo: lo + lv.shape[0]].to(device=device, dtype=dtype).contiguous()
        else:
            assert nz == N, "the built-in reference solution is that of the cubic slabs"
            ref_u = slab.hat_reference_slab(lv, N, rank, world, dtype, device)
            self.rhs, _ = self.ops.poisson_residual(ref_u, torch.zeros_like(ref_u), self.h2)
        self.status = dict()

    def owned(self, a):
        return self.lv.owned(a)

    def _halo(self, comm, a):
        """One boundary plane of `a` to each neighbour's inner ghost plane."""
        lv = self.lv
        lo = a[lv.g_lo] if self.rank > 0 else None
        hi = a[lv.g_lo + lv.nz - 1] if self.rank < self.world - 1 else None
        recv_lo, recv_hi = comm.exchange("halo", lo, hi)
        if recv_lo is not None:
            a[lv.g_lo - 1].copy_(recv_lo.view(lv.ny, lv.nx))
        if recv_hi is not None:
            a[lv.g_lo + lv.nz].copy_(recv_hi.view(lv.ny, lv.nx))

    def _dot(self, comm, a, b):
        part = (self.owned(a).to(torch.float64) * self.owned(b).to(torch.float64)).sum().reshape(1)
        return float(comm.exchange("gather", part, None).sum())

    def residual(self, comm, u, out):
        """out = f(u) = Lap(u) - rhs on the owned planes; -> global mean of its squares."""
        self._halo(comm, u)
        lv = self.lv
        self.ops.poisson_residual(u, self.rhs, self.h2, fu=out, loss=self.part, zrange=(lv.g_lo, lv.g_lo + lv.nz),
                                  denom=self.global_cells)
        return float(comm.exchange("gather", self.part.to(torch.float64).reshape(1), None).sum())

    def normal_apply(self, comm, p, out, damp2=0.0):
        """out = M^T M p + damp2 p on the owned planes (ghost planes of p and of M p refreshed on the way)."""
        self._halo(comm, p)
        self.ops.poisson_residual(p, self.zero, self.h2, fu=self.q, loss=self.part)
        self._halo(comm, self.q)
        self.ops.poisson_adjoint(self.q, self.h2, 1.0, out=out)
        if damp2:
            out.add_(p, alpha=damp2)

    def step(self, comm, maxiter=100, tol=1e-10, damp=0.0):
        """One Newton step u <- u + dx; -> (loss before, loss after).  self.status: CG iterations and relative residual."""
        loss0 = self.residual(comm, self.u, self.f)
        # b = -M^T f
        self._halo(comm, self.f)
        b = self.z
        self.ops.poisson_adjoint(self.f, self.h2, -1.0, out=b)
        dx = torch.zeros_like(self.u)
        r, p, ap = self.r, self.p, torch.zeros_like(self.u)
        r.copy_(b)
        p.copy_(b)
        rr = self._dot(comm, r, r)
        bb, it = rr, 0
        while it < maxiter and rr > tol * tol * bb and rr > 0.0:
            self.normal_apply(comm, p, ap, damp * damp)
            alpha = rr / self._dot(comm, p, ap)
            self.owned(dx).add_(self.owned(p), alpha=alpha)
            self.owned(r).add_(self.owned(ap), alpha=-alpha)
            rr_new = self._dot(comm, r, r)
            beta = rr_new / rr
            self.owned(p).mul_(beta).add_(self.owned(r))
            rr = rr_new
            it += 1
        self.owned(self.u).add_(self.owned(dx))
        loss1 = self.residual(comm, self.u, self.f)
        self.status = dict(niter=it, residual=float(np.sqrt(rr / bb)) if bb > 0 else 0.0)
        return loss0, loss1


class SlabPoissonNewtonGMG(SlabPoissonNewtonCG):
    """The Newton step of the slab-decomposed Poisson problem solved by GEOMETRIC MULTIGRID -- the slab form of
    `gmg.PoissonGMG` ("Newton: matrix-free M / M^T apply = same halo pattern"; the reference's Newton driver,
    src/odil/util.py:152-187, solves M^T M delta = -M^T f with SuperLU, linsolver.py:17-26; for the square nonsingular
    Laplacian M delta = -f has the same solution).  `SlabPoissonNewtonCG` above is unpreconditioned CG on the normal
    equations: O(N) iterations; this one needs ~12 cycles at any size.

    Levels: the box (world nz, N, N) is coarsened by 2 along every axis while every rank keeps >= 2 planes and the
    cross-section extents stay even; level arrays are ghost-extended like the epochs' (`slab.SlabLevel`, G = 2 planes per
    interior interface).  One V(2, 2) cycle per level, everything with the unmodified single-GPU kernels on the extended
    arrays (an array end that is a ghost plane is treated as a wall by the kernel: only the ghost planes themselves see that):
      * two planes of x to each neighbour, then TWO Chebyshev-weighted Jacobi sweeps (the first leaves owned + inner ghost
        planes valid, the second the owned planes);
      * one plane of x, the residual A x - b on the owned planes (its squared norm summed over the ranks by one all-gather),
        full-weighting restriction of the owned planes (pairs of owned planes: no exchange), one plane of the coarse
        right-hand side to each neighbour;
      * the coarse correction; one plane of it to each neighbour, then x += P x_c by the multigrid decomposition's
        prolongation on the ghost-extended coarse array (`SlabLevel.inner`: exact on owned + inner ghost planes);
      * two planes of x, two sweeps.
    Five exchanges per level and cycle, 1 - 2 planes each (512^2 f64 planes: 2 - 4 MB at the finest level, halved twice per
    level).  Below the last slab level the problem is AGGLOMERATED: the coarse right-hand side is all-gathered
    (a few thousand numbers), every rank runs the remaining cycle on the whole coarse box and keeps its planes -- the same
    bits on every rank, no further exchange."""

    def __init__(self, N, rank, world, dtype=torch.float64, device=None, rhs_global=None, nz=None, nu=2, agg_cells=32**3,
                 pair_min_cells=128**3):
        """agg_cells: a level is a SLAB level only while a rank holds more cells of it than this -- an exchange costs ~65 us of
        host time whatever its size (`profiles/r06_rccl_selfloop_sweep.txt`), four of them per level and cycle, and a level of
        32^3 cells per rank is sooner solved redundantly on the whole agglomerated box (0: down to two planes per rank)."""
        super().__init__(N, rank, world, dtype=dtype, device=device, rhs_global=rhs_global, nz=nz)
        self.nu = nu
        npdt = np.float64 if dtype == torch.float64 else np.float32
        nz = self.lv.nz
        self.mlv, self.mh2 = [self.lv], [list(self.h2)]
        shape = (nz, N, N)
        # (cross-sections of >= 4 cells on every slab level, so that the agglomerated box below the last one still has two)
        while (all(s % 2 == 0 for s in shape) and shape[0] // 2 >= 2 and min(shape[1], shape[2]) // 2 >= 4
               and int(np.prod(shape)) // 8 > agg_cells):
            shape = tuple(s // 2 for s in shape)
            self.mlv.append(slab.SlabLevel(shape[0], shape[1], shape[2], rank, world))
            self.mh2.append([v * npdt(4) for v in self.mh2[-1]])
        # the agglomerated box below the last slab level (None when that level cannot be coarsened at all)
        last = self.mlv[-1]
        self.agg_shape = None
        if last.nz % 2 == 0 and last.ny % 2 == 0 and last.nx % 2 == 0 and min(last.ny, last.nx) >= 4:
            self.agg_shape = (world * last.nz // 2, last.ny // 2, last.nx // 2)
            self.agg_h2 = [v * npdt(4) for v in self.mh2[-1]]
        mk = lambda lv: torch.zeros(lv.shape, dtype=dtype, device=device)
        self.mx = [None] + [mk(lv) for lv in self.mlv[1:]]
        self.mb = [None] + [mk(lv) for lv in self.mlv[1:]]
        self.spare = [mk(lv) for lv in self.mlv]
        self.res = [mk(lv) for lv in self.mlv]
        self._dense = dict()
        self.pair_min_cells = pair_min_cells
        # with the library's own kernels underneath the cycle is the single-GPU one's: sweeps in pairs
        # (`odil_poisson_jacobi2`), the residual restricted in the pass that forms it (`odil_poisson_residual_restrict`),
        # the agglomerated box cycled by `gmg.PoissonGMG`; a stand-in `ops` (the host tests) keeps the plain launches
        self.native = getattr(self.ops, "__name__", "") == "odil_amd.ops"
        self.agg_gmg = None
        if self.native and self.agg_shape is not None and min(self.agg_shape) >= 4:
            from . import gmg

            self.agg_gmg = gmg.PoissonGMG(self.agg_shape, self.agg_h2, dtype, device)

    # ---- pieces -------------------------------------------------------------------------------------------------------
    @staticmethod
    def weights(n):
        lo, hi = 1.0 / 3.0, 2.0  # the part of the spectrum of D^-1 A the coarse grid cannot see (gmg.PoissonGMG.weights)
        mid, half = 0.5 * (hi + lo), 0.5 * (hi - lo)
        return [1.0 / (mid - half * np.cos(np.pi * (2 * k + 1) / (2 * n))) for k in range(n)]

    def _halo_planes(self, comm, a, lv, planes):
        """`planes` (1 or 2) boundary planes of `a` into the neighbours' ghost planes nearest the interface."""
        lo = a[lv.g_lo: lv.g_lo + planes] if self.rank > 0 else None
        hi = a[lv.g_lo + lv.nz - planes: lv.g_lo + lv.nz] if self.rank < self.world - 1 else None
        recv_lo, recv_hi = comm.exchange("halo", lo, hi)
        if recv_lo is not None:
            a[lv.g_lo - planes: lv.g_lo].copy_(recv_lo.view(planes, lv.ny, lv.nx))
        if recv_hi is not None:
            a[lv.g_lo + lv.nz: lv.g_lo + lv.nz + planes].copy_(recv_hi.view(planes, lv.ny, lv.nx))

    def _smooth(self, comm, l, x, b, zero=False):
        """`nu` sweeps, two per exchange of two planes; returns the tensor holding the iterate (owned planes valid).
        zero: x is zero on every rank -- its ghost planes are right as they are, the first exchange is skipped."""
        lv, w = self.mlv[l], self.weights(self.nu)
        k = 0
        while k < len(w):
            pair = w[k: k + 2]
            if not (zero and k == 0):
                self._halo_planes(comm, x, lv, min(2, lv.nz) if len(pair) == 2 else 1)
            # (native kernels do not read a zero iterate: u = None -- the bits of the same launch on an array of zeros)
            src = None if (zero and k == 0 and self.native) else x
            if (self.native and len(pair) == 2 and self.dtype == torch.float64 and lv.size >= self.pair_min_cells
                    and self.ops.jacobi2_supported(tuple(x.shape), self.dtype)):
                y = self.spare[l]
                self.ops.poisson_jacobi2(src, b, self.mh2[l], pair[0], pair[1], out=y)  # (== the two sweeps below, bit for bit)
                self.spare[l] = x
                x = y
                pair = []
            for wk in pair:
                y = self.spare[l]
                self.ops.poisson_jacobi(src, b, self.mh2[l], wk, out=y)
                self.spare[l] = x
                x, src = y, y
            k += 2
        return x

    def _residual(self, comm, l, x, b, out):
        """out = A x - b on the owned planes; -> its squared norm over all ranks (finest level only)."""
        lv = self.mlv[l]
        self._halo_planes(comm, x, lv, 1)
        self.ops.poisson_residual(x, b, self.mh2[l], fu=out, loss=self.part, zrange=(lv.g_lo, lv.g_lo + lv.nz), denom=1.0)
        if l:  # (a coarser level's norm is nobody's measure: no collective, no read-back)
            return None
        return float(comm.exchange("gather", self.part.to(torch.float64).reshape(1), None).sum())

    def _restricted_residual(self, comm, l, x, b, out):
        """out = -R (A x - b) of level l's owned planes in ONE pass over the ghost-extended arrays (the fine residual is never
        stored): `out` has half the extended planes -- the owned coarse planes and one plane per interface formed from the
        fine ghost planes, which nobody reads.  -> the squared norm of A x - b over all ranks' owned planes (finest level
        only; `odil_poisson_residual_restrict_slab` counts the rank's own planes)."""
        lv = self.mlv[l]
        self._halo_planes(comm, x, lv, 1)
        self.ops.poisson_residual_restrict(x, b, self.mh2[l], -0.125, out, self.part, zrange=(lv.g_lo, lv.g_lo + lv.nz),
                                           denom=1.0)
        if l:
            return None
        return float(comm.exchange("gather", self.part.to(torch.float64).reshape(1), None).sum())

    def _fused_restriction(self, l, x):
        lv = self.mlv[l]
        shape = tuple(x.shape)
        return (self.native and shape[0] % 2 == 0 and lv.g_lo % 2 == 0
                and self.ops.residual_restrict_supported(shape, self.dtype))

    def _dense_solve(self, b, h2):
        """A^-1 b on a box of at most 512 cells (the residual kernel applied to unit vectors, factorised once)."""
        shape = tuple(b.shape)
        inv = self._dense.get(shape)
        if inv is None:
            n = int(np.prod(shape))
            eye = torch.eye(n, dtype=self.dtype, device=self.device)
            zero = torch.zeros(shape, dtype=self.dtype, device=self.device)
            cols = [self.ops.poisson_residual(eye[j].view(shape).contiguous(), zero, h2)[0].reshape(-1) for j in range(n)]
            amat = torch.stack(cols, dim=1).cpu().numpy().astype(np.float64)
            inv = self._dense[shape] = torch.as_tensor(np.linalg.inv(amat), dtype=self.dtype).to(self.device)
        return (inv @ b.reshape(-1)).view(shape)

    def _local_cycle(self, x, b, h2):
        """One V(nu, nu) cycle on a WHOLE box held by this rank (the agglomerated coarse problem): returns the iterate."""
        shape = tuple(b.shape)
        if int(np.prod(shape)) <= 512 or any(s % 2 or s // 2 < 2 for s in shape):
            if int(np.prod(shape)) <= 512:
                return self._dense_solve(b, h2)
            for _ in range(20):  # (cannot coarsen further: by iteration)
                for wk in self.weights(2):
                    x = self.ops.poisson_jacobi(x, b, h2, wk, out=torch.empty_like(x))
            return x
        for wk in self.weights(self.nu):
            x = self.ops.poisson_jacobi(x, b, h2, wk, out=torch.empty_like(x))
        r, _ = self.ops.poisson_residual(x, b, h2)
        bc = self.ops.restrict_to_coarser(r, "ccc").mul_(-1.0)
        xc = self._local_cycle(torch.zeros_like(bc), bc, [v * 4 for v in h2])
        x = self.ops.interp_add(xc.contiguous(), "ccc", add=x)
        for wk in self.weights(self.nu):
            x = self.ops.poisson_jacobi(x, b, h2, wk, out=torch.empty_like(x))
        return x

    def _coarse_correction(self, comm, l, r):
        """x_c with A_c x_c ~= -R r for the residual r = A x - b of level l (owned planes valid); returned on level l + 1's
        ghost-extended array with owned + inner ghost planes valid."""
        lv = self.mlv[l]
        if l + 1 < len(self.mlv):
            lc = self.mlv[l + 1]
            bc, xc = self.mb[l + 1], self.mx[l + 1]
            if r is not None:  # (else: `_restricted_residual` has written the owned planes of bc)
                lc.owned(bc).copy_(self.ops.restrict_to_coarser(lv.owned(r).contiguous(), "ccc")).mul_(-1.0)
            self._halo_planes(comm, bc, lc, 1)
            if not self.native:
                xc.zero_()
            xc = self._vcycle(comm, l + 1, xc, bc, zero=True)
            self.mx[l + 1] = xc
            self._halo_planes(comm, xc, lc, 1)
            return lc.inner(xc)
        # agglomerated: every rank gets the whole coarse right-hand side and solves the whole coarse problem alike
        if r is not None:
            part = self.ops.restrict_to_coarser(lv.owned(r).contiguous(), "ccc").mul_(-1.0)
        else:
            part = self._agg_part[lv.g_lo // 2: lv.g_lo // 2 + lv.nz // 2]
        bc = comm.exchange("gather", part.contiguous(), None).reshape(self.agg_shape).contiguous()
        if self.agg_gmg is not None:
            xc = self.agg_gmg.vcycle(0, torch.zeros_like(bc), bc, zero=True)
        else:
            xc = self._local_cycle(torch.zeros_like(bc), bc, self.agg_h2)
        nzc = lv.nz // 2
        lo = self.rank * nzc - (1 if self.rank > 0 else 0)
        hi = (self.rank + 1) * nzc + (1 if self.rank < self.world - 1 else 0)
        return xc[lo:hi]

    def _vcycle(self, comm, l, x, b, pre=True, zero=False):
        """One V(nu, nu) cycle on level l of the slab hierarchy: returns the tensor holding the iterate (owned planes)."""
        lv = self.mlv[l]
        if pre:
            x = self._smooth(comm, l, x, b, zero=zero)
        r = self.res[l]
        coarser = l + 1 < len(self.mlv) or self.agg_shape is not None
        if coarser and self._fused_restriction(l, x):
            if l + 1 < len(self.mlv):  # straight into the coarse right-hand side: its owned planes + one per interface
                lc = self.mlv[l + 1]
                target = self.mb[l + 1][lc.g_lo - lv.g_lo // 2: lc.g_lo + lc.nz + lv.g_hi // 2]
            else:
                if getattr(self, "_agg_part", None) is None:
                    self._agg_part = torch.empty(tuple(n // 2 for n in x.shape), dtype=self.dtype, device=self.device)
                target = self._agg_part
            res2 = self._restricted_residual(comm, l, x, b, target)
            r = None
        else:
            res2 = self._residual(comm, l, x, b, r)
        if l == 0:  # (the coarser levels' residuals belong to THEIR systems: only the finest one is the solve's measure)
            self._last_res2 = res2
        if coarser:
            xc = self._coarse_correction(comm, l, r)
            y = self.spare[l]
            self.ops.interp_add(xc.contiguous(), "ccc", add=x, out=y)
            self.spare[l] = x
            x = y
        return self._smooth(comm, l, x, b)

    # ---- the Newton step ----------------------------------------------------------------------------------------------
    def step(self, comm, maxiter=40, tol=1e-10, damp=0.0):
        """One Newton step u <- u + delta with A delta = -f(u) by V-cycles; -> (loss before, loss after).
        self.status: cycles, relative residual of the linear system, converged."""
        assert not damp, "the multigrid step solves the undamped system"
        loss0 = self.residual(comm, self.u, self.f)
        lv = self.lv
        b = self.z
        b.copy_(self.f).mul_(-1.0)  # owned planes valid; the neighbours' planes of b follow
        self._halo_planes(comm, b, lv, 1)
        bb = float(comm.exchange("gather", (lv.owned(b).to(torch.float64) ** 2).sum().reshape(1), None).sum())
        x = torch.zeros_like(self.u)
        it, rel = 0, 1.0
        while it < maxiter:
            x = self._vcycle(comm, 0, x, b)
            it += 1
            # (the residual a cycle forms on its way down belongs to its pre-smoothed iterate: a cheap, slightly
            # pessimistic convergence test that costs no pass of its own)
            rel = float(np.sqrt(self._last_res2 / bb)) if bb > 0 else 0.0
            if rel <= tol:
                break
        self.owned(self.u).add_(self.owned(x))
        loss1 = self.residual(comm, self.u, self.f)
        self.status = dict(niter=it, residual=rel, converged=rel <= tol, method="slab gmg-vcycle ({} slab levels{})".format(
            len(self.mlv), " + agglomerated {}".format(self.agg_shape) if self.agg_shape else ""))
        return loss0, loss1


class SlabStencilGMG:
    """Geometric multigrid for ANY (2 d + 1)-point operator with variable coefficients on a slab-decomposed 3-D grid -- the
    slab form of `gmg.StencilGMG` (the Newton system M delta = -r of a single-field operator from its Jacobian's coefficient
    arrays, reference src/odil/core.py:1113-1217, linsolver.py:17-26; SURVEY 8 E: "Newton: matrix-free M / M^T apply = same
    halo pattern").  The global grid (world nz, ny, nx) is cut along axis 0, whose two ENDS are walls (zero coefficients
    towards them; a periodic cut axis would need the ring closure).

    Every rank holds its planes of the 7 coefficient arrays.  Levels are coarsened by 2 along every axis while a rank keeps
    >= 2 planes and the cross-section >= 4 cells; level arrays and coefficient arrays are ghost-extended (`slab.SlabLevel`,
    G = 2 planes per interior interface).  Set-up, once per solve (the coefficients change with the state): two boundary
    planes of the seven arrays to each neighbour (ONE packed message per level); the coarse operators by
    `odil_stencil_var_coarsen` on the extended arrays -- aggregates of 2^3 cells never straddle an interface, the
    matrix-symmetric split reads the neighbour's coefficient one plane into the valid ghosts.  Below the last slab level the
    problem is AGGLOMERATED: coefficient arrays and right-hand side are all-gathered and every rank runs the rest of the
    cycle on the whole coarse box (the same bits on every rank).  One V(nu, nu) cycle per level, with the unmodified
    single-GPU kernels on the extended arrays (their periodic wrap at an array end only reaches the outer ghost plane):
    two planes of x, two sweeps (as ONE pass, `odil_stencil_var_smooth2`, on large levels); one plane of x, the residual on
    the owned planes (its norm by one all-gather), the mean of the children as the coarse right-hand side, one plane of
    it; the coarse correction, one plane of it, x += P x_c; two planes, two sweeps."""

    def __init__(self, coeffs, rank, world, ops=None, nu=2, pair_min_cells=128**3, agg_cells=32**3):
        """agg_cells: as SlabPoissonNewtonGMG -- levels of at most this many cells per rank are not slab levels but part of
        the agglomerated box (0: slab levels down to two planes per rank)."""
        assert coeffs.dim() == 4 and coeffs.shape[0] == 7 and coeffs.is_contiguous()
        self.ops = ops or hip_ops
        self._native = getattr(self.ops, "__name__", "") == "odil_amd.ops"  # (the library's kernels, not a stand-in)
        self.rank, self.world, self.nu, self.pair_min_cells = rank, world, nu, pair_min_cells
        self.dtype, self.device = coeffs.dtype, coeffs.device
        nz, ny, nx = (int(v) for v in coeffs.shape[1:])
        shape = (nz, ny, nx)
        self.mlv = [slab.SlabLevel(nz, ny, nx, rank, world)]
        while (all(v % 2 == 0 for v in shape) and shape[0] // 2 >= 2 and min(shape[1], shape[2]) // 2 >= 4
               and int(np.prod(shape)) // 8 > agg_cells):
            shape = tuple(v // 2 for v in shape)
            self.mlv.append(slab.SlabLevel(shape[0], shape[1], shape[2], rank, world))
        self.c_owned = coeffs
        self.mc = None  # ghost-extended coefficient arrays per slab level, built by setup(comm)
        self.status = dict()

    @staticmethod
    def weights(n):
        lo, hi = 1.0 / 3.0, 2.0  # (gmg.PoissonGMG.weights, d = 3)
        mid, half = 0.5 * (hi + lo), 0.5 * (hi - lo)
        return [1.0 / (mid - half * np.cos(np.pi * (2 * k + 1) / (2 * n))) for k in range(n)]

    # ---- exchanges ----------------------------------------------------------------------------------------------------
    def _halo_planes(self, comm, a, lv, planes):
        """`planes` boundary planes of `a` ([..., z, y, x]: one array or a stack of arrays) into the neighbours' ghosts."""
        lo = a[..., lv.g_lo: lv.g_lo + planes, :, :].contiguous() if self.rank > 0 else None
        hi = a[..., lv.g_lo + lv.nz - planes: lv.g_lo + lv.nz, :, :].contiguous() if self.rank < self.world - 1 else None
        recv_lo, recv_hi = comm.exchange("halo", lo, hi)
        if recv_lo is not None:
            a[..., lv.g_lo - planes: lv.g_lo, :, :].copy_(recv_lo.view(lo.shape))
        if recv_hi is not None:
            a[..., lv.g_lo + lv.nz: lv.g_lo + lv.nz + planes, :, :].copy_(recv_hi.view(hi.shape))

    def setup(self, comm):
        """The operators of every level (see the class text); the agglomerated hierarchy below the last slab level."""
        mk = lambda lv, lead=(): torch.zeros(tuple(lead) + lv.shape, dtype=self.dtype, device=self.device)
        self.mc = []
        c = mk(self.mlv[0], (7,))
        c[:, self.mlv[0].g_lo: self.mlv[0].g_lo + self.mlv[0].nz].copy_(self.c_owned)
        self._unit_ghost_diagonal(c, self.mlv[0])
        self._halo_planes(comm, c, self.mlv[0], min(slab.G, self.mlv[0].nz))
        self.mc.append(c)
        for l in range(1, len(self.mlv)):
            fine, lf, lc = self.mc[-1], self.mlv[l - 1], self.mlv[l]
            cc_all = self.ops.stencil_var_coarsen(fine.contiguous())
            c = mk(lc, (7,))
            c[:, lc.g_lo: lc.g_lo + lc.nz].copy_(cc_all[:, lf.g_lo // 2: lf.g_lo // 2 + lc.nz])
            self._unit_ghost_diagonal(c, lc)
            self._halo_planes(comm, c, lc, min(slab.G, lc.nz))
            self.mc.append(c)
        self.mx = [None] + [mk(lv) for lv in self.mlv[1:]]
        self.mb = [None] + [mk(lv) for lv in self.mlv[1:]]
        self.spare = [mk(lv) for lv in self.mlv]
        self.res = [mk(lv) for lv in self.mlv]
        # the agglomerated box: every rank gets the whole coarse operator of the level below the last slab level
        last, ll = self.mc[-1], self.mlv[-1]
        self.agg = None
        if ll.nz % 2 == 0 and ll.ny % 2 == 0 and ll.nx % 2 == 0 and min(ll.ny, ll.nx) >= 4:
            cc_all = self.ops.stencil_var_coarsen(last.contiguous())
            part = cc_all[:, ll.g_lo // 2: ll.g_lo // 2 + ll.nz // 2].contiguous()
            rows = comm.exchange("gather", part, None)  # (world, 7, nz / 2, ny / 2, nx / 2)
            whole = torch.cat([rows[r] for r in range(self.world)], dim=1).contiguous()
            self.agg = [whole]
            while all(v % 2 == 0 and v // 2 >= 2 for v in self.agg[-1].shape[1:]) and self.agg[-1][0].numel() > 512:
                self.agg.append(self.ops.stencil_var_coarsen(self.agg[-1]))
            self._agg_inv = None
            # with the library's own kernels underneath, the box is cycled by the single-GPU solver (its paired sweeps and
            # one-launch coarse tail); a stand-in `ops` (the host tests) keeps the plain recursion of `_agg_cycle`
            self.agg_gmg = None
            if getattr(self.ops, "__name__", "") == "odil_amd.ops" and min(whole.shape[1:]) >= 4:
                from . import gmg

                self.agg_gmg = gmg.StencilGMG(whole)
        self.part = torch.zeros((), dtype=self.dtype, device=self.device)

    @staticmethod
    def _unit_ghost_diagonal(c, lv):
        """Ghost rows that no neighbour fills (the outer planes of a thin level) keep a unit diagonal: the sweeps divide by it."""
        if lv.g_lo:
            c[0, : lv.g_lo].fill_(1.0)
        if lv.g_hi:
            c[0, lv.g_lo + lv.nz:].fill_(1.0)

    # ---- pieces -------------------------------------------------------------------------------------------------------
    def _sweeps(self, l, x, b, weights, zero=False):
        """zero (the library's own kernels only): the iterate is the zero vector -- the first launch does not read x."""
        c, size = self.mc[l], self.mlv[l].size
        weights = list(weights)
        pair = hasattr(self.ops, "stencil_var_smooth2") and size >= self.pair_min_cells and x.shape[-1] % 2 == 0
        while weights:
            y = self.spare[l]
            src = None if zero else x
            zero = False
            if pair and len(weights) >= 2:
                self.ops.stencil_var_smooth2(c, src, b, weights[0], weights[1], out=y)
                weights = weights[2:]
            else:
                self.ops.stencil_var_smooth(c, src, b, weights[0], out=y)
                weights = weights[1:]
            self.spare[l] = x
            x = y
        return x

    def _smooth(self, comm, l, x, b, zero=False):
        lv, w = self.mlv[l], self.weights(self.nu)
        k = 0
        while k < len(w):
            pair = w[k: k + 2]
            if not (zero and k == 0):  # (a zero iterate has the right ghost planes on every rank)
                self._halo_planes(comm, x, lv, min(2, lv.nz) if len(pair) == 2 else 1)
            x = self._sweeps(l, x, b, pair, zero=zero and k == 0 and self._native)
            k += 2
        return x

    def _residual(self, comm, l, x, b, out):
        """out = b - A x on the owned planes; -> its squared norm over all ranks (finest level only)."""
        lv = self.mlv[l]
        self._halo_planes(comm, x, lv, 1)
        self.ops.stencil_var_residual(self.mc[l], x, b, out=out)
        if l:  # (a coarser level's norm is nobody's measure: no collective, no read-back)
            return None
        mine = (lv.owned(out).to(torch.float64) ** 2).sum().reshape(1)
        return float(comm.exchange("gather", mine, None).sum())

    def _agg_cycle(self, k, x, b):
        """One V(nu, nu) cycle on level k of the agglomerated hierarchy (the whole coarse box on this rank)."""
        c = self.agg[k]
        if k == len(self.agg) - 1:
            if self._agg_inv is None:
                shape = tuple(c.shape[1:])
                n = int(np.prod(shape))
                eye = torch.eye(n, dtype=self.dtype, device=self.device)
                zero = torch.zeros(shape, dtype=self.dtype, device=self.device)
                cols = [-self.ops.stencil_var_residual(c, eye[j].view(shape).contiguous(), zero).reshape(-1) for j in range(n)]
                amat = torch.stack(cols, dim=1).cpu().numpy().astype(np.float64)
                self._agg_inv = torch.as_tensor(np.linalg.pinv(amat, rcond=1e-12), dtype=self.dtype).to(self.device)
            return (self._agg_inv @ b.reshape(-1)).view(b.shape)
        for wk in self.weights(self.nu):
            x = self.ops.stencil_var_smooth(c, x, b, wk, out=torch.empty_like(x))
        r = self.ops.stencil_var_residual(c, x, b)
        bc = self.ops.restrict_to_coarser(r, "ccc")
        xc = self._agg_cycle(k + 1, torch.zeros_like(bc), bc.contiguous())
        x = self.ops.interp_add(xc.contiguous(), "ccc", add=x)
        for wk in self.weights(self.nu):
            x = self.ops.stencil_var_smooth(c, x, b, wk, out=torch.empty_like(x))
        return x

    def _coarse_correction(self, comm, l, r):
        lv = self.mlv[l]
        if l + 1 < len(self.mlv):
            lc = self.mlv[l + 1]
            bc, xc = self.mb[l + 1], self.mx[l + 1]
            if r is not None:  # (else: the fused pass has written the owned planes of bc)
                lc.owned(bc).copy_(self.ops.restrict_to_coarser(lv.owned(r).contiguous(), "ccc"))
            self._halo_planes(comm, bc, lc, 1)
            if not self._native:
                xc.zero_()
            xc = self._vcycle(comm, l + 1, xc, bc, zero=True)
            if l == 0 and len(self.mlv) + (len(self.agg) if self.agg else 0) > 2:
                # TWO cycles on the first coarse level, as gmg.StencilGMG.finish_cycle: the aggregation-built coarse
                # operators are less accurate at the walls than a rediscretisation (0.24 -> 0.14 per cycle there)
                xc = self._vcycle(comm, l + 1, xc, bc)
            self.mx[l + 1] = xc
            self._halo_planes(comm, xc, lc, 1)
            return lc.inner(xc)
        if r is not None:
            part = self.ops.restrict_to_coarser(lv.owned(r).contiguous(), "ccc")
        else:
            part = self._agg_part[lv.g_lo // 2: lv.g_lo // 2 + lv.nz // 2]
        bc = comm.exchange("gather", part.contiguous(), None).reshape(tuple(self.agg[0].shape[1:]))
        bc = bc.contiguous()
        # (the box is the FIRST coarse level when there is one slab level: two cycles there, as above)
        twice = l == 0 and len(self.agg) > 1
        if self.agg_gmg is not None:
            xc = self.agg_gmg.vcycle(0, torch.zeros_like(bc), bc, zero=True)
            if twice:
                xc = self.agg_gmg.vcycle(0, xc, bc)
        else:
            xc = self._agg_cycle(0, torch.zeros_like(bc), bc)
            if twice:
                xc = self._agg_cycle(0, xc, bc)
        nzc = lv.nz // 2
        lo = self.rank * nzc - (1 if self.rank > 0 else 0)
        hi = (self.rank + 1) * nzc + (1 if self.rank < self.world - 1 else 0)
        return xc[lo:hi]

    def _vcycle(self, comm, l, x, b, zero=False):
        x = self._smooth(comm, l, x, b, zero=zero)
        r, lv = self.res[l], self.mlv[l]
        coarser = l + 1 < len(self.mlv) or self.agg is not None
        if (coarser and getattr(self.ops, "__name__", "") == "odil_amd.ops" and x.shape[0] % 2 == 0 and lv.g_lo % 2 == 0
                and x.shape[2] % 2 == 0):
            # residual and restriction in ONE pass over the ghost-extended arrays (the fine residual is never stored),
            # straight into the coarse right-hand side: its owned planes + one plane per interface that nobody reads
            if l + 1 < len(self.mlv):
                lc = self.mlv[l + 1]
                target = self.mb[l + 1][lc.g_lo - lv.g_lo // 2: lc.g_lo + lc.nz + lv.g_hi // 2]
            else:
                if getattr(self, "_agg_part", None) is None:
                    self._agg_part = torch.empty(tuple(n // 2 for n in x.shape), dtype=self.dtype, device=self.device)
                target = self._agg_part
            self._halo_planes(comm, x, lv, 1)
            self.ops.stencil_var_residual_restrict(self.mc[l], x, b, 0.125, target, self.part,
                                                   zrange=(lv.g_lo, lv.g_lo + lv.nz), denom=1.0)
            res2 = None if l else float(comm.exchange("gather", self.part.to(torch.float64).reshape(1), None).sum())
            r = None
        else:
            res2 = self._residual(comm, l, x, b, r)
        if l == 0:
            self._last_res2 = res2
        if coarser:
            xc = self._coarse_correction(comm, l, r)
            y = self.spare[l]
            self.ops.interp_add(xc.contiguous(), "ccc", add=x, out=y)
            self.spare[l] = x
            x = y
        return self._smooth(comm, l, x, b)

    def solve(self, comm, b_owned, tol=1e-10, maxiter=40):
        """x (owned planes) with A x = b to |b - A x| <= tol |b| over all ranks; self.status: niter, residual, converged."""
        if self.mc is None:
            self.setup(comm)
        lv = self.mlv[0]
        b = torch.zeros(lv.shape, dtype=self.dtype, device=self.device)
        lv.owned(b).copy_(b_owned)
        self._halo_planes(comm, b, lv, 1)
        bb = float(comm.exchange("gather", (b_owned.to(torch.float64) ** 2).sum().reshape(1), None).sum())
        x = torch.zeros_like(b)
        it, rel = 0, 1.0
        while it < maxiter:
            x = self._vcycle(comm, 0, x, b)
            it += 1
            rel = float(np.sqrt(self._last_res2 / bb)) if bb > 0 else 0.0  # (of the cycle's pre-smoothed iterate)
            if rel <= tol:
                break
        self.status = dict(niter=it, residual=rel, converged=rel <= tol, method="slab variable-coefficient gmg ({} slab levels{})".format(
            len(self.mlv), " + agglomerated {}".format(tuple(self.agg[0].shape[1:])) if self.agg else ""))
        return lv.owned(x).clone()


class ReplicatedTailVectors(SlabLbfgsVectors):
    """`SlabLbfgsVectors` for a local vector [owned entries | entries EVERY rank holds] (coarse levels agglomerated on
    every rank, network parameters): the tail takes part in every rank's vector algebra -- all ranks then update it alike,
    from the same combined scalars -- but must count ONCE in the reductions: every rank enters its tail products with
    the weight 1 / world (whole-vector product minus (1 - 1 / world) of the tail's, two launches instead of one)."""

    def __init__(self, n, m, device, comm, n_own, world):
        super().__init__(n, m, device, comm)
        self.n_own, self.excess = n_own, 1.0 - 1.0 / world
        self.tail = n > n_own and world > 1
        self.tscal = torch.zeros(8, dtype=torch.float64, device=device)

    def probe_direction(self, d, g):
        super().probe_direction(d, g)
        if self.tail:
            k = self.n_own
            hip_ops.dots3(d[None, k:], [d[k:], g[k:]], out=self.tscal[0:3].view(3, 1))

    def probe_eval(self, f, g, d):
        super().probe_eval(f, g, d)
        if self.tail:
            k = self.n_own
            hip_ops.dots3(g[None, k:], [d[k:], g[k:]], out=self.tscal[3:6].view(3, 1))  # <g, d>, <g, g> of the tail

    def reduce_probes(self, scal):
        if self.tail:
            fixed = scal.clone()
            fixed[0:2] -= self.excess * self.tscal[0:2]
            fixed[3:5] -= self.excess * self.tscal[3:5]
            scal = fixed
        return super().reduce_probes(scal)

    def history_products(self, nphys, bs):
        if nphys == 0 or not self.tail:
            return super().history_products(nphys, bs)
        k = self.n_own
        whole = hip_ops.dots3(self.w[: 2 * nphys], bs)
        tail = hip_ops.dots3(self.w[: 2 * nphys, k:], [b[k:] for b in bs])
        out = self.reduce_sums(whole - self.excess * tail).cpu().numpy()[: len(bs)]
        return out[:, 0::2], out[:, 1::2]

    def dot(self, a, b):
        k = self.n_own
        whole = hip_ops.dots3(a[None], [b])
        if self.tail:
            whole = whole - self.excess * hip_ops.dots3(a[None, k:], [b[k:]])
        return float(self.reduce_sums(whole).cpu().numpy()[0, 0])


class SlabTracedLbfgs:
    """L-BFGS-B for ANY traced operator on the slab decomposition of `slab_traced.SlabTracedAdam` (whose evaluation it
    borrows: `epoch_gen(update=False)`).  The local vector: this rank's owned planes of every level of every field,
    then what every rank holds whole (agglomerated coarse levels, network / `Array` parameters)."""

    def __init__(self, run):
        self.run = run
        own, rep = [], []
        dev = run.device
        for e in run.entries:
            shapes = [lv.shape for lv in e["levels"]] if "levels" in e else e["shapes"]
            pos = e["start"]
            for k, shape in enumerate(shapes):
                cnt = int(np.prod(shape)) if len(shape) else 1
                index = torch.arange(pos, pos + cnt, dtype=torch.int64, device=dev).view(tuple(shape))
                if "levels" in e and not e["levels"][k].replicated:
                    own.append(e["levels"][k].owned(index).reshape(-1))
                else:
                    rep.append(index.reshape(-1))
                pos += cnt
        self.n_own = int(sum(t.numel() for t in own))
        self.index = torch.cat(own + rep) if own or rep else torch.zeros(0, dtype=torch.int64, device=dev)
        self.n = int(self.index.numel())
        self.nfev = 0

    def pack(self, flat, out=None):
        res = flat.index_select(0, self.index).to(torch.float64)
        if out is None:
            return res
        out.copy_(res)
        return out

    def unpack(self, vec, flat):
        flat.index_copy_(0, self.index, vec.to(flat.dtype))

    def local_loss(self):
        """This rank's share of the loss of the last evaluation (the ranks' shares add up to the loss)."""
        run = self.run
        part = run.kern.partial_terms().to(torch.float64).sum()
        if getattr(run.kern, "par_outputs", None) is not None and hasattr(run.kern, "pout") and run.rank == 0:
            for q in range(len(run.kern.par_outputs)):  # evaluated by every rank alike: counted once
                part = part + run.kern.pout[2 * q].to(torch.float64)
        return part

    def minimize(self, comm, maxiter, m=50, maxls=50, pgtol=1e-16, factr=0.0, callback=None, vectors=None):
        """-> the dict of `lbfgsb_minimize` (f = GLOBAL loss); the unknowns of `self.run` hold the result."""
        run = self.run
        vec = vectors or ReplicatedTailVectors(self.n, m, run.device, comm, self.n_own, run.world)
        x = self.pack(run.x)
        gflat = torch.empty(self.n, dtype=torch.float64, device=run.device)

        def fg(xflat):
            self.unpack(xflat, run.x)
            drive(run.epoch_gen(update=False), comm)
            self.nfev += 1
            return self.local_loss(), self.pack(run.g, out=gflat)

        res = lbfgsb_minimize(x, fg, vec, maxiter, m=m, maxls=maxls, pgtol=pgtol, factr=factr, callback=callback)
        self.unpack(x, run.x)
        run._x_synced = False  # the ghost planes of the unknowns are refreshed by whoever evaluates next
        return res
