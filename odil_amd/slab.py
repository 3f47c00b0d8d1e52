"""Slab decomposition of the Poisson multigrid hot path over the GPUs of one node
(no reference counterpart: the reference is single-device).

Layout.  The global grid (P*Nz, Ny, Nx) is cut along axis 0; rank r owns Nz planes of
every multigrid level (levels halve all axes, and the level count is set by the smallest
axis, so every level keeps >= 2 planes per rank: no agglomeration is needed).  Each level
array is stored ghost-extended: G = 2 extra planes at every interior interface, none at a
wall.  Only the INNER ghost plane ever needs valid data; the outer one exists so that the
unmodified single-GPU kernels, run on the extended array as if it were a whole domain,
compute exact values on all owned planes (their wall formulas only ever touch the discarded
ghost planes, except at true walls where they are the physics).  Two kernels know about
the cut: P^T drops the wall weights at a cut end (`odil_interp_adj_cut`), and the residual
restricts its loss sum to the owned planes (`odil_poisson_residual_slab`).

Per epoch each rank exchanges single planes with its two neighbours over RCCL/xGMI
(point-to-point, one direct link per neighbour pair):
  1. the first / last owned plane of every level of w, packed in ONE message per neighbour;
     P then produces u's inner ghost plane by itself (no exchange for u);
  2. the first / last owned plane of fu (the adjoint reads fu at z +- 1);
  3. before each level of the P^T chain, the first / last owned plane of that level's
     cotangent (8 small messages).
No field-sized collective exists; the loss needs one scalar all-reduce, issued only when a
loss value is actually asked for.  Adam is local.

The epoch is written as a generator that yields at every exchange, so the same code runs
(a) one rank per GPU under torch.distributed (RCCL), (b) over gloo on CPU in the tests with
an oracle-backed `ops` double, and (c) with several ranks emulated in one process on one
GPU (tests), where the HIP kernels themselves are checked against the undivided domain.
"""

import math

import numpy as np
import torch

from . import ops as hip_ops
from .poisson_path import mg_cshapes

G = 2  # ghost planes per interior interface


class SlabLevel:
    def __init__(self, nz, ny, nx, rank, world):
        self.nz, self.ny, self.nx = nz, ny, nx
        self.g_lo = 0 if rank == 0 else G
        self.g_hi = 0 if rank == world - 1 else G
        self.shape = (self.g_lo + nz + self.g_hi, ny, nx)
        self.size = math.prod(self.shape)
        self.plane = ny * nx

    def owned(self, a):
        return a[self.g_lo : self.g_lo + self.nz]

    def inner(self, a):
        """View with ONE ghost plane per interior interface (what P takes as its coarse operand)."""
        lo = self.g_lo - 1 if self.g_lo else 0
        hi = a.shape[0] - (self.g_hi - 1 if self.g_hi else 0)
        return a[lo:hi]


def hat_reference_slab(levels0, N, rank, world, dtype, device):
    """'hat' reference solution (reference examples/poisson/poisson.py:21-24) in normalised
    coordinates of the global box, evaluated on this rank's ghost-extended planes."""
    lv = levels0
    nzg = N * world
    z = (torch.arange(lv.shape[0], dtype=torch.float64, device=device) + (rank * N - lv.g_lo) + 0.5) / nzg
    y = (torch.arange(lv.ny, dtype=torch.float64, device=device) + 0.5) / lv.ny
    x = (torch.arange(lv.nx, dtype=torch.float64, device=device) + 0.5) / lv.nx
    u = torch.ones(lv.shape, dtype=torch.float64, device=device)
    for c, shp in ((z, (-1, 1, 1)), (y, (1, -1, 1)), (x, (1, 1, -1))):
        u = u * ((1 - c) * c * 5).reshape(shp)
    p = 5
    return ((u**p / (1 + u**p)) ** (1 / p)).to(dtype)


class SlabPoissonAdam:
    """One rank of the slab-decomposed Poisson multigrid Adam loop (3-D, all cell-centred)."""

    def __init__(self, N, rank, world, dtype=torch.float64, device=None, lr=0.005, beta_1=0.9, beta_2=0.999,
                 epsilon=1e-7, rhs_global=None):
        self.ops = hip_ops  # the HIP kernels (tests of the exchange logic without a GPU swap this module attribute)
        self.N, self.rank, self.world = N, rank, world
        self.dtype, self.device = dtype, device
        self.npdt = np.float64 if dtype == torch.float64 else np.float32
        cglobal = (N * world, N, N)
        self.global_cells = math.prod(cglobal)
        self.local_cells = N**3
        shapes_global = mg_cshapes(cglobal)
        self.nlvl = len(shapes_global)
        self.levels = [SlabLevel(s[0] // world, s[1], s[2], rank, world) for s in shapes_global]
        for lv in self.levels:
            assert lv.nz >= 2, "every level needs >= 2 owned planes per rank"
        self.h2 = [self.npdt(1.0 / N) ** 2] * 3  # box (world, 1, 1): uniform spacing 1/N
        sizes = [lv.size for lv in self.levels]
        n = sum(sizes)
        self.n_unknowns_local = sum(lv.nz * lv.plane for lv in self.levels)
        mk = lambda: torch.zeros(n, dtype=dtype, device=device)
        self.x, self.m, self.v, self.g = mk(), mk(), mk(), mk()
        split = lambda f: [t.view(lv.shape) for t, lv in zip(f.split(sizes), self.levels)]
        self.w, self.gw = split(self.x), split(self.g)
        self.mw, self.vw = split(self.m), split(self.v)
        # positions, in the packed state, of the boundary planes of EVERY level: the start-of-epoch exchange of
        # all level arrays is then one gather and one scatter per neighbour instead of ~20 slice copies
        starts = np.concatenate([[0], np.cumsum(sizes)[:-1]])

        def plane_index(plane_of):
            parts = []
            for st, lv in zip(starts, self.levels):
                k = plane_of(lv)
                if k is None:
                    return None
                parts.append(int(st) + k * lv.plane + np.arange(lv.plane, dtype=np.int64))
            return torch.as_tensor(np.concatenate(parts), device=device)

        self._own_idx = dict(lo=plane_index(lambda lv: lv.g_lo if rank > 0 else None),
                             hi=plane_index(lambda lv: lv.g_lo + lv.nz - 1 if rank < world - 1 else None))
        self._ghost_idx = dict(lo=plane_index(lambda lv: lv.g_lo - 1 if rank > 0 else None),
                               hi=plane_index(lambda lv: lv.g_lo + lv.nz if rank < world - 1 else None))
        l0 = self.levels[0]
        self._u = None  # synthesised field of the two-kernel path (the fused residual never stores it)
        self.fu = torch.zeros(l0.shape, dtype=dtype, device=device)
        self.work = [None] + [torch.zeros(lv.shape, dtype=dtype, device=device) for lv in self.levels[1:-1]] + [None]
        self.loss_part = torch.zeros((), dtype=dtype, device=device)
        if rhs_global is not None:
            lo = rank * N - l0.g_lo
            self.rhs = rhs_global[lo : lo + l0.shape[0]].to(device=device, dtype=dtype).contiguous()
        else:
            ref_u = hat_reference_slab(l0, N, rank, world, dtype, device)
            self.rhs, _ = self.ops.poisson_residual(ref_u, torch.zeros_like(ref_u), self.h2)
        self.lr, self.b1, self.b2, self.eps = self.npdt(lr), self.npdt(beta_1), self.npdt(beta_2), epsilon
        self.t = 0
        self.scale = self.npdt(2) / self.npdt(self.global_cells)
        import os

        self.fuse_transpose = bool(int(os.environ.get("ODIL_FUSE_TRANSPOSE", 1)))

    @property
    def u(self):
        if self._u is None:
            self._u = torch.zeros(self.levels[0].shape, dtype=self.dtype, device=self.device)
        return self._u

    # ---- plane packing -----------------------------------------------------------------
    def _pack(self, arrays, levels, side, depth=1):
        """The first (side 'lo') or last ('hi') `depth` OWNED planes of each array, concatenated; None at a wall."""
        if (side == "lo" and self.rank == 0) or (side == "hi" and self.rank == self.world - 1):
            return None
        parts = []
        for a, lv in zip(arrays, levels):
            k = lv.g_lo if side == "lo" else lv.g_lo + lv.nz - depth
            parts.append(a[k : k + depth].reshape(-1))
        return torch.cat(parts) if len(parts) > 1 else parts[0].clone()

    def _unpack(self, buf, arrays, levels, side, depth=1):
        """Received planes -> the `depth` ghost planes next to the owned ones on `side` of each array."""
        if buf is None:
            return
        off = 0
        for a, lv in zip(arrays, levels):
            k = lv.g_lo - depth if side == "lo" else lv.g_lo + lv.nz
            a[k : k + depth].copy_(buf[off : off + depth * lv.plane].view(depth, lv.ny, lv.nx))
            off += depth * lv.plane

    def _exchange_state(self):
        """Generator step: the boundary planes of all level arrays of the packed state x, one message per
        neighbour (same planes, same order as _exchange(self.w, self.levels))."""
        lo, hi = self._own_idx["lo"], self._own_idx["hi"]
        recv_lo, recv_hi = yield ("halo", None if lo is None else self.x.index_select(0, lo),
                                  None if hi is None else self.x.index_select(0, hi))
        if recv_lo is not None:
            self.x.index_copy_(0, self._ghost_idx["lo"], recv_lo)
        if recv_hi is not None:
            self.x.index_copy_(0, self._ghost_idx["hi"], recv_hi)

    def _exchange(self, arrays, levels, depth=1):
        """Generator step: swap boundary planes of `arrays` with both neighbours."""
        recv_lo, recv_hi = yield ("halo", self._pack(arrays, levels, "lo", depth), self._pack(arrays, levels, "hi", depth))
        self._unpack(recv_lo, arrays, levels, "lo", depth)
        self._unpack(recv_hi, arrays, levels, "hi", depth)

    # ---- one epoch -----------------------------------------------------------------------
    def epoch_gen(self, timers=None):
        ops, lv = self.ops, self.levels
        L = self.nlvl

        def tic(name):
            if timers is None:
                return None
            a, b = timers.section(name)
            a.record()
            return b

        def toc(b):
            if b is not None:
                b.record()

        b = tic("halo")
        yield from self._exchange_state()
        toc(b)
        # u = w_0 + P(w_1 + P(...)): coarse operand with one ghost plane -> fine with two
        # the last prolongation is fused into the residual when the kernel set has it (u never stored)
        fused_last = hasattr(ops, "poisson_residual_synth") and L >= 2
        b = tic("mg_synth")
        coarse = lv[L - 1].inner(self.w[L - 1])
        for l in range(L - 2, 0 if fused_last else -1, -1):
            out = self.u if l == 0 else self.work[l]
            ops.interp_add(coarse.contiguous(), "ccc", add=self.w[l], out=out)
            coarse = lv[l].inner(out)
        toc(b)
        b = tic("residual")
        l0 = lv[0]
        if fused_last:
            ops.poisson_residual_synth(coarse.contiguous(), self.w[0], self.rhs, self.h2, fu=self.fu,
                                       loss=self.loss_part, zrange=(l0.g_lo, l0.g_lo + l0.nz),
                                       denom=self.global_cells)
        else:
            ops.poisson_residual(self.u, self.rhs, self.h2, fu=self.fu, loss=self.loss_part,
                                 zrange=(l0.g_lo, l0.g_lo + l0.nz), denom=self.global_cells)
        toc(b)
        n0 = lv[0].size
        fuse0 = hasattr(ops, "poisson_adjoint_adam")
        # stencil adjoint + first transposed prolongation + Adam of levels 0 and 1 in one launch, as on one GPU
        # (fused.py): the finest-level gradient is never stored, so its halo exchange disappears; the residual
        # then needs BOTH ghost planes (g0 of the inner ghost plane reads the outer one).
        fuse_t = (fuse0 and L >= 2 and self.fuse_transpose and hasattr(ops, "poisson_adjoint_transpose")
                  and G >= 2 and lv[0].nz >= 2 and ops.adjoint_transpose_supported(tuple(lv[0].shape))
                  and tuple(lv[1].inner(self.gw[1]).shape) == tuple(n // 2 for n in lv[0].shape))
        b = tic("halo")
        yield from self._exchange([self.fu], [l0], depth=2 if fuse_t else 1)
        toc(b)
        self.t += 1
        t = self.npdt(self.t)
        alpha = self.lr * np.sqrt(1 - self.b2**t) / (1 - self.b1**t)
        omb1, omb2 = 1 - self.b1, 1 - self.b2
        cut = (self.rank > 0, self.rank < self.world - 1)
        first = 1  # first level whose transposed prolongation is still to do
        if fuse_t:
            # The interfaces are `cut` ends: interior stencil rows there, and the owned coarse planes only
            # read fine planes whose g0 is complete (the outermost ghost plane has zero weight).
            b = tic("adjoint_transpose")
            ops.poisson_adjoint_transpose(
                self.fu, self.h2, self.scale, lv[1].inner(self.gw[1]), g0=None,
                adam0=(self.w[0], self.m[:n0].view(lv[0].shape), self.v[:n0].view(lv[0].shape)),
                adam1=(lv[1].inner(self.w[1]), lv[1].inner(self.mw[1]), lv[1].inner(self.vw[1])),
                alpha=alpha, one_minus_b1=omb1, one_minus_b2=omb2, eps=self.eps, cut=cut)
            toc(b)
            first = 2
        b = tic("adjoint") if first == 1 else None
        if first == 2:
            pass
        elif fuse0:
            # Adam of the finest level inside the adjoint launch (as on one GPU, poisson_path.py)
            ops.poisson_adjoint_adam(self.fu, self.h2, self.scale, self.gw[0], self.w[0],
                                     self.m[:n0].view(lv[0].shape), self.v[:n0].view(lv[0].shape), alpha, omb1, omb2,
                                     self.eps)
        else:
            ops.poisson_adjoint(self.fu, self.h2, self.scale, out=self.gw[0])
        toc(b)
        for l in range(first, L):
            b = tic("halo")
            yield from self._exchange([self.gw[l - 1]], [lv[l - 1]])
            toc(b)
            b = tic("mg_synth_adj")
            cview = lv[l].inner(self.gw[l])
            if fuse0 and hasattr(ops, "interp_adj_adam"):
                ops.interp_adj_adam(self.gw[l - 1], "ccc", tuple(cview.shape), cview, lv[l].inner(self.w[l]),
                                    lv[l].inner(self.mw[l]), lv[l].inner(self.vw[l]), alpha, omb1, omb2, self.eps,
                                    cut=cut)
            else:
                ops.interp_adj(self.gw[l - 1], "ccc", tuple(cview.shape), out=cview, cut=cut)
            toc(b)
        if fuse0 and hasattr(ops, "interp_adj_adam"):
            return  # every level was updated inside the launch that formed its gradient
        b = tic("adam")
        if fuse0:
            ops.adam_step(self.x[n0:], self.m[n0:], self.v[n0:], self.g[n0:], alpha, omb1, omb2, self.eps)
        else:
            ops.adam_step(self.x, self.m, self.v, self.g, alpha, omb1, omb2, self.eps)
        toc(b)

    def epoch(self, comm, timers=None):
        gen = self.epoch_gen(timers)
        try:
            msg = next(gen)
            while True:
                msg = gen.send(comm.exchange(*msg))
        except StopIteration:
            pass

    def last_loss(self, comm=None):
        """Global loss of the last epoch (sum of the ranks' partial means)."""
        part = self.loss_part.clone()
        if comm is not None:
            part = comm.exchange("sum", part, None)
        return float(part)

    def owned_levels(self):
        return [lv.owned(w) for lv, w in zip(self.levels, self.w)]


class TorchDistComm:
    """The exchanges of the slab epochs over torch.distributed (backend nccl = RCCL on ROCm; gloo in tests).

    exchange(kind, send_lo, send_hi):
      "halo"  planes to the lower / upper neighbour (None at an end of the decomposition); returns what the
              neighbours sent: (recv_lo, recv_hi), same sizes as the sends;
      "wrap"  the periodic closure: the FIRST rank's send_lo goes to the LAST rank (arriving as its recv_hi),
              the last rank's send_hi to the first (its recv_lo); other ranks pass (None, None);
      "sum"   all-reduce of send_lo (a few scalars); returns the tensor."""

    def __init__(self, rank, world):
        import torch.distributed as dist

        self.dist, self.rank, self.world = dist, rank, world
        # gloo moves host memory: device planes are staged (tests only; RCCL sends device memory)
        self.stage = dist.get_backend() == "gloo"

    def exchange(self, kind, send_lo, send_hi):
        dist = self.dist
        if kind == "sum":
            t = send_lo
            if self.stage and t.is_cuda:
                h = t.cpu()
                dist.all_reduce(h)
                return h.to(t.device)
            dist.all_reduce(t)
            return t
        if kind == "wrap":
            if self.world == 1:
                return send_hi, send_lo
            peer_lo = self.world - 1 if self.rank == 0 else None
            peer_hi = 0 if self.rank == self.world - 1 else None
        else:
            peer_lo = self.rank - 1 if self.rank > 0 else None
            peer_hi = self.rank + 1 if self.rank < self.world - 1 else None
        dev = None
        if self.stage:
            for t in (send_lo, send_hi):
                if t is not None and t.is_cuda:
                    dev = t.device
            if dev is not None:
                send_lo = send_lo.cpu() if send_lo is not None else None
                send_hi = send_hi.cpu() if send_hi is not None else None
        ops, recv_lo, recv_hi = [], None, None
        if send_lo is not None and peer_lo is not None:
            recv_lo = torch.empty_like(send_lo)
            ops += [dist.P2POp(dist.isend, send_lo, peer_lo), dist.P2POp(dist.irecv, recv_lo, peer_lo)]
        if send_hi is not None and peer_hi is not None:
            recv_hi = torch.empty_like(send_hi)
            ops += [dist.P2POp(dist.isend, send_hi, peer_hi), dist.P2POp(dist.irecv, recv_hi, peer_hi)]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if dev is not None:
            recv_lo = recv_lo.to(dev) if recv_lo is not None else None
            recv_hi = recv_hi.to(dev) if recv_hi is not None else None
        return recv_lo, recv_hi


def run_lockstep(ranks, nepochs=1, timers=None):
    """Several ranks emulated in ONE process (one GPU): advances every rank's epoch generator to
    its next exchange, moves the planes by device copies, continues."""
    P = len(ranks)
    for _ in range(nepochs):
        gens = [r.epoch_gen(timers if i == 0 else None) for i, r in enumerate(ranks)]
        msgs = [next(g) for g in gens]
        alive = True
        while alive:
            kind = msgs[0][0]
            assert all(m[0] == kind for m in msgs), "ranks out of step"
            clone = lambda t: None if t is None else t.clone()
            if kind == "sum":
                total = msgs[0][1].clone()
                for m in msgs[1:]:
                    total = total + m[1]
                replies = [total.clone() for _ in range(P)]
            elif kind == "wrap":
                replies = [(None, None)] * P
                replies[0] = (clone(msgs[P - 1][2]), None)
                replies[P - 1] = (replies[P - 1][0] if P == 1 else None, clone(msgs[0][1]))
            else:
                replies = [(clone(msgs[i - 1][2]) if i > 0 else None, clone(msgs[i + 1][1]) if i + 1 < P else None)
                           for i in range(P)]
            new = []
            for g, rep in zip(gens, replies):
                try:
                    new.append(g.send(rep))
                except StopIteration:
                    alive = False
            msgs = new
