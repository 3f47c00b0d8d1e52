"""Slab decomposition of the Poisson multigrid hot path over the GPUs of one node
(no reference counterpart: the reference is single-device).

Layout.  The global grid (P*Nz, Ny, Nx) is cut along axis 0; rank r owns Nz planes of
every multigrid level (levels halve all axes, and the level count is set by the smallest
axis, so every level keeps >= 2 planes per rank: no agglomeration is needed).  Each level
array is stored ghost-extended: G = 2 extra planes at every interior interface, none at a
wall (only the INNER one ever holds valid data; the outer one exists so that a coarse array with
one ghost plane prolongates to a fine array of the stored shape).  The unmodified single-GPU kernels run on the extended arrays as if they were whole
domains: their wall formulas only ever reach the outermost ghost planes, which nobody reads,
except at true walls where they are the physics.  Two kernels know about the cut: P^T and the
transposed stencil drop the wall rows at a cut end (`cut` flags), and the residual restricts its
loss sum to the owned planes.

TWO exchanges per epoch and neighbour (RCCL point-to-point over the direct xGMI link) instead of one
before every kernel that looks across the interface (round 1: nine):

  1. two planes of fu after the residual (the stencil-adjoint launch forms g0 of the inner ghost
     plane from them, which the first transposed prolongation needs for the owned planes of level 1;
     u itself is never exchanged: P regenerates its ghost plane from the coarse ghost planes);
  2. ONE packed message at the end of the epoch.  P^T is linear: below level 1 every rank pushes only
     ITS OWN contributions down the levels (owned planes of g1, ghost planes zero; a coarse ghost
     plane then holds what this rank contributes to the neighbour's boundary plane) without waiting
     for anybody.  The message carries the updated boundary planes of w0 and w1 (their Adam update
     happened inside the stencil-adjoint launch) and, for every level >= 2, the ghost plane of
     contributions plus this rank's own boundary plane of the partial gradient.  The receiver
     completes the gradient of its boundary plane (own + received contribution) AND of its ghost
     plane (received own part + what it contributed itself: the same two numbers added in the other
     order, so both ranks hold the same bits); the Adam update of levels >= 2 then runs on owned and
     ghost planes alike, and the ghost planes of those levels stay equal to the neighbour's planes
     without ever being sent.
One scalar all-reduce when a loss value is actually asked for.  Adam is local.

The epoch is written as a generator that yields at every exchange, so the same code runs
(a) one rank per GPU under torch.distributed (RCCL), (b) over gloo on CPU in the tests with
oracle doubles of the kernels, and (c) with several ranks emulated in one process on one
GPU (tests), where the HIP kernels themselves are checked against the undivided domain.
"""

import math
import os

import numpy as np
import torch

from . import ops as hip_ops
from .poisson_path import mg_cshapes

G = 2  # ghost planes per interior interface
V = 1  # of which valid (kept equal to the neighbour's plane)


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
        """View with the V valid ghost planes per interior interface (coarse operand of P, result of P^T):
        half the fine array's G."""
        lo = self.g_lo - V if self.g_lo else 0
        hi = a.shape[0] - (self.g_hi - V if self.g_hi else 0)
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
                 epsilon=1e-7, rhs_global=None, moments=True):
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
        assert self.nlvl >= 2
        self.h2 = [self.npdt(1.0 / N) ** 2] * 3  # box (world, 1, 1): uniform spacing 1/N
        sizes = [lv.size for lv in self.levels]
        n = sum(sizes)
        self.n_unknowns_local = sum(lv.nz * lv.plane for lv in self.levels)
        mk = lambda: torch.zeros(n, dtype=dtype, device=device)
        self.x, self.g = mk(), mk()
        self.m, self.v = (mk(), mk()) if moments else (None, None)  # (the quasi-Newton drivers of slab_solvers.py keep none)
        split = lambda f: [t.view(lv.shape) for t, lv in zip(f.split(sizes), self.levels)]
        self.w, self.gw = split(self.x), split(self.g)
        self.mw, self.vw = (split(self.m), split(self.v)) if moments else (None, None)
        self.n01 = sizes[0] + sizes[1]  # levels 0 and 1: updated before the exchange; the rest after it
        self._index_tables(sizes)
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
        self.fuse_transpose = bool(int(os.environ.get("ODIL_FUSE_TRANSPOSE", 1)))
        self.synced = False  # ghost planes of the initial state

    @property
    def u(self):
        if self._u is None:
            self._u = torch.zeros(self.levels[0].shape, dtype=self.dtype, device=self.device)
        return self._u

    # ---- positions of the exchanged planes in the packed vectors ------------------------------
    def _index_tables(self, sizes):
        """Per side ('lo' = towards rank - 1): where the message is read from and where the neighbour's message
        goes.  Message = [x planes of levels 0 and 1 | g planes that are ADDED | g planes that are COPIED]."""
        starts = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        dev = self.device

        def planes(level, rel):
            """flat positions of the planes at owned-relative z positions `rel` of `level`"""
            lv = self.levels[level]
            return np.concatenate([int(starts[level]) + (lv.g_lo + k) * lv.plane + np.arange(lv.plane, dtype=np.int64)
                                   for k in rel])

        self.tab = dict()
        for side in ("lo", "hi"):
            if (side == "lo" and self.rank == 0) or (side == "hi" and self.rank == self.world - 1):
                self.tab[side] = None
                continue
            first = lambda lv, k: list(range(k)) if side == "lo" else list(range(lv.nz - 1, lv.nz - 1 - k, -1))
            ghost = lambda lv, k: [-1 - i for i in range(k)] if side == "lo" else [lv.nz + i for i in range(k)]
            # order within a message: planes counted AWAY from the interface, so that the sender's k-th owned
            # plane lands on the receiver's k-th ghost plane
            depth = [V, V]
            sx = [planes(l, first(self.levels[l], depth[l])) for l in (0, 1)]
            rx = [planes(l, ghost(self.levels[l], depth[l])) for l in (0, 1)]
            deep = range(2, self.nlvl)
            # added at the receiver: [ghost contribution -> its boundary plane, own plane 0 -> its ghost plane 0]
            s_add = [planes(l, ghost(self.levels[l], 1) + first(self.levels[l], 1)) for l in deep]
            r_add = [planes(l, first(self.levels[l], 1) + ghost(self.levels[l], 1)) for l in deep]
            # copied: nothing (planes further from the interface would be, were more than one ghost plane valid)
            s_cp, r_cp = [], []
            cat = lambda parts: torch.as_tensor(np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64), device=dev)
            t = dict(sx=cat(sx), rx=cat(rx), s_add=cat(s_add), r_add=cat(r_add), s_cp=cat(s_cp), r_cp=cat(r_cp))
            t["nx"], t["nadd"], t["ncp"] = int(t["sx"].numel()), int(t["s_add"].numel()), int(t["s_cp"].numel())
            t["send"] = torch.empty(t["nx"] + t["nadd"] + t["ncp"], dtype=self.dtype, device=dev)
            # the same planes as strided descriptors for odil_planes_copy: one launch packs the x part, one the g part,
            # one unpacks, one adds -- no index tables read per element (a plane of an axis-0 cut is one contiguous run)
            def runs(level, rel):
                lv = self.levels[level]
                return [(int(starts[level]) + (lv.g_lo + k) * lv.plane, 1, lv.plane, lv.plane) for k in rel]

            PL = getattr(self.ops, "PlaneList", None)
            if PL is not None and t["ncp"] == 0:
                pl = dict()
                pl["sx"] = PL([d for l in (0, 1) for d in runs(l, first(self.levels[l], depth[l]))], dev)
                pl["rx"] = PL([d for l in (0, 1) for d in runs(l, ghost(self.levels[l], depth[l]))], dev)
                sa = [d for l in deep for d in runs(l, ghost(self.levels[l], 1) + first(self.levels[l], 1))]
                ra = [d for l in deep for d in runs(l, first(self.levels[l], 1) + ghost(self.levels[l], 1))]
                pl["s_add"] = PL(sa, dev, start=t["nx"]) if sa else None
                pl["r_add"] = PL(ra, dev, start=t["nx"]) if ra else None
                t["planes"] = pl
            # the initial state: V (3 on level 0) boundary planes of every level
            full_s = [planes(l, first(self.levels[l], V)) for l in range(self.nlvl)]
            full_r = [planes(l, ghost(self.levels[l], V)) for l in range(self.nlvl)]
            t["full_s"], t["full_r"] = cat(full_s), cat(full_r)
            self.tab[side] = t

    def _sync_state(self):
        """Generator step, once: ghost planes of the initial state."""
        msg = [None if t is None else self.x.index_select(0, t["full_s"]) for t in (self.tab["lo"], self.tab["hi"])]
        recv = yield ("halo", msg[0], msg[1])
        for t, r in zip((self.tab["lo"], self.tab["hi"]), recv):
            if t is not None:
                self.x.index_copy_(0, t["full_r"], r)
        self.synced = True

    def _exchange_fu(self, depth):
        """Generator step: the `depth` boundary planes of fu -> the neighbours' ghost planes."""
        l0 = self.levels[0]
        lo = self.fu[l0.g_lo : l0.g_lo + depth] if self.rank > 0 else None
        hi = self.fu[l0.g_lo + l0.nz - depth : l0.g_lo + l0.nz] if self.rank < self.world - 1 else None
        recv_lo, recv_hi = yield ("halo", lo, hi)
        if recv_lo is not None:
            self.fu[l0.g_lo - depth : l0.g_lo].copy_(recv_lo.view(depth, l0.ny, l0.nx))
        if recv_hi is not None:
            self.fu[l0.g_lo + l0.nz : l0.g_lo + l0.nz + depth].copy_(recv_hi.view(depth, l0.ny, l0.nx))

    def _exchange(self):
        """Generator step: the packed exchange at the end of the epoch (see the module docstring)."""
        msg = []
        for t in (self.tab["lo"], self.tab["hi"]):
            if t is None:
                msg.append(None)
                continue
            buf, nx, na = t["send"], t["nx"], t["nadd"]
            pl = t.get("planes")
            if pl is not None:
                pl["sx"].pack(self.x, buf)
                if pl["s_add"] is not None:
                    pl["s_add"].pack(self.g, buf)
                msg.append(buf)
                continue
            torch.index_select(self.x, 0, t["sx"], out=buf[:nx])
            if na:
                torch.index_select(self.g, 0, t["s_add"], out=buf[nx:nx + na])
            if t["ncp"]:
                torch.index_select(self.g, 0, t["s_cp"], out=buf[nx + na:])
            msg.append(buf)
        recv = yield ("halo", msg[0], msg[1])
        for t, r in zip((self.tab["lo"], self.tab["hi"]), recv):
            if t is None:
                continue
            nx, na = t["nx"], t["nadd"]
            pl = t.get("planes")
            if pl is not None:
                r = r.reshape(-1)
                pl["rx"].unpack(self.x, r)
                if pl["r_add"] is not None:
                    pl["r_add"].unpack_add(self.g, r)
                continue
            self.x.index_copy_(0, t["rx"], r[:nx])
            if t["ncp"]:
                self.g.index_copy_(0, t["r_cp"], r[nx + na:])
            if na:
                self.g.index_add_(0, t["r_add"], r[nx:nx + na])

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

        if not self.synced:
            yield from self._sync_state()
        # u = w_0 + P(w_1 + P(...)): coarse operand with one ghost plane -> fine with two
        # the last prolongation is fused into the residual when the kernel set has it (u never stored)
        fused_last = hasattr(ops, "poisson_residual_synth")
        b = tic("mg_synth")
        coarse = lv[L - 1].inner(self.w[L - 1])
        for l in range(L - 2, 0 if fused_last else -1, -1):
            out = self.u if l == 0 else self.work[l]
            ops.interp_add(coarse.contiguous(), "ccc", add=self.w[l], out=out)
            coarse = lv[l].inner(out)
        toc(b)
        b = tic("residual")
        l0, l1 = lv[0], lv[1]
        if fused_last:
            ops.poisson_residual_synth(coarse.contiguous(), self.w[0], self.rhs, self.h2, fu=self.fu,
                                       loss=self.loss_part, zrange=(l0.g_lo, l0.g_lo + l0.nz),
                                       denom=self.global_cells)
        else:
            ops.poisson_residual(self.u, self.rhs, self.h2, fu=self.fu, loss=self.loss_part,
                                 zrange=(l0.g_lo, l0.g_lo + l0.nz), denom=self.global_cells)
        toc(b)
        n0 = l0.size
        self.t += 1
        t = self.npdt(self.t)
        alpha = self.lr * np.sqrt(1 - self.b2**t) / (1 - self.b1**t)
        omb1, omb2 = 1 - self.b1, 1 - self.b2
        cut = (self.rank > 0, self.rank < self.world - 1)
        g1 = l1.inner(self.gw[1])
        adam1 = (l1.inner(self.w[1]), l1.inner(self.mw[1]), l1.inner(self.vw[1]))
        m0, v0 = self.m[:n0].view(l0.shape), self.v[:n0].view(l0.shape)
        # stencil adjoint + first transposed prolongation + Adam of levels 0 and 1 in one launch, as on one GPU
        # (fused.py): the interfaces are `cut` ends (interior stencil rows there; the owned coarse planes only
        # read fine planes whose g0 is complete)
        fuse_t = (self.fuse_transpose and hasattr(ops, "poisson_adjoint_transpose")
                  and ops.adjoint_transpose_supported(tuple(l0.shape))
                  and tuple(g1.shape) == tuple(n // 2 for n in l0.shape))
        # the adjoint reads fu at z +- 1; the one-launch form makes g0 of the inner ghost plane itself and needs both
        b = tic("halo")
        yield from self._exchange_fu(2 if fuse_t else 1)
        toc(b)
        if fuse_t:
            b = tic("adjoint_transpose")
            ops.poisson_adjoint_transpose(self.fu, self.h2, self.scale, g1, g0=None, adam0=(self.w[0], m0, v0),
                                          adam1=adam1, alpha=alpha, one_minus_b1=omb1, one_minus_b2=omb2,
                                          eps=self.eps, cut=cut)
            toc(b)
        else:
            b = tic("adjoint")
            if hasattr(ops, "poisson_adjoint_adam"):
                ops.poisson_adjoint_adam(self.fu, self.h2, self.scale, self.gw[0], self.w[0], m0, v0, alpha, omb1, omb2,
                                         self.eps)
            else:
                ops.poisson_adjoint(self.fu, self.h2, self.scale, out=self.gw[0])
                ops.adam_step(self.x[:n0], self.m[:n0], self.v[:n0], self.g[:n0], alpha, omb1, omb2, self.eps)
            toc(b)
            # (fallback for shapes the one-launch kernel does not take: g0 is stored, its ghost plane comes
            # from the neighbour -- one more exchange)
            b = tic("halo")
            recv_lo, recv_hi = yield ("halo", self.gw[0][l0.g_lo] if cut[0] else None,
                                      self.gw[0][l0.g_lo + l0.nz - 1] if cut[1] else None)
            if recv_lo is not None:
                self.gw[0][l0.g_lo - 1].copy_(recv_lo)
            if recv_hi is not None:
                self.gw[0][l0.g_lo + l0.nz].copy_(recv_hi)
            toc(b)
            b = tic("mg_synth_adj")
            if hasattr(ops, "interp_adj_adam"):
                ops.interp_adj_adam(self.gw[0], "ccc", tuple(g1.shape), g1, *adam1, alpha, omb1, omb2, self.eps, cut=cut)
            else:
                ops.interp_adj(self.gw[0], "ccc", tuple(g1.shape), out=g1, cut=cut)
                n1 = lv[1].size
                ops.adam_step(self.x[n0:n0 + n1], self.m[n0:n0 + n1], self.v[n0:n0 + n1], self.g[n0:n0 + n1], alpha,
                              omb1, omb2, self.eps)
            toc(b)
        b = tic("mg_synth_adj")
        # from here down every rank carries only its own contributions: g1 is complete on the owned planes,
        # so its ghost planes (written above with partial sums nobody needs) count as zero
        if cut[0]:
            self.gw[1][: l1.g_lo].zero_()
        if cut[1]:
            self.gw[1][l1.g_lo + l1.nz :].zero_()
        for l in range(2, L):
            cview = lv[l].inner(self.gw[l])
            ops.interp_adj(self.gw[l - 1], "ccc", tuple(cview.shape), out=cview, cut=cut)
        toc(b)
        b = tic("halo")
        yield from self._exchange()
        toc(b)
        if L > 2:
            b = tic("adam")
            k = self.n01
            ops.adam_step(self.x[k:], self.m[k:], self.v[k:], self.g[k:], alpha, omb1, omb2, self.eps)
            toc(b)

    # ---- loss and gradient alone (the quasi-Newton drivers of slab_solvers.py) ----------------------------
    def loss_grad_gen(self):
        """Generator (yields at exchanges like `epoch_gen`): from the owned planes of self.w to this rank's share of the
        loss (self.loss_part) and the complete gradient on the owned planes of every level (self.gw).  One exchange per
        level instead of the Adam epoch's packed one: the boundary planes of g_{l-1} travel to the neighbours' ghost
        planes, then P^T with `cut` ends completes g_l on the owned planes."""
        ops, lv, L = self.ops, self.levels, self.nlvl
        yield from self._sync_state()  # the unknowns changed: their ghost planes follow (one plane per level and side)
        coarse = lv[L - 1].inner(self.w[L - 1])
        for l in range(L - 2, -1, -1):
            out = self.u if l == 0 else self.work[l]
            ops.interp_add(coarse.contiguous(), "ccc", add=self.w[l], out=out)
            coarse = lv[l].inner(out)
        l0 = lv[0]
        ops.poisson_residual(self.u, self.rhs, self.h2, fu=self.fu, loss=self.loss_part,
                             zrange=(l0.g_lo, l0.g_lo + l0.nz), denom=self.global_cells)
        yield from self._exchange_fu(1)
        ops.poisson_adjoint(self.fu, self.h2, self.scale, out=self.gw[0])
        cut = (self.rank > 0, self.rank < self.world - 1)
        for l in range(1, L):
            f, g = lv[l - 1], self.gw[l - 1]
            recv_lo, recv_hi = yield ("halo", g[f.g_lo] if cut[0] else None, g[f.g_lo + f.nz - 1] if cut[1] else None)
            if recv_lo is not None:
                g[f.g_lo - 1].copy_(recv_lo.view(f.ny, f.nx))
            if recv_hi is not None:
                g[f.g_lo + f.nz].copy_(recv_hi.view(f.ny, f.nx))
            cview = lv[l].inner(self.gw[l])
            ops.interp_adj(g, "ccc", tuple(cview.shape), out=cview, cut=cut)

    def pack_owned(self, arrays, out=None):
        """The owned planes of the level arrays `arrays` (self.w / self.gw) as ONE flat float64 vector, finest level
        first -- the layout of the undivided unknown vector restricted to this rank (reference core.py:436-469)."""
        if out is None:
            out = torch.empty(self.n_unknowns_local, dtype=torch.float64, device=self.device)
        off = 0
        for lv, a in zip(self.levels, arrays):
            n = lv.nz * lv.plane
            out[off:off + n].view(lv.nz, lv.ny, lv.nx).copy_(lv.owned(a))
            off += n
        return out

    def unpack_owned(self, flat, arrays=None):
        off = 0
        for lv, a in zip(self.levels, arrays or self.w):
            n = lv.nz * lv.plane
            lv.owned(a).copy_(flat[off:off + n].view(lv.nz, lv.ny, lv.nx))
            off += n

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
      "sum"   all-reduce of send_lo (a few scalars); returns the tensor;
      "gather" all-gather of send_lo: returns the (world, ...) stack of every rank's tensor, rank order (reductions formed
              from it are the same bits on every rank -- the quasi-Newton drivers branch on them)."""

    def __init__(self, rank, world, self_loop=False):
        import torch.distributed as dist

        self.dist, self.rank, self.world = dist, rank, world
        # self_loop (world 1 only; tests/test_rccl_selfloop_gpu.py): the rank is its own lower AND upper neighbour -- a ring
        # of one -- so that the messages really travel through the backend's send / receive on a box with a single GPU
        self.self_loop = bool(self_loop) and world == 1
        # gloo moves host memory: device planes are staged (tests only; RCCL sends device memory)
        self.stage = dist.get_backend() == "gloo"
        self._recv = dict()  # receive buffers, kept: (side, numel, dtype, device) -> tensor

    def _recv_like(self, side, t):
        """The receive buffer for a message of t's size from `side` (allocated once; every consumer copies the
        planes out on the stream the next receive is ordered behind)."""
        key = (side, t.numel(), t.dtype, t.device)
        buf = self._recv.get(key)
        if buf is None:
            buf = self._recv[key] = torch.empty(t.numel(), dtype=t.dtype, device=t.device)
        return buf.view(t.shape)

    def exchange(self, kind, send_lo, send_hi):
        dist = self.dist
        if kind == "wait":  # completes a posted halo exchange: (recv_lo, recv_hi)
            reqs, recv_lo, recv_hi, dev, keep = send_lo
            for req in reqs:
                req.wait()
            if dev is not None:
                recv_lo = recv_lo.to(dev) if recv_lo is not None else None
                recv_hi = recv_hi.to(dev) if recv_hi is not None else None
            return recv_lo, recv_hi
        if kind == "sum":
            t = send_lo
            if self.stage and t.is_cuda:
                h = t.cpu()
                dist.all_reduce(h)
                return h.to(t.device)
            dist.all_reduce(t)
            return t
        if kind == "gather":
            t = send_lo.contiguous()
            src = t.cpu() if (self.stage and t.is_cuda) else t
            parts = [torch.empty_like(src) for _ in range(self.world)]
            dist.all_gather(parts, src)
            return torch.stack(parts).to(t.device)
        if kind == "wrap":
            if self.world == 1 and not self.self_loop:
                return send_hi, send_lo
            peer_lo = self.world - 1 if self.rank == 0 else None
            peer_hi = 0 if self.rank == self.world - 1 else None
        else:  # "halo" / "post" 
            peer_lo = self.rank - 1 if self.rank > 0 else None
            peer_hi = self.rank + 1 if self.rank < self.world - 1 else None
        if self.self_loop:
            peer_lo = peer_hi = 0
        dev = None
        if self.stage:
            for t in (send_lo, send_hi):
                if t is not None and t.is_cuda:
                    dev = t.device
            if dev is not None:
                send_lo = send_lo.cpu() if send_lo is not None else None
                send_hi = send_hi.cpu() if send_hi is not None else None
        ops, recv_lo, recv_hi = [], None, None
        tag = "post-" if kind == "post" else ""  # (a posted message keeps its receive buffer until it is waited for)
        if self.self_loop and send_lo is not None and send_hi is not None:
            # a ring of one: what goes down arrives from above and vice versa (messages between one pair of ranks match
            # in the order they were issued)
            send_lo, send_hi = send_lo.contiguous(), send_hi.contiguous()
            recv_lo, recv_hi = self._recv_like(tag + "lo", send_hi), self._recv_like(tag + "hi", send_lo)
            ops += [dist.P2POp(dist.isend, send_lo, 0), dist.P2POp(dist.irecv, recv_hi, 0),
                    dist.P2POp(dist.isend, send_hi, 0), dist.P2POp(dist.irecv, recv_lo, 0)]
            send_lo_done = send_hi_done = True
        else:
            send_lo_done = send_hi_done = False
        if not send_lo_done and send_lo is not None and peer_lo is not None:
            send_lo = send_lo.contiguous()
            recv_lo = self._recv_like(tag + "lo", send_lo)
            ops += [dist.P2POp(dist.isend, send_lo, peer_lo), dist.P2POp(dist.irecv, recv_lo, peer_lo)]
        if not send_hi_done and send_hi is not None and peer_hi is not None:
            send_hi = send_hi.contiguous()
            recv_hi = self._recv_like(tag + "hi", send_hi)
            ops += [dist.P2POp(dist.isend, send_hi, peer_hi), dist.P2POp(dist.irecv, recv_hi, peer_hi)]
        if kind == "post":
            # "post": the halo exchange is STARTED (RCCL orders it behind what the stream holds now and runs it on its
            # own stream); kernels launched from here on overlap it; "wait" orders the stream behind its completion
            reqs = dist.batch_isend_irecv(ops) if ops else []
            return (reqs, recv_lo, recv_hi, dev, (send_lo, send_hi))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if dev is not None:
            recv_lo = recv_lo.to(dev) if recv_lo is not None else None
            recv_hi = recv_hi.to(dev) if recv_hi is not None else None
        return recv_lo, recv_hi


class LocalComm:
    """The exchanges of a slab epoch when there is ONE rank: the periodic closure is the rank itself."""

    def exchange(self, kind, a, b):
        if kind == "sum":
            return a
        if kind == "gather":
            return a[None]
        if kind == "wait":
            return a
        return (b, a) if kind == "wrap" else (None, None)


def init_distributed():
    """(rank, world, comm) from the environment a launcher sets (torch.distributed.run: RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR / MASTER_PORT): one process per GPU, backend nccl (= RCCL on ROCm) unless
    ODIL_DIST_BACKEND says otherwise; a single process gets the local closure.  A rank that cannot initialise its
    backend raises (no silent fallback)."""
    import os

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    if world == 1:
        return 0, 1, LocalComm()
    import torch.distributed as dist

    local = int(os.environ.get("LOCAL_RANK", rank)) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if not dist.is_initialized():
        backend = os.environ.get("ODIL_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, TorchDistComm(rank, world)


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
            if kind == "wait":  # the planes of a posted exchange were moved when it was posted
                replies = [m[1] for m in msgs]
            elif kind == "sum":
                total = msgs[0][1].clone()
                for m in msgs[1:]:
                    total = total + m[1]
                replies = [total.clone() for _ in range(P)]
            elif kind == "gather":
                stack = torch.stack([m[1] for m in msgs])
                replies = [stack.clone() for _ in range(P)]
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
