"""Optimizers of the hot loop (reference src/odil/optimizer.py:1-357), device-resident.

  AdamNativeOptimizer  optimizer.py:280-341  -> odil_adam_step (one launch over the packed state)
  GdOptimizer          optimizer.py:256-277  -> odil_axpy
  LbfgsbOptimizer      optimizer.py:29-117   -> the unbounded case of L-BFGS-B 3.0, which the
      reference reaches through scipy.optimize.fmin_l_bfgs_b (scipy 1.16.2 pinned, uv.lock:1499),
      restated here from the published algorithm (Byrd, Lu, Nocedal & Zhu 1995; Morales &
      Nocedal 2011; line search dcsrch/dcstep of More' & Thuente 1994): compact
      representation B = theta I - W M W^T, step d = -B^{-1} g, identical update / skip /
      restart rules and the same dcsrch constants (ftol 1e-3, gtol 0.9, xtol 0.1).  The n-vectors
      (x, g, d and the 2m history vectors) never leave the device: the reference round-trips
      the full vector through host float64 on every evaluation (optimizer.py:63-88); here only
      O(m) dot products and the scalar line-search state are on the host.
`adam_tf` and `lbfgs` (TensorFlow-Probability) are TensorFlow-specific duplicates and not provided.
"""

import math
import os
from argparse import Namespace

import numpy as np
import torch

from . import ops


class Optimizer:
    """Base of the optimizers: a name pair, the dtype and an evaluation counter (interface of reference
    optimizer.py:8-19); `run` of the base makes no step."""

    def __init__(self, name=None, displayname=None, dtype=None):
        self.name, self.displayname = name, displayname or name
        self.dtype, self.pinfo, self.evals = dtype, None, 0

    def run(self, x0, loss_grad, epochs, callback=None, epoch_start=0, **kwargs):
        return x0, Namespace(evals=0, epochs=0)


class EarlyStopError(Exception):
    def __init__(self, msg, optinfo):
        super().__init__(msg)
        self.optinfo = optinfo


# --------------------------------------------------------------------------------------
# Packed state: all unknown arrays in one flat device buffer, arrays are views
# (the layout of Domain.pack_state, reference core.py:436-443).
# --------------------------------------------------------------------------------------
def pack_like(arrays, dtype=None):
    sizes = [int(a.numel()) for a in arrays]
    dtype = dtype or arrays[0].dtype
    flat = torch.empty(sum(sizes), dtype=dtype, device=arrays[0].device)
    views = [t.view(a.shape) for t, a in zip(flat.split(sizes), arrays)]
    return flat, views


def flat_base(arrays):
    """If `arrays` are adjacent views of one contiguous buffer (in order), returns that flat view."""
    if not arrays:
        return None
    a0 = arrays[0]
    ptr = a0.data_ptr()
    esize = a0.element_size()
    storage = a0.untyped_storage()
    total = 0
    for a in arrays:
        if not a.is_contiguous() or a.dtype != a0.dtype or a.untyped_storage().data_ptr() != storage.data_ptr():
            return None
        if a.data_ptr() != ptr + total * esize:
            return None
        total += a.numel()
    offset = (ptr - storage.data_ptr()) // esize
    return torch.empty(0, dtype=a0.dtype, device=a0.device).set_(storage, offset, (total,), (1,))


def copy_into(flat, views, arrays):
    base = flat_base(arrays)
    if base is not None and base.dtype == flat.dtype:
        if base.data_ptr() != flat.data_ptr():
            flat.copy_(base)
        return
    for v, a in zip(views, arrays):
        v.copy_(a)


_graph_runtime_ready = False  # the first capture of a process pays ~0.15 s of one-time initialisation


def _graph_wanted(numel, loss_grad, epochs):
    """ODIL_GRAPH=1 / 0 forces graph replay of Adam epochs on / off; the default replays problems of at
    most 4 M unknowns whose evaluator is one of the kernel paths (fused Poisson, traced operator), when
    the run is long enough to amortise the first capture (1-D Poisson N=256: 0.119 -> 0.058 ms / epoch,
    heat 256x512: 0.160 -> 0.106)."""
    mode = os.environ.get("ODIL_GRAPH", "auto")
    if mode in ("0", "1"):
        return mode == "1" and torch.cuda.is_available()
    safe = getattr(loss_grad, "graph_safe", None)
    long_enough = epochs >= 500 or (_graph_runtime_ready and epochs >= 20)
    return torch.cuda.is_available() and numel <= (1 << 22) and long_enough and safe is not None and safe()


class _EpochGraph:
    """One optimizer epoch captured into a hipGraph.  The per-epoch step sizes are computed on the host
    exactly as in the eager loop and kept in a device table; a captured index_select moves the current
    one into the scalar the kernels read, a captured increment advances the index."""

    def __init__(self, step, step_sizes, dtype, device, loss_grad):
        self.step = step
        self.refresh = getattr(loss_grad, "refresh", None)  # called before every replay
        self.begin, self.end = getattr(loss_grad, "graph_begin", None), getattr(loss_grad, "graph_end", None)
        self.nrows = len(step_sizes)
        self.table = torch.tensor(np.array(step_sizes, dtype=np.float64), dtype=dtype, device=device)
        self.index = torch.zeros(1, dtype=torch.int64, device=device)
        self.alpha = torch.zeros(1, dtype=dtype, device=device)
        self.graph, self.pinfo = None, None

    def _body(self):
        torch.index_select(self.table, 0, self.index, out=self.alpha)
        self.index.add_(1)
        return self.step(self.alpha)

    def capture(self):
        from .util import printlog

        try:
            graph = torch.cuda.CUDAGraph()
            if self.begin is not None:
                self.begin(self.nrows)
            with torch.cuda.graph(graph):
                pinfo = self._body()
            # the report of an epoch is a dict of DEVICE scalars that a lazy wrapper converts (and
            # caches) on first read: keep the raw tensors, hand out a fresh wrapper per replay
            self.pinfo_type = type(pinfo) if isinstance(pinfo, dict) else None
            self.pinfo = {k: dict.__getitem__(pinfo, k) for k in dict.keys(pinfo)} if self.pinfo_type else pinfo
        except Exception as e:  # capture is an optimisation: never a reason to fail
            printlog("odil_amd: hipGraph capture of the Adam epoch failed ({}: {}); running eagerly".format(
                type(e).__name__, str(e).splitlines()[0] if str(e) else ""))
            torch.cuda.synchronize()
            self.index.zero_()
            self.close()
            return False
        self.graph = graph
        global _graph_runtime_ready
        _graph_runtime_ready = True
        return True

    def close(self):
        if self.end is not None:
            self.end()

    def replay(self):
        if self.refresh is not None:
            self.refresh()
        self.graph.replay()
        return self.pinfo_type(**self.pinfo) if self.pinfo_type else self.pinfo


class AdamNativeOptimizer(Optimizer):
    def __init__(self, dtype=None, mod=None, **kwargs):
        super().__init__(name="adamn", displayname="AdamNative", dtype=dtype)
        self.mod = mod

    def run(self, x0, loss_grad, epochs=None, callback=None, lr=1e-3, epoch_start=0, beta_1=0.9, beta_2=0.999,
            epsilon=1e-7, jit=True, moments=None, steps_done=0, **kwargs):
        """Keras-convention Adam (epsilon outside the sqrt), reference optimizer.py:286-341.

        Beyond the reference: `moments=(m, v)` (lists shaped like x0) and `steps_done` resume a run where another
        left it -- the reference restarts m = v = 0 and the bias correction with every call (optimizer.py:327-334),
        which is what the defaults do.  The final moments are returned in `optinfo.m / optinfo.v`."""
        tdtype = x0[0].dtype
        npdt = np.float64 if tdtype == torch.float64 else np.float32
        lr, beta_1, beta_2 = npdt(lr), npdt(beta_1), npdt(beta_2)
        xf, x = pack_like(x0)
        copy_into(xf, x, x0)
        mf = torch.zeros_like(xf)
        vf = torch.zeros_like(xf)
        mviews = [t.view(a.shape) for t, a in zip(mf.split([a.numel() for a in x]), x)]
        vviews = [t.view(a.shape) for t, a in zip(vf.split([a.numel() for a in x]), x)]
        if moments is not None:
            copy_into(mf, mviews, moments[0])
            copy_into(vf, vviews, moments[1])
        scratch = dict(gf=None, gviews=None)

        def step(alpha):
            """One epoch: loss + gradient, update.  `alpha`: host number or one-element device tensor."""
            # A recognised problem may apply the update of its leading arrays inside its own
            # gradient launch (core.Problem / fused.py): it then tells how many were done.
            fused = getattr(loss_grad, "fused_adam", None)
            done = 0
            if fused is not None:
                res = fused(x, mviews, vviews, alpha, 1 - beta_1, 1 - beta_2, epsilon)
                if res is not None:
                    loss, grads, pinfo, done = res
            if not done:
                loss, grads, pinfo = loss_grad(x)
            g = flat_base(grads)
            if g is None or g.dtype != tdtype or g.numel() != xf.numel():
                if scratch["gf"] is None:
                    scratch["gf"], scratch["gviews"] = pack_like(x)
                copy_into(scratch["gf"], scratch["gviews"], grads)
                g = scratch["gf"]
            n0 = sum(int(a.numel()) for a in x[:done])
            if n0 < xf.numel():
                ops.adam_step(xf[n0:], mf[n0:], vf[n0:], g[n0:], alpha, 1 - beta_1, 1 - beta_2, epsilon)
            return pinfo

        def step_size(epoch):
            t = npdt(epoch - epoch_start + steps_done)
            return lr * np.sqrt(1 - beta_2**t) / (1 - beta_1**t)  # optimizer.py:313-315

        first, last = epoch_start + 1, epoch_start + epochs
        epoch = first
        # Problems small enough for ONE workgroup (1-D / 2-D Poisson: fused.PoissonEvaluator.small_epochs) run whole epochs
        # in one launch -- as many as lie before the callback's next active epoch (a callback that cannot tell is called
        # after every epoch, as the reference does, optimizer.py:331-336).
        small = getattr(loss_grad, "small_epochs", None)
        runner = None
        if small is not None and epochs > 0:
            table = torch.tensor(np.array([step_size(e) for e in range(first, last + 1)], dtype=np.float64), dtype=tdtype,
                                 device=xf.device)
            runner = small(x, mviews, vviews, table, 1 - beta_1, 1 - beta_2, epsilon)
        while runner is not None and epoch <= last:
            nxt = getattr(callback, "next_active", None) if callback is not None else (lambda e: last)
            stop = min(last, max(epoch, nxt(epoch - 1)) if nxt is not None else epoch, epoch + 4095)
            pinfo = runner(epoch - first, stop - epoch + 1)
            self.evals += stop - epoch + 1
            epoch = stop + 1
            if callback is not None and stop > 0:
                captured = [a.data_ptr() for a in x]
                callback(x, stop, pinfo)
                if [a.data_ptr() for a in x] != captured:
                    runner = None  # (the callback swapped arrays: the packed vectors are no longer the state)
        # Launch-bound problems (a few million unknowns or fewer: dozens of launches of a few
        # microseconds each) replay the epoch as ONE hipGraph: the first two epochs run eagerly (lazy
        # initialisation, allocator warm-up), the third is captured with the step size read from
        # device memory (`alpha_dev` of the *_adam kernels), the rest are replays.
        graph = None
        if last - epoch + 1 > 4 and _graph_wanted(xf.numel(), loss_grad, epochs):  # (epochs that are LEFT: none after whole-epoch launches)
            while epoch < first + 2:
                self.evals += 1
                pinfo = step(step_size(epoch))
                if callback is not None:
                    callback(x, epoch, pinfo)
                epoch += 1
            graph = _EpochGraph(step, [step_size(e) for e in range(epoch, last + 1)], tdtype, xf.device, loss_grad)
            if not graph.capture():
                graph = None
        captured = [a.data_ptr() for a in x]
        while epoch <= last:
            self.evals += 1
            pinfo = graph.replay() if graph is not None else step(step_size(epoch))
            if epoch > 0 and callback is not None:
                callback(x, epoch, pinfo)
                if graph is not None and [a.data_ptr() for a in x] != captured:
                    graph.close()
                    graph = None  # the callback swapped arrays (callback_update_state): replays would miss them
            epoch += 1
        if graph is not None:
            graph.close()
        optinfo = Namespace()
        optinfo.epochs = epochs
        optinfo.evals = self.evals
        optinfo.m, optinfo.v = mviews, vviews
        return x, optinfo


class GdOptimizer(Optimizer):
    def __init__(self, dtype=None, mod=None, **kwargs):
        super().__init__(name="gd", displayname="GD", dtype=dtype)
        self.mod = mod

    def run(self, x0, loss_grad, epochs=None, callback=None, lr=1e-3, epoch_start=0, **kwargs):
        xf, x = pack_like(x0)
        copy_into(xf, x, x0)
        for epoch in range(epoch_start + 1, epoch_start + epochs + 1):
            self.evals += 1
            loss, grads, pinfo = loss_grad(x)
            for xi, gi in zip(x, grads):
                ops.axpy(xi, gi.contiguous(), -float(lr))  # x -= lr * g (optimizer.py:270)
            if epoch > 0 and callback is not None:
                callback(x, epoch, pinfo)
        optinfo = Namespace()
        optinfo.epochs = epochs
        optinfo.evals = self.evals
        return x, optinfo


# --------------------------------------------------------------------------------------
# More'-Thuente line search (MINPACK-2 dcsrch / dcstep), scalar state on the host.
# --------------------------------------------------------------------------------------
def _dcstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stpmin, stpmax):
    sgnd = dp * (dx / abs(dx))
    if fp > fx:  # case 1: higher function value -> minimum bracketed
        theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp
        s = max(abs(theta), abs(dx), abs(dp))
        gamma = s * math.sqrt((theta / s) ** 2 - (dx / s) * (dp / s))
        if stp < stx:
            gamma = -gamma
        p = (gamma - dx) + theta
        q = ((gamma - dx) + gamma) + dp
        r = p / q
        stpc = stx + r * (stp - stx)
        stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx)
        if abs(stpc - stx) < abs(stpq - stx):
            stpf = stpc
        else:
            stpf = stpc + (stpq - stpc) / 2.0
        brackt = True
    elif sgnd < 0.0:  # case 2: derivatives of opposite sign
        theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp
        s = max(abs(theta), abs(dx), abs(dp))
        gamma = s * math.sqrt((theta / s) ** 2 - (dx / s) * (dp / s))
        if stp > stx:
            gamma = -gamma
        p = (gamma - dp) + theta
        q = ((gamma - dp) + gamma) + dx
        r = p / q
        stpc = stp + r * (stx - stp)
        stpq = stp + (dp / (dp - dx)) * (stx - stp)
        stpf = stpc if abs(stpc - stp) > abs(stpq - stp) else stpq
        brackt = True
    elif abs(dp) < abs(dx):  # case 3: derivative magnitude decreases
        theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp
        s = max(abs(theta), abs(dx), abs(dp))
        gamma = s * math.sqrt(max(0.0, (theta / s) ** 2 - (dx / s) * (dp / s)))
        if stp > stx:
            gamma = -gamma
        p = (gamma - dp) + theta
        q = (gamma + (dx - dp)) + gamma
        r = p / q
        if r < 0.0 and gamma != 0.0:
            stpc = stp + r * (stx - stp)
        elif stp > stx:
            stpc = stpmax
        else:
            stpc = stpmin
        stpq = stp + (dp / (dp - dx)) * (stx - stp)
        if brackt:
            stpf = stpc if abs(stpc - stp) < abs(stpq - stp) else stpq
            if stp > stx:
                stpf = min(stp + 0.66 * (sty - stp), stpf)
            else:
                stpf = max(stp + 0.66 * (sty - stp), stpf)
        else:
            stpf = stpc if abs(stpc - stp) > abs(stpq - stp) else stpq
            stpf = min(stpmax, stpf)
            stpf = max(stpmin, stpf)
    else:  # case 4
        if brackt:
            theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp
            s = max(abs(theta), abs(dy), abs(dp))
            gamma = s * math.sqrt((theta / s) ** 2 - (dy / s) * (dp / s))
            if stp > sty:
                gamma = -gamma
            p = (gamma - dp) + theta
            q = ((gamma - dp) + gamma) + dy
            r = p / q
            stpf = stp + r * (sty - stp)
        elif stp > stx:
            stpf = stpmax
        else:
            stpf = stpmin
    if fp > fx:
        sty, fy, dy = stp, fp, dp
    else:
        if sgnd < 0.0:
            sty, fy, dy = stx, fx, dx
        stx, fx, dx = stp, fp, dp
    return stx, fx, dx, sty, fy, dy, stpf, brackt


class _Dcsrch:
    """One line search: call start() then step(f, g) until task is not 'FG'."""

    XTRAPL, XTRAPU = 1.1, 4.0

    def __init__(self, ftol=1e-3, gtol=0.9, xtol=0.1, stpmin=0.0, stpmax=1e10):
        self.ftol, self.gtol, self.xtol, self.stpmin, self.stpmax = ftol, gtol, xtol, stpmin, stpmax

    def start(self, stp, f, g):
        if stp < self.stpmin:
            return stp, "ERROR: STP .LT. STPMIN"
        if stp > self.stpmax:
            return stp, "ERROR: STP .GT. STPMAX"
        if g >= 0.0:
            return stp, "ERROR: INITIAL G .GE. ZERO"
        self.brackt = False
        self.stage = 1
        self.finit, self.ginit = f, g
        self.gtest = self.ftol * g
        self.width = self.stpmax - self.stpmin
        self.width1 = self.width / 0.5
        self.stx, self.fx, self.gx = 0.0, f, g
        self.sty, self.fy, self.gy = 0.0, f, g
        self.stmin = 0.0
        self.stmax = stp + self.XTRAPU * stp
        return stp, "FG"

    def step(self, stp, f, g):
        ftest = self.finit + stp * self.gtest
        if self.stage == 1 and f <= ftest and g >= 0.0:
            self.stage = 2
        task = "FG"
        if self.brackt and (stp <= self.stmin or stp >= self.stmax):
            task = "WARNING: ROUNDING ERRORS PREVENT PROGRESS"
        if self.brackt and self.stmax - self.stmin <= self.xtol * self.stmax:
            task = "WARNING: XTOL TEST SATISFIED"
        if stp == self.stpmax and f <= ftest and g <= self.gtest:
            task = "WARNING: STP = STPMAX"
        if stp == self.stpmin and (f > ftest or g >= self.gtest):
            task = "WARNING: STP = STPMIN"
        if f <= ftest and abs(g) <= self.gtol * (-self.ginit):
            task = "CONVERGENCE"
        if task != "FG":
            return stp, task
        if self.stage == 1 and f <= self.fx and f > ftest:
            fm = f - stp * self.gtest
            fxm = self.fx - self.stx * self.gtest
            fym = self.fy - self.sty * self.gtest
            gm = g - self.gtest
            gxm = self.gx - self.gtest
            gym = self.gy - self.gtest
            self.stx, fxm, gxm, self.sty, fym, gym, stp, self.brackt = _dcstep(
                self.stx, fxm, gxm, self.sty, fym, gym, stp, fm, gm, self.brackt, self.stmin, self.stmax
            )
            self.fx = fxm + self.stx * self.gtest
            self.fy = fym + self.sty * self.gtest
            self.gx = gxm + self.gtest
            self.gy = gym + self.gtest
        else:
            self.stx, self.fx, self.gx, self.sty, self.fy, self.gy, stp, self.brackt = _dcstep(
                self.stx, self.fx, self.gx, self.sty, self.fy, self.gy, stp, f, g, self.brackt, self.stmin, self.stmax
            )
        if self.brackt:
            if abs(self.sty - self.stx) >= 0.66 * self.width1:
                stp = self.stx + 0.5 * (self.sty - self.stx)
            self.width1 = self.width
            self.width = abs(self.sty - self.stx)
        if self.brackt:
            self.stmin = min(self.stx, self.sty)
            self.stmax = max(self.stx, self.sty)
        else:
            self.stmin = stp + self.XTRAPL * (stp - self.stx)
            self.stmax = stp + self.XTRAPU * (stp - self.stx)
        stp = max(stp, self.stpmin)
        stp = min(stp, self.stpmax)
        if (self.brackt and (stp <= self.stmin or stp >= self.stmax)) or (
            self.brackt and self.stmax - self.stmin <= self.xtol * self.stmax
        ):
            stp = self.stx
        return stp, "FG"


class LbfgsVectors:
    """The n-vector algebra L-BFGS needs, on HIP kernels.  (Tests substitute a NumPy double
    to exercise the scalar logic on CPU.)

    The history lives in ONE matrix with s_k and y_k interleaved (row 2k = s_k, row 2k + 1 = y_k),
    so that the products of the whole history with the new vectors are one launch and one host
    read, and the direction is one linear combination.  Everything the host needs after an
    evaluation (loss, <g, d>, max |g|, and the <d, d>, <g_old, d> launched before it) sits in one
    small device buffer that is read with a single copy: two host reads per iteration."""

    def __init__(self, n, m, device):
        self.n, self.m, self.device = n, m, device
        self.w = torch.zeros((2 * m, n), dtype=torch.float64, device=device)
        self.ws = self.w[0::2]
        self.wy = self.w[1::2]
        self.scal = torch.zeros(8, dtype=torch.float64, device=device)  # dtd, gd_old, -, gd, gg, gmax, f
        self.scal_host = torch.zeros(8, dtype=torch.float64).pin_memory()
        self.coef_host = torch.zeros(2 * m, dtype=torch.float64).pin_memory()
        self.coef = torch.zeros(2 * m, dtype=torch.float64, device=device)

    def new(self):
        return torch.zeros(self.n, dtype=torch.float64, device=self.device)

    def copy(self, dst, src):
        dst.copy_(src)

    def set_axpy(self, out, t, d, a):
        """out = t + a * d."""
        out.copy_(t)
        ops.axpy(out, d, a)

    def scale_into(self, dst, src, a):
        ops.scale(src, a, out=dst)

    def sub_into(self, dst, a, b):
        """dst = a - b (dst may alias b)."""
        if dst.data_ptr() == b.data_ptr():
            ops.scale(dst, -1.0, out=dst)
            ops.axpy(dst, a, 1.0)
        else:
            dst.copy_(a)
            ops.axpy(dst, b, -1.0)

    # ---- scalars: launched where their operands are ready, read together ------------------------
    def probe_direction(self, d, g):
        """Launches <d, d> and <g, d> for the direction just formed."""
        ops.dots3(d[None], [d, g], out=self.scal[0:3].view(3, 1))

    def probe_eval(self, f, g, d):
        """Launches <g, d>, <g, g>, max |g| for a new evaluation and places its loss beside them."""
        ops.lbfgs_probe(g, d, self.scal[3:6])
        if torch.is_tensor(f):
            self.scal[6:7].copy_(f.reshape(1))
        else:
            self.scal[6:7].fill_(float(f))

    # ---- hooks of a decomposed domain (slab_solvers.SlabLbfgsVectors): a rank's vectors hold its share of the unknowns,
    # its reductions are partial until they have been combined over the ranks.  One rank: nothing to do.
    def reduce_probes(self, scal):
        return scal

    def reduce_sums(self, t):
        return t

    def read_probes(self):
        """-> (dtd, gd_direction, gd, gmax, f) with one device-to-host copy."""
        self.scal_host.copy_(self.reduce_probes(self.scal), non_blocking=True)
        torch.cuda.current_stream().synchronize()
        h = self.scal_host
        return float(h[0]), float(h[1]), float(h[3]), float(h[5]), float(h[6])

    # ---- history ------------------------------------------------------------------------------------
    def store_pair(self, slot, s, y):
        self.w[2 * slot].copy_(s)
        self.w[2 * slot + 1].copy_(y)

    def history_products(self, nphys, bs):
        """[<s_k, b>], [<y_k, b>] for every b in bs (<= 3 vectors) and every stored pair k < nphys:
        one pass over the history, one host read.  -> (S-products, Y-products), each (len(bs), nphys)."""
        if nphys == 0:
            z = np.zeros((len(bs), 0))
            return z, z
        out = self.reduce_sums(ops.dots3(self.w[: 2 * nphys], bs)).cpu().numpy()[: len(bs)]
        return out[:, 0::2], out[:, 1::2]

    def dot(self, a, b):
        """<a, b> on the host (warm start of the memory only: not on the per-iteration path)."""
        return float(self.reduce_sums(ops.dots3(a[None], [b])).cpu().numpy()[0, 0])

    def history_lincomb(self, y, nphys, cs, cy):
        """y += sum_k cs[k] s_k + cy[k] y_k in one pass over the history."""
        if nphys == 0:
            return
        c = self.coef_host.numpy()
        c[0 : 2 * nphys : 2] = cs
        c[1 : 2 * nphys : 2] = cy
        # the pinned staging buffer is rewritten only after the next read_probes(), which waits for this copy
        self.coef[: 2 * nphys].copy_(self.coef_host[: 2 * nphys], non_blocking=True)
        ops.lincomb(y, 1.0, self.w[: 2 * nphys], self.coef[: 2 * nphys])


def lbfgsb_minimize(x, fg, vec, maxiter, m=50, maxls=50, pgtol=1e-16, factr=0.0, maxfun=math.inf, callback=None,
                    history=None):
    """L-BFGS-B 3.0 without bounds.  `x`: flat vector (updated in place); `fg(x) -> (f, g)`
    with g written/returned as a flat vector (f may stay on the device: it reaches the host through
    `vec.read_probes`); `vec`: vector backend.  Returns dict(task, warnflag, nit, funcalls, f).

    The host reads device results twice per iteration: once after the evaluation (loss, <g, d>,
    max |g| together with the <d, d> and <g_old, d> of the direction), once after the pass over the
    history that forms the new rows of S^T Y, S^T S, Y^T Y.  From the second iteration on the first
    trial step is 1, so the first evaluation of a line search is launched before <g_old, d> is
    known; a non-descent direction (never seen with a positive-definite memory) discards it.

    `history`: warm start -- [(s_i, y_i, <g_i, s_i>)] oldest first, the correction pairs of earlier iterations
    (s_i = x_{i+1} - x_i, y_i = g_{i+1} - g_i): they pass through the same acceptance test and matrix update as
    pairs formed here, and the first line search starts from step 1 as every iteration but the very first does.
    (Resuming a run; the teacher-forced parity test hands over the reference's own iterates this way.)"""
    from scipy.linalg import solve_triangular

    epsmch = np.finfo(np.float64).eps
    sy = np.zeros((m, m))  # sy[i, j] = s_i . y_j (logical order, oldest first)
    ss = np.zeros((m, m))
    yy = np.zeros((m, m))
    slots = []  # physical row of logical history entry i
    theta = 1.0
    col = 0
    nit = 0
    nfev = 0
    nskip = 0
    big = 1e10
    t = vec.new()  # x at the start of the line search
    r = vec.new()  # g at the start of the line search, later y
    d = vec.new()
    g = vec.new()

    def evaluate(step):
        """x = t + step d, evaluation there, probes launched: -> nothing; read with read_probes()."""
        vec.set_axpy(x, t, d, step)
        fobj, gnew = fg(x)
        vec.copy(g, gnew)
        vec.probe_eval(fobj, g, d)

    pending = None  # (Y^T g, S^T g) for the next direction, when already known

    def push_pair(s, y, dr, ss_new):
        """matupd: the accepted pair (s, y), dr = <s, y>, enters the memory; one pass over the history gives the new
        rows of S^T Y, S^T S, Y^T Y and, for the next direction, S^T g and Y^T g (the history is the dominant
        traffic of an iteration)."""
        nonlocal col, theta, pending
        if col < m:
            slot = col
            slots.append(slot)
            col += 1
        else:
            slot = slots.pop(0)
            slots.append(slot)
            sy[:-1, :-1] = sy[1:, 1:]
            ss[:-1, :-1] = ss[1:, 1:]
            yy[:-1, :-1] = yy[1:, 1:]
        vec.store_pair(slot, s, y)
        (s_y, s_s, s_g), (y_y, y_s, y_g) = vec.history_products(len(slots), [y, s, g])
        pending = (y_g, s_g)
        rr = float(y_y[slot])  # y_new . y_new is one of the products of that pass: no separate reduction
        theta = rr / dr
        c = col - 1
        order = np.asarray(slots)
        idx = np.arange(col)
        sy[idx, c] = s_y[order]
        sy[c, idx] = y_s[order]
        ss[idx, c] = ss[c, idx] = s_s[order]
        yy[idx, c] = yy[c, idx] = y_y[order]
        sy[c, c] = dr
        ss[c, c] = float(s_s[slot]) if ss_new is None else ss_new
        yy[c, c] = rr

    fobj, gnew = fg(x)
    vec.copy(g, gnew)
    nfev += 1
    vec.probe_eval(fobj, g, g)
    _, _, _, sbgnrm, f = vec.read_probes()
    if sbgnrm <= pgtol:
        return dict(task="CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL", warnflag=0, nit=0, funcalls=nfev, f=f)
    first = True  # the very first line search of a cold start scales its trial step by 1 / |d|
    for s_old, y_old, gs_old in history or []:
        first = False
        s_y = vec.dot(s_old, y_old)
        if s_y <= epsmch * (-float(gs_old)):
            nskip += 1
            continue
        push_pair(s_old, y_old, s_y, None)

    while True:
        # ---- search direction d = -B^{-1} g (compact representation) --------------------
        vec.scale_into(d, g, -1.0 / theta)
        if col > 0:
            order = np.asarray(slots)  # logical -> physical
            if pending is not None:
                p1, p2 = pending  # Y^T g, S^T g came out of the pass that formed the update rows
            else:
                sg_, yg_ = vec.history_products(len(slots), [g])
                p1, p2 = yg_[0], sg_[0]
            pending = None
            yg = np.asarray(p1)[order]
            sg = np.asarray(p2)[order] * theta
            # [[-D - Y^T Y / theta, -Rbar^T], [-Rbar, 0]] q = [yg, sg] with Rbar = triu(S^T Y): the zero
            # block makes it two triangular solves
            rbar = np.triu(sy[:col, :col])  # s_i.y_j for i <= j
            amat = np.diag(np.diag(sy[:col, :col])) + yy[:col, :col] / theta
            q1 = -solve_triangular(rbar, sg, lower=False, check_finite=False)
            q2 = -solve_triangular(rbar, yg + amat @ q1, trans=1, lower=False, check_finite=False)
            # d = -g/theta - (1/theta^2) (Y q1 + theta S q2)
            cy = np.zeros(len(slots))
            cs = np.zeros(len(slots))
            cy[order] = -q1 / theta**2
            cs[order] = -q2 / theta
            vec.history_lincomb(d, len(slots), cs, cy)

        # ---- line search (lnsrlb) ------------------------------------------------------------
        vec.probe_direction(d, g)
        vec.copy(t, x)
        vec.copy(r, g)
        fold = f
        stpmx = big
        launched = False
        if first:
            first = False
            dtd, gd = vec.read_probes()[:2]
            stp = min(1.0 / math.sqrt(dtd), stpmx)
        else:
            stp = 1.0
            evaluate(stp)  # before <g_old, d> is on the host
            launched = True
            dtd, gd, gd_new, gmax_new, f_new = vec.read_probes()
        gdold = gd
        ls_failed = False
        if gd >= 0.0:
            ls_failed = True  # not a descent direction (info = -4)
        else:
            search = _Dcsrch(stpmax=stpmx)
            stp, task = search.start(stp, f, gd)
            ifun = 0
            while task == "FG":
                ifun += 1
                nfev += 1
                iback = ifun - 1
                if iback >= maxls:
                    ls_failed = True
                    break
                if not launched:
                    evaluate(stp)
                    _, _, gd_new, gmax_new, f_new = vec.read_probes()
                launched = False
                f, gd, sbgnrm = f_new, gd_new, gmax_new
                stp, task = search.step(stp, f, gd)
                if nfev > maxfun:
                    break
            if task.startswith("ERROR"):
                ls_failed = True
        if ls_failed:
            vec.copy(x, t)
            vec.copy(g, r)
            f = fold
            if col == 0:
                return dict(task="ABNORMAL_TERMINATION_IN_LNSRCH", warnflag=2, nit=nit, funcalls=nfev, f=f)
            # refresh the memory and restart with the steepest-descent step
            col, slots, theta = 0, [], 1.0
            pending = None
            continue

        # ---- new iterate ---------------------------------------------------------------------
        nit += 1
        if callback is not None:
            callback(x)
        if nit >= maxiter:
            return dict(task="STOP: TOTAL NO. of ITERATIONS REACHED LIMIT", warnflag=1, nit=nit, funcalls=nfev, f=f)
        if nfev > maxfun:
            return dict(task="STOP: TOTAL NO. of f AND g EVALUATIONS EXCEEDS LIMIT", warnflag=1, nit=nit,
                        funcalls=nfev, f=f)
        if sbgnrm <= pgtol:
            return dict(task="CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL", warnflag=0, nit=nit, funcalls=nfev, f=f)
        ddum = max(abs(fold), abs(f), 1.0)
        if (fold - f) <= epsmch * factr * ddum:
            return dict(task="CONVERGENCE: REL_REDUCTION_OF_F_<=_FACTR*EPSMCH", warnflag=0, nit=nit, funcalls=nfev, f=f)

        # ---- BFGS update (matupd) ----------------------------------------------------------------
        vec.sub_into(r, g, r)  # y = g_new - g_old
        if stp == 1.0:
            dr = gd - gdold
            ddum = -gdold
        else:
            dr = (gd - gdold) * stp
            vec.scale_into(d, d, stp)  # s = stp * d
            ddum = -gdold * stp
        if dr <= epsmch * ddum:
            nskip += 1
            pending = None
            continue
        push_pair(d, r, dr, stp * stp * dtd if stp != 1.0 else dtd)


class LbfgsbOptimizer(Optimizer):
    def __init__(self, pgtol=1e-16, m=50, maxls=50, factr=0, dtype=None, mod=None, **kwargs):
        super().__init__(name="lbfgsb", displayname="L-BFGS-B", dtype=dtype)
        self.mod = mod
        self.pgtol, self.m, self.maxls, self.factr = pgtol, m, maxls, factr

    def run(self, x0, loss_grad, epochs=None, callback=None, epoch_start=0, **kwargs):
        self.epoch = epoch_start
        tdtype = x0[0].dtype
        device = x0[0].device
        # evaluation buffer in the domain dtype, L-BFGS vectors in float64 (optimizer.py:63-71, 90-91)
        xe, xviews = pack_like(x0)
        copy_into(xe, xviews, x0)
        n = xe.numel()
        x = xe if tdtype == torch.float64 else xe.to(torch.float64)
        gbuf = torch.zeros(n, dtype=torch.float64, device=device)
        gviews = [t.view(a.shape) for t, a in zip(gbuf.split([a.numel() for a in x0]), x0)]

        # Launch-bound problems (<= 4 M unknowns: a 2-D 1024^2 evaluation is ~25 launches of 5 - 10 us each) replay the
        # EVALUATION as a hipGraph: the optimizer always evaluates at the same buffer `xe` and the evaluator writes the same
        # gradient buffer, so the first two evaluations run eagerly (lazy initialisation), the third is captured, the rest
        # are replays.  The vector algebra between evaluations depends on the host's line-search decisions and stays eager.
        ev = dict(count=0, graph=None, out=None, raw=None, begun=False)
        want_graph = _graph_wanted(n, loss_grad, epochs or 0)
        refresh, begin, end = (getattr(loss_grad, k, None) for k in ("refresh", "graph_begin", "graph_end"))

        def close_graph():
            if ev["begun"] and end is not None:
                end()
            ev.update(graph=None, begun=False)

        def evaluate():
            if ev["graph"] is not None:
                try:
                    if refresh is not None:
                        refresh()
                except RuntimeError:  # (more replays than rows of host scalars were provided for: eager from here on)
                    close_graph()
                    return loss_grad(xviews)
                ev["graph"].replay()
                loss, grads, raw = ev["out"]
                return loss, grads, (raw[0](**raw[1]) if raw[0] is not None else raw[1])
            ev["count"] += 1
            if want_graph and ev["count"] == 3:
                from .util import printlog

                try:
                    graph = torch.cuda.CUDAGraph()
                    if begin is not None:
                        begin(4 * (epochs or 0) + 64)
                        ev["begun"] = True
                    if refresh is not None:
                        refresh()
                    with torch.cuda.graph(graph):
                        loss, grads, pinfo = loss_grad(xviews)
                    kind = type(pinfo) if isinstance(pinfo, dict) else None
                    raw = {k: dict.__getitem__(pinfo, k) for k in dict.keys(pinfo)} if kind else pinfo
                    ev.update(graph=graph, out=(loss, grads, (kind, raw)))
                    global _graph_runtime_ready
                    _graph_runtime_ready = True
                    graph.replay()
                    return loss, grads, (kind(**raw) if kind else raw)
                except Exception as e:  # capture is an optimisation: never a reason to fail
                    printlog("odil_amd: hipGraph capture of the L-BFGS evaluation failed ({}: {}); running eagerly".format(
                        type(e).__name__, str(e).splitlines()[0] if str(e) else ""))
                    torch.cuda.synchronize()
                    close_graph()
            return loss_grad(xviews)

        def fg(xflat):
            self.evals += 1
            if xflat.data_ptr() != xe.data_ptr():
                xe.copy_(xflat)  # cast to the evaluation dtype
            loss, grads, pinfo = evaluate()
            self.pinfo = pinfo
            base = flat_base(grads)
            if base is not None and base.dtype == torch.float64 and base.numel() == n:
                g = base
            else:
                for gv, gi in zip(gviews, grads):
                    gv.copy_(gi)
                g = gbuf
            return loss, g  # the loss reaches the host through vec.read_probes()

        def callback_wrap(xflat):
            self.epoch += 1
            if callback:
                if xflat.data_ptr() != xe.data_ptr():
                    xe.copy_(xflat)
                callback(xviews, self.epoch, self.pinfo)

        vec = LbfgsVectors(n, self.m, device)
        try:
            res = lbfgsb_minimize(
                x, fg, vec, maxiter=epochs, m=self.m, maxls=self.maxls, pgtol=self.pgtol, factr=self.factr,
                callback=callback_wrap,
            )
        finally:
            close_graph()
        if x.data_ptr() != xe.data_ptr():
            xe.copy_(x)
        optinfo = Namespace()
        optinfo.warnflag = res["warnflag"]
        optinfo.task = res["task"]
        optinfo.evals = res["funcalls"]
        optinfo.epochs = res["nit"]
        if optinfo.warnflag not in [0, 1] or optinfo.epochs < epochs:
            raise EarlyStopError(
                ", ".join("{:}={:}".format(k, res.get(k, "")) for k in ["warnflag", "task", "funcalls", "nit"]),
                optinfo,
            )
        return xviews, optinfo


def make_optimizer(name, dtype=None, mod=None, **kwargs):
    if name == "lbfgsb":
        return LbfgsbOptimizer(dtype=dtype, mod=mod, **kwargs)
    elif name == "adam" or name == "adamn":
        return AdamNativeOptimizer(dtype=dtype, mod=mod, **kwargs)
    elif name == "gd":
        return GdOptimizer(dtype=dtype, mod=mod, **kwargs)
    elif name in ("lbfgs", "adam_tf"):
        raise ValueError("Optimizer '{}' is TensorFlow-specific; use 'lbfgsb' / 'adam'".format(name))
    raise ValueError("Unknown optimizer '{}'".format(name))
