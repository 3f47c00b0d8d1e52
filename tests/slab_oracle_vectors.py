"""Test double of `odil_amd.slab_solvers.SlabLbfgsVectors` on torch CPU tensors: the same interface and the same
reductions through the communicator, immediate float64 arithmetic instead of the HIP kernels."""

import numpy as np
import torch


class TorchCpuSlabVectors:
    def __init__(self, n, m, comm):
        self.n, self.m, self.comm = n, m, comm
        self.w = torch.zeros((2 * m, n), dtype=torch.float64)
        self.scal = torch.zeros(8, dtype=torch.float64)

    def new(self):
        return torch.zeros(self.n, dtype=torch.float64)

    def copy(self, dst, src):
        dst.copy_(src)

    def set_axpy(self, out, t, d, a):
        out.copy_(t + a * d)

    def scale_into(self, dst, src, a):
        dst.copy_(a * src)

    def sub_into(self, dst, a, b):
        dst.copy_(a - b)

    def probe_direction(self, d, g):
        self.scal[0], self.scal[1] = d @ d, g @ d

    def probe_eval(self, f, g, d):
        self.scal[3], self.scal[4], self.scal[5], self.scal[6] = g @ d, g @ g, g.abs().max(), float(f)

    def read_probes(self):
        rows = self.comm.exchange("gather", self.scal.clone(), None)
        h = rows.sum(dim=0)
        h[5] = rows[:, 5].max()
        return float(h[0]), float(h[1]), float(h[3]), float(h[5]), float(h[6])

    def store_pair(self, slot, s, y):
        self.w[2 * slot] = s
        self.w[2 * slot + 1] = y

    def _sums(self, t):
        return self.comm.exchange("gather", t, None).sum(dim=0)

    def history_products(self, nphys, bs):
        if nphys == 0:
            z = np.zeros((len(bs), 0))
            return z, z
        out = self._sums(torch.stack([self.w[: 2 * nphys] @ b for b in bs])).numpy()
        return out[:, 0::2], out[:, 1::2]

    def dot(self, a, b):
        return float(self._sums((a @ b).reshape(1)))

    def history_lincomb(self, y, nphys, cs, cy):
        if nphys:
            c = torch.zeros(2 * nphys, dtype=torch.float64)
            c[0::2], c[1::2] = torch.as_tensor(cs), torch.as_tensor(cy)
            y += c @ self.w[: 2 * nphys]


class TorchCpuTailVectors(TorchCpuSlabVectors):
    """Double of `odil_amd.slab_solvers.ReplicatedTailVectors`: the tail [n_own:] is held by every rank and enters the
    reductions with the weight 1 / world."""

    def __init__(self, n, m, comm, n_own, world):
        super().__init__(n, m, comm)
        self.weight = torch.ones(n, dtype=torch.float64)
        self.weight[n_own:] = 1.0 / world

    def probe_direction(self, d, g):
        self.scal[0], self.scal[1] = (self.weight * d) @ d, (self.weight * g) @ d

    def probe_eval(self, f, g, d):
        self.scal[3], self.scal[4], self.scal[5], self.scal[6] = (self.weight * g) @ d, (self.weight * g) @ g, g.abs().max(), float(f)

    def history_products(self, nphys, bs):
        if nphys == 0:
            z = np.zeros((len(bs), 0))
            return z, z
        out = self._sums(torch.stack([self.w[: 2 * nphys] @ (self.weight * b) for b in bs])).numpy()
        return out[:, 0::2], out[:, 1::2]

    def dot(self, a, b):
        return float(self._sums(((self.weight * a) @ b).reshape(1)))
