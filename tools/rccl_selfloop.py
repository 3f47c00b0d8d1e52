#!/usr/bin/env python3
"""First contact of the slab paths' communication layer with RCCL on a box with ONE GPU: a process group of one rank
(backend nccl = RCCL) whose rank is its own neighbour.  `TorchDistComm(self_loop=True)` sends every message through
RCCL's send / receive kernels to itself, so that what the multi-GPU epochs rely on is exercised for real:

  * device tensors through `batch_isend_irecv`, `req.wait()` ordering the compute stream behind the transfer;
  * a POSTED exchange overlapping kernels on the compute stream, receive buffers that are reused every epoch,
    send buffers rewritten right after the wait (a stale or torn plane shows as a wrong checksum);
  * the scalar all-reduce and the all-gather of the quasi-Newton drivers' partial reductions;
  * a Poisson slab epoch whose exchanges run through this comm equals the one with the local closure bit for bit.

Prints one line `rccl self-loop ok ...` and exits 0, or raises.
"""
import os
import socket
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from odil_amd.slab import TorchDistComm

    comm = TorchDistComm(0, 1, self_loop=True)
    gen = torch.Generator(device=dev).manual_seed(1)
    n = 1 << 20  # 4 MB planes (the Poisson path's are 2 MB, config 5's packed messages 135 MB)
    lo_src, hi_src = torch.empty(n, device=dev), torch.empty(n, device=dev)
    work = torch.zeros(64 << 20, device=dev)
    for it in range(200):
        # new contents in the SAME send buffers every iteration, written right after the previous wait
        lo_src.copy_(torch.randn(n, device=dev, generator=gen))
        hi_src.copy_(torch.randn(n, device=dev, generator=gen))
        want_lo, want_hi = hi_src.clone(), lo_src.clone()  # a ring of one: lo arrives from above, hi from below
        kind = "post" if it % 2 else "halo"
        if kind == "post":
            token = comm.exchange("post", lo_src, hi_src)
            for _ in range(4):
                work.add_(1.0)  # compute-stream work the transfer overlaps
            recv_lo, recv_hi = comm.exchange("wait", token, None)
        else:
            recv_lo, recv_hi = comm.exchange("halo", lo_src, hi_src)
        assert torch.equal(recv_lo, want_lo) and torch.equal(recv_hi, want_hi), "iteration {}: wrong planes".format(it)
    total = comm.exchange("sum", torch.arange(5, dtype=torch.float64, device=dev), None)
    assert torch.equal(total.cpu(), torch.arange(5, dtype=torch.float64))
    rl, rh = comm.exchange("wrap", lo_src, hi_src)
    assert torch.equal(rl, hi_src) and torch.equal(rh, lo_src)
    rows = comm.exchange("gather", torch.arange(8, dtype=torch.float64, device=dev), None)
    assert tuple(rows.shape) == (1, 8) and torch.equal(rows[0].cpu(), torch.arange(8, dtype=torch.float64))
    # the quasi-Newton driver of the slab path, its reductions through RCCL's all-gather: the same iterates as with the
    # local closure
    from odil_amd.slab import LocalComm
    from odil_amd.slab_solvers import SlabPoissonLbfgs

    plain = TorchDistComm(0, 1)  # (no self-loop: a single rank of this driver has no neighbour planes to send)
    outs = []
    for c in (plain, LocalComm()):
        run = SlabPoissonLbfgs(32, 0, 1, dtype=torch.float64, device=dev)
        res = run.minimize(c, 6, m=4)
        outs.append((res["f"], res["funcalls"], run.x.clone()))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1] and torch.equal(outs[0][2], outs[1][2])
    torch.cuda.synchronize()
    if "--sweep" in sys.argv:
        # per-exchange cost of the communication layer as the slab epochs use it (`batch_isend_irecv` of a pair of planes +
        # wait, enqueued back to back: the device-side time per exchange, the host's enqueue time beside it) by message
        # size -- the latency term of DESIGN section 6's weak-scaling prediction.  (One device: RCCL's send / receive
        # kernels copy through its own buffers; no xGMI link is crossed.)
        import time

        for nbytes in (8, 4096, 65536, 1 << 20, 4 << 20, 32 << 20, 128 << 20):
            k = max(2, nbytes // 4)
            a, b = torch.zeros(k, device=dev), torch.zeros(k, device=dev)
            reps = 200 if nbytes <= (4 << 20) else 30
            for _ in range(5):
                comm.exchange("halo", a, b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(reps):
                comm.exchange("halo", a, b)
            e1.record()
            host = (time.perf_counter() - t0) / reps
            torch.cuda.synchronize()
            dev_s = e0.elapsed_time(e1) * 1e-3 / reps
            print("exchange of 2 x {:>9d} B: {:8.1f} us on the device ({:6.1f} GB/s per direction), host enqueue {:6.1f} us".format(
                4 * k, dev_s * 1e6, 4 * k / dev_s / 1e9, host * 1e6), flush=True)
        for nbytes in (8, 64):
            t = torch.zeros(nbytes // 8, dtype=torch.float64, device=dev)
            for _ in range(5):
                comm.exchange("sum", t, None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                comm.exchange("sum", t, None)
            torch.cuda.synchronize()
            print("all-reduce of {} B: {:.1f} us per call (enqueue + completion, back to back)".format(
                nbytes, (time.perf_counter() - t0) / 200 * 1e6), flush=True)
    dist.destroy_process_group()
    print("rccl self-loop ok: 200 exchanges of 2 x {} MB (halo / post + wait), all-reduce, all-gather, wrap, "
          "6 L-BFGS iterations".format(4 * n >> 20))


if __name__ == "__main__":
    main()
