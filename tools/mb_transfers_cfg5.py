"""The finest multigrid transfers of config 5 (one rank's ghost-extended arrays) in isolation: u = w0 + P(c1) and g1 = P^T g0
of a (129, 36, 256, 256) float array on the layout 'nccc', one field and four fields (on four streams, as the epoch runs
them).  python3 tools/mb_transfers_cfg5.py"""
import sys

import torch

sys.path.insert(0, ".")
from odil_amd import ops

dev = torch.device("cuda:0")
LOC = "nccc"
FS, CS = (129, 36, 256, 256), (65, 18, 128, 128)


def t(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


K = 4
w0 = [torch.randn(FS, dtype=torch.float32, device=dev) for _ in range(K)]
c1 = [torch.randn(CS, dtype=torch.float32, device=dev) for _ in range(K)]
u = [torch.empty(FS, dtype=torch.float32, device=dev) for _ in range(K)]
g1 = [torch.empty(CS, dtype=torch.float32, device=dev) for _ in range(K)]
fb, cb = w0[0].numel() * 4, c1[0].numel() * 4
ms = t(lambda: ops.interp_add(c1[0], LOC, add=w0[0], out=u[0]))
print("P   one field : %.3f ms  %.2f TB/s" % (ms, (2 * fb + cb) / ms / 1e9))
ms = t(lambda: ops.interp_adj_best(w0[0], LOC, CS, out=g1[0]))
print("P^T one field : %.3f ms  %.2f TB/s" % (ms, (fb + cb) / ms / 1e9))
streams = [torch.cuda.Stream() for _ in range(K)]


def four(f):
    cur = torch.cuda.current_stream()
    for i, s in enumerate(streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            f(i)
    for s in streams:
        cur.wait_stream(s)


ms = t(lambda: four(lambda i: ops.interp_add(c1[i], LOC, add=w0[i], out=u[i])))
print("P   4 fields, 4 streams: %.3f ms  %.2f TB/s" % (ms, K * (2 * fb + cb) / ms / 1e9))
ms = t(lambda: four(lambda i: ops.interp_adj_best(w0[i], LOC, CS, out=g1[i])))
print("P^T 4 fields, 4 streams: %.3f ms  %.2f TB/s" % (ms, K * (fb + cb) / ms / 1e9))
ms = t(lambda: [ops.interp_add(c1[i], LOC, add=w0[i], out=u[i]) for i in range(K)])
print("P   4 fields, 1 stream : %.3f ms  %.2f TB/s" % (ms, K * (2 * fb + cb) / ms / 1e9))
# (every field with its OWN intermediate: four calls that reuse one scratch buffer -- written, read, overwritten right away --
# run 3.4x slower, docs/rounds/kernel_log_r04.md)
space = [torch.empty((FS[0],) + CS[1:], dtype=torch.float32, device=dev) for _ in range(K)]


def adj_split(i):
    ops.interp_adj(w0[i], "." + LOC[1:], (FS[0],) + CS[1:], out=space[i])
    ops.interp_adj(space[i], "n...", CS, out=g1[i])


ms = t(lambda: [adj_split(i) for i in range(K)])
print("P^T 4 fields, 1 stream : %.3f ms  %.2f TB/s" % (ms, K * (fb + cb) / ms / 1e9))
# reference points: a plain copy and a read-only reduction of the same bytes
ms = t(lambda: u[0].copy_(w0[0]))
print("copy one field: %.3f ms  %.2f TB/s" % (ms, 2 * fb / ms / 1e9))
