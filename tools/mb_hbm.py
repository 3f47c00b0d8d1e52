"""HBM streams on this GPU: pure write (fill), copy, triad-like (2 reads + 1 write), 17-stream write."""
import torch, time
dev = torch.device("cuda:0")
n = 1 << 30  # 4 GB of float32
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
x = torch.empty(n, dtype=torch.float32, device=dev); y = torch.empty_like(x); z = torch.empty_like(x)
ms = t(lambda: x.fill_(1.0)); print("fill   4 GB: %.3f ms  %.2f TB/s written" % (ms, 4.295 / ms))
ms = t(lambda: y.copy_(x)); print("copy   4+4 GB: %.3f ms  %.2f TB/s total" % (ms, 8.59 / ms))
ms = t(lambda: torch.add(x, y, out=z)); print("add    8+4 GB: %.3f ms  %.2f TB/s total" % (ms, 12.885 / ms))
ms = t(lambda: x.sum()); print("sum    4 GB read: %.3f ms  %.2f TB/s" % (ms, 4.295 / ms))
