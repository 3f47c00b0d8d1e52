"""Two smoothing sweeps as two launches against the one-pass kernels (csrc/smooth2.hip), per level size.

    python tools/mb_smooth2.py [poisson|svar] [N ...]

Prints per size: ms of two single sweeps, ms of the fused pair for several chunk lengths, the bytes each must move
and the rate on them."""

import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from odil_amd import ops  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "poisson"
    sizes = [int(v) for v in sys.argv[2:]] or [512, 256, 128, 64]
    dev = torch.device("cuda:0")
    for dtype in (torch.float64, torch.float32):
        for n in sizes:
            shape = (n, n, n)
            g = torch.Generator(device=dev).manual_seed(0)
            x = torch.randn(shape, dtype=dtype, device=dev, generator=g)
            b = torch.randn(shape, dtype=dtype, device=dev, generator=g)
            y, z = torch.empty_like(x), torch.empty_like(x)
            h2 = [1.0 / n**2] * 3
            w = x.element_size()
            if kind == "poisson":
                t1 = timeit(lambda: (ops.poisson_jacobi(x, b, h2, 0.9, y), ops.poisson_jacobi(y, b, h2, 0.6, z)))
                line = "poisson {} {}^3: two launches {:.3f} ms ({:.2f} TB/s on 6 words)".format(
                    str(dtype)[6:], n, t1, 6 * w * x.numel() / t1 / 1e9)
                for zc in (0, 16, 32, 64, 128):
                    t2 = timeit(lambda: ops.poisson_jacobi2(x, b, h2, 0.9, 0.6, z, zc_hint=zc))
                    line += " | zc {} {:.3f} ms ({:.2f} TB/s on 3)".format(zc, t2, 3 * w * x.numel() / t2 / 1e9)
                print(line, flush=True)
                xc = torch.randn(tuple(v // 2 for v in shape), dtype=dtype, device=dev, generator=g)
                t1 = timeit(lambda: (ops.poisson_jacobi_synth(xc, x, b, h2, 0.9, y), ops.poisson_jacobi(y, b, h2, 0.6, z)))
                line = "   with the correction: synth + sweep, sweep {:.3f} ms".format(t1)
                for zc in (0, 16, 32, 64, 128):
                    t2 = timeit(lambda: ops.poisson_jacobi2_synth(xc, x, b, h2, 0.9, 0.6, z, zc_hint=zc))
                    line += " | zc {} {:.3f} ms ({:.2f} TB/s on 3 1/8)".format(zc, t2, 3.125 * w * x.numel() / t2 / 1e9)
                print(line, flush=True)
            else:
                c = ops.poisson_jac_coeffs(shape, [np.float64(v) for v in h2], dtype, dev)
                t1 = timeit(lambda: (ops.stencil_var_smooth(c, x, b, 0.9, out=y), ops.stencil_var_smooth(c, y, b, 0.6, out=z)))
                line = "svar {} {}^3: two launches {:.3f} ms ({:.2f} TB/s on 20 words)".format(
                    str(dtype)[6:], n, t1, 20 * w * x.numel() / t1 / 1e9)
                for zc in (0, 16, 32, 64, 128):
                    t2 = timeit(lambda: ops.stencil_var_smooth2(c, x, b, 0.9, 0.6, out=z, zc_hint=zc))
                    line += " | zc {} {:.3f} ms ({:.2f} TB/s on 10)".format(zc, t2, 10 * w * x.numel() / t2 / 1e9)
                print(line, flush=True)


if __name__ == "__main__":
    main()
