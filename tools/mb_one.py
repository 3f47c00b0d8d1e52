"""One smoothing configuration a few times (for the profiler): python tools/mb_one.py poisson|svar N [f32]"""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from odil_amd import ops  # noqa: E402

kind, n = sys.argv[1], int(sys.argv[2])
dtype = torch.float32 if len(sys.argv) > 3 and sys.argv[3] == "f32" else torch.float64
dev = torch.device("cuda:0")
shape = (n, n, n)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(shape, dtype=dtype, device=dev, generator=g)
b = torch.randn(shape, dtype=dtype, device=dev, generator=g)
y, z = torch.empty_like(x), torch.empty_like(x)
h2 = [1.0 / n**2] * 3
for _ in range(5):
    if kind == "poisson":
        ops.poisson_jacobi(x, b, h2, 0.9, y)
        ops.poisson_jacobi2(x, b, h2, 0.9, 0.6, z)
    else:
        c = ops.poisson_jac_coeffs(shape, [np.float64(v) for v in h2], dtype, dev)
        ops.stencil_var_smooth(c, x, b, 0.9, out=y)
        ops.stencil_var_smooth2(c, x, b, 0.9, 0.6, out=z)
torch.cuda.synchronize()
