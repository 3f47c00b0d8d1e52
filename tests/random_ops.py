"""Random straight-line stencil operators shared by the GPU parity test of the generated kernels
(tests/test_workloads_gpu.py) and the CPU check of the symbolic gradient expressions (tests/test_stencil_grad_host.py)."""

import numpy as np


def random_operator(seed):
    """A random straight-line program over shifted reads of two fields, index masks, constants and
    the elementwise vocabulary of `mod` (smooth where it has to be: divisors and log / sqrt arguments
    are kept away from zero)."""
    rng = np.random.default_rng(seed)
    plan = []
    for _ in range(int(rng.integers(6, 14))):
        kind = rng.choice(["unary", "binary", "where", "minmax", "pow", "div", "roll", "rows", "param"])
        plan.append((kind, int(rng.integers(0, 1000)), int(rng.integers(0, 1000)), int(rng.integers(0, 1000)),
                     float(rng.uniform(-1.5, 1.5))))
    shifts = [(int(a), int(b)) for a, b in rng.integers(-2, 3, size=(5, 2))]
    frozen = [bool(v) for v in rng.integers(0, 4, size=5) == 0]

    def operator(ctx):
        m = ctx.mod
        it, ix = ctx.indices()
        x, y = ctx.points()
        vals = [ctx.field("a" if k % 2 == 0 else "b", *shifts[k], frozen=frozen[k] and k > 1) for k in range(5)]
        vals += [x * 0.7 + y, ctx.cast(0.3)]
        coeff = ctx.field("coeff")
        rows = ctx.extra
        for kind, i, j, k, c in plan:
            p, q, r = vals[i % len(vals)], vals[j % len(vals)], vals[k % len(vals)]
            if kind == "unary":
                f = [m.sin, m.cos, m.tanh, m.abs, m.square, lambda z: m.exp(m.clip(z, -3, 3)),
                     lambda z: m.sqrt(m.abs(z) + 0.5), lambda z: m.log(m.abs(z) + 0.5), m.relu, m.sigmoid][i % 10]
                vals.append(f(p * c))
            elif kind == "binary":
                vals.append([p + q, p - q * c, p * q, c - p, p * c + q][j % 5])
            elif kind == "where":
                cond = [(it + ix) % 2 == 0 if False else it > 2, ix == 0, p > q, (q < c) & (ix != 3), ~(p >= 0.1) | (it == 1)][k % 5]
                vals.append(m.where(cond, p, r * c))
            elif kind == "minmax":
                vals.append(m.maximum(p, q) - m.minimum(r, c))
            elif kind == "pow":
                vals.append([p**2, (m.abs(p) + 0.5) ** c, 1.5**(m.clip(q, -2, 2))][i % 3])
            elif kind == "roll":
                vals.append(m.roll(p + vals[0] * 0, (i % 5 - 2, j % 3 - 1), axis=(0, 1)) * c)
            elif kind == "rows":  # first / last rows imposed by concatenation (as heat_tmax / infer_constant do)
                full = p + vals[1] * 0
                vals.append(m.concatenate([rows[None, :], full[1:-1] * c, rows[None, :] * 0.5], axis=0))
            elif kind == "param":
                vals.append(p * coeff[k % 3] + coeff[(k + 1) % 3])
            else:
                vals.append(p / (2 + m.abs(q)) + c / (1.5 + q * q))
        outs = [("f{}".format(n), v + vals[n] * 0.1) for n, v in enumerate(vals[-3:])]  # every output on the grid
        if seed % 3 == 0:
            outs.append(("w", outs[0][1][1:, :-1] - outs[1][1][:-1, 1:]))
        return outs

    return operator
