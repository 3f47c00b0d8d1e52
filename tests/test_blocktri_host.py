"""Host logic of odil_amd/blocktri.py (torch-CPU tensors): the normal operator S^T S as stencil coefficient arrays, its
dense block-tridiagonal form and the block cyclic reduction, against explicitly assembled dense matrices -- for the
shift pattern of the implicit heat operator (reference examples/heat/heat.py:36-137: two time levels, three space
points each) with rows that do not couple across the ends of the time axis."""

import argparse

import numpy as np
import pytest
import torch

from odil_amd import blocktri


def fake_op(shape, outputs, seed):
    """An object with the attributes blocktri reads from core.LinearizedOperator: stencil blocks with random
    coefficients; `outputs`: per output a list of shifts.  Rows at the low end of axis 0 do not read level -1."""
    gen = torch.Generator().manual_seed(seed)
    size = int(np.prod(shape))
    field = argparse.Namespace(array=torch.zeros(shape, dtype=torch.float64), loc="c" * len(shape))
    blocks, dense = [], []
    for o, shifts in enumerate(outputs):
        S = torch.zeros((size, size), dtype=torch.float64)
        for s in shifts:
            c = torch.randn(shape, generator=gen, dtype=torch.float64)
            if s[0] < 0:
                c[: -s[0]] = 0  # nothing below the first level
            if s[0] > 0:
                c[-s[0]:] = 0
            blocks.append((o * size, size, "stencil", "u", (c.clone(), s, field.loc, shape)))
            idx = torch.arange(size).reshape(shape)
            cols = torch.roll(idx, shifts=tuple(-v for v in s), dims=tuple(range(len(shape))))  # column of i: i + s
            S[torch.arange(size), cols.reshape(-1)] += c.reshape(-1)
        dense.append(S)
    op = argparse.Namespace(key_to_field={"u": field}, blocks=blocks)
    return op, torch.cat(dense, dim=0)


HEAT = [[(0, 0), (0, 1), (0, -1), (-1, 0), (-1, 1), (-1, -1)], [(0, 0)], [(0, 0), (0, -1)], [(0, 0), (-1, 0)]]


@pytest.mark.parametrize("shape", [(8, 6), (7, 5), (5, 3, 4)])
def test_normal_stencil_and_blocks_equal_dense(shape):
    outputs = HEAT if len(shape) == 2 else [[(0, 0, 0), (0, 1, 0), (0, 0, -1), (-1, 0, 0), (-1, 0, 1)], [(0, 0, 0), (-1, 0, 0)]]
    op, S = fake_op(shape, outputs, 3)
    A = S.t() @ S + 0.3**2 * torch.eye(S.shape[1], dtype=torch.float64)
    A = A + 0.2**2 * torch.diag(torch.diag(A))  # reference linsolver.py:19-23: dampdiag acts on the damped matrix
    normal = blocktri.normal_stencil(op, "u", damp=0.3, dampdiag=0.2)
    axis = blocktri.recognise(normal, shape)
    assert axis == 0
    L, D, U = blocktri.dense_blocks(normal, shape, axis)
    n, nb = shape[0], int(np.prod(shape[1:]))
    full = torch.zeros((n * nb, n * nb), dtype=torch.float64)
    for i in range(n):
        full[i * nb:(i + 1) * nb, i * nb:(i + 1) * nb] = D[i]
        if i > 0:
            full[i * nb:(i + 1) * nb, (i - 1) * nb:i * nb] = L[i]
        if i + 1 < n:
            full[i * nb:(i + 1) * nb, (i + 1) * nb:(i + 2) * nb] = U[i]
    assert float((full - A).abs().max()) < 1e-12 * float(A.abs().max())
    assert float(L[0].abs().max()) == 0 and float(U[n - 1].abs().max()) == 0


@pytest.mark.parametrize("n", [1, 2, 3, 8, 13, 32])
def test_block_cyclic_reduction_vs_dense_solve(n):
    gen = torch.Generator().manual_seed(n)
    nb, k = 6, 3
    R = torch.randn((n * nb, n * nb + 4), generator=gen, dtype=torch.float64)
    A = R @ R.t()
    for i in range(n):  # keep the block-tridiagonal part of an SPD matrix, made diagonally heavy
        for j in range(n):
            if abs(i - j) > 1:
                A[i * nb:(i + 1) * nb, j * nb:(j + 1) * nb] = 0
    A += 10 * n * torch.eye(n * nb, dtype=torch.float64)
    L = torch.stack([A[i * nb:(i + 1) * nb, (i - 1) * nb:i * nb] if i > 0 else torch.zeros(nb, nb, dtype=torch.float64) for i in range(n)])
    D = torch.stack([A[i * nb:(i + 1) * nb, i * nb:(i + 1) * nb] for i in range(n)])
    U = torch.stack([A[i * nb:(i + 1) * nb, (i + 1) * nb:(i + 2) * nb] if i + 1 < n else torch.zeros(nb, nb, dtype=torch.float64) for i in range(n)])
    B = torch.randn((n, nb, k), generator=gen, dtype=torch.float64)
    X = blocktri.solve_block_tridiagonal(L, D, U, B)
    want = torch.linalg.solve(A, B.reshape(n * nb, k)).reshape(n, nb, k)
    assert float((X - want).abs().max()) < 1e-11 * float(want.abs().max())


def test_solver_object_solves_the_normal_equations():
    shape = (16, 9)
    op, S = fake_op(shape, HEAT, 5)
    solver = blocktri.BlockTridiagonalNormal(op, "u")
    assert solver.ok and solver.axis == 0
    gen = torch.Generator().manual_seed(1)
    b = torch.randn((4, S.shape[1]), generator=gen, dtype=torch.float64)
    x = solver.solve(b)
    want = torch.linalg.solve(S.t() @ S, b.t()).t()
    assert float((x - want).abs().max()) < 1e-9 * float(want.abs().max())
    # shifts reaching two levels: not block tridiagonal along that axis; blocks of the other axis are tried instead
    op2, _ = fake_op((4, 4), [[(0, 0), (-2, 0), (0, 2)]], 2)
    assert not blocktri.BlockTridiagonalNormal(op2, "u").ok
