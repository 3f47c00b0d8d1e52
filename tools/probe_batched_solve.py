"""Which batched dense solve of torch-ROCm is usable for the block cyclic reduction's shapes (128 x 512 x 512, 1025 rhs)."""
import os, sys, time, torch
dev = torch.device("cuda:0")
def mk(batch, nb, k):
    g = torch.Generator(device=dev).manual_seed(0)
    R = torch.randn((batch, nb, nb), dtype=torch.float64, device=dev, generator=g)
    A = R @ R.transpose(1, 2) + nb * torch.eye(nb, dtype=torch.float64, device=dev)
    B = torch.randn((batch, nb, k), dtype=torch.float64, device=dev, generator=g)
    return A, B
def trial(name, fn, A, B):
    try:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        X = fn(A, B); torch.cuda.synchronize(); t1 = time.perf_counter()
        X = fn(A, B); torch.cuda.synchronize(); t2 = time.perf_counter()
        err = float((A @ X - B).abs().max())
        print("%-28s batch %4d nb %4d k %5d: %.1f ms (first %.1f)  resid %.1e" % (name, A.shape[0], A.shape[1], B.shape[2], (t2 - t1) * 1e3, (t1 - t0) * 1e3, err), flush=True)
    except Exception as e:
        print("%-28s batch %4d nb %4d k %5d: FAILED %s" % (name, A.shape[0], A.shape[1], B.shape[2], str(e).splitlines()[0][:120]), flush=True)
def lu(A, B):
    f, p = torch.linalg.lu_factor(A); return torch.linalg.lu_solve(f, p, B)
def lu_chunk(c):
    def fn(A, B):
        f, p = torch.linalg.lu_factor(A)
        return torch.cat([torch.linalg.lu_solve(f[i:i + c], p[i:i + c], B[i:i + c]) for i in range(0, A.shape[0], c)])
    return fn
def chol(A, B):
    return torch.cholesky_solve(B, torch.linalg.cholesky(A))
def inv(A, B):
    return torch.linalg.inv(A) @ B
def chol_inv(A, B):
    return torch.cholesky_inverse(torch.linalg.cholesky(A)) @ B
which = sys.argv[1:] or ["lu", "solve", "chol", "inv", "chol_inv", "lu8", "lu32"]
fns = dict(lu=lu, solve=torch.linalg.solve, chol=chol, inv=inv, chol_inv=chol_inv, lu8=lu_chunk(8), lu32=lu_chunk(32))
for batch, nb, k in ((8, 24, 49), (64, 256, 513), (128, 512, 1025), (32, 512, 1025), (4, 512, 1025), (1, 512, 1025), (32, 1024, 2049)):
    A, B = mk(batch, nb, k)
    for w in which:
        trial(w, fns[w], A, B)
