"""Cycles and time of every multigrid solve of config 4b (Newton 512^3): python3 tools/newton_cycles.py [scale]"""
import sys, time, torch
sys.path.insert(0, '.')
import bench_configs
from odil_amd import gmg
orig = gmg.PoissonGMG.solve
def solve(self, b, **kw):
    st = kw.get("status") if kw.get("status") is not None else {}
    kw["status"] = st
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = orig(self, b, **kw)
    torch.cuda.synchronize()
    print("solve: shape %s cycles %s residual %.3e converged %s  %.1f ms  (tol %s maxiter %s)" % (
        tuple(b.shape), st.get("niter"), st.get("residual"), st.get("converged"), 1e3 * (time.perf_counter() - t0),
        kw.get("tol"), kw.get("maxiter")), flush=True)
    return x
gmg.PoissonGMG.solve = solve
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
res = bench_configs.run_config("4b", scale)
print({k: res[k] for k in ("ms_per_epoch", "loss", "epochs")})
