import sys, os, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["bench.py"]
import bench
from odil_amd import gmg
args = bench.parse_args()
dev = torch.device("cuda:0")
for label, pm, tail in (("pairs on", 128**3, 8192), ("pairs on", 128**3, 8192)):
    gmg.PoissonGMG.pair_min_cells = pm
    gmg.PoissonGMG.tail_max_cells = tail
    out = bench.other_configs(args, dev)
    print(label, {k: round(out[k]["ms_per_step"], 2) for k in ("4b", "4b_general", "4b_varcoef", "4c_diffusion") if "ms_per_step" in out.get(k, {})}, flush=True)
