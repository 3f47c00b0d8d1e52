"""Host-side enqueue time vs GPU time of the emulated slab epoch (is the host or the GPU the limit?)."""
import sys, time, torch
sys.path.insert(0, '.')
from odil_amd.slab import SlabPoissonAdam, run_lockstep
dev = torch.device('cuda:0')
N, world = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 2
ranks = [SlabPoissonAdam(N, r, world, dtype=torch.float64, device=dev) for r in range(world)]
run_lockstep(ranks, 3); torch.cuda.synchronize()
for rep in range(3):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record(); run_lockstep(ranks, 10); b.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("enqueue %.3f ms / rank-epoch, wall %.3f, events %.3f ms / rank-epoch" % ((t1 - t0) * 1e3 / 10 / world, (t2 - t0) * 1e3 / 10 / world, a.elapsed_time(b) / 10 / world))
