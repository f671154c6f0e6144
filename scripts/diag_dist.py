"""Orchestration overhead of the Python-sequenced distributed cycle at world_size 1 vs the C library."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multigrid_jl_amd as mg
from multigrid_jl_amd import distributed as dd
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lv = {32: 3, 64: 4, 128: 5, 256: 6}[cells]
A, mesh = mg.poisson_shifted([cells] * 3)
p = mg.getMGparam(np.float64, np.int64, lv, 8, 10, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
mg.MGsetup(A, mesh, p, 1)
be = dd.HipBackend(0)
H = dd.DistributedHierarchy.from_global(p, dd.SingleComm(), be, dd.box_owner(mesh.n + 1, [1, 1, 1]), 1)
b = H.scatter_fine(mg.seeded_rhs(A)); x = torch.zeros_like(b)
for i in range(3):
    x.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    it, rv = H.solve(b, x, 0.0, 10); torch.cuda.synchronize()
    print(f"distributed(world=1) solve: {(time.perf_counter()-t0)/10*1e3:.3f} ms/step, relres {rv[-1]/rv[0]:.3e}", flush=True)
# CPU-side enqueue cost only (no per-step sync): time to enqueue 10 cycles
x.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10): H.cycle(b, x, i == 0)
t_enq = time.perf_counter() - t0; torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"10 cycles: enqueue {t_enq/10*1e3:.3f} ms/cycle (CPU), complete {t_all/10*1e3:.3f} ms/cycle", flush=True)
