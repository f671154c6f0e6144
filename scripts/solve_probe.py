"""A few solves of the C2 workload (for rocprofv3 --kernel-trace).  usage: python3 scripts/solve_probe.py [cells] [steps] [solves]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multigrid_jl_amd as mg
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
solves = int(sys.argv[3]) if len(sys.argv) > 3 else 3
lv = {32: 3, 64: 4, 128: 5, 256: 6, 400: 7, 512: 7}.get(cells, 4)
A, mesh = mg.poisson_shifted([cells] * 3)
p = mg.getMGparam(np.float64, np.int64, lv, 8, steps, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0, "FullWeighting")
mg.MGsetup(A, mesh, p, 1)
h = mg.to_device(p)
b = torch.from_numpy(mg.seeded_rhs(A, 1)).cuda()
x = torch.zeros_like(b)
for s in range(solves):
    x.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    h.solve_dev(b, x, 0.0, steps); torch.cuda.synchronize()
    print(f"solve {s}: {(time.perf_counter() - t0) / steps * 1e3:.4f} ms per step", flush=True)
