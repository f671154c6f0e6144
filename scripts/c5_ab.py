"""A/B of library builds on C5 (256^3, 16 right-hand sides): ms per solveMG step.  usage: AB_LIB=libmgvcycle_x.so python scripts/c5_ab.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multigrid_jl_amd as mg
from multigrid_jl_amd import device as D
if os.environ.get("AB_LIB"):
    D._lib = D.load_library(os.path.join(os.path.dirname(D.LIB_PATH), os.environ["AB_LIB"]))
    print("library:", os.environ["AB_LIB"], flush=True)
nrhs = 16
A, mesh = mg.poisson_shifted([256] * 3)
p = mg.getMGparam(np.float64, np.int64, 6, 8, 10, 1e-10, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
mg.MGsetup(A, mesh, p, nrhs)
b = torch.from_numpy(np.ascontiguousarray(mg.seeded_rhs(A, nrhs))).cuda()
h = mg.device.DeviceHierarchy(p, device_id=0, nrhs=nrhs)
x = torch.zeros_like(b)
h.solve_dev(b, x, 0.0, 2)
best = 1e9
for rep in range(3):
    x.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, res = h.solve_dev(b, x, 0.0, 6)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 6 * 1e3)
h.profile_reset(); h.profile_enable(True); x.zero_(); h.solve_dev(b, x, 0.0, 4); h.profile_enable(False)
pr = h.profile()
print("step %.3f ms" % best, "relres %.3e" % (res[-1] / res[0]), {f"L{k[0]}:{k[1]}": round(v[0] / max(v[1], 1), 3) for k, v in sorted(pr.items()) if k[0] <= 2}, flush=True)
