"""A/B of library options (environment) on C2: ms per solveMG step (best of 5 x 20 steps) and HIP-event time per level.
usage: python scripts/env_ab.py ENV=VAL[,ENV=VAL] ...   ("-" = defaults)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multigrid_jl_amd as mg
from multigrid_jl_amd import device as D
if os.environ.get("AB_LIB"):     # a `make variant` build: libmgvcycle_<NAME>.so
    D._lib = D.load_library(os.path.join(os.path.dirname(D.LIB_PATH), os.environ["AB_LIB"]))
    print("library:", os.environ["AB_LIB"], flush=True)

cells = int(os.environ.get("AB_CELLS", "256"))
A, mesh = mg.poisson_shifted([cells] * 3)
p = mg.getMGparam(np.float64, np.int64, 6, 8, 20, 1e-10, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
mg.MGsetup(A, mesh, p, 1)
b = torch.from_numpy(mg.seeded_rhs(A, 1)).cuda()
for spec in sys.argv[1:]:
    envs = dict(kv.split("=") for kv in spec.split(",") if "=" in kv)
    for k, v in envs.items():
        os.environ[k] = v
    h = mg.device.DeviceHierarchy(p, device_id=0, nrhs=1)
    x = torch.zeros_like(b)
    h.solve_dev(b, x, 0.0, 3)
    best = 1e9
    for rep in range(5):
        x.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, res = h.solve_dev(b, x, 0.0, 20)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    h.profile_reset(); h.profile_enable(True); x.zero_(); h.solve_dev(b, x, 0.0, 10); h.profile_enable(False)
    pr = h.profile()
    lev = {}
    for (l, name), v in pr.items():
        lev[l] = lev.get(l, 0.0) + v[0] / 10 * 1e3
    print(spec, "step %.4f ms" % best, "relres %.3e" % (res[-1] / res[0]), {f"L{l}": round(t, 1) for l, t in sorted(lev.items())}, flush=True)
    if os.environ.get("AB_DETAIL"):
        print("   ", {f"L{k[0]}:{k[1]}": round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in sorted(pr.items()) if k[0] >= 2}, flush=True)
    h.close()
    for k in envs:
        del os.environ[k]
