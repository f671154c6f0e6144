"""Time of ONE coarsest-level solve from sparse factors (33^3 Poisson level = 35 937 rows) on the device:
single-workgroup level sweep against the chip-wide form.  Usage: python scripts/lu_coarse_time.py"""
import os, sys, time
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_jl_amd as mg
from multigrid_jl_amd import device as D
if os.environ.get("AB_LIB"):
    D._lib = D.load_library(os.path.join(os.path.dirname(D.LIB_PATH), os.environ["AB_LIB"]))
    print("library:", os.environ["AB_LIB"], flush=True)

def run(multi):
    os.environ["MG_LU_MULTI_MIN_ROWS"] = "0" if multi else "1000000000"
    os.environ["MG_DEBUG_FORMAT"] = "1"
    A, M = mg.poisson_shifted([32, 32, 32])
    q = mg.getMGparam(np.float64, np.int64, 1, 8, 1, 1e-10, "Jac", 0.8, 2, 1)
    t0 = time.time()
    from multigrid_jl_amd.mgsetup import coarse_lu
    q.As, q.Ps, q.Rs, q.relaxPrecs, q.Meshes, q.levels = [A], [], [], [], [M], 1
    q.LU = coarse_lu(A)
    t1 = time.time()
    rng = np.random.default_rng(5)
    B = rng.standard_normal(A.shape[0])
    t2 = time.time()
    X = mg.recursiveCycle(q, B.copy(), np.zeros_like(B), 1)
    print(f"upload + first cycle {time.time()-t2:.2f}s")
    Xo = q.LU.solve(B)
    err = np.abs(X - Xo).max() / np.abs(Xo).max()
    H = q.device
    b = torch.from_numpy(B).cuda(); x = torch.zeros_like(b)
    torch.cuda.synchronize()
    t = []
    for _ in range(5):
        torch.cuda.synchronize(); s = time.perf_counter()
        H.cycle_dev(b, x, 1, 1)
        torch.cuda.synchronize(); t.append(time.perf_counter() - s)
    print(f"multi={multi} n={A.shape[0]} factor {t1-t0:.1f}s err {err:.2e} cycle(ms) {[round(1e3*v,2) for v in t]}", flush=True)
    mg.clear_(q)

for m in (False, True):
    run(m)
