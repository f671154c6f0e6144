"""Time the device-resident Krylov drivers on C2 (one cycle from x = 0 per iteration as preconditioner):
usage: python scripts/diag_pcg.py [cells]   (environment switches such as MG_NO_MARCH2_ZERO=1 apply)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multigrid_jl_amd as mg

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lv = {32: 3, 64: 4, 128: 5, 256: 6}[cells]
A, mesh = mg.poisson_shifted([cells] * 3)
p = mg.getMGparam(np.float64, np.int64, lv, 8, 10, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
mg.MGsetup(A, mesh, p, 1)
h = mg.to_device(p)
b = torch.from_numpy(mg.seeded_rhs(A, 1)).cuda()
n = A.shape[0]
# untimed pre-warm (a one-off 70-80 ms stall lands somewhere in the first tens of ms of GPU activity of a fresh process)
_x = torch.zeros_like(b)
_t0 = time.perf_counter()
while time.perf_counter() - _t0 < 0.6:
    h.cycle_dev(b, _x, 1)
for name, fn in (("cycle from x = 0 (preconditioner call)", None), ("solveCG_MG (mg_pcg_dev)", "pcg"), ("solveGMRES_MG (mg_fgmres_dev, inner 5)", "fgmres")):
    x = torch.zeros_like(b)
    if fn is None:
        for _ in range(3):
            h.cycle_dev(b, x, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 20
        for _ in range(K):
            h.cycle_dev(b, x, 1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print(f"{name}: {dt*1e3:.4f} ms per cycle = {n/dt/1e9:.2f} G DoF-updates/s", flush=True)
    elif fn == "fgmres":
        h.fgmres_dev(b, x, 5, 0.0, 1)
        x.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R = 3
        out = h.fgmres_dev(b, x, 5, 0.0, R)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (5 * R)
        print(f"{name}: {dt*1e3:.4f} ms per inner step (cycle from x = 0 + A*z + modified Gram-Schmidt), {5 * R} inner steps", flush=True)
    else:
        h.pcg_dev(b, x, 0.0, 3)
        x.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 12
        out = h.pcg_dev(b, x, 0.0, K)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print(f"{name}: {dt*1e3:.4f} ms per iteration (cycle from x = 0 + A*p with p'Ap + update with ||r|| + z'r + update of p), {K} iterations: {out[:2] if isinstance(out, tuple) else out}", flush=True)
h.close()
