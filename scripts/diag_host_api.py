"""PCIe-inclusive rate of the host-pointer API (what a Julia caller sees): mg_solve_FP64 with numpy buffers."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import multigrid_jl_amd as mg
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lv = {32: 3, 64: 4, 128: 5, 256: 6}[cells]
A, mesh = mg.poisson_shifted([cells] * 3)
p = mg.getMGparam(np.float64, np.int64, lv, 8, 10, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
mg.MGsetup(A, mesh, p, 1)
b = mg.seeded_rhs(A)
x = np.zeros_like(b)
mg.solveMG(p, b, x)                    # warm-up (upload + first call)
for k in (1, 10):
    p.maxOuterIter = k
    x[...] = 0
    t0 = time.perf_counter(); mg.solveMG(p, b, x); dt = time.perf_counter() - t0
    n = A.shape[0]
    print(f"host API solveMG, {k} cycles: {dt*1e3:.2f} ms total -> {n*k/dt/1e9:.2f} G DoF-updates/s "
          f"(b up + x up + x down = {3*8*n/1e6:.0f} MB over PCIe)", flush=True)
# the same with the caller's long-lived arrays page-locked (mg_host_register), and one preconditioner-style cycle
from multigrid_jl_amd import device as D
D.host_register(b); D.host_register(x)
mg.solveMG(p, b, x)
for k in (1, 10):
    p.maxOuterIter = k
    x[...] = 0
    t0 = time.perf_counter(); mg.solveMG(p, b, x); dt = time.perf_counter() - t0
    print(f"host API solveMG, PINNED b/x, {k} cycles: {dt*1e3:.2f} ms total -> {n*k/dt/1e9:.2f} G DoF-updates/s", flush=True)
z = np.zeros_like(b)
D.host_register(z)
mg.recursiveCycle(p, b, z, 1)
dt = 0.0
for _ in range(5):
    z[...] = 0
    t0 = time.perf_counter()
    mg.recursiveCycle(p, b, z, 1)
    dt += (time.perf_counter() - t0) / 5
print(f"host API recursiveCycle (preconditioner call), PINNED: {dt*1e3:.2f} ms -> {n/dt/1e9:.2f} G DoF-updates/s", flush=True)
D.host_unregister(b); D.host_unregister(x); D.host_unregister(z)
