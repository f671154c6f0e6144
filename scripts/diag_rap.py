"""replaceMatrixInHierarchy: device numeric RAP (mg_rap_FP64) vs the host path, C2 size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import multigrid_jl_amd as mg
from multigrid_jl_amd.mgsetup import galerkin
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lv = {32: 3, 64: 4, 128: 5, 256: 6}[cells]
A, mesh = mg.poisson_shifted([cells] * 3)
p = mg.getMGparam(np.float64, np.int64, lv, 8, 3, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
mg.MGsetup(A, mesh, p, 1)
b = mg.seeded_rhs(A); x = np.zeros_like(b); mg.solveMG(p, b, x)
A2 = A.copy(); A2.data = A.data * 1.01
t0 = time.perf_counter(); mg.replaceMatrixInHierarchy(p, A2); t_dev = time.perf_counter() - t0
h = p.device; lib = h.lib
import ctypes as C
nz = np.ascontiguousarray(A2.data); om = np.full(lv, 0.8); done = C.c_longlong(0)
t0 = time.perf_counter(); lib.mg_rap_FP64(h.handle, nz.ctypes.data_as(C.POINTER(C.c_double)), nz.size, 0, om.ctypes.data_as(C.POINTER(C.c_double)), C.byref(done)); t_k = time.perf_counter() - t0
t0 = time.perf_counter(); Ac = galerkin(p.Rs[0], A2, p.Ps[0]); t_host1 = time.perf_counter() - t0
print(f"device replaceMatrixInHierarchy (RAP all levels + host refresh + coarse LU): {t_dev:.3f} s; mg_rap alone (incl. 947 MB H2D): {t_k*1e3:.1f} ms; "
      f"host Galerkin product of level 1 alone (row-parallel SpGEMM): {t_host1:.2f} s; max diff level 2: {np.abs(Ac.data - p.As[1].data).max()/np.abs(Ac.data).max():.2e}")
