"""Time the fine-level kernels of C2 (row-class format) with many repetitions.  usage: diag_rc.py [cells]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multigrid_jl_amd as mg
from multigrid_jl_amd import device as D
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lv = {32: 3, 64: 4, 128: 5, 256: 6}[cells]
A, mesh = mg.poisson_shifted([cells] * 3)
p = mg.getMGparam(np.float64, np.int64, lv, 8, 2, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
mg.MGsetup(A, mesh, p, 1)
h = mg.to_device(p)
b = torch.from_numpy(np.ascontiguousarray(mg.seeded_rhs(A, 1))).cuda()
x = torch.zeros_like(b)
h.solve_dev(b, x, 0.0, 3)
for lvl in (1, 2):
    for k in (D.MG_K_SMOOTH, D.MG_K_RESIDUAL, D.MG_K_PROLONG, D.MG_K_RESTRICT):
        h.time_op(lvl, k, 20)
        ms, bts = h.time_op(lvl, k, 200)
        print(f"level {lvl} {D.KERNEL_NAMES[k]:9s}: {ms*1e3:8.1f} us", flush=True)
