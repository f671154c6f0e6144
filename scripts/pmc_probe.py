"""Launch each fine-level kernel of the C2 workload a few times (for rocprofv3 --pmc passes).
usage: python3 scripts/pmc_probe.py [cells] [nrhs]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multigrid_jl_amd as mg
from multigrid_jl_amd import device as D
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nrhs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lv = {32: 3, 64: 4, 128: 5, 256: 6, 400: 7}[cells]
A, mesh = mg.poisson_shifted([cells] * 3)
p = mg.getMGparam(np.float64, np.int64, lv, 8, 2, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
mg.MGsetup(A, mesh, p, nrhs)
h = mg.to_device(p)
b = torch.from_numpy(np.ascontiguousarray(mg.seeded_rhs(A, nrhs))).cuda()
x = torch.zeros_like(b)
torch.cuda.synchronize()
h.solve_dev(b, x, 0.0, 2)
for lvl in (1, 2):
    for k in (D.MG_K_SMOOTH, D.MG_K_RESIDUAL, D.MG_K_SMOOTH_RESIDUAL, D.MG_K_FOUR_STAGE, D.MG_K_PROLONG, D.MG_K_RESTRICT, D.MG_K_DSCALE, D.MG_K_NORM):
        try:
            ms, bts = h.time_op(lvl, k, 3)
        except D.MGDeviceError:
            continue   # (the two-stage kernel does not serve this level / this nrhs)
        print(f"level {lvl} kernel {D.KERNEL_NAMES[k]}: {ms:.4f} ms, algorithmic {bts/1e6:.1f} MB, {bts/ms/1e6:.0f} GB/s", flush=True)
