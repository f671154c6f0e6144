"""Launch the fine-level and level-2 kernels of the div-sigma-grad workload (jInv's: testGMG.jl:57-75) a few times, for rocprofv3 --pmc passes.
usage: python3 scripts/divsiggrad_probe.py [cells]"""
import os, sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multigrid_jl_amd as mg
from multigrid_jl_amd import device as D
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lv = {32: 3, 64: 4, 128: 5, 256: 6}[cells]
mesh = mg.getRegularMesh([0.0, 1.0] * 3, [cells] * 3)
sigma = np.exp(np.random.default_rng(5).standard_normal(cells ** 3))
A = mg.getNodalDivSigGradMatrix(mesh, sigma)
A = (A + 1e-3 * abs(A).sum(axis=0).max() * sp.identity(A.shape[0], format="csr")).tocsr()
A.sort_indices()
p = mg.getMGparam(np.float64, np.int64, lv, 8, 2, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
mg.MGsetup(A, mesh, p, 1)
h = mg.to_device(p)
b = torch.from_numpy(np.ascontiguousarray(mg.seeded_rhs(A, 1))).cuda()
x = torch.zeros_like(b)
torch.cuda.synchronize()
h.solve_dev(b, x, 0.0, 4)
for lvl in (1, 2):
    for k in (D.MG_K_SMOOTH, D.MG_K_RESIDUAL, D.MG_K_SMOOTH_RESIDUAL):
        try:
            ms, bts = h.time_op(lvl, k, 3)
        except D.MGDeviceError:
            continue
        print(f"level {lvl} kernel {D.KERNEL_NAMES[k]}: {ms:.4f} ms", flush=True)
