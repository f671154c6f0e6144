"""A/B timing of experiment builds of libmgvcycle (make variant NAME=.. DEFS=..): the two-stage marching kernel and the
two single-stage launches it replaces, C2 fine level."""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multigrid_jl_amd as mg
from multigrid_jl_amd import device as D

cells = int(os.environ.get("AB_CELLS", "256"))
A, mesh = mg.poisson_shifted([cells] * 3)
p = mg.getMGparam(np.float64, np.int64, 6 if cells == 256 else 4, 8, 10, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0, "FullWeighting")
mg.MGsetup(A, mesh, p, 1)
b = torch.from_numpy(mg.seeded_rhs(A, 1)).cuda()
libs = sorted(glob.glob(os.path.join(os.path.dirname(D.LIB_PATH), "libmgvcycle*.so")))
for path in libs:
    D._lib = D.load_library(path)
    h = D.DeviceHierarchy(p, 0, 1)
    x = torch.zeros_like(b)
    h.solve_dev(b, x, 0.0, 3)
    out = [os.path.basename(path)]
    for k, name in ((D.MG_K_SMOOTH, "smooth"), (D.MG_K_RESIDUAL, "resid"), (D.MG_K_SMOOTH_RESIDUAL, "smooth+resid")):
        try:
            ms, _ = h.time_op(1, k, 30)
            out.append(f"L1:{name} {ms*1e3:.1f}us")
        except Exception as e:
            out.append(f"L1:{name} n/a")
    print("  ".join(out), flush=True)
    h.close()
