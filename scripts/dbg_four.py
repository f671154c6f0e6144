import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update({"MG_ROWCLASS_MIN_ROWS": "0", "MG_ROWCLASS_MAX_PASSES": "64", "MG_ROWCLASS_MIN_COVER": "0.05", "MG_MARCH_MIN_WG": "0", "MG_MARCH_MAX_LEN": "64"})
cells, nt, k1, tx, tym = eval(sys.argv[1]) if len(sys.argv) > 1 else ([36, 44, 10], 1024, 2, 0, 7)
os.environ.update({"MG_MARCH4_NT": str(nt), "MG_MARCH4_K1": str(k1), "MG_MARCH4_TILES_X": str(tx), "MG_MARCH4_TY_MAX": str(tym), "MG_DEBUG_FORMAT": "1"})
import torch
import multigrid_jl_amd as mg
A, mesh = mg.poisson_shifted(cells)
p = mg.getMGparam(np.float64, np.int64, 2, 8, 6, 1e-10, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
mg.MGsetup(A, mesh, p, 1)
h = mg.to_device(p)
print(h.four_stage_form(1))
n = A.shape[0]
rng = np.random.default_rng(1)
xh, bh = rng.standard_normal(n), rng.standard_normal(n)
x, bb = torch.from_numpy(xh).cuda(), torch.from_numpy(bh).cuda()
tp, rp = torch.full_like(x, np.nan), torch.full_like(x, np.nan)
h.four_stage_dev(1, bb, x, tp, rp)
t, xn = torch.zeros_like(x), torch.zeros_like(x)
h.sweep_residual_dev(1, bb, x, t, None, xn, True)
t2, r2 = torch.zeros_like(x), torch.zeros_like(x)
h.sweep_residual_dev(1, bb, xn, t2, r2)
n1, n2, n3 = [c + 1 for c in cells]
for name, u, v in (("tp", tp, t2), ("rp", rp, r2)):
    bad = torch.nonzero(u != v).flatten().cpu().numpy()
    print(name, "mismatches", len(bad), "of", n)
    if len(bad):
        zz, rem = bad // (n1 * n2), bad % (n1 * n2)
        yy, xx_ = rem // n1, rem % n1
        print(" x range", xx_.min(), xx_.max(), "distinct", np.unique(xx_)[:40])
        print(" y range", yy.min(), yy.max(), "distinct", np.unique(yy)[:60])
        print(" z range", zz.min(), zz.max(), "distinct", np.unique(zz)[:40])
