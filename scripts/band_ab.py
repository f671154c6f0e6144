"""A/B of band-form geometries on the C2 operator with row classes off: ms per solveMG step and per pair launch.
usage: python scripts/band_ab.py K1:tiles_x[:lockstep] ..."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multigrid_jl_amd as mg
from multigrid_jl_amd import device as D
if os.environ.get("AB_LIB"):
    D._lib = D.load_library(os.path.join(os.path.dirname(D.LIB_PATH), os.environ["AB_LIB"]))
    print("library:", os.environ["AB_LIB"], flush=True)

cells = int(os.environ.get("AB_CELLS", "256"))
A, mesh = mg.poisson_shifted([cells] * 3)
p = mg.getMGparam(np.float64, np.int64, 6, 8, 20, 1e-10, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
mg.MGsetup(A, mesh, p, 1)
b = torch.from_numpy(mg.seeded_rhs(A, 1)).cuda()
for spec in sys.argv[1:]:
    f = spec.split(":")
    os.environ["MG_MARCH3_K1"] = f[0]
    os.environ["MG_MARCH3_TILES_X"] = f[1]
    if len(f) > 2:
        os.environ["MG_NO_MARCH3_LOCKSTEP"] = "1" if f[2] == "0" else "0"
    opts = {"no_rowclass": 1}
    if f[0] == "csr":
        opts["no_band"] = 1
        os.environ["MG_MARCH3_K1"] = "0"
    h = mg.device.DeviceHierarchy(p, device_id=0, nrhs=1, options=opts)
    x = torch.zeros_like(b)
    h.solve_dev(b, x, 0.0, 3)
    best = 1e9
    for rep in range(3):
        x.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h.solve_dev(b, x, 0.0, 20)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    h.profile_reset(); h.profile_enable(True); x.zero_(); h.solve_dev(b, x, 0.0, 10); h.profile_enable(False)
    pr = h.profile()
    form, geo = h.sweep_residual_form(1)
    line = {k[1]: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in pr.items() if k[0] == 1}
    print(spec, "form", form, "geo", geo[:6], "step %.4f ms" % best, line, flush=True)
    h.close()
