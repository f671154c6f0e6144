"""A/B timing of one fine-level kernel slot under option variants.  usage: python3 scripts/op_ab.py <cells> <kernel> opt=val,opt=val ..."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multigrid_jl_amd as mg
from multigrid_jl_amd import device as D
if os.environ.get("AB_LIB"):
    D._lib = D.load_library(os.path.join(os.path.dirname(D.LIB_PATH), os.environ["AB_LIB"]))
    print("library:", os.environ["AB_LIB"], flush=True)
cells = int(sys.argv[1])
kern = {"restrict": D.MG_K_RESTRICT, "prolong": D.MG_K_PROLONG, "smooth": D.MG_K_SMOOTH, "residual": D.MG_K_RESIDUAL,
        "pair": D.MG_K_SMOOTH_RESIDUAL}[sys.argv[2]]
A, mesh = mg.poisson_shifted([cells] * 3)
lv = {32: 3, 64: 4, 128: 5, 256: 6, 400: 5, 512: 7}.get(cells, 4)
p = mg.getMGparam(np.float64, np.int64, lv, 8, 10, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0, "FullWeighting")
mg.MGsetup(A, mesh, p, 1)
b = torch.from_numpy(mg.seeded_rhs(A, 1)).cuda()
for v in sys.argv[3:] or [""]:
    opts = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in v.split(",") if kv}
    h = D.DeviceHierarchy(p, 0, 1, options=opts)
    x = torch.zeros_like(b)
    h.solve_dev(b, x, 0.0, 3)
    out = []
    for lvl in (1, 2):
        try:
            ms, _ = h.time_op(lvl, kern, 30)
            out.append(f"L{lvl} {ms*1e3:7.1f} us")
        except Exception as e:
            out.append(f"L{lvl} n/a")
    import time
    x.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter(); h.solve_dev(b, x, 0.0, 20); torch.cuda.synchronize()
    print(f"{v or 'default':40s} {'  '.join(out)}   step {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms", flush=True)
    h.close()
