"""A/B timing of the fused sweep + residual pass on the C2 fine level: the 1-D chunk form (march2) against the 2-D tile
form (march3) over tile geometries (mg_set_option march3_tiles_x / march3_k1).  usage: python3 scripts/march3_ab.py [cells]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multigrid_jl_amd as mg
from multigrid_jl_amd import device as D

if os.environ.get("AB_LIB"):
    D._lib = D.load_library(os.path.join(os.path.dirname(D.LIB_PATH), os.environ["AB_LIB"]))
    print("library:", os.environ["AB_LIB"], flush=True)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
variants = sys.argv[2:] or ["m2", "3:0:0", "3:2:2:1024:0", "3:2:2:1024:2", "3:4:2:768:0", "3:4:2:768:2", "3:3:2:768:0", "3:3:2:768:2", "3:4:1:768:0", "3:4:3:768:0", "3:4:2:768:2:1"]
A, mesh = mg.poisson_shifted([cells] * 3)
lv = {32: 3, 64: 4, 128: 5, 256: 6, 400: 5, 512: 7}.get(cells, 4)
p = mg.getMGparam(np.float64, np.int64, lv, 8, 10, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0, "FullWeighting")
mg.MGsetup(A, mesh, p, 1)
b = torch.from_numpy(mg.seeded_rhs(A, 1)).cuda()
for v in variants:
    if v == "m2":
        opts = {"no_march3": 1}
    else:
        f = v.split(":")          # 3:K1:tiles_x[:NT[:lockstep 0/1/2 (2 = forced)[:no_dead_t]]]
        opts = {"march3_k1": int(f[1]), "march3_tiles_x": int(f[2])}
        if len(f) > 3:
            opts["march3_nt"] = int(f[3])
        if len(f) > 4:
            opts["no_march3_lockstep"] = 1 if f[4] == "0" else 0
            opts["march3_lockstep_force"] = 1 if f[4] == "2" else 0
        if len(f) > 5:
            opts["no_dead_t"] = int(f[5])
    h = D.DeviceHierarchy(p, 0, 1, options=opts)
    x = torch.zeros_like(b)
    it, res = h.solve_dev(b, x, 0.0, 3)
    form, geo = h.sweep_residual_form(1)
    try:
        ms, _ = h.time_op(1, D.MG_K_SMOOTH_RESIDUAL, 30)
        torch.cuda.synchronize()
        import time
        x.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter(); h.solve_dev(b, x, 0.0, 20); torch.cuda.synchronize()
        step = (time.perf_counter() - t0) / 20 * 1e3
        print(f"{v:8s} form {form} geo {geo}  pass {ms*1e3:7.1f} us   step {step:.4f} ms  relres {res[-1]/res[0]:.3e}", flush=True)
    except Exception as e:
        print(f"{v:8s} form {form}: {e}", flush=True)
    h.close()
