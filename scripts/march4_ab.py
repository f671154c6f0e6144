"""A/B timing of the four-stage pass on the C2 fine level over workgroup shapes / tile geometries (mg_set_option march4_nt,
march4_tiles_x) and attribution builds (AB_LIB=libmgvcycle_<name>.so from `make variant DEFS=-DMG_M4_EXP=k`).
usage: python3 scripts/march4_ab.py [cells] [nt:tiles_x ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multigrid_jl_amd as mg
from multigrid_jl_amd import device as D

if os.environ.get("AB_LIB"):
    D._lib = D.load_library(os.path.join(os.path.dirname(D.LIB_PATH), os.environ["AB_LIB"]))
    print("library:", os.environ["AB_LIB"], flush=True)
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
variants = sys.argv[2:] or ["1024:0", "768:0", "512:0"]
A, mesh = mg.poisson_shifted([cells] * 3)
lv = {32: 3, 64: 4, 128: 5, 256: 6, 400: 5, 512: 7}.get(cells, 4)
p = mg.getMGparam(np.float64, np.int64, lv, 8, 10, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0, "FullWeighting")
mg.MGsetup(A, mesh, p, 1)
b = torch.from_numpy(mg.seeded_rhs(A, 1)).cuda()
for v in variants:
    if v == "no4":
        h = D.DeviceHierarchy(p, 0, 1, options={"no_march4": 1})
        x = torch.zeros_like(b)
        h.solve_dev(b, x, 0.0, 3)
        ms2, _ = h.time_op(1, D.MG_K_SMOOTH_RESIDUAL, 30)
        x.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter(); h.solve_dev(b, x, 0.0, 20); torch.cuda.synchronize()
        print(f"no4      two passes per step (two-stage pass {ms2*1e3:6.1f} us)  step {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms", flush=True)
        h.close()
        continue
    f = v.split(":")
    opts = {"march4_nt": int(f[0]), "march4_tiles_x": int(f[1]) if len(f) > 1 else 0}
    if len(f) > 2:
        opts["march4_k1"] = int(f[2])
    h = D.DeviceHierarchy(p, 0, 1, options=opts)
    x = torch.zeros_like(b)
    it, res = h.solve_dev(b, x, 0.0, 3)
    ok, geo = h.four_stage_form(1)
    try:
        ms, _ = h.time_op(1, D.MG_K_FOUR_STAGE, 30)
        ms2, _ = h.time_op(1, D.MG_K_SMOOTH_RESIDUAL, 30)
        torch.cuda.synchronize()
        x.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter(); h.solve_dev(b, x, 0.0, 20); torch.cuda.synchronize()
        step = (time.perf_counter() - t0) / 20 * 1e3
        print(f"{v:8s} geo {geo}  four-stage {ms*1e3:7.1f} us  (two-stage pass {ms2*1e3:6.1f} us)  step {step:.4f} ms  relres {res[-1]/res[0]:.3e}", flush=True)
    except Exception as e:
        print(f"{v:8s} ok {ok}: {e}", flush=True)
    h.close()
