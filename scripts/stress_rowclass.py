"""Randomised stress of the row-class kernel variants against scipy (run on a GPU box).
usage: stress_rowclass.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
os.environ.update(MG_ROWCLASS_MIN_ROWS="0", MG_ROWCLASS_MAX_PASSES="64", MG_ROWCLASS_MIN_COVER="0.05",
                  MG_WINDOW_MIN_WG="0", MG_STAGE_MIN_LEN="0", MG_PAIR_MIN_ROWS="0", MG_MARCH_MIN_WG="0", MG_WINP_MIN_ROWS="0")
import multigrid_jl_amd as mg
from multigrid_jl_amd import device as D

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for case in range(ncases):
    dim = int(rng.choice([2, 3]))
    cells = [int(rng.integers(3, 90)) for _ in range(dim)] if dim == 2 else \
            [int(rng.integers(3, 70)), int(rng.integers(3, 40)), int(rng.integers(2, 14))]
    levels = int(rng.integers(2, 4))
    env = dict(MG_NO_TILE=str(int(rng.integers(0, 2))), MG_NO_WINDOW=str(int(rng.integers(0, 2))),
               MG_NO_IMPLICIT_FIRST=str(int(rng.integers(0, 2))), MG_NO_CLASS_D=str(int(rng.integers(0, 2))),
               MG_ROWCLASS_KEEP_SINGLETONS=str(int(rng.choice([0, 1024]))), MG_NO_PAIR=str(int(rng.integers(0, 2))),
               # round 2: the marching kernels (single- and two-stage, from-zero form), the tile kernel's per-lane walk and its
               # 256-row form, the staged prolongation, the restriction's second output
               MG_NO_MARCH=str(int(rng.integers(0, 2))), MG_MARCH_MAX_LEN=str(int(rng.choice([8, 64]))),
               MG_NO_MARCH2=str(int(rng.integers(0, 2))), MG_NO_MARCH2_ZERO=str(int(rng.integers(0, 2))),
               MG_NO_TILE_LANE=str(int(rng.integers(0, 2))), MG_NO_TILE_SMALL=str(int(rng.integers(0, 2))),
               MG_TILE_MIN_WG=str(int(rng.choice([0, 24]))), MG_NO_WINP=str(int(rng.integers(0, 2))),
               MG_NO_RESTRICT_SCALE=str(int(rng.integers(0, 2))))
    os.environ.update(env)
    relax = str(rng.choice(["Jac", "SPAI"]))
    A, mesh = mg.poisson_shifted(cells)
    p = mg.getMGparam(np.float64, np.int64, levels, 8, 4, 1e-10, relax, 0.8 if relax == "Jac" else 1.0, 2, 1)
    mg.MGsetup(A, mesh, p, 1)
    h = mg.to_device(p)
    for l in range(1, p.levels):
        Al, Pl, Rl, dl = p.As[l - 1], p.Ps[l - 1], p.Rs[l - 1], p.relaxPrecs[l - 1]
        xn, bn = rng.standard_normal(Al.shape[0]), rng.standard_normal(Al.shape[0])
        x, bb = torch.from_numpy(xn).cuda(), torch.from_numpy(bn).cuda()
        out = torch.zeros_like(x)
        h.fused_dev(l, D.MG_K_RESIDUAL, bb, x, out)
        w = bn - Al @ xn
        e1 = np.abs(out.cpu().numpy() - w).max() / np.abs(w).max()
        h.fused_dev(l, D.MG_K_SMOOTH, bb, x, out)
        w = xn + dl * (bn - Al @ xn)
        e2 = np.abs(out.cpu().numpy() - w).max() / np.abs(w).max()
        xc = rng.standard_normal(Pl.shape[1])
        e3 = np.abs(mg.SpMatMul(p, l, "P", xc, xn.copy(), 1.0, 1.0) - (xn + Pl @ xc)).max()
        e4 = np.abs(mg.SpMatMul(p, l, "R", xn, np.zeros(Rl.shape[0]), 1.0, 0.0) - Rl @ xn).max() / np.abs(Rl @ xn).max()
        e5 = 0.0
        try:   # the two-stage pass, where it serves the level
            t, r, xn2 = torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(x)
            nrm = h.sweep_residual_dev(l, bb, x, t, r, xn2, True)
            tw = xn + dl * (bn - Al @ xn)
            rw = bn - Al @ tw
            e5 = max(np.abs(t.cpu().numpy() - tw).max() / np.abs(tw).max(), np.abs(r.cpu().numpy() - rw).max() / np.abs(rw).max(),
                     np.abs(xn2.cpu().numpy() - (tw + dl * rw)).max() / np.abs(tw).max(), abs(nrm - np.linalg.norm(rw)) / np.linalg.norm(rw))
        except D.MGDeviceError:
            pass
        worst = max(worst, e1, e2, e3, e4, e5)
        if max(e1, e2, e3, e4, e5) > 1e-12:
            print("FAIL", cells, levels, env, l, e1, e2, e3, e4, e5, flush=True)
            sys.exit(1)
    b = mg.seeded_rhs(A, 1)
    xs = np.zeros_like(b)
    mg.solveMG(p, b, xs)
    rr = np.linalg.norm(A @ xs - b) / np.linalg.norm(b)
    if not np.isfinite(rr) or abs(rr - p.resvec[-1] / 1.0) > 1e-9 * max(1.0, p.resvec[0]):
        print("FAIL solve", cells, levels, env, rr, p.resvec[-1], flush=True)
        sys.exit(1)
    from oracle import mg_oracle as orc   # (checker) the same solve on the CPU restatement
    xo, hist = np.zeros_like(b), {}
    orc.solveMG(p, b, xo, False, hist)
    if np.abs(xs - xo).max() > 1e-10 * np.abs(xo).max() or np.abs(p.resvec - hist["resvec"]).max() > 1e-10 * hist["resvec"][0]:
        print("FAIL oracle", cells, levels, env, np.abs(xs - xo).max() / np.abs(xo).max(), flush=True)
        sys.exit(1)
    mg.clear_(p)
print(f"{ncases} cases ok, worst kernel error {worst:.2e}")
