"""Diagnostic: sparse-factor coarse solve on the device, repeated (run on a GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import multigrid_jl_amd as mg
from multigrid_jl_amd import device as D

D.DENSE_COARSE_MAX = 0
for cells in ([24, 24, 24], [32, 32, 32], [40, 40, 40], [64, 64, 64]):
    A, mesh = mg.poisson_shifted(cells)
    p = mg.getMGparam(np.float64, np.int64, 2, 8, 4, 1e-10, "Jac", 0.8, 2, 1)
    mg.MGsetup(A, mesh, p, 1)
    Ac = p.As[-1]
    q = mg.getMGparam(np.float64, np.int64, 1, 8, 1, 1e-10, "Jac", 0.8, 2, 1)
    q.As, q.Ps, q.Rs, q.relaxPrecs, q.LU, q.Meshes, q.levels, q.nrhs = [Ac], [], [], [], p.LU, [p.Meshes[-1]], 1, 1
    rng = np.random.default_rng(5)
    B = rng.standard_normal(Ac.shape[0])
    Xo = p.LU.solve(B)
    for rep in range(4):
        t0 = time.perf_counter()
        X = mg.recursiveCycle(q, B.copy(), np.zeros_like(B), 1)
        dt = time.perf_counter() - t0
        bad = np.abs(X - Xo) > 1e-10 * np.abs(Xo).max()
        print(cells, "nc", Ac.shape[0], "L nnz", p.LU.L.nnz, "rep", rep, "err", np.abs(X - Xo).max() / np.abs(Xo).max(),
              "bad", int(bad.sum()), "zeros", int((X == 0).sum()), "nan", int(np.isnan(X).sum()), f"{dt*1e3:.2f} ms", flush=True)
    mg.clear_(q); mg.clear_(p)
