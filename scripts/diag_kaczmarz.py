import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import numpy as np, scipy.sparse as sp
import multigrid_jl_amd as mg
from oracle import mg_oracle as orc
from test_kaczmarz import _problem
from test_reference_parrelax import ref_apply
A, mesh, b = _problem(mg, [64, 64], 2, 3)
hk = mg.getHybridKaczmarz(np.float64, np.int64, A, mesh, [4, 4], mg.getNodalIndicesOfCell, 0.8, 4, 5)
for seq in (True, False):
    hk.sequential = seq
    prec = mg.getHybridKaczmarzPrecond(hk, A, 2)
    z = prec(b).copy()
    print("seq", seq, "one application: ||A z - b|| =", np.linalg.norm(A @ z - b))
    out = orc.FGMRES_relaxation(lambda z: A @ z, b.copy(), np.zeros_like(b), 5, lambda r: prec(r).copy(), 1e-5)
    x = out[0] if isinstance(out, tuple) else out
    print("   fgmres res", np.linalg.norm(A @ x - b))
for nc in (1, 4):
    def prec_ref(r):
        z = np.zeros_like(r, order="F")
        return ref_apply(A, hk.ArrIdxs, z, np.asfortranarray(r), hk.invDiag, 5, nc)
    z = prec_ref(b)
    print("ref cores", nc, "one application:", np.linalg.norm(A @ z - b))
    out = orc.FGMRES_relaxation(lambda z: A @ z, b.copy(), np.zeros_like(b), 5, prec_ref, 1e-5)
    x = out[0] if isinstance(out, tuple) else out
    print("   fgmres res", np.linalg.norm(A @ x - b))
