#!/usr/bin/env python3
"""Generate the committed golden fixtures from the CPU oracle (oracle/mg_oracle.py).

The reference holds no golden vectors and cannot be executed here (no Julia), so these
are ORACLE outputs on seeded inputs - they pin the oracle against silent drift and give the GPU tests
(-m gpu) a comparand that does not depend on the oracle code at run time.
Re-run:  python tests/golden/make_golden.py      (rewrites tests/golden/*.npz)
Each case stores: the solver parameters, b, the residual history, x after the first and the last cycle,
and per-level fingerprints of the hierarchy the host setup produced (shape, nnz, sum|a_ij|).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import multigrid_jl_amd as mg            # host setup (CPU)
from oracle import mg_oracle as orc      # the checker

# name: (cells, levels, relaxType, omega, pre, post, cycle, nrhs, maxIter)
CASES = {
    "gmg_poisson9_v21_jac": ([8, 8, 8], 2, "Jac", 0.8, 2, 1, "V", 1, 10),
    "gmg_poisson17_v21_jac": ([16, 16, 16], 3, "Jac", 0.8, 2, 1, "V", 1, 10),
    "gmg_poisson33_v21_jac": ([32, 32, 32], 3, "Jac", 0.8, 2, 1, "V", 1, 10),          # BASELINE.json configs[0] (C1)
    "gmg_poisson17_v11_spai": ([16, 16, 16], 3, "SPAI", 1.0, 1, 1, "V", 1, 10),
    "gmg_poisson17_w_jac_nrhs2": ([16, 16, 16], 4, "Jac", 0.8, 2, 1, "W", 2, 6),
    "gmg_poisson17_f_jac_nrhs16": ([16, 16, 16], 4, "Jac", 0.8, 2, 1, "F", 16, 6),
    "gmg_poisson33x33x17_v11_nrhs2": ([32, 32, 16], 4, "Jac", 0.75, 1, 1, "V", 2, 5),  # testGMGRAPforPoisson.jl:60-78 shape
    "gmg_poisson2d_129_v11_jac": ([128, 128], 4, "Jac", 0.8, 1, 1, "V", 1, 5),         # testGMG.jl:21-37 shape (shifted)
    "gmg_poisson_even_15x12x9": ([15, 12, 9], 3, "Jac", 0.8, 2, 1, "V", 1, 8),         # even node counts: algebraic branch
}


# SA-AMG cases ("a 17^3 SA-AMG log-normal sigma case"): name -> (cells, shift, levels, nrhs, cycle, maxIter)
SA_CASES = {
    "sa_divsiggrad17_v11_spai_nrhs3": ([16, 16, 16], 1e-6, 3, 3, "V", 5),        # testSAforDivSigGrad.jl:96-112 shape
    "sa_divsiggrad2d_51_v11_spai": ([50, 50], 1e-8, 3, 1, "V", 5),               # testSAforDivSigGrad.jl:9-38 shape
}


def build_sa_case(name):
    import scipy.sparse as sp
    from multigrid_jl_amd.operators import getRegularMesh, getNodalDivSigGradMatrix, entrynorm1
    cells, shift, levels, nrhs, cyc, maxit = SA_CASES[name]
    rng = np.random.default_rng(42)
    mesh = getRegularMesh([0, 1] * len(cells), cells)
    m = np.exp(rng.standard_normal(mesh.nc))
    A = getNodalDivSigGradMatrix(mesh, m)
    A = (A + shift * entrynorm1(A) * sp.identity(A.shape[0])).tocsr()
    A.sort_indices()
    p = mg.getMGparam(np.float64, np.int64, levels, 2, maxit, 1e-10, "SPAI", 1.0, 1, 1, cyc, "Julia")
    mg.SA_AMGsetup(A, p, True, nrhs)
    b = mg.seeded_rhs(A, nrhs)
    return A, p, b


def fingerprint(param):
    fp = []
    for name in ("As", "Ps", "Rs"):
        for M in getattr(param, name):
            fp.append([M.shape[0], M.shape[1], M.nnz, float(abs(M).sum())])
    return np.array(fp)


def build_case(name):
    cells, levels, rt, om, pre, post, cyc, nrhs, maxit = CASES[name]
    A, mesh = mg.poisson_shifted(cells)
    p = mg.getMGparam(np.float64, np.int64, levels, 8, maxit, 1e-10, rt, om, pre, post, cyc, "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p, nrhs)
    b = mg.seeded_rhs(A, nrhs)
    return A, p, b


def main():
    for name in CASES:
        A, p, b = build_case(name)
        x = np.zeros_like(b)
        hist = {}
        _, _, it = orc.solveMG(p, b, x, False, hist)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), b=b, resvec=hist["resvec"], x_first=hist["xs"][0],
                            x_last=hist["xs"][-1], iters=it, fingerprint=fingerprint(p),
                            relaxPrec0=p.relaxPrecs[0])
        print(f"{name}: {it} cycles, relres {hist['resvec'][-1] / hist['resvec'][0]:.3e}")
    for name in SA_CASES:
        A, p, b = build_sa_case(name)
        x = np.zeros_like(b)
        hist = {}
        _, _, it = orc.solveMG(p, b, x, False, hist)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), b=b, resvec=hist["resvec"], x_first=hist["xs"][0],
                            x_last=hist["xs"][-1], iters=it, fingerprint=fingerprint(p), relaxPrec0=p.relaxPrecs[0])
        print(f"{name}: {it} cycles, relres {hist['resvec'][-1] / hist['resvec'][0]:.3e}")


if __name__ == "__main__":
    main()
