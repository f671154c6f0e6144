"""Pin of the coarsest solve `z = param.LU \\ b` (MGcycle.jl:177) against the REFERENCE'S OWN compiled code.

oracle/_ref/parLU.so is the reference's deps/src/parLU.cpp built in place by oracle/Makefile (never copied
into the repo).  Its applyLUsolve_FP64_INT64 (parLU.cpp:52-63,120-190) is the "Julia factors, native applies"
path of src/ParallelJuliaSolver: x[q] = U \\ (L \\ b[p]) on CSR factors with 1-based Int64 indices, L's diagonal
last and U's diagonal first in every row.  The same factors (here from SuperLU instead of UMFPACK) fed to the
reference binary must reproduce the oracle's solveCoarsest and the dense inverse the device applies.
"""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import mg_oracle as orc

REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "parLU.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/parLU.so not built (reference tree absent)")

_i64p = C.POINTER(C.c_longlong)
_f64p = C.POINTER(C.c_double)


def _ref_lu_solve(lu, B, doTranspose=0):
    """Call the reference's applyLUsolve_FP64_INT64 with SuperLU's factors in the layout setupLUFactor produces
    (parallelJuliaSolver.jl:113-148, convertCSC2MyCSR)."""
    lib = C.CDLL(REF)
    f = lib.applyLUsolve_FP64_INT64
    f.restype = None
    f.argtypes = [_i64p, _f64p, _i64p, _i64p, _f64p, _i64p, _i64p, _i64p, _i64p, _i64p, _f64p, _f64p,
                  C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong, C.c_longlong]
    n = lu.shape[0]
    L = sp.csr_matrix(lu.L)
    U = sp.csr_matrix(lu.U)
    L.sort_indices()
    U.sort_indices()                                            # lower: diagonal last; upper: diagonal first
    p = (np.argsort(lu.perm_r) + 1).astype(np.int64)            # A[p, q] = L U  (1-based)
    q = (np.argsort(lu.perm_c) + 1).astype(np.int64)
    arr = lambda a, t: np.ascontiguousarray(a, dtype=t)
    Lp, Lc, Lv = arr(L.indptr + 1, np.int64), arr(L.indices + 1, np.int64), arr(L.data, np.float64)
    Up, Uc, Uv = arr(U.indptr + 1, np.int64), arr(U.indices + 1, np.int64), arr(U.data, np.float64)
    B = np.asfortranarray(B, dtype=np.float64)
    nrhs = 1 if B.ndim == 1 else B.shape[1]
    # parLU.cpp:143-145 advances the RHS offset by n[rhsIdx] (not n[LUIdx]): with one factorisation the array
    # must therefore hold n at every index < num_rhs
    nn = np.full(nrhs + 1, n, dtype=np.int64)
    nnz = np.full(nrhs + 1, max(L.nnz, U.nnz), dtype=np.int64)
    X = np.zeros_like(B, order="F")
    Bw = B.copy(order="F")                                      # the reference uses b as workspace
    P = lambda a: a.ctypes.data_as(_i64p)
    F = lambda a: a.ctypes.data_as(_f64p)
    f(P(Lp), F(Lv), P(Lc), P(Up), F(Uv), P(Uc), P(p), P(q), P(nn), P(nnz), F(X), F(Bw), 1, nrhs, 1, 1, int(doTranspose))
    return X


def _nonsymmetric(mg, n, seed):
    """G'*m*G of testParallelJuliaSolver.jl:13-21 plus an unsymmetric sparse perturbation (the transposed solve is
    only a test when A != A')."""
    rng = np.random.default_rng(seed)
    Mr = mg.getRegularMesh([0.0, 1.0, 0.0, 1.0], n)
    G = mg.getNodalGradientMatrix(Mr)
    m = sp.diags(np.exp(rng.standard_normal(G.shape[0])))
    Ar = (G.T @ m @ G).tocsc()
    Ar = Ar + 1e-1 * np.abs(Ar).sum(axis=0).max() * sp.identity(Ar.shape[1])
    E = sp.random(Ar.shape[0], Ar.shape[1], density=3.0 / Ar.shape[0], random_state=seed, format="csc")
    return (Ar + 0.05 * np.abs(Ar).max() * E).tocsc()


@pytest.mark.parametrize("nrhs", [1, 5])
def test_transposed_solve_of_the_reference_binary(mg, built, nrhs):
    """doTranspose = 1 (parLU.cpp:194-260): x[p] = L' \\ (U' \\ b[q]) solves A' x = b.  Pins the layout the device
    applier's transposed factors are derived from (mg_lu_solve_FP64)."""
    import scipy.sparse.linalg as spla
    A = _nonsymmetric(mg, [20, 23], 7)
    lu = spla.splu(A, permc_spec="MMD_AT_PLUS_A")
    rng = np.random.default_rng(11)
    B = rng.standard_normal((A.shape[0], nrhs)) if nrhs > 1 else rng.standard_normal(A.shape[0])
    X0 = _ref_lu_solve(lu, B, 0)
    X1 = _ref_lu_solve(lu, B, 1)
    assert np.abs(A @ X0 - B).max() <= 1e-10 * np.abs(B).max()
    assert np.abs(A.T @ X1 - B).max() <= 1e-10 * np.abs(B).max()
    assert np.abs(X1 - lu.solve(B, trans="T")).max() <= 1e-12 * np.abs(X1).max()


@pytest.mark.parametrize("cells,levels,nrhs", [([16, 16, 16], 3, 1), ([32, 32], 4, 3)])
def test_coarse_solve_matches_reference_binary(mg, built, cells, levels, nrhs):
    A, mesh = mg.poisson_shifted(cells)
    p = mg.getMGparam(np.float64, np.int64, levels, 8, 5, 1e-8, "Jac", 0.8, 2, 1)
    mg.MGsetup(A, mesh, p, nrhs)
    Ac = p.As[-1]
    rng = np.random.default_rng(3)
    B = rng.standard_normal((Ac.shape[0], nrhs)) if nrhs > 1 else rng.standard_normal(Ac.shape[0])
    Xref = _ref_lu_solve(p.LU, B)                               # the reference's compiled triangular solves
    Xo = orc.solveCoarsest(p, B, np.zeros_like(B))              # oracle (MGcycle.jl:177)
    assert np.abs(Xref - Xo).max() <= 1e-12 * np.abs(Xo).max()
    Ainv = p.LU.solve(np.eye(Ac.shape[0]))                      # what device.py uploads (mg_set_coarse_dense_inverse)
    assert np.abs(Ainv @ B - Xref).max() <= 1e-11 * np.abs(Xref).max()
    assert np.abs(Ac @ Xref - B).max() <= 1e-10 * np.abs(B).max()
