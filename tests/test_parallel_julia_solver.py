"""Device sparse-factor applier (mg_lu_* <-> applyLUsolve_FP64_INT64, deps/src/parLU.cpp) behind the mirror of
src/ParallelJuliaSolver: the reference's own test replayed with its thresholds, and value parity with the reference's
compiled binary where it was built (oracle/_ref/parLU.so), for the plain and the transposed solve."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from test_reference_parlu import REF, _nonsymmetric, _ref_lu_solve


@pytest.mark.gpu
def test_reference_test_replayed_real_symmetric(mg, built):
    """test/ParallelJuliaSolver/testParallelJuliaSolver.jl:11-57 (Float64 part): 20 x 23 cells, G'*m*G + shift, one and
    five right-hand sides, every back end value, residual < 1e-8."""
    PJ = mg.ParallelJuliaSolver
    rng = np.random.default_rng(0)
    Mr = mg.getRegularMesh([0.0, 1.0, 0.0, 1.0], [20, 23])
    G = mg.getNodalGradientMatrix(Mr)
    m = sp.diags(np.exp(rng.standard_normal(G.shape[0])))
    Ar = (G.T @ m @ G).tocsc()
    Ar = Ar + 1e-1 * np.abs(Ar).sum(axis=0).max() * sp.identity(Ar.shape[1])
    N = Ar.shape[1]
    Bs = (Ar @ rng.random(N), np.asfortranarray(Ar @ rng.random((N, 5))))
    for backend, cores in ((1, 1), (2, 1), (3, 2)):
        LU = PJ.getParallelJuliaSolver(np.float64, np.int64, numCores=cores, backend=backend)
        for B in Bs:
            x, LU = PJ.solveLinearSystem(Ar, B, LU)
            assert np.abs(Ar @ x - B).max() / np.abs(B).max() < 1e-8
        assert LU.nFac == 1 and LU.nSolve == 2
        PJ.clear_(LU)


@pytest.mark.gpu
@pytest.mark.parametrize("multi", [False, True])
@pytest.mark.parametrize("nrhs", [1, 5, 6])
def test_plain_and_transposed_solve_match_reference_binary(mg, built, nrhs, multi, monkeypatch):
    """x[q] = U\\(L\\b[p]) and, with doTranspose, x[p] = L'\\(U'\\b[q]) on an UNSYMMETRIC matrix, against scipy and (where
    built) against the reference's applyLUsolve_FP64_INT64; `multi` forces the chip-wide form of the factors."""
    PJ = mg.ParallelJuliaSolver
    monkeypatch.setenv("MG_LU_MULTI_MIN_ROWS", "0" if multi else "1000000000")
    monkeypatch.setenv("MG_LU_DENSE_TAIL_MIN", "4")
    A = _nonsymmetric(mg, [20, 23], 7)
    rng = np.random.default_rng(11)
    B = np.asfortranarray(rng.standard_normal((A.shape[0], nrhs))) if nrhs > 1 else rng.standard_normal(A.shape[0])
    LU = PJ.getParallelJuliaSolver(np.float64, np.int64, numCores=2, backend=3)
    X0, LU = PJ.solveLinearSystem(A, B, LU, 0)
    X1, LU = PJ.solveLinearSystem(A, B, LU, 1)
    X0b, LU = PJ.solveLinearSystem(A, B, LU, 0)              # back to the plain factors after a transposed solve
    assert LU.nFac == 1 and LU.nSolve == 3
    assert np.array_equal(X0, X0b)
    assert np.abs(A @ X0 - B).max() <= 1e-10 * np.abs(B).max()
    assert np.abs(A.T @ X1 - B).max() <= 1e-10 * np.abs(B).max()
    import scipy.sparse.linalg as spla
    lu = spla.splu(A, permc_spec="MMD_AT_PLUS_A")
    assert np.abs(X0 - lu.solve(B)).max() <= 1e-12 * np.abs(X0).max()
    assert np.abs(X1 - lu.solve(B, trans="T")).max() <= 1e-12 * np.abs(X1).max()
    if os.path.exists(REF):
        assert np.abs(X0 - _ref_lu_solve(lu, B, 0)).max() <= 1e-12 * np.abs(X0).max()
        assert np.abs(X1 - _ref_lu_solve(lu, B, 1)).max() <= 1e-12 * np.abs(X1).max()
    PJ.clear_(LU)


def test_mirror_rejects_what_the_device_cannot_do(mg, built):
    PJ = mg.ParallelJuliaSolver
    with pytest.raises(TypeError):
        PJ.getParallelJuliaSolver(np.float32, np.uint32)
    LU = PJ.getParallelJuliaSolver()
    with pytest.raises(RuntimeError):
        PJ.solve(np.zeros(3), np.zeros(3), LU)
