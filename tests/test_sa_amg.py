"""SA-AMG (reference src/Multigrid/SA-AMG.jl): host setup vs the oracle's literal-loop restatement (CPU),
the reference's known-answer thresholds with seeded inputs (CPU), and device parity (-m gpu)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import mg_oracle as orc


def _divsiggrad(mg, n, shift, seed=42):
    from multigrid_jl_amd.operators import getRegularMesh, getNodalDivSigGradMatrix, entrynorm1
    rng = np.random.default_rng(seed)
    mesh = getRegularMesh([0, 1] * len(n), n)
    m = np.exp(rng.standard_normal(mesh.nc))                 # exp.(randn) (testSAforDivSigGrad.jl:12,98)
    A = getNodalDivSigGradMatrix(mesh, m)
    A = (A + shift * entrynorm1(A) * sp.identity(A.shape[0])).tocsr()      # norm(Ar,1): entry-wise (l.14,100)
    A.sort_indices()
    return A


def _rhs(A, nrhs, seed=1):
    rng = np.random.default_rng(seed)
    b = A @ rng.random((A.shape[0], nrhs))
    return np.asfortranarray(b / np.linalg.norm(b))


@pytest.mark.parametrize("n,theta", [([12, 12], 0.4), ([8, 7, 6], 0.4), ([14, 9], 0.25), ([6, 6, 5], 0.6)])
def test_sa_setup_matches_loop_restatement(mg, built, n, theta):
    A = _divsiggrad(mg, n, 1e-8, seed=3)
    S = mg.getStrengthMatrix(A, theta)
    S2 = orc.getStrengthMatrix_loops(A, theta)
    assert np.array_equal(S.toarray(), S2)
    ag = mg.neighborhoodAggregationNew(S)
    assert np.array_equal(ag, orc.neighborhoodAggregationNew_loops(S2))
    assert np.array_equal(mg.aggrArray2P(ag).toarray(), orc.aggrArray2P_loops(ag))
    p = mg.getMGparam(np.float64, np.int64, 4, 2, 5, 1e-4, "SPAI", 1.0, 1, 1, "V", "Julia", theta)
    mg.SA_AMGsetup(A, p)
    h = orc.SA_AMGsetup_dense(A.toarray(), 4, "SPAI", 1.0, theta, 1, 1)
    assert p.levels == len(p.As) == len(h.As) and len(p.relaxPrecs) == len(h.relaxPrecs) == len(p.As) - 1
    for a, b in zip(p.As, h.dense_As):
        assert np.allclose(a.toarray(), b, rtol=0, atol=1e-13 * abs(b).max())
    for a, b in zip(p.Ps, h.dense_Ps):
        assert np.allclose(a.toarray(), b, rtol=0, atol=1e-14)
    for a, b in zip(p.Rs, h.dense_Rs):
        assert np.allclose(a.toarray(), b, rtol=0, atol=1e-14)


def test_small_problem_is_not_coarsened(mg, built):
    """n <= 100 -> identity aggregation -> 'Stopped Coarsening' (SA-AMG.jl:79-81,35-42)."""
    A = _divsiggrad(mg, [6, 6], 1e-8)
    p = mg.getMGparam(levels=3)
    mg.SA_AMGsetup(A, p)
    assert p.levels == 1 and len(p.As) == 1 and len(p.Ps) == 0


@pytest.mark.parametrize("n,shift,thr", [([50, 50], 1e-8, 0.01), ([32, 32, 16], 1e-6, 0.005)])
def test_reference_thresholds_sa(mg, built, n, shift, thr):
    """testSAforDivSigGrad.jl:9-38 (2-D, < 0.01) and l.96-112 (3-D, < 0.005): 3 levels, SPAI w=1, V(1,1),
    nrhs=3, maxIter 5, tol 1e-4."""
    A = _divsiggrad(mg, n, shift)
    p = mg.getMGparam(np.float64, np.int64, 3, 2, 5, 1e-4, "SPAI", 1.0, 1, 1, "V", "Julia")
    mg.SA_AMGsetup(A, p, True, 3)
    b = _rhs(A, 3)
    x = np.zeros_like(b)
    orc.solveMG(p, b, x)
    assert np.linalg.norm(A @ x - b) < thr


@pytest.mark.gpu
@pytest.mark.parametrize("n,shift,nrhs,cyc", [([50, 50], 1e-8, 3, "V"), ([32, 32, 16], 1e-6, 3, "V"),
                                              ([24, 24, 24], 1e-6, 1, "W"), ([20, 20, 20], 1e-6, 16, "F")])
def test_sa_amg_device_parity(mg, built, n, shift, nrhs, cyc):
    A = _divsiggrad(mg, n, shift)
    p = mg.getMGparam(np.float64, np.int64, 3, 2, 5, 1e-8, "SPAI", 1.0, 1, 1, cyc, "Julia")
    mg.SA_AMGsetup(A, p, True, nrhs)
    b = _rhs(A, nrhs)
    b = b[:, 0].copy() if nrhs == 1 else b
    x = np.zeros_like(b)
    xo = np.zeros_like(b)
    _, _, it = mg.solveMG(p, b, x)
    hist = {}
    _, _, ito = orc.solveMG(p, b, xo, False, hist)
    assert it == ito
    assert np.abs(p.resvec - hist["resvec"]).max() / hist["resvec"][0] < 1e-10
    assert np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()
    mg.clear_(p)


@pytest.mark.gpu
def test_anisotropic_c3_operator_small(mg, built):
    """BASELINE.json configs[2] at a size the oracle finishes in seconds: anisotropic diffusion as general CSR."""
    A, _ = mg.anisotropic_divsiggrad([24, 24, 24])
    p = mg.getMGparam(np.float64, np.int64, 8, 8, 6, 1e-10, "SPAI", 1.0, 1, 1, "V", "Julia", 0.4)
    mg.SA_AMGsetup(A, p)
    assert p.As[-1].shape[0] <= 100
    b = mg.seeded_rhs(A)
    x = np.zeros_like(b)
    xo = np.zeros_like(b)
    mg.solveMG(p, b, x)
    hist = {}
    orc.solveMG(p, b, xo, False, hist)
    assert np.abs(p.resvec - hist["resvec"]).max() / hist["resvec"][0] < 1e-10
    mg.clear_(p)
