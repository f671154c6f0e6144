"""solveCG_MG (reference SolveFuncs.jl:104-116; KrylovMethods.cg is un-vendored, restated in the oracle).
CPU: the oracle's PCG against scipy's CG algebra and the reference's threshold; GPU: device PCG vs oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import mg_oracle as orc


def _sa_problem(mg, n, shift, levels=3):
    from tests_helpers import divsiggrad
    A = divsiggrad(n, shift)
    p = mg.getMGparam(np.float64, np.int64, levels, 2, 5, 1e-4, "SPAI", 1.0, 1, 1, "V", "Julia")
    mg.SA_AMGsetup(A, p)
    rng = np.random.default_rng(1)
    b = A @ rng.random(A.shape[0])
    return A, p, b / np.linalg.norm(b)


def test_oracle_cg_is_pcg(mg, built):
    """With M = I the restated cg must follow textbook CG: residuals orthogonal, monotone A-norm error."""
    A, mesh = mg.poisson_shifted([6, 6, 6])
    rng = np.random.default_rng(0)
    xs = rng.standard_normal(A.shape[0])
    b = A @ xs
    x, flag, rn, it, rv = orc.cg(lambda v: A @ v, b, tol=1e-12, maxIter=400, M=None, x=np.zeros_like(b))
    assert flag == 0 and np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-11
    assert orc.cg(lambda v: A @ v, np.zeros(5), M=None)[1] == -9


def test_reference_threshold_cg_sa(mg, built):
    """testSAforDivSigGrad.jl:41-44: CG preconditioned with SA-AMG (3 levels, SPAI, V(1,1), tol 1e-4, 5 its)
    -> ||Ax-b|| < 0.005 (one right-hand side here; the reference uses blockCG on 3)."""
    A, p, b = _sa_problem(mg, [50, 50], 1e-8)
    x, flag, it, rv = orc.solveCG_MG(p, b, np.zeros_like(b))
    assert np.linalg.norm(A @ x - b) < 0.005


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sa2d", "gmg3d"])
def test_device_pcg_matches_oracle(mg, built, kind):
    if kind == "sa2d":
        A, p, b = _sa_problem(mg, [50, 50], 1e-8)
        p.relativeTol, p.maxOuterIter = 1e-9, 40
    else:
        A, mesh = mg.poisson_shifted([24, 24, 24])
        p = mg.getMGparam(np.float64, np.int64, 3, 8, 12, 1e-9, "Jac", 0.8, 2, 2, "V", "NoMUMPS", 0.5, 0.0)
        mg.MGsetup(A, mesh, p)
        b = mg.seeded_rhs(A)
    x = np.zeros_like(b)
    xin = x
    x, _, it = mg.solveCG_MG(A, p, b, x)
    assert x is xin
    xo, flag, ito, rvo = orc.solveCG_MG(p, b, np.zeros_like(b))
    assert it == ito and p.flag == flag == 0
    assert np.abs(p.resvec - rvo).max() / rvo[0] < 1e-9
    assert np.abs(x - xo).max() <= 1e-9 * np.abs(xo).max()
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-8
    mg.clear_(p)
