"""-m gpu, BASELINE.json's full size (C2: 3-D Poisson 256^3 cells, 16 974 593 DoF, 6 levels, V(2,1) Jacobi).
The numpy oracle is too slow here; parity is checked through (a) the C/OpenMP oracle's residual history on two
steps, and size-independent properties of the path: (b) a checksum - A*1 equals the host's row sums, P*1 == 1 for
full weighting; (c) linearity of the cycle in b; (d) the residual the library reports is the true residual."""
import numpy as np
import pytest

from oracle import c_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2(mg, built):
    A, mesh = mg.poisson_shifted([256, 256, 256])
    p = mg.getMGparam(np.float64, np.int64, 6, 8, 2, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p)
    b = mg.seeded_rhs(A)
    yield A, p, b
    mg.clear_(p)


def test_c2_residual_history_matches_c_oracle(mg, c2):
    A, p, b = c2
    x = np.zeros_like(b)
    mg.solveMG(p, b, x)
    co = c_oracle.COracle(p, 1)
    xo = np.zeros_like(b)
    it, rv = co.solveMG(b, xo, 0.0, 2, c_oracle.max_threads())
    assert it == 2 and np.abs(rv - p.resvec).max() / rv[0] < 1e-10
    assert np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()
    # (d) the reported residual norm is the true one
    assert abs(np.linalg.norm(b - A @ x) - p.resvec[-1]) <= 1e-10 * p.resvec[0]


def test_c2_checksums(mg, c2):
    A, p, b = c2
    ones = np.ones(A.shape[0])
    y = np.zeros_like(ones)
    mg.SpMatMul(p, 1, "A", ones, y)
    rs = np.asarray(A.sum(axis=1)).ravel()
    assert np.abs(y - rs).max() <= 1e-12 * np.abs(rs).max()
    yc = np.zeros(p.Ps[0].shape[0])
    mg.SpMatMul(p, 1, "P", np.ones(p.Ps[0].shape[1]), yc)
    assert np.abs(yc - 1.0).max() < 1e-14                       # full-weighting interpolation reproduces constants
    # a checksum of checksums: 1'(A x) == (A'1)'x for a random x (A symmetric here)
    rng = np.random.default_rng(3)
    xr = rng.standard_normal(A.shape[0])
    y2 = np.zeros_like(xr)
    mg.SpMatMul(p, 1, "A", xr, y2)
    assert abs(y2.sum() - rs @ xr) <= 1e-9 * np.abs(rs).max() * np.sqrt(A.shape[0])


def test_c2_cycle_is_linear_in_b(mg, c2):
    A, p, b = c2
    x1 = np.zeros_like(b)
    x2 = np.zeros_like(b)
    mg.recursiveCycle(p, b, x1, 1)
    mg.recursiveCycle(p, -3.0 * b, x2, 1)
    assert np.abs(x2 + 3.0 * x1).max() <= 1e-13 * np.abs(x1).max()
