"""transposeHierarchy (MGsetup.jl:274-318) on the resident hierarchy: the operators are transposed in HBM (mg_transpose_hierarchy)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import mg_oracle as orc

RES_TOL = 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("cells,levels,relax", [([16, 12, 10], 3, "Jac"), ([24, 24], 3, "SPAI"), ([20, 20, 20], 2, "Jac")])
def test_transpose_hierarchy_on_the_device(mg, built, cells, levels, relax):
    """A NON-symmetric operator (Laplacian + a one-sided convection term): solve A x = b on the device, then transposeHierarchy -
    As[l] <- As[l]', Ps[l] <- Rs[l]' (the reference's second assignment leaves Rs[l] as it was), the dense coarsest inverse transposed -
    WITHOUT dropping the resident hierarchy, and solve A' x = b: iterate and residual history against the oracle on the transposed
    host hierarchy (1e-10); the values the device holds after the transpose equal the host transposes bit for bit; a second transpose
    (As back, Ps[l] = Rs[l]' once more - the reference's literal assignments) against the oracle as well."""
    from multigrid_jl_amd import device as D
    A, mesh = mg.poisson_shifted(cells)
    n = A.shape[0]
    N = sp.diags([np.linspace(0.05, 0.3, n - 1)], [1], format="csr") * abs(A[0, 0]) * 0.2
    A = (A + N).tocsr()
    A.sort_indices()
    p = mg.getMGparam(np.float64, np.int64, levels, 8, 6, 1e-10, relax, 0.8 if relax == "Jac" else 1.0, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p, 1)
    b = mg.seeded_rhs(A, 1)
    x = np.zeros_like(b)
    mg.solveMG(p, b, x)
    hist0 = np.asarray(p.resvec).copy()
    hist = {}
    xo = np.zeros_like(b)
    orc.solveMG(p, b, xo, False, hist)
    assert np.abs(hist0 - hist["resvec"]).max() / hist["resvec"][0] < RES_TOL
    handle = p.device
    assert handle is not None
    mg.transposeHierarchy(p)
    assert p.device is handle                     # the resident hierarchy was transposed, not dropped
    for l in range(1, levels + 1):
        got = handle.get_values(l, D.MG_OP_A)
        assert np.array_equal(got, p.As[l - 1].data)
        if l < levels:
            assert np.array_equal(handle.get_values(l, D.MG_OP_P), p.Ps[l - 1].data)
            assert np.array_equal(handle.get_values(l, D.MG_OP_R), p.Rs[l - 1].data)
    x = np.zeros_like(b)
    mg.solveMG(p, b, x)
    hist_t = {}
    xo = np.zeros_like(b)
    orc.solveMG(p, b, xo, False, hist_t)
    assert np.abs(p.resvec - hist_t["resvec"]).max() / hist_t["resvec"][0] < RES_TOL
    assert np.abs(x - xo).max() <= RES_TOL * np.abs(xo).max()
    assert abs(np.linalg.norm(p.As[0] @ x - b) - p.resvec[-1]) <= 1e-8 * p.resvec[0]     # (it is the transposed system that was solved)
    # once more: As are back, but Ps[l] = Rs[l]' again (the reference's literal assignments - not the hierarchy the setup built,
    # whose P was 2^dim times Rs[l]'): the device must follow the host mirror there too
    mg.transposeHierarchy(p)
    assert p.device is handle and p.doTranspose == 0
    for l in range(1, levels):
        assert np.array_equal(handle.get_values(l, D.MG_OP_P), p.Ps[l - 1].data)
    x = np.zeros_like(b)
    mg.solveMG(p, b, x)
    hist_b = {}
    xo = np.zeros_like(b)
    orc.solveMG(p, b, xo, False, hist_b)
    assert np.abs(p.resvec - hist_b["resvec"]).max() / hist_b["resvec"][0] < RES_TOL
    assert np.abs(x - xo).max() <= RES_TOL * np.abs(xo).max()
    mg.clear_(p)
