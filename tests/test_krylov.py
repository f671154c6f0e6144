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


def test_reference_threshold_bicgstab_sa(mg, built):
    """testSAforDivSigGrad.jl:47-50: BiCGSTAB preconditioned with SA-AMG -> ||Ax-b|| < 0.005 (tol 1e-4, 5 iterations)."""
    A, p, b = _sa_problem(mg, [50, 50], 1e-8)
    x, flag, it, rv = orc.solveBiCGSTAB_MG(p, b, np.zeros_like(b))
    assert np.linalg.norm(A @ x - b) < 0.005


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["sa2d", "gmg3d"])
def test_device_bicgstab_matches_oracle(mg, built, kind):
    if kind == "sa2d":
        A, p, b = _sa_problem(mg, [50, 50], 1e-8)
        p.relativeTol, p.maxOuterIter = 1e-9, 30
    else:
        A, mesh = mg.poisson_shifted([24, 24, 24])
        p = mg.getMGparam(np.float64, np.int64, 3, 8, 10, 1e-9, "Jac", 0.8, 2, 2, "V", "NoMUMPS", 0.5, 0.0)
        mg.MGsetup(A, mesh, p)
        b = mg.seeded_rhs(A)
    x = np.zeros_like(b)
    x, _, it, nprec = mg.solveBiCGSTAB_MG(A, p, b, x)
    xo, flag, ito, rvo = orc.solveBiCGSTAB_MG(p, b, np.zeros_like(b))
    assert it == ito and p.flag == flag and flag in (0, -3)
    assert len(p.resvec) == len(rvo)
    assert np.abs(p.resvec - rvo).max() / rvo[0] < 1e-8
    assert np.abs(x - xo).max() <= 1e-8 * np.abs(xo).max()
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-8
    mg.clear_(p)


# ---- jInv solver wrappers (MGWrapper.jl, SAAMGWrapper.jl) -------------------------------------------------------
def test_wrapper_host_logic(mg, built):
    A, mesh = mg.poisson_shifted([8, 8])
    p = mg.getMGparam(levels=3, relaxType="Jac", relaxParam=0.8, maxIter=7)
    s = mg.getMGsolver(p, mesh, 1, "PCG")
    assert s.Krylov == "PCG" and s.tol == p.relativeTol and s.MG.Meshes[0] is mesh and s.nIter == 0
    c = mg.copySolverWrapper(s)
    assert c.MG is not s.MG and c.MG.maxOuterIter == 7 and not mg.hierarchyExists(c.MG)
    X = np.ones(A.shape[0])
    mg.solveLinearSystem_(A, np.zeros(A.shape[0]), X, s)          # norm(B) == 0 -> X = 0, no setup (l.37-40)
    assert not X.any() and not mg.hierarchyExists(s.MG)
    mg.setupSolver(A, s)
    assert mg.hierarchyExists(s.MG)
    mg.clearSolver_(s)
    assert not mg.hierarchyExists(s.MG) and s.doClear == 0
    g = mg.getMGsolver(mg.getMGparam(levels=2, relaxType="Jac"), mesh, 1)
    assert g.Krylov == "GMRES"                                                  # the reference's default (MGWrapper.jl:22)


@pytest.mark.gpu
@pytest.mark.parametrize("kry", ["PCG", "BiCGSTAB", "GMRES", "MG"])
def test_wrapper_solves_like_the_reference_tests(mg, built, kry):
    """testLinSolveMGWrapper.jl:13-39 shape: 2-D 51^2 nodes, 5 levels, V(2,2) SPAI, relres < tol (1e-2)."""
    from multigrid_jl_amd.operators import getRegularMesh, getNodalLaplacianMatrix, opnorm1
    mesh = getRegularMesh([0, 1, 0, 1], [48, 48])
    A = getNodalLaplacianMatrix(mesh)
    A = (A + 1e-4 * opnorm1(A) * sp.identity(A.shape[0])).tocsr()
    p = mg.getMGparam(np.float64, np.int64, 5, 8, 10, 1e-2, "SPAI", 1.0, 2, 2, "V", "NoMUMPS")
    s = mg.getMGsolver(p, mesh, 1, kry)
    rng = np.random.default_rng(4)
    B = A @ rng.random(A.shape[0])
    X = np.zeros_like(B)
    Xr, s = mg.solveLinearSystem_(A, B, X, s)
    assert Xr is X and s.nIter > 0 and s.timeSetup > 0 and s.timeSolve > 0
    assert np.linalg.norm(A @ X - B) / np.linalg.norm(B) < 1e-2
    sa = mg.getSA_AMGsolver(mg.getMGparam(np.float64, np.int64, 3, 8, 10, 1e-2, "SPAI", 1.0, 1, 1, "V", "Julia"), "PCG")
    X2 = np.zeros_like(B)
    mg.solveLinearSystem_(A, B, X2, sa)
    assert np.linalg.norm(A @ X2 - B) / np.linalg.norm(B) < 1e-2
    mg.clearSolver_(s)
    mg.clearSolver_(sa)


def test_reference_threshold_gmres_gmg(mg, built):
    """testGMGRAPforPoisson.jl:48-55: GMRES(10) preconditioned with GMG (Jac-GMRES smoother) -> ||Ax-b|| < 0.001
    (one right-hand side here; the reference uses blockFGMRES on 2)."""
    A, mesh = mg.poisson_shifted([128, 128])
    p = mg.getMGparam(np.float64, np.int64, 4, 8, 5, 1e-10, "Jac-GMRES", 0.75, 1, 1, "V", "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p)
    rng = np.random.default_rng(1)
    b = A @ rng.random(A.shape[0])
    b /= np.linalg.norm(b)
    x, flag, it, rv = orc.solveGMRES_MG(p, b, np.zeros_like(b), 10)
    assert np.linalg.norm(A @ x - b) < 0.001


@pytest.mark.gpu
@pytest.mark.parametrize("inner", [2, 5])
def test_device_fgmres_matches_oracle(mg, built, inner):
    A, mesh = mg.poisson_shifted([24, 24, 24])
    p = mg.getMGparam(np.float64, np.int64, 3, 8, 12, 1e-9, "Jac", 0.8, 1, 1, "V", "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p)
    b = mg.seeded_rhs(A)
    x = np.zeros_like(b)
    x, _, it, rv = mg.solveGMRES_MG(A, p, b, x, True, inner)
    xo, flag, ito, rvo = orc.solveGMRES_MG(p, b, np.zeros_like(b), inner)
    assert it == ito and p.flag == flag == 0 and len(rv) == len(rvo)
    assert np.abs(rv - rvo).max() / rvo[0] < 1e-8
    assert np.abs(x - xo).max() <= 1e-8 * np.abs(xo).max()
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-8
    mg.clear_(p)


# ---- block branches (SolveFuncs.jl:95,113,130), mixed precision hook (l.52-58), wrapper transposition -----------------
def test_oracle_block_methods_converge(mg, built):
    """The restated blockCG / blockBiCGSTB / blockFGMRES with the oracle's multigrid cycle as preconditioner solve a
    4-column system (the reference's wrapper test uses 4 right-hand sides, testLinSolveMGWrapper.jl:19)."""
    A, mesh = mg.poisson_shifted([16, 16])
    B = A @ np.random.default_rng(0).random((A.shape[0], 4))
    p = mg.getMGparam(np.float64, np.int64, 3, 8, 15, 1e-8, "SPAI", 1.0, 2, 2, "V", "Julia")
    mg.MGsetup(A, mesh, p, 4)
    M = orc.getMultigridPreconditioner(p, B)
    Af = lambda V: A @ V
    for out in (orc.blockCG(Af, B, 1e-8, 15, M), orc.blockBiCGSTB(Af, B, 1e-8, 15, M), orc.blockFGMRES(Af, B, 5, 1e-8, 15, M)):
        assert out[1] == 0 and np.linalg.norm(A @ out[0] - B) / np.linalg.norm(B) < 1e-7
    # rank-deficient block (two equal columns): the pseudo-inverse / semi-definite Cholesky paths
    B2 = np.column_stack([B[:, 0], B[:, 0], B[:, 1]])
    X, flag, _, _ = orc.blockCG(Af, B2, 1e-8, 30, orc.getMultigridPreconditioner(p, B2))
    assert flag == 0 and np.linalg.norm(A @ X - B2) / np.linalg.norm(B2) < 1e-7
    X, flag, _, _ = orc.blockFGMRES(Af, B2, 5, 1e-8, 30, orc.getMultigridPreconditioner(p, B2))
    assert flag == 0 and np.linalg.norm(A @ X - B2) / np.linalg.norm(B2) < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("nrhs", [2, 4])
def test_device_block_krylov_matches_oracle(mg, built, nrhs):
    A, mesh = mg.poisson_shifted([20, 18, 16])
    p = mg.getMGparam(np.float64, np.int64, 3, 8, 12, 1e-9, "Jac", 0.8, 2, 2, "V", "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p, nrhs)
    B = np.asfortranarray(mg.seeded_rhs(A, nrhs))
    Af = lambda V: A @ V
    tol = 1e-8
    # blockCG
    X = np.zeros_like(B, order="F")
    _, _, it = mg.solveCG_MG(A, p, B, X)
    Xo, flag, resmat, ito = orc.blockCG(Af, B, p.relativeTol, p.maxOuterIter, orc.getMultigridPreconditioner(p, B))
    assert it == ito and p.flag == flag == 0
    assert np.abs(p.resvec - resmat.max(axis=1)).max() < tol
    assert np.abs(X - Xo).max() <= tol * np.abs(Xo).max()
    # blockBiCGSTB
    X = np.zeros_like(B, order="F")
    _, _, it, nprec = mg.solveBiCGSTAB_MG(A, p, B, X)
    Xo, flag, ito, rvo = orc.blockBiCGSTB(Af, B, p.relativeTol, p.maxOuterIter, orc.getMultigridPreconditioner(p, B))
    assert it == ito and p.flag == flag and flag in (0, -3)
    assert len(p.resvec) == len(rvo) and np.abs(p.resvec - rvo).max() < tol
    assert np.abs(X - Xo).max() <= tol * np.abs(Xo).max()
    # blockFGMRES
    X = np.zeros_like(B, order="F")
    _, _, it, rv = mg.solveGMRES_MG(A, p, B, X, True, 5)
    Xo, flag, ito, rvo = orc.blockFGMRES(Af, B, 5, p.relativeTol, p.maxOuterIter, orc.getMultigridPreconditioner(p, B))
    assert it == ito and p.flag == flag == 0
    assert len(rv) == len(rvo) and np.abs(rv - rvo).max() < tol
    assert np.abs(X - Xo).max() <= tol * np.abs(Xo).max()
    assert np.linalg.norm(A @ X - B) / np.linalg.norm(B) < 1e-8
    mg.clear_(p)


@pytest.mark.gpu
def test_reference_wrapper_test_with_four_rhs(mg, built):
    """test/Multigrid/testLinSolveMGWrapper.jl:13-39 with its own parameters: 2-D 50^2 cells, G'G + 1e-2*norm(.,1)*I,
    B = Ar*rand(N,4), 5 levels, SPAI w=1, V(2,2), tol 1e-2, maxIter 15: MG-GMRES and MG-PCG through the jInv wrapper,
    both on the device with the block drivers, both pass the reference's assertion relres < tol."""
    import scipy.sparse as sp
    mesh = mg.getRegularMesh([0.0, 1.0, 0.0, 1.0], [50, 50])
    G = mg.getNodalGradientMatrix(mesh)
    Ar = (G.T @ G).tocsr()
    Ar = (Ar + 1e-2 * abs(Ar).sum() * sp.identity(Ar.shape[0])).tocsr()       # norm(Ar,1): entry-wise (SURVEY note N1)
    N = Ar.shape[0]
    B = Ar @ np.random.default_rng(42).random((N, 4))
    for krylov in ("GMRES", "PCG"):
        MG = mg.getMGparam(np.float64, np.int64, 5, 8, 15, 1e-2, "SPAI", 1.0, 2, 2, "V", "Julia")
        s = mg.getMGsolver(MG, mesh, 1, krylov, out=-1)
        X = np.zeros((N, 4), order="F")
        X, s = mg.solveLinearSystem_(Ar, np.asfortranarray(B), X, s)
        assert np.linalg.norm(Ar @ X - B) / np.linalg.norm(B) < s.tol
        mg.clearSolver_(s)


@pytest.mark.gpu
def test_wrapper_transposition_for_nonsymmetric_operator(mg, built):
    """solveLinearSystem!(A,B,X,param,doTranspose) solves A X = B for doTranspose = 0 and A' X = B for 1
    (MGWrapper.jl:27-86), also when the matrix handed over is flagged as already transposed (isTranspose)."""
    import scipy.sparse as sp
    A0, mesh = mg.poisson_shifted([20, 20])
    n = A0.shape[0]
    conv = sp.diags([0.35 * A0.diagonal().mean() * np.ones(n - 1)], [1], format="csr")   # upwind-like: A != A'
    A = (A0 + conv).tocsr()
    assert abs(A - A.T).max() > 1e-3
    b = np.random.default_rng(8).random(n)
    for is_t in (False, True):
        for doT in (0, 1):
            MG = mg.getMGparam(np.float64, np.int64, 3, 8, 60, 1e-9, "Jac", 0.7, 2, 2, "V", "Julia")
            s = mg.getMGsolver(MG, mesh, 0, "BiCGSTAB", out=-1)
            s.isTranspose = is_t
            x = np.zeros(n)
            x, s = mg.solveLinearSystem_(A, b, x, s, doT)
            # isTranspose: the matrix given is A' of the system the flags talk about
            Aeff = A.T if is_t else A
            op = Aeff.T if doT == 1 else Aeff
            assert np.linalg.norm(op @ x - b) / np.linalg.norm(b) < 1e-7, (is_t, doT)
            mg.clearSolver_(s)


@pytest.mark.gpu
def test_mixed_precision_preconditioner_and_schedule_resync(mg, built):
    """Float32 right-hand sides against the Float64 hierarchy (SolveFuncs.jl:52-58), one and three columns; and a change
    of param.cycleType / relaxPre after the upload is followed by the next cycle (MGcycle.jl reads them every time)."""
    A, mesh = mg.poisson_shifted([16, 16, 16])
    p = mg.getMGparam(np.float64, np.int64, 3, 8, 5, 1e-10, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p, 1)
    for nrhs in (1, 3):
        B = mg.seeded_rhs(A, nrhs)
        B32 = np.asfortranarray(B.astype(np.float32))
        M32 = mg.getMultigridPreconditioner(p, B32)
        z32 = M32(B32).copy()
        assert z32.dtype == np.float32
        zo = orc.recursiveCycle(p, np.asarray(B32, dtype=np.float64), np.zeros(B32.shape), 1)
        assert np.abs(z32 - zo).max() <= 2e-6 * np.abs(zo).max()
    b = mg.seeded_rhs(A, 1)
    x = np.zeros_like(b)
    mg.recursiveCycle(p, b, x, 1)
    p.cycleType = "W"
    p.relaxPre = lambda l: 3
    x = np.zeros_like(b)
    mg.recursiveCycle(p, b, x, 1)
    xo = orc.recursiveCycle(p, b, np.zeros_like(b), 1)
    assert np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()
    mg.clear_(p)
