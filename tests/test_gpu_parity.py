"""-m gpu: the HIP path through the C ABI against the oracle on the same seeded inputs.

Tolerances: fp64, residual-norm history within 1e-10 relative of the oracle (BASELINE.json
north_star); kernel-level SpMV within 1e-13 relative (reassociated in-row sums only).
"""
import numpy as np
import pytest

from oracle import mg_oracle as orc

pytestmark = pytest.mark.gpu

RES_TOL = 1e-10
KERNEL_TOL = 1e-13


def _setup(mg, ncells, levels, relaxType="Jac", omega=0.8, pre=2, post=1, cyc="V", maxIter=8, nrhs=1):
    A, mesh = mg.poisson_shifted(ncells)
    p = mg.getMGparam(np.float64, np.int64, levels, 8, maxIter, 1e-10, relaxType, omega, pre, post, cyc,
                      "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p, nrhs)
    b = mg.seeded_rhs(A, nrhs)
    return A, p, b


def _compare_solve(mg, p, b):
    x = np.zeros_like(b)
    xo = np.zeros_like(b)
    _, _, it = mg.solveMG(p, b, x)
    hist = {}
    _, _, ito = orc.solveMG(p, b, xo, False, hist)
    assert it == ito
    rel = np.abs(p.resvec - hist["resvec"]).max() / hist["resvec"][0]
    assert rel < RES_TOL, rel
    assert np.abs(x - xo).max() <= RES_TOL * np.abs(xo).max()
    return x, hist


@pytest.mark.parametrize("ncells,levels", [([8, 8, 8], 2), ([16, 16, 16], 3), ([32, 32, 32], 3), ([32, 32, 16], 4),
                                           ([15, 12, 9], 3), ([64, 64], 4), ([129, 65], 4)])
def test_solveMG_gmg_matches_oracle(mg, built, ncells, levels):
    A, p, b = _setup(mg, ncells, levels)
    x, hist = _compare_solve(mg, p, b)
    # in-place contract + it is really a solve
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-3
    mg.clear_(p)


@pytest.mark.parametrize("cyc", ["V", "W", "F"])
@pytest.mark.parametrize("relaxType,omega,pre,post", [("Jac", 0.8, 1, 1), ("SPAI", 1.0, 2, 2), ("Jac", 0.75, 3, 0)])
def test_cycle_types_and_smoothers(mg, built, cyc, relaxType, omega, pre, post):
    A, p, b = _setup(mg, [16, 16, 16], 4, relaxType, omega, pre, post, cyc, maxIter=5)
    _compare_solve(mg, p, b)
    mg.clear_(p)


@pytest.mark.parametrize("nrhs", [2, 3, 4, 16])
def test_block_rhs(mg, built, nrhs):
    A, p, b = _setup(mg, [16, 16, 8], 3, "Jac", 0.8, 2, 1, "V", 6, nrhs)
    _compare_solve(mg, p, b)
    mg.clear_(p)


def test_nonzero_initial_guess_and_single_cycle(mg, built):
    A, p, b = _setup(mg, [16, 16, 16], 3, maxIter=3)
    rng = np.random.default_rng(5)
    x0 = rng.standard_normal(b.shape)
    x = x0.copy()
    xo = x0.copy()
    mg.solveMG(p, b, x)
    hist = {}
    orc.solveMG(p, b, xo, False, hist)
    assert np.abs(p.resvec - hist["resvec"]).max() / hist["resvec"][0] < RES_TOL
    # one recursiveCycle from a non-zero x, and from zero (preconditioner closure)
    x = x0.copy()
    mg.recursiveCycle(p, b, x, 1)
    xo = orc.recursiveCycle(p, b, x0.copy(), 1)
    assert np.abs(x - xo).max() <= RES_TOL * np.abs(xo).max()
    M = mg.getMultigridPreconditioner(p, b)
    z = M(b).copy()
    zo = orc.recursiveCycle(p, b, np.zeros_like(b), 1)
    assert np.abs(z - zo).max() <= RES_TOL * np.abs(zo).max()
    mg.clear_(p)


@pytest.mark.parametrize("nrhs", [1, 3])
def test_spmatmul_all_operators(mg, built, nrhs):
    A, p, b = _setup(mg, [16, 12, 10], 3, nrhs=nrhs)
    rng = np.random.default_rng(11)
    for level in (1, 2):
        for which, M in (("A", p.As[level - 1]), ("P", p.Ps[level - 1]), ("R", p.Rs[level - 1])):
            shp = (M.shape[1],) if nrhs == 1 else (M.shape[1], nrhs)
            oshp = (M.shape[0],) if nrhs == 1 else (M.shape[0], nrhs)
            x = np.asfortranarray(rng.standard_normal(shp))
            for alpha, beta in ((1.0, 0.0), (-1.0, 1.0), (-1.0, 0.0), (1.0, 1.0), (0.5, -2.0)):
                y = np.asfortranarray(rng.standard_normal(oshp))
                ref = orc.SpMatMul(alpha, M, x, beta, y.copy())
                got = mg.SpMatMul(p, level, which, x, y, alpha, beta)
                scale = np.abs(ref).max() + 1e-300
                assert np.abs(got - ref).max() / scale < KERNEL_TOL
    mg.clear_(p)


def test_irregular_rows_and_long_rows(mg, built):
    """General CSR: empty rows, ragged rows and rows longer than one LDS chunk (long-row path)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(3)
    n = 6000
    A = sp.random(n, n, density=2e-3, random_state=1, format="lil")
    A[7, :] = rng.standard_normal(n)            # dense row: 6000 nnz > chunk
    A[4000, :3000] = 1.0
    A[10, :] = 0.0                              # empty row
    A = (A + sp.identity(n) * 50).tocsr()
    A[10, 10] = 0.0
    A.eliminate_zeros()
    A.sort_indices()
    p = mg.getMGparam(np.float64, np.int64, 1, 8, 1, 1e-10, "Jac", 0.8, 1, 1)
    p.As = [A]
    import scipy.sparse.linalg as spla
    p.LU = spla.splu(sp.csc_matrix(sp.identity(n)))    # single level: coarse inverse = I
    p.nrhs = 1
    # single-level hierarchies enter through mg_spmv only (the coarse solve is the identity here)
    for nrhs in (1, 4):
        x = np.asfortranarray(rng.standard_normal((n, nrhs))) if nrhs > 1 else rng.standard_normal(n)
        y = np.zeros_like(x)
        got = mg.SpMatMul(p, 1, "A", x, y)
        ref = A @ x
        assert np.abs(got - ref).max() / np.abs(ref).max() < KERNEL_TOL
    mg.clear_(p)


def test_api_errors_are_loud(mg, built):
    A, p, b = _setup(mg, [8, 8, 8], 2)
    dev = mg.to_device(p)
    with pytest.raises(mg.device.MGDeviceError):
        dev.cycle(np.zeros(5), np.zeros(5))              # wrong n
    with pytest.raises(mg.device.MGDeviceError):
        dev.spmv(7, 0, 1.0, b, 0.0, b.copy())            # bad level
    mg.clear_(p)


# ---- committed golden fixtures (tests/golden/*.npz, generated by tests/golden/make_golden.py) ---------------
import glob
import os

_GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(_GOLD, "*.npz"))), ids=lambda p: os.path.basename(p)[:-4])
def test_hip_path_reproduces_golden(mg, built, path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(_GOLD, "make_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    name = os.path.basename(path)[:-4]
    g = np.load(path)
    A, p, b = mk.build_sa_case(name) if name in mk.SA_CASES else mk.build_case(name)
    assert np.allclose(b, g["b"], rtol=1e-12, atol=1e-15)      # regenerated RHS (last bits vary with the host BLAS)
    b = np.asfortranarray(g["b"])                               # the fixture is the input
    x = np.zeros_like(b)
    _, _, it = mg.solveMG(p, b, x)
    assert it == int(g["iters"])
    assert np.abs(p.resvec - g["resvec"]).max() / g["resvec"][0] < RES_TOL
    assert np.abs(x - g["x_last"]).max() <= RES_TOL * np.abs(g["x_last"]).max()
    # first cycle alone, through mg_cycle with x = 0
    x1 = np.zeros_like(b)
    mg.recursiveCycle(p, b, x1, 1)
    assert np.abs(x1 - g["x_first"]).max() <= RES_TOL * np.abs(g["x_first"]).max()
    mg.clear_(p)


@pytest.mark.parametrize("nrhs,cells", [(1, [20, 18, 16]), (4, [16, 16, 16]), (1, [33, 31])])
def test_tiled_block_schedule_changes_nothing(mg, built, nrhs, cells, monkeypatch):
    """mg_set_grid_hint + a tiny L2 budget force the y-tiled row-block order on every level: results must be
    IDENTICAL to the natural order (pure scheduling) and match the oracle."""
    A, mesh = mg.poisson_shifted(cells)
    b = mg.seeded_rhs(A, nrhs)
    xs = []
    for budget in ("1e12", "2000"):
        monkeypatch.setenv("MG_SCHED_BUDGET", budget)
        p = mg.getMGparam(np.float64, np.int64, 3, 8, 4, 1e-10, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
        mg.MGsetup(A, mesh, p, nrhs)
        x = np.zeros_like(b)
        mg.solveMG(p, b, x)
        xs.append((x, p.resvec.copy()))
        if budget == "2000":
            xo = np.zeros_like(b)
            hist = {}
            orc.solveMG(p, b, xo, False, hist)
            assert np.abs(p.resvec - hist["resvec"]).max() / hist["resvec"][0] < RES_TOL
        mg.clear_(p)
    assert np.array_equal(xs[0][0], xs[1][0]) and np.array_equal(xs[0][1], xs[1][1])


@pytest.mark.parametrize("cyc,relaxType,pre,post,nrhs", [("V", "Jac-GMRES", 1, 1, 1), ("V", "Jac-GMRES", 2, 1, 2),
                                                        ("K", "Jac", 1, 1, 1), ("K", "Jac-GMRES", 1, 1, 3),
                                                        ("W", "Jac-GMRES", 2, 2, 1)])
def test_jac_gmres_and_kcycle(mg, built, cyc, relaxType, pre, post, nrhs):
    """FGMRES_relaxation smoother (FGMRES.jl:48-126) and the K-cycle recursion (MGcycle.jl:72-76)."""
    A, p, b = _setup(mg, [16, 16, 16], 4, relaxType, 0.75, pre, post, cyc, maxIter=5, nrhs=nrhs)
    _compare_solve(mg, p, b)
    mg.clear_(p)


def test_reference_test_verbatim_gmgrap_poisson(mg, built):
    """test/Multigrid/testGMGRAPforPoisson.jl:59-78 with its own parameters (Jac-GMRES 0.75, V(1,1), 4 levels,
    nrhs 2, 5 cycles, seeded RHS): the device solve passes the reference's assertion ||AX-B|| < 0.01."""
    A, p, b = _setup(mg, [32, 32, 16], 4, "Jac-GMRES", 0.75, 1, 1, "V", maxIter=5, nrhs=2)
    x, hist = _compare_solve(mg, p, b)
    assert np.linalg.norm(A @ x - b) < 0.01
    mg.clear_(p)


def test_pattern_coded_kernel_paths(mg, built, monkeypatch):
    """csr_pattern_spmv (dictionary-coded column indices) vs csr_stream_spmv (plain CSR) on the same operators,
    including a banded operator whose identical rows are LONGER than one LDS chunk (long-row path) and every
    fused epilogue, through the stand-alone operator entry points."""
    import torch
    import scipy.sparse as sp
    from multigrid_jl_amd import device as D
    monkeypatch.setenv("MG_NO_ROWCLASS", "1")            # this test is about the two STREAMING formats
    rng = np.random.default_rng(7)
    nr, L = 700, 2500                                    # 2500 > CHUNK-2: every row takes the long-row path
    rows = np.repeat(np.arange(nr), L)
    cols = rows + np.tile(np.arange(L), nr)
    band = sp.csr_matrix((rng.standard_normal(nr * L), (rows, cols)), shape=(nr, nr + L))
    A, _ = mg.poisson_shifted([14, 12, 10])
    for M, square in ((band, False), (A, True)):
        x = torch.from_numpy(rng.standard_normal(M.shape[1])).cuda()
        b = torch.from_numpy(rng.standard_normal(M.shape[0])).cuda()
        d = torch.from_numpy(rng.standard_normal(M.shape[0])).cuda()
        outs = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("MG_NO_PATTERN", mode)
            op = D.DeviceOperator(M, 0)
            res = []
            y = torch.from_numpy(rng.standard_normal(M.shape[0]) * 0 + 1.0).cuda()
            op.apply(D.MG_K_SPMV, x, y, alpha=-0.5, beta=2.0)
            res.append(y.cpu().numpy().copy())
            op.apply(D.MG_K_RESIDUAL, x, y, b=b)
            res.append(y.cpu().numpy().copy())
            if square:
                op.apply(D.MG_K_SMOOTH, x, y, b=b, d=d)
                res.append(y.cpu().numpy().copy())
            torch.cuda.synchronize()
            outs[mode] = res
            op.close()
        xn, bn, dn = x.cpu().numpy(), b.cpu().numpy(), d.cpu().numpy()
        want = [-0.5 * (M @ xn) + 2.0, bn - M @ xn] + ([xn + dn * (bn - M @ xn)] if square else [])
        for got0, got1, w in zip(outs["0"], outs["1"], want):
            scale = np.abs(w).max()
            assert np.abs(got0 - w).max() / scale < KERNEL_TOL
            assert np.abs(got1 - w).max() / scale < KERNEL_TOL


@pytest.mark.parametrize("relaxType,omega,cells,chunk", [("Jac", 0.8, [16, 16, 16], 0), ("SPAI", 1.0, [24, 20], 0), ("Jac", 0.8, [12, 12, 12], 10)])
def test_replace_matrix_on_device(mg, built, monkeypatch, relaxType, omega, cells, chunk):
    """replaceMatrixInHierarchy (MGsetup.jl:226-270) with a resident hierarchy: the numeric Galerkin products and
    relaxPrecs are recomputed on the device (mg_rap_FP64) and must equal the host products; the following solve must
    match the oracle on the refreshed host hierarchy."""
    import scipy.sparse as sp
    from multigrid_jl_amd.mgsetup import galerkin
    if chunk:      # coarse rows longer than the accumulator (SA-AMG middle levels: thousands of entries): 27-entry rows, 10 at a time
        monkeypatch.setenv("MG_RAP_CHUNK", str(chunk))
    A, mesh = mg.poisson_shifted(cells)
    p = mg.getMGparam(np.float64, np.int64, 3, 8, 6, 1e-10, relaxType, omega, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p)
    b = mg.seeded_rhs(A)
    x = np.zeros_like(b)
    mg.solveMG(p, b, x)                                   # uploads the hierarchy
    assert p.device is not None
    rng = np.random.default_rng(8)
    A2 = A.copy()
    A2.data = A.data * (1.0 + 0.3 * rng.random(A.nnz))    # same sparsity, new (non-symmetric) values
    A2 = (A2 + sp.diags(np.full(A.shape[0], A.diagonal().max()))).tocsr()
    A2.sort_indices()
    assert np.array_equal(A2.indices, A.indices)
    dev_before = p.device
    mg.replaceMatrixInHierarchy(p, A2)
    assert p.device is dev_before                         # stayed resident: the device path was taken
    Al = A2
    for l in range(len(p.As) - 1):
        assert np.allclose(p.relaxPrecs[l], mg.getRelaxPrec(Al, relaxType, omega), rtol=1e-13)
        Al = galerkin(p.Rs[l], Al, p.Ps[l])
        assert np.array_equal(Al.indices, p.As[l + 1].indices)
        assert np.abs(Al.data - p.As[l + 1].data).max() <= 1e-13 * np.abs(Al.data).max()
    b2 = A2 @ rng.random(A.shape[0])
    b2 /= np.linalg.norm(b2)
    _compare_solve(mg, p, b2)
    # the device product is deterministic (no atomics): a second pass over the same values gives the same bits
    first = [M.data.copy() for M in p.As[1:]]
    first_d = [np.array(d, copy=True) for d in p.relaxPrecs[:len(p.As) - 1]]
    mg.replaceMatrixInHierarchy(p, A2)
    for v0, M in zip(first, p.As[1:]):
        assert np.array_equal(v0, M.data)
    # ... and so are the relaxPrecs: SPAI's column sums of squares are ordered sums (ascending rows, the order of Julia's
    # sum(AT.^2, dims=2) and of the host's bincount), not atomics - the same bits as the host's
    for l, d0 in enumerate(first_d):
        assert np.array_equal(d0, p.relaxPrecs[l])
    if relaxType == "SPAI":
        assert np.array_equal(p.relaxPrecs[0], mg.getRelaxPrec(A2, relaxType, omega))
    mg.clear_(p)


def test_c_abi_error_codes(mg, built):
    """Every misuse returns a status + message (include/mgvcycle.h), never a crash or a silent fallback."""
    import ctypes as C
    from multigrid_jl_amd import device as D
    lib = D.load_library()
    h = C.c_void_p()
    assert lib.mg_create(0, 1, 0, C.byref(h)) == 1 and b"nlevels" in lib.mg_last_error()          # MG_ERR_INVALID
    assert lib.mg_create(2, 1, 99, C.byref(h)) == 1
    assert lib.mg_create(2, 1, 0, C.byref(h)) == 0
    keep = []                                   # the arrays must outlive the calls

    def i64(a):
        keep.append(np.ascontiguousarray(a, dtype=np.int64))
        return keep[-1].ctypes.data_as(C.POINTER(C.c_longlong))

    def f64(a):
        keep.append(np.ascontiguousarray(a, dtype=np.float64))
        return keep[-1].ctypes.data_as(C.POINTER(C.c_double))
    # 0-based pointer array, out-of-range column, non-monotone pointers
    assert lib.mg_set_operator_FP64_INT64(h, 1, 0, 2, 2, i64([0, 1, 2]), i64([1, 2]), f64([1.0, 1.0])) == 1
    assert lib.mg_set_operator_FP64_INT64(h, 1, 0, 2, 2, i64([1, 2, 3]), i64([1, 3]), f64([1.0, 1.0])) == 1
    assert lib.mg_set_operator_FP64_INT64(h, 1, 0, 2, 2, i64([1, 3, 2]), i64([1, 2]), f64([1.0, 1.0])) == 1
    assert lib.mg_set_operator_FP64_INT64(h, 2, 1, 2, 2, i64([1, 2, 3]), i64([1, 2]), f64([1.0, 1.0])) == 1   # P on the coarsest
    assert lib.mg_set_operator_FP64_INT64(h, 1, 0, 2, 2, i64([1, 2, 3]), i64([1, 2]), f64([1.0, 1.0])) == 0
    assert lib.mg_finalize(h) == 3 and b"not set" in lib.mg_last_error()                       # MG_ERR_STATE
    b = np.zeros(2)
    assert lib.mg_cycle_FP64(h, f64(b), f64(b), 2, 1, 1) == 3                                       # not finalized
    assert lib.mg_set_cycle_type(h, ord("Z")) == 1
    assert lib.mg_set_relax_type(h, 7) == 1
    assert lib.mg_destroy(h) == 0


def test_rediscretised_hierarchy_on_device(mg, built):
    """The geometric / re-discretisation mode of MGsetup (multilevelOperatorConstructor, testGMG.jl:70-75)."""
    from test_oracle import _rediscretisation_problem
    mesh, sig, op, restrict = _rediscretisation_problem(mg)
    p = mg.getMGparam(np.float64, np.int64, 4, 2, 5, 1e-10, "Jac", 0.8, 1, 1, "V", "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(mg.getMultilevelOperatorConstructor(sig, op, restrict), mesh, p)
    rng = np.random.default_rng(0)
    b = p.As[0] @ rng.random(p.As[0].shape[0])
    b /= np.linalg.norm(b)
    _compare_solve(mg, p, b)
    mg.clear_(p)


@pytest.mark.gpu
@pytest.mark.parametrize("multi", [False, True])
@pytest.mark.parametrize("cells,levels,nrhs", [([16, 16, 16], 3, 1), ([32, 32], 3, 3), ([24, 24, 24], 2, 2), ([24, 24, 24], 2, 6)])
def test_sparse_lu_coarse_solve(mg, built, cells, levels, nrhs, multi, monkeypatch):
    """Coarsest levels above the dense-inverse cap are solved on the device from the SPARSE factors in the layout of
    the reference's native applier (mg_set_coarse_lu_FP64_INT64 <-> deps/src/parLU.cpp:120-190).  Forced here on small
    hierarchies: the whole solve must match the oracle exactly as the dense-inverse path does, and (where the
    reference binary was built) the coarse solve alone must match applyLUsolve_FP64_INT64."""
    from multigrid_jl_amd import device as D
    monkeypatch.setattr(D, "DENSE_COARSE_MAX", 0)
    # multi: the chip-wide form (one launch per dependency level, the trailing chain of single-row levels through the
    # explicit inverse of its dense block) that factors of >= 4096 rows get, forced on these small ones
    monkeypatch.setenv("MG_LU_MULTI_MIN_ROWS", "0" if multi else "1000000000")
    monkeypatch.setenv("MG_LU_DENSE_TAIL_MIN", "4")
    A, p, b = _setup(mg, cells, levels, nrhs=nrhs)
    _compare_solve(mg, p, b)
    # the coarse solve in isolation: a 1-level "hierarchy" is just x = LU \ b (recursiveCycle at the coarsest level)
    Ac = p.As[-1]
    q = mg.getMGparam(np.float64, np.int64, 1, 8, 1, 1e-10, "Jac", 0.8, 2, 1)
    q.As, q.Ps, q.Rs, q.relaxPrecs, q.LU, q.Meshes, q.levels = [Ac], [], [], [], p.LU, [p.Meshes[-1]], 1
    q.nrhs = nrhs
    rng = np.random.default_rng(5)
    B = np.asfortranarray(rng.standard_normal((Ac.shape[0], nrhs))) if nrhs > 1 else rng.standard_normal(Ac.shape[0])
    X = mg.recursiveCycle(q, B.copy(order="F"), np.zeros_like(B, order="F"), 1)
    Xo = p.LU.solve(B)
    assert np.abs(X - Xo).max() <= 1e-12 * np.abs(Xo).max()
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "parLU.so")
    if os.path.exists(ref):
        from test_reference_parlu import _ref_lu_solve
        Xr = _ref_lu_solve(p.LU, B)
        assert np.abs(X - Xr).max() <= 1e-12 * np.abs(Xr).max()
    mg.clear_(q)
    mg.clear_(p)


@pytest.mark.gpu
def test_large_coarsest_level_uses_sparse_factors(mg, built):
    """levels=2 on 64^3 leaves a 33^3 = 35 937-row coarsest level (> 16 384): the reference handles any size through
    its sparse LU (MGsetup.jl:350), so must the device path."""
    A, p, b = _setup(mg, [64, 64, 64], 2, maxIter=4)
    assert p.As[-1].shape[0] == 33 ** 3
    _compare_solve(mg, p, b)
    mg.clear_(p)


@pytest.mark.gpu
def test_rowclass_kernel_paths(mg, built, monkeypatch):
    """csr_rowclass_spmv (rows stored as {first column, class id}, offsets AND values from a dictionary) against the
    streaming kernels on the same operators: every fused epilogue, A / P / R of a GMG hierarchy, the fused ||r||^2,
    a solve under both settings, and an operator WITHOUT redundant rows (must not be stored that way)."""
    import torch
    import scipy.sparse as sp
    from multigrid_jl_amd import device as D
    # by default only operators of >= 100 000 rows with <= 4 classes per wavefront are stored this way
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_STAGE_MIN_LEN", "0")          # LDS-staged variants also on the 7-point level
    monkeypatch.setenv("MG_TILE_MIN_WG", "0")            # ... and on levels with few workgroups
    monkeypatch.setenv("MG_WINDOW_MIN_WG", "0")
    rng = np.random.default_rng(17)
    A, _ = mg.poisson_shifted([14, 12, 10])
    Arand = A.copy()
    Arand.data = rng.standard_normal(A.nnz)              # same pattern, all rows distinct
    P = mg.getFWInterp(np.array([15, 13, 11]))[0]
    R = (0.125 * P.T).tocsr()
    for M, square in ((A, True), (P, False), (R, False), (Arand, True)):
        x = torch.from_numpy(rng.standard_normal(M.shape[1])).cuda()
        b = torch.from_numpy(rng.standard_normal(M.shape[0])).cuda()
        d = torch.from_numpy(rng.standard_normal(M.shape[0])).cuda()
        outs = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("MG_NO_ROWCLASS", mode)
            op = D.DeviceOperator(M, 0)
            res = []
            y = torch.ones(M.shape[0], dtype=torch.float64, device="cuda")
            op.apply(D.MG_K_SPMV, x, y, alpha=-0.5, beta=2.0)
            res.append(y.cpu().numpy().copy())
            op.apply(D.MG_K_SPMV, x, y, alpha=1.0, beta=0.0)
            res.append(y.cpu().numpy().copy())
            op.apply(D.MG_K_RESIDUAL, x, y, b=b)
            res.append(y.cpu().numpy().copy())
            if square:
                op.apply(D.MG_K_SMOOTH, x, y, b=b, d=d)
                res.append(y.cpu().numpy().copy())
            torch.cuda.synchronize()
            outs[mode] = res
            op.close()
        xn, bn, dn = x.cpu().numpy(), b.cpu().numpy(), d.cpu().numpy()
        want = [-0.5 * (M @ xn) + 2.0, M @ xn, bn - M @ xn] + ([xn + dn * (bn - M @ xn)] if square else [])
        for got0, got1, w in zip(outs["0"], outs["1"], want):
            scale = np.abs(w).max()
            assert np.abs(got0 - w).max() / scale < KERNEL_TOL
            assert np.abs(got1 - w).max() / scale < KERNEL_TOL
    # inside a hierarchy: which operators are stored as row classes, and the solve is the same either way
    hist = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MG_NO_ROWCLASS", mode)
        A, p, b = _setup(mg, [20, 18, 16], 3)
        x = np.zeros_like(b)
        mg.solveMG(p, b, x)
        fm = {(l, w): p.device.operator_rowclasses(l, k)[0] for l in (1, 2)
              for w, k in (("A", D.MG_OP_A), ("P", D.MG_OP_P), ("R", D.MG_OP_R))}
        if mode == "0":
            # constant-coefficient GMG: the fine operators qualify (tiny coarse ones may not: dictionary > nnz/16)
            assert all(v > 0 for (l, w), v in fm.items() if l == 1), fm
        else:
            assert all(v == 0 for v in fm.values()), fm
        hist[mode] = (x.copy(), np.asarray(p.resvec).copy())
        _compare_solve(mg, p, b)
        mg.clear_(p)
    assert np.abs(hist["0"][1] - hist["1"][1]).max() <= RES_TOL * hist["1"][1][0]
    assert np.abs(hist["0"][0] - hist["1"][0]).max() <= RES_TOL * np.abs(hist["1"][0]).max()
    # refinements: implicit first column (square operators) and relaxPrec from the class dictionary, each switchable
    monkeypatch.setenv("MG_NO_ROWCLASS", "0")
    for relaxType, omega in (("Jac", 0.8), ("SPAI", 1.0)):
        for no_first, no_d, no_win, no_tile in (("0", "0", "0", "0"), ("1", "0", "0", "0"), ("0", "1", "0", "0"),
                                                ("0", "0", "1", "1"), ("0", "1", "1", "1"), ("0", "0", "0", "1"),
                                                ("0", "1", "0", "1")):
            monkeypatch.setenv("MG_NO_IMPLICIT_FIRST", no_first)
            monkeypatch.setenv("MG_NO_CLASS_D", no_d)
            monkeypatch.setenv("MG_NO_WINDOW", no_win)        # csr_rowclass_window_spmv (LDS windows) on / off
            monkeypatch.setenv("MG_NO_TILE", no_tile)         # csr_rowclass_tile_spmv (plane tiles from the grid hint)
            A, p, b = _setup(mg, [32, 24, 8], 3, relaxType=relaxType, omega=omega)   # 33x25 = 825 rows per plane
            _compare_solve(mg, p, b)
            fa = p.device.operator_rowclass_flags(1, D.MG_OP_A)
            fp = p.device.operator_rowclass_flags(1, D.MG_OP_P)
            assert fa[0] == (no_first == "0") and fp == (False, False), (fa, fp)
            if relaxType == "Jac":
                assert fa[1] == (no_d == "0"), fa             # omega/a_ii is constant per class by construction
            mg.clear_(p)
    monkeypatch.delenv("MG_NO_IMPLICIT_FIRST")
    monkeypatch.delenv("MG_NO_CLASS_D")
    monkeypatch.delenv("MG_NO_WINDOW")
    monkeypatch.delenv("MG_NO_TILE")
    # a hierarchy on the non-redundant operator keeps the streaming formats
    monkeypatch.setenv("MG_NO_ROWCLASS", "0")
    Ad, mesh = mg.poisson_shifted([12, 12, 12])
    sig = sp.diags(1.0 + rng.random(Ad.shape[0]))
    Av = (sig @ Ad @ sig).tocsr()                         # SPD, variable coefficients
    p = mg.getMGparam(np.float64, np.int64, 2, 8, 4, 1e-10, "Jac", 0.8, 2, 1)
    mg.MGsetup(Av, mesh, p, 1)
    bv = mg.seeded_rhs(Av, 1)
    _compare_solve(mg, p, bv)
    assert p.device.operator_rowclasses(1, D.MG_OP_A)[0] == 0
    assert p.device.operator_rowclasses(1, D.MG_OP_P)[0] > 0
    mg.clear_(p)
    # default thresholds: a 64x64x48-cell grid qualifies on the fine level only (small levels stay streaming)
    monkeypatch.delenv("MG_ROWCLASS_MIN_ROWS")
    monkeypatch.delenv("MG_ROWCLASS_MAX_PASSES")
    monkeypatch.delenv("MG_STAGE_MIN_LEN")
    A, p, b = _setup(mg, [64, 64, 48], 4, maxIter=4)
    _compare_solve(mg, p, b)
    assert p.device.operator_rowclasses(1, D.MG_OP_A)[0] > 0
    assert p.device.operator_rowclasses(3, D.MG_OP_A)[0] == 0
    mg.clear_(p)


@pytest.mark.gpu
@pytest.mark.parametrize("n_odd", [0, 6])          # 0: 3 % of the rows (own launch); 6: handled inside the main kernel
def test_rowclass_exception_rows(mg, built, monkeypatch, n_odd):
    """A mostly regular operator: 3 % of the rows of a Poisson matrix get unique values, a few more get an extra
    entry.  The popular classes stay in the dictionary, the odd rows become exception rows computed from the CSR
    arrays (csr_rows_spmv); every fused epilogue and the fused ||r||^2 must match scipy."""
    import torch
    import scipy.sparse as sp
    from multigrid_jl_amd import device as D
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_ROWCLASS_KEEP_SINGLETONS", "0")    # unique rows become exception rows, not dictionary classes
    monkeypatch.setenv("MG_WINDOW_MIN_WG", "0")
    monkeypatch.setenv("MG_TILE_MIN_WG", "0")
    rng = np.random.default_rng(23)
    A, mesh = mg.poisson_shifted([24, 20, 18])
    A = A.tolil()
    n = A.shape[0]
    odd = rng.choice(n, size=(n_odd or n // 33), replace=False)
    for i in odd[: len(odd) // 2]:
        A[i, i] = A[i, i] * (1.0 + rng.random())               # unique values
    for i in odd[len(odd) // 2:]:
        A[i, (i * 7 + 3) % n] = 0.125 * rng.random()           # an extra entry somewhere
    A = sp.csr_matrix(A)
    A.sort_indices()
    x = torch.from_numpy(rng.standard_normal(n)).cuda()
    b = torch.from_numpy(rng.standard_normal(n)).cuda()
    d = torch.from_numpy(rng.standard_normal(n)).cuda()
    for stage in ("0", "1"):                                    # staged (window) and plain row-class kernels
        monkeypatch.setenv("MG_NO_WINDOW", stage)
        op = D.DeviceOperator(A, 0)
        y = torch.ones(n, dtype=torch.float64, device="cuda")
        xn, bn, dn = x.cpu().numpy(), b.cpu().numpy(), d.cpu().numpy()
        op.apply(D.MG_K_SPMV, x, y, alpha=-0.5, beta=2.0)
        want = -0.5 * (A @ xn) + 2.0
        assert np.abs(y.cpu().numpy() - want).max() / np.abs(want).max() < KERNEL_TOL
        op.apply(D.MG_K_RESIDUAL, x, y, b=b)
        want = bn - A @ xn
        assert np.abs(y.cpu().numpy() - want).max() / np.abs(want).max() < KERNEL_TOL
        op.apply(D.MG_K_SMOOTH, x, y, b=b, d=d)
        want = xn + dn * (bn - A @ xn)
        assert np.abs(y.cpu().numpy() - want).max() / np.abs(want).max() < KERNEL_TOL
        op.close()
    # inside a hierarchy (fused residual + norm, class-constant relaxPrec with exception rows, xpdr): vs the oracle
    p = mg.getMGparam(np.float64, np.int64, 2, 8, 6, 1e-10, "Jac", 0.8, 2, 1)
    mg.MGsetup(A, mesh, p, 1)
    bb = mg.seeded_rhs(A, 1)
    _compare_solve(mg, p, bb)
    assert p.device.operator_rowclasses(1, D.MG_OP_A)[0] > 0
    nexc = p.device.operator_kernel_info(1, D.MG_OP_A)[1]
    assert (0 < nexc <= 256) if n_odd else nexc > 256          # in-kernel handling / own launch
    mg.clear_(p)


@pytest.mark.gpu
@pytest.mark.parametrize("cells,levels", [([33, 25, 7], 2), ([40, 30, 5], 3), ([63, 17, 9], 2), ([31, 37], 3), ([129, 9], 2),
                                           ([23, 23, 23], 3), ([70, 10, 3], 2)])
def test_rowclass_variants_on_odd_grids(mg, built, cells, levels, monkeypatch):
    """Every row-class kernel variant (plain / window / plane tile, with and without the implicit first column and the
    dictionary relaxPrec) on grids whose sizes are not multiples of anything convenient (partial chunks, partial
    planes groups, planes smaller or larger than a workgroup, 2-D): fused residual, fused sweep and the transfer
    products of every level against scipy, through the device-resident entry points."""
    import torch
    from multigrid_jl_amd import device as D
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_ROWCLASS_MIN_COVER", "0.05")
    monkeypatch.setenv("MG_TILE_MIN_WG", "0")
    monkeypatch.setenv("MG_WINDOW_MIN_WG", "0")
    monkeypatch.setenv("MG_PAIR_MIN_ROWS", "0")          # paired-rows variant of the plain kernel for P (alternating classes)
    rng = np.random.default_rng(sum(cells))
    seen = set()
    for no_tile, no_win, no_first, no_lane in (("0", "0", "0", "0"), ("1", "0", "0", "0"), ("1", "1", "0", "0"),
                                               ("1", "1", "1", "0"), ("1", "1", "0", "1"), ("1", "1", "1", "1")):
        monkeypatch.setenv("MG_NO_TILE", no_tile)
        monkeypatch.setenv("MG_NO_WINDOW", no_win)
        monkeypatch.setenv("MG_NO_IMPLICIT_FIRST", no_first)
        monkeypatch.setenv("MG_NO_LANE", no_lane)              # csr_rowclass_lane_spmv / the waterfall kernel
        A, p, b = _setup(mg, cells, levels)
        h = mg.to_device(p)
        for l in range(1, p.levels):
            Al, Pl, Rl, dl = p.As[l - 1], p.Ps[l - 1], p.Rs[l - 1], p.relaxPrecs[l - 1]
            seen.add(h.operator_kernel_variant(l, D.MG_OP_A))
            xn, bn = rng.standard_normal(Al.shape[0]), rng.standard_normal(Al.shape[0])
            x, bb = torch.from_numpy(xn).cuda(), torch.from_numpy(bn).cuda()
            out = torch.zeros_like(x)
            h.fused_dev(l, D.MG_K_RESIDUAL, bb, x, out)
            want = bn - Al @ xn
            assert np.abs(out.cpu().numpy() - want).max() / np.abs(want).max() < KERNEL_TOL
            h.fused_dev(l, D.MG_K_SMOOTH, bb, x, out)
            want = xn + dl * (bn - Al @ xn)
            assert np.abs(out.cpu().numpy() - want).max() / np.abs(want).max() < KERNEL_TOL
            xc = rng.standard_normal(Pl.shape[1])
            got = mg.SpMatMul(p, l, "P", xc, xn.copy(), 1.0, 1.0)
            want = xn + Pl @ xc
            assert np.abs(got - want).max() / np.abs(want).max() < KERNEL_TOL
            got = mg.SpMatMul(p, l, "R", xn, np.zeros(Rl.shape[0]), 1.0, 0.0)
            want = Rl @ xn
            assert np.abs(got - want).max() / np.abs(want).max() < KERNEL_TOL
        _compare_solve(mg, p, b)
        mg.clear_(p)
    assert 0 in seen and 4 in seen        # the waterfall and the per-lane row-class kernels both ran


@pytest.mark.gpu
@pytest.mark.parametrize("cyc,cells,levels", [("V", [16, 16, 16], 3), ("W", [24, 20], 3), ("V", [32, 32, 32], 2)])
def test_gmres_coarse_solve(mg, built, cyc, cells, levels):
    """coarseSolveType "GMRES": Jacobi-preconditioned FGMRES(10), one restart, tol 1e-2, on the coarsest level
    (mg_set_coarse_gmres_FP64 <-> MGcycle.jl:152-168) against the oracle's restatement, one right-hand side (fgmres) and
    blocks (blockFGMRES, l.164-166)."""
    A, mesh = mg.poisson_shifted(cells)
    p = mg.getMGparam(np.float64, np.int64, levels, 8, 8, 1e-10, "Jac", 0.8, 2, 1, cyc, "GMRES", 0.5, 0.0)
    mg.MGsetup(A, mesh, p, 1)
    b = mg.seeded_rhs(A, 1)
    _compare_solve(mg, p, b)
    mg.clear_(p)
    # a block of right-hand sides takes the blockFGMRES branch (MGcycle.jl:164-166)
    for nrhs in (2, 3):
        mg.MGsetup(A, mesh, p, nrhs)
        b2 = mg.seeded_rhs(A, nrhs)
        _compare_solve(mg, p, b2)
        mg.clear_(p)


@pytest.mark.gpu
def test_wrong_grid_hint_changes_nothing(mg, built, monkeypatch):
    """The grid hint only selects a row partition and what is staged in LDS: with a hint that multiplies out to the
    row count but describes the wrong grid (dimensions permuted), entries whose shift is not in the staged set gather
    from global memory and the solve is still the oracle's."""
    from multigrid_jl_amd import device as D
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_TILE_MIN_WG", "0")
    A, p, b = _setup(mg, [40, 30, 5], 2)                    # nodes 41 x 31 x 6
    h = mg.to_device(p)
    assert h.operator_kernel_variant(1, D.MG_OP_A) == 2     # plane tiles with the true hint
    for wrong in ((6, 31, 41), (31, 41, 6), (41 * 31, 3, 2)):
        rc = h.lib.mg_set_grid_hint(h.handle, 1, *wrong)
        assert rc == 0
        assert h.lib.mg_finalize(h.handle) == 0
        _compare_solve(mg, p, b)
    mg.clear_(p)


@pytest.mark.gpu
@pytest.mark.parametrize("cells,levels,cyc,pre,post", [([33, 25, 7], 2, "V", 2, 1), ([40, 30, 9], 3, "W", 1, 2), ([23, 23, 23], 3, "V", 2, 1),
                                                       ([70, 10, 12], 2, "F", 2, 2), ([64, 64, 20], 3, "V", 2, 1), ([31, 37], 3, "V", 2, 1)])
def test_march_kernel_vs_plane_tiles(mg, built, monkeypatch, cells, levels, cyc, pre, post):
    """csr_rowclass_march_spmv (z-marching ring of slabs): kernel-level products against scipy, the solve against the oracle, and
    bit-identical iterates against the plane-tile kernel (same products, same order).  (Until round 5 this test also forced the
    coarse-grid correction fused into the staging of the first post-smoothing sweep - measured slower in every round, retired in round 6.)"""
    import torch
    from multigrid_jl_amd import device as D
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_ROWCLASS_MIN_COVER", "0.05")
    monkeypatch.setenv("MG_TILE_MIN_WG", "0")
    monkeypatch.setenv("MG_WINDOW_MIN_WG", "0")
    monkeypatch.setenv("MG_MARCH_MIN_WG", "0")
    monkeypatch.setenv("MG_MARCH_MAX_LEN", "64")         # also on the 27-point coarse levels
    monkeypatch.setenv("MG_PAIR_MIN_ROWS", "0")
    rng = np.random.default_rng(sum(cells) + 1)
    runs = {}
    for name, no_march in (("march", "0"), ("tile", "1")):
        monkeypatch.setenv("MG_NO_MARCH", no_march)
        A, p, b = _setup(mg, cells, levels, "Jac", 0.8, pre, post, cyc, maxIter=5)
        h = mg.to_device(p)
        var = h.operator_kernel_variant(1, D.MG_OP_A)
        if len(cells) == 3:
            assert var == (3 if no_march == "0" else 2), var
        for l in range(1, p.levels):
            Al, dl = p.As[l - 1], p.relaxPrecs[l - 1]
            xn, bn = rng.standard_normal(Al.shape[0]), rng.standard_normal(Al.shape[0])
            x, bb = torch.from_numpy(xn).cuda(), torch.from_numpy(bn).cuda()
            out = torch.zeros_like(x)
            h.fused_dev(l, D.MG_K_RESIDUAL, bb, x, out)
            want = bn - Al @ xn
            assert np.abs(out.cpu().numpy() - want).max() / np.abs(want).max() < KERNEL_TOL
            h.fused_dev(l, D.MG_K_SMOOTH, bb, x, out)
            want = xn + dl * (bn - Al @ xn)
            assert np.abs(out.cpu().numpy() - want).max() / np.abs(want).max() < KERNEL_TOL
            y = torch.from_numpy(bn.copy()).cuda()
            h.spmv_dev(l, D.MG_OP_A, -0.5, x, 2.0, y)
            want = -0.5 * (Al @ xn) + 2.0 * bn
            assert np.abs(y.cpu().numpy() - want).max() / np.abs(want).max() < KERNEL_TOL
        x, hist = _compare_solve(mg, p, b)
        # non-zero initial guess (the first sweep is a full fused sweep then) through the device entry point
        x0 = np.random.default_rng(99).standard_normal(b.shape)
        x1 = x0.copy()
        mg.recursiveCycle(p, b, x1, 1)
        xo = orc.recursiveCycle(p, b, x0.copy(), 1)
        assert np.abs(x1 - xo).max() <= RES_TOL * np.abs(xo).max()
        runs[name] = (x.copy(), np.asarray(p.resvec).copy(), x1.copy())
        mg.clear_(p)
    assert np.array_equal(runs["march"][0], runs["tile"][0])          # same products, same order
    assert np.array_equal(runs["march"][2], runs["tile"][2])
    assert np.abs(runs["march"][1] - runs["tile"][1]).max() <= 1e-14 * runs["tile"][1][0]


@pytest.mark.gpu
@pytest.mark.parametrize("cells,levels,cyc,pre,post", [([33, 25, 7], 2, "V", 2, 1), ([40, 30, 9], 3, "W", 1, 2), ([23, 23, 23], 3, "V", 2, 1),
                                                       ([70, 10, 12], 2, "F", 2, 2), ([64, 64, 20], 3, "V", 2, 1), ([36, 36, 40], 3, "V", 3, 3),
                                                       ([20, 20, 3], 2, "V", 1, 1)])
def test_march2_sweep_and_residual_in_one_pass(mg, built, monkeypatch, cells, levels, cyc, pre, post):
    """csr_rowclass_march2_spmv (temporal blocking: the last sweep of relax and the residual that follows it in one walk
    along z): kernel-level t, r, t + d.*r against numpy and BIT-identical to the two single-stage launches; the solve
    against the oracle and bit-identical iterates against the unfused path (MG_NO_MARCH2=1)."""
    import torch
    from multigrid_jl_amd import device as D
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_ROWCLASS_MIN_COVER", "0.05")
    monkeypatch.setenv("MG_MARCH_MIN_WG", "0")
    monkeypatch.setenv("MG_MARCH_MAX_LEN", "64")         # also on the 27-point coarse levels
    rng = np.random.default_rng(sum(cells) + 7)
    runs = {}
    for name, no2, nozero in (("fused", "0", "0"), ("unfused", "1", "0"), ("fused, dscale launched", "0", "1")):
        monkeypatch.setenv("MG_NO_MARCH2", no2)
        monkeypatch.setenv("MG_NO_MARCH2_ZERO", nozero)   # two sweeps from x = 0: x1 = d.*b formed inside the pass, or by dscale
        monkeypatch.setenv("MG_NO_RESTRICT_SCALE", nozero)   # the restriction also writes the coarse level's d.*bc, or dscale does
        A, p, b = _setup(mg, cells, levels, "Jac", 0.8, pre, post, cyc, maxIter=5)
        h = mg.to_device(p)
        assert h.operator_kernel_variant(1, D.MG_OP_A) == 3
        served = 0
        for l in range(1, p.levels):
            Al, dl = p.As[l - 1], p.relaxPrecs[l - 1]
            xn_, bn = rng.standard_normal(Al.shape[0]), rng.standard_normal(Al.shape[0])
            x, bb = torch.from_numpy(xn_).cuda(), torch.from_numpy(bn).cuda()
            t, r, xn = torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(x)
            try:
                nrm = h.sweep_residual_dev(l, bb, x, t, r, xn, True)
            except D.MGDeviceError:
                assert no2 == "1" or h.operator_kernel_variant(l, D.MG_OP_A) != 3
                continue
            assert no2 == "0"
            served += 1
            t_want = xn_ + dl * (bn - Al @ xn_)
            r_want = bn - Al @ t_want
            assert np.abs(t.cpu().numpy() - t_want).max() / np.abs(t_want).max() < KERNEL_TOL
            assert np.abs(r.cpu().numpy() - r_want).max() / np.abs(r_want).max() < 10 * KERNEL_TOL
            assert np.abs(xn.cpu().numpy() - (t_want + dl * r_want)).max() / np.abs(t_want).max() < 10 * KERNEL_TOL
            assert abs(nrm - np.linalg.norm(r_want)) < 1e-12 * np.linalg.norm(r_want)
            t1, r1 = torch.zeros_like(x), torch.zeros_like(x)
            h.fused_dev(l, D.MG_K_SMOOTH, bb, x, t1)
            h.fused_dev(l, D.MG_K_RESIDUAL, bb, t1, r1)
            assert torch.equal(t, t1) and torch.equal(r, r1)          # same products, same order
            # outputs that are not asked for are not written
            t2 = torch.zeros_like(x)
            h.sweep_residual_dev(l, bb, x, t2)
            assert torch.equal(t2, t)
        if no2 == "0":
            assert served >= 1
        x, hist = _compare_solve(mg, p, b)
        x0 = np.random.default_rng(99).standard_normal(b.shape)
        x1 = x0.copy()
        mg.recursiveCycle(p, b, x1, 1)
        xo = orc.recursiveCycle(p, b, x0.copy(), 1)
        assert np.abs(x1 - xo).max() <= RES_TOL * np.abs(xo).max()
        # a solve from a non-zero initial guess (first step: full residual, then the fused tail)
        x2 = x0.copy()
        mg.solveMG(p, b, x2)
        runs[name] = (x.copy(), np.asarray(p.resvec).copy(), x1.copy(), x2.copy())
        mg.clear_(p)
    for other in ("unfused", "fused, dscale launched"):
        assert np.array_equal(runs["fused"][0], runs[other][0]), other
        assert np.array_equal(runs["fused"][2], runs[other][2]), other
        assert np.array_equal(runs["fused"][3], runs[other][3]), other
        assert np.abs(runs["fused"][1] - runs[other][1]).max() <= 1e-14 * runs[other][1][0], other


@pytest.mark.gpu
@pytest.mark.parametrize("cells,k1,nt,tiles_x,lockstep", [([33, 25, 7], 2, 1024, 0, 0), ([33, 25, 7], 3, 768, 2, 1), ([40, 30, 9], 4, 768, 3, 0),
                                                          ([23, 23, 23], 3, 1024, 1, 1), ([40, 32, 10], 4, 768, 0, 1), ([20, 20, 3], 2, 1024, 1, 1),
                                                          ([257, 9, 4], 3, 1024, 0, 1)])
def test_march3_two_stage_pass_on_inplane_tiles(mg, built, monkeypatch, cells, k1, nt, tiles_x, lockstep):
    """csr_rowclass_march3_spmv (sweep + residual in one pass on 2-D in-plane tiles; the z-1 / z+1 entries from registers):
    t, r, t + d.*r and ||r|| against numpy, BIT-identical to the 1-D chunk form (MG_NO_MARCH3=1) and to the two
    single-stage launches, from a given x and from x = 0 (x1 = d.*b formed inside the pass); the solve against the oracle
    with bit-identical iterates.  Tile geometries forced through the options: partial tiles at both far edges, one tile
    per line, tiles narrower than a wavefront's worth of lanes, two to four rows per lane, 768 and 1024 threads, the
    lockstep and the balanced schedule.  The class ids come from the verified product map, never from a stream."""
    import torch
    from multigrid_jl_amd import device as D
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_ROWCLASS_MIN_COVER", "0.05")
    monkeypatch.setenv("MG_MARCH_MIN_WG", "0")
    monkeypatch.setenv("MG_MARCH_MAX_LEN", "64")
    monkeypatch.setenv("MG_MARCH3_K1", str(k1))
    monkeypatch.setenv("MG_MARCH3_NT", str(nt))
    monkeypatch.setenv("MG_MARCH3_TILES_X", str(tiles_x))
    monkeypatch.setenv("MG_NO_MARCH3_LOCKSTEP", "0" if lockstep else "1")
    monkeypatch.setenv("MG_MARCH3_LOCKSTEP_FORCE", "1" if lockstep else "0")
    rng = np.random.default_rng(sum(cells) + 11)
    xn_ = bn = None
    runs, outs = {}, {}
    for name, no3 in (("tiles", "0"), ("chunks", "1")):
        monkeypatch.setenv("MG_NO_MARCH3", no3)
        A, p, b = _setup(mg, cells, 2, "Jac", 0.8, 2, 1, "V", maxIter=5)
        h = mg.to_device(p)
        form, geo = h.sweep_residual_form(1)
        assert form == (3 if no3 == "0" else 2), (form, geo)
        if no3 == "0":
            assert geo[4] == k1 and geo[8] == nt and (tiles_x == 0 or geo[0] == tiles_x) and (geo[9] > 0) == bool(lockstep), geo
        Al, dl = p.As[0], p.relaxPrecs[0]
        if xn_ is None:
            xn_, bn = rng.standard_normal(Al.shape[0]), rng.standard_normal(Al.shape[0])
        x, bb = torch.from_numpy(xn_).cuda(), torch.from_numpy(bn).cuda()
        t, r, xn = torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(x)
        nrm = h.sweep_residual_dev(1, bb, x, t, r, xn, True)
        t_want = xn_ + dl * (bn - Al @ xn_)
        r_want = bn - Al @ t_want
        assert np.abs(t.cpu().numpy() - t_want).max() / np.abs(t_want).max() < KERNEL_TOL
        assert np.abs(r.cpu().numpy() - r_want).max() / np.abs(r_want).max() < 10 * KERNEL_TOL
        assert np.abs(xn.cpu().numpy() - (t_want + dl * r_want)).max() / np.abs(t_want).max() < 10 * KERNEL_TOL
        assert abs(nrm - np.linalg.norm(r_want)) < 1e-12 * np.linalg.norm(r_want)
        t1, r1 = torch.zeros_like(x), torch.zeros_like(x)
        h.fused_dev(1, D.MG_K_SMOOTH, bb, x, t1)
        h.fused_dev(1, D.MG_K_RESIDUAL, bb, t1, r1)
        assert torch.equal(t, t1) and torch.equal(r, r1)          # same products, same order
        # single outputs (the template variants that keep only one pending store per row)
        t2, r2 = torch.zeros_like(x), torch.zeros_like(x)
        h.sweep_residual_dev(1, bb, x, t2, r2)
        assert torch.equal(t2, t) and torch.equal(r2, r)
        t3, x3 = torch.zeros_like(x), torch.zeros_like(x)
        h.sweep_residual_dev(1, bb, x, t3, None, x3)
        assert torch.equal(t3, t) and torch.equal(x3, xn)
        outs[name] = (t.cpu().numpy(), r.cpu().numpy(), xn.cpu().numpy())
        x_, hist = _compare_solve(mg, p, b)
        x0 = np.random.default_rng(99).standard_normal(b.shape)
        x1 = x0.copy()
        mg.recursiveCycle(p, b, x1, 1)
        xo = orc.recursiveCycle(p, b, x0.copy(), 1)
        assert np.abs(x1 - xo).max() <= RES_TOL * np.abs(xo).max()
        x2 = x0.copy()
        mg.solveMG(p, b, x2)
        runs[name] = (x_.copy(), np.asarray(p.resvec).copy(), x1.copy(), x2.copy())
        mg.clear_(p)
    for k in range(3):
        assert np.array_equal(outs["tiles"][k], outs["chunks"][k]), k
    for k in (0, 2, 3):
        assert np.array_equal(runs["tiles"][k], runs["chunks"][k]), k
    assert np.abs(runs["tiles"][1] - runs["chunks"][1]).max() <= 1e-14 * runs["chunks"][1][0]


def _setup_divsiggrad(mg, cells, levels, relaxType="Jac", omega=0.8, pre=2, post=1, cyc="V", maxIter=6, seed=5):
    """Nodal div sigma grad with a log-normal cell coefficient (testGMG.jl:57-75 / testSAforDivSigGrad.jl:96-100 idiom) + a
    small shift: every row of the 7-point operator has its own values - no row classes."""
    import scipy.sparse as sp
    mesh = mg.getRegularMesh([0.0, 1.0] * len(cells), cells)
    sigma = np.exp(np.random.default_rng(seed).standard_normal(int(np.prod(cells))))
    A = mg.getNodalDivSigGradMatrix(mesh, sigma)
    A = (A + 1e-3 * abs(A).sum(axis=0).max() * sp.identity(A.shape[0], format="csr")).tocsr()
    A.sort_indices()
    p = mg.getMGparam(np.float64, np.int64, levels, 8, maxIter, 1e-10, relaxType, omega, pre, post, cyc, "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p, 1)
    return A, p, mg.seeded_rhs(A, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("cells,k1,tiles_x,lockstep,relax", [([33, 25, 7], 2, 0, 0, "Jac"), ([40, 30, 9], 3, 2, 1, "Jac"), ([23, 23, 23], 2, 1, 1, "SPAI"),
                                                             ([130, 9, 5], 2, 0, 0, "Jac")])
def test_band_form_variable_coefficients(mg, built, monkeypatch, cells, k1, tiles_x, lockstep, relax):
    """Grid operators whose coefficients differ from row to row (div sigma grad: what jInv feeds the package) have no row
    classes; round 2 ran them through the pattern-coded CSR kernels only.  Band form: structure classes as a verified product
    map, the values in 7 planar arrays, relaxPrec per row, sweep + residual in ONE pass of csr_rowclass_march3_spmv<VAR>
    (the matrix is streamed once for both stages).  t, r, t + d.*r and ||r|| against numpy and against the CSR kernels
    (MG_NO_BAND=1; those add rounded products, this one fused multiply-adds: 1e-13, not bits); solve and cycle against the
    oracle, from x = 0 (x1 = d.*b formed inside the pass) and from a given x."""
    import torch
    import scipy.sparse as sp
    from multigrid_jl_amd import device as D
    monkeypatch.setenv("MG_BAND_MIN_ROWS", "0")
    monkeypatch.setenv("MG_MARCH_MIN_WG", "0")
    monkeypatch.setenv("MG_MARCH3_K1", str(k1))
    monkeypatch.setenv("MG_MARCH3_TILES_X", str(tiles_x))
    monkeypatch.setenv("MG_NO_MARCH3_LOCKSTEP", "0" if lockstep else "1")
    monkeypatch.setenv("MG_MARCH3_LOCKSTEP_FORCE", "1" if lockstep else "0")
    rng = np.random.default_rng(sum(cells) + 3)
    outs, runs = {}, {}
    xn_ = bn = None
    for name, off in (("band", "0"), ("band7", "0"), ("csr", "1")):
        monkeypatch.setenv("MG_NO_BAND", off)
        monkeypatch.setenv("MG_NO_BAND_SYM", "1" if name == "band7" else "0")
        A, p, b = _setup_divsiggrad(mg, cells, 2, relax, 0.8 if relax == "Jac" else 1.0)
        h = mg.to_device(p)
        form, geo = h.sweep_residual_form(1)
        assert form == (4 if off == "0" else 0), (form, geo)
        # G' diag(sigma) G is symmetric entry by entry: the pass reads the lower entries from the neighbours' upper ones (4 planes of 7)
        assert h.band_form(1) == ([1, 1, 1, 4] if name == "band" else [1, 1, 0, 7] if name == "band7" else [0, 0, 0, 0])
        assert h.operator_rowclasses(1, D.MG_OP_A)[0] == 0          # really no row classes
        Al, dl = p.As[0], p.relaxPrecs[0]
        if xn_ is None:
            xn_, bn = rng.standard_normal(Al.shape[0]), rng.standard_normal(Al.shape[0])
        t_want = xn_ + dl * (bn - Al @ xn_)
        r_want = bn - Al @ t_want
        if off == "0":
            assert geo[4] == k1 and geo[8] == 512 and (tiles_x == 0 or geo[0] == tiles_x) and (geo[9] > 0) == bool(lockstep), geo
            x, bb = torch.from_numpy(xn_).cuda(), torch.from_numpy(bn).cuda()
            t, r, xn = torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(x)
            nrm = h.sweep_residual_dev(1, bb, x, t, r, xn, True)
            assert np.abs(t.cpu().numpy() - t_want).max() / np.abs(t_want).max() < KERNEL_TOL
            assert np.abs(r.cpu().numpy() - r_want).max() / np.abs(r_want).max() < 10 * KERNEL_TOL
            assert np.abs(xn.cpu().numpy() - (t_want + dl * r_want)).max() / np.abs(t_want).max() < 10 * KERNEL_TOL
            assert abs(nrm - np.linalg.norm(r_want)) < 1e-12 * np.linalg.norm(r_want)
            t1, r1 = torch.zeros_like(x), torch.zeros_like(x)
            h.fused_dev(1, D.MG_K_SMOOTH, bb, x, t1)
            h.fused_dev(1, D.MG_K_RESIDUAL, bb, t1, r1)
            assert (t - t1).abs().max().item() <= KERNEL_TOL * np.abs(t_want).max()
            assert (r - r1).abs().max().item() <= 10 * KERNEL_TOL * np.abs(r_want).max()
            t2, r2 = torch.zeros_like(x), torch.zeros_like(x)
            h.sweep_residual_dev(1, bb, x, t2, r2)
            assert torch.equal(t2, t) and torch.equal(r2, r)
            t3, x3 = torch.zeros_like(x), torch.zeros_like(x)
            h.sweep_residual_dev(1, bb, x, t3, None, x3)
            assert torch.equal(t3, t) and torch.equal(x3, xn)
        x_, hist = _compare_solve(mg, p, b)
        x0 = np.random.default_rng(99).standard_normal(b.shape)
        x1 = x0.copy()
        mg.recursiveCycle(p, b, x1, 1)
        xo = orc.recursiveCycle(p, b, x0.copy(), 1)
        assert np.abs(x1 - xo).max() <= RES_TOL * np.abs(xo).max()
        runs[name] = (x_.copy(), np.asarray(p.resvec).copy(), x1.copy())
        if off == "0":
            # sigma changes (what replaceMatrixInHierarchy exists for, MGsetup.jl:226-270): same pattern, new values - the
            # planar value arrays of the resident hierarchy must follow
            mesh = mg.getRegularMesh([0.0, 1.0] * len(cells), cells)
            A2 = mg.getNodalDivSigGradMatrix(mesh, np.exp(np.random.default_rng(77).standard_normal(int(np.prod(cells)))))
            A2 = (A2 + 1e-3 * abs(A2).sum(axis=0).max() * sp.identity(A2.shape[0], format="csr")).tocsr()
            A2.sort_indices()
            assert np.array_equal(A2.indices, A.indices)
            dev_before = p.device
            mg.replaceMatrixInHierarchy(p, A2)
            assert p.device is dev_before and p.device.sweep_residual_form(1)[0] == 4
            _compare_solve(mg, p, mg.seeded_rhs(A2, 1))
        mg.clear_(p)
    for q in range(3):        # the symmetric reads substitute equal values: the same bits as the 7-plane reads
        assert np.array_equal(runs["band"][q], runs["band7"][q])
    assert np.abs(runs["band"][0] - runs["csr"][0]).max() <= 1e-11 * np.abs(runs["csr"][0]).max()
    assert np.abs(runs["band"][1] - runs["csr"][1]).max() <= 1e-11 * runs["csr"][1][0]
    assert np.abs(runs["band"][2] - runs["csr"][2]).max() <= 1e-11 * np.abs(runs["csr"][2]).max()


@pytest.mark.gpu
def test_band_form_of_a_nonsymmetric_operator_reads_seven_planes(mg, built, monkeypatch):
    """The symmetric reads of the band form are decided by a device check of the VALUES: rows of div sigma grad scaled by a random
    diagonal keep the 7-point structure (canonical slots) but are no longer symmetric - the pass must stream all 7 planes, and
    t, r, ||r|| must still agree with numpy; after the symmetric values are put back (replaceMatrixInHierarchy) it reads 4."""
    import torch
    import scipy.sparse as sp
    monkeypatch.setenv("MG_BAND_MIN_ROWS", "0")
    monkeypatch.setenv("MG_MARCH_MIN_WG", "0")
    cells = [33, 25, 9]
    A, p, b = _setup_divsiggrad(mg, cells, 2)
    rng = np.random.default_rng(8)
    Ans = (sp.diags(np.exp(0.3 * rng.standard_normal(A.shape[0]))) @ A).tocsr()
    Ans.sort_indices()
    mesh = mg.getRegularMesh([0.0, 1.0] * 3, cells)
    mg.MGsetup(Ans, mesh, p, 1)
    h = mg.to_device(p)
    assert h.sweep_residual_form(1)[0] == 4 and h.band_form(1) == [1, 1, 0, 7]
    Al, dl = p.As[0], p.relaxPrecs[0]
    xn_, bn = rng.standard_normal(Al.shape[0]), rng.standard_normal(Al.shape[0])
    t_want = xn_ + dl * (bn - Al @ xn_)
    r_want = bn - Al @ t_want
    x, bb = torch.from_numpy(xn_).cuda(), torch.from_numpy(bn).cuda()
    t, r = torch.zeros_like(x), torch.zeros_like(x)
    nrm = h.sweep_residual_dev(1, bb, x, t, r, None, True)
    assert np.abs(t.cpu().numpy() - t_want).max() / np.abs(t_want).max() < KERNEL_TOL
    assert np.abs(r.cpu().numpy() - r_want).max() / np.abs(r_want).max() < 10 * KERNEL_TOL
    assert abs(nrm - np.linalg.norm(r_want)) < 1e-12 * np.linalg.norm(r_want)
    _compare_solve(mg, p, mg.seeded_rhs(Ans, 1))
    mg.replaceMatrixInHierarchy(p, A)                       # symmetric values, same pattern: the check runs again
    assert p.device.band_form(1) == [1, 1, 1, 4]
    _compare_solve(mg, p, b)
    mg.clear_(p)


@pytest.mark.gpu
def test_march3_serves_513_node_lines(mg, built):
    """A 512^3-cell grid on ONE GPU has 513-node lines: the 1-D chunk form of the two-stage pass cannot stage them (its halo
    is a whole line: LDS) and round 2 fell back to two launches of the lane kernel there.  The 2-D tile form cuts the line
    into tiles, so the same kernel serves it - checked on a 512 x 512 x 8 slab of such a grid with DEFAULT thresholds: the
    fused pass is selected, t and r agree with scipy and are bit-identical to the two single-stage launches."""
    import torch
    from multigrid_jl_amd import device as D
    A, p, b = _setup(mg, [512, 512, 8], 5, "Jac", 0.8, 2, 1, "V", maxIter=2)      # (coarsest: 33 x 33 x 2 nodes)
    h = mg.to_device(p)
    form, geo = h.sweep_residual_form(1)
    assert form == 3 and geo[0] >= 4 and geo[2] <= 160, (form, geo)         # >= 4 tiles per 513-node line
    Al, dl = p.As[0], p.relaxPrecs[0]
    rng = np.random.default_rng(5)
    xn_, bn = rng.standard_normal(Al.shape[0]), rng.standard_normal(Al.shape[0])
    x, bb = torch.from_numpy(xn_).cuda(), torch.from_numpy(bn).cuda()
    t, r = torch.zeros_like(x), torch.zeros_like(x)
    nrm = h.sweep_residual_dev(1, bb, x, t, r, None, True)
    t_want = xn_ + dl * (bn - Al @ xn_)
    r_want = bn - Al @ t_want
    assert np.abs(t.cpu().numpy() - t_want).max() / np.abs(t_want).max() < KERNEL_TOL
    assert np.abs(r.cpu().numpy() - r_want).max() / np.abs(r_want).max() < 10 * KERNEL_TOL
    assert abs(nrm - np.linalg.norm(r_want)) < 1e-12 * np.linalg.norm(r_want)
    t1, r1 = torch.zeros_like(x), torch.zeros_like(x)
    h.fused_dev(1, D.MG_K_SMOOTH, bb, x, t1)
    h.fused_dev(1, D.MG_K_RESIDUAL, bb, t1, r1)
    assert torch.equal(t, t1) and torch.equal(r, r1)
    mg.clear_(p)


@pytest.mark.gpu
@pytest.mark.parametrize("cells,levels", [([32, 32, 32], 3), ([33, 25, 7], 2), ([40, 30, 9], 3), ([23, 23, 23], 3), ([31, 16, 12], 2),
                                          ([64, 64, 20], 3), ([70, 10, 12], 2)])
def test_prolongation_with_staged_coarse_windows(mg, built, monkeypatch, cells, levels):
    """csr_rowclass_winp_spmv (x += P*xc with the coarse windows of a workgroup's rows staged in LDS, tables derived from
    P's pattern at upload): products against scipy for several (alpha, beta), bit-identical to the lane kernel's
    (MG_NO_WINP=1), and the solve against the oracle - on odd and even node counts (the even-count variants of
    getFWInterp, GeometricTransferOperators.jl:35-36, either fit the windows or fall back)."""
    import torch
    from multigrid_jl_amd import device as D
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_ROWCLASS_MIN_COVER", "0.05")
    monkeypatch.setenv("MG_WINP_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_CELL_PROLONG", "1")    # (... and the lane-per-coarse-cell kernel, which takes x += P xc of odd-count grid pairs)
    monkeypatch.setenv("MG_NO_WAVE_RESTRICT", "1")
    outs = {}
    for name, no in (("windows", "0"), ("lane", "1")):
        rng = np.random.default_rng(sum(cells) + 3)           # the same vectors for both runs
        monkeypatch.setenv("MG_NO_WINP", no)
        A, p, b = _setup(mg, cells, levels, maxIter=5)
        h = mg.to_device(p)
        res = []
        for l in range(1, p.levels):
            P = p.Ps[l - 1]
            Pm = P if P.shape[0] == p.As[l - 1].shape[0] else P.T          # fine x coarse
            R = p.Rs[l - 1]
            Rm = R if R.shape[1] == p.As[l - 1].shape[0] else R.T          # coarse x fine
            for which, M in ((D.MG_OP_P, Pm), (D.MG_OP_R, Rm)):
                xin = rng.standard_normal(M.shape[1])
                y0 = rng.standard_normal(M.shape[0])
                for alpha, beta in ((1.0, 1.0), (1.0, 0.0), (-0.5, 2.0)):
                    y = torch.from_numpy(y0.copy()).cuda()
                    h.spmv_dev(l, which, alpha, torch.from_numpy(xin).cuda(), beta, y)
                    want = alpha * (M @ xin) + beta * y0
                    got = y.cpu().numpy()
                    assert np.abs(got - want).max() / np.abs(want).max() < KERNEL_TOL
                    res.append(got)
        x, hist = _compare_solve(mg, p, b)
        outs[name] = (res, x.copy(), np.asarray(p.resvec).copy())
        mg.clear_(p)
    for a, bb in zip(outs["windows"][0], outs["lane"][0]):
        assert np.array_equal(a, bb)
    assert np.array_equal(outs["windows"][1], outs["lane"][1])
    assert np.array_equal(outs["windows"][2], outs["lane"][2])


@pytest.mark.gpu
@pytest.mark.parametrize("cells,levels", [([56, 56, 16], 3), ([48, 40, 20], 3), ([33, 65, 12], 3)])
def test_plane_tiles_of_256_rows_for_small_levels(mg, built, monkeypatch, cells, levels):
    """csr_rowclass_tile_spmv<..., 256>: a level whose 1024-row tiles would be too few for the chip takes 256-row tiles
    (same kernel, a quarter of the threads per workgroup).  Forced here by the workgroup threshold; bit-identical to the
    lane kernel that serves the level otherwise, and to the oracle within the solve tolerance."""
    from multigrid_jl_amd import device as D
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_ROWCLASS_MIN_COVER", "0.05")
    monkeypatch.setenv("MG_NO_MARCH", "1")
    monkeypatch.setenv("MG_TILE_MIN_WG", "32")
    runs = {}
    for name, no_small in (("tiles256", "0"), ("lane", "1")):
        monkeypatch.setenv("MG_NO_TILE_SMALL", no_small)
        A, p, b = _setup(mg, cells, levels, maxIter=5)
        h = mg.to_device(p)
        var = h.operator_kernel_variant(1, D.MG_OP_A)
        assert var == (2 if no_small == "0" else 4), var          # plane tiles / lane kernel
        x, hist = _compare_solve(mg, p, b)
        runs[name] = (x.copy(), np.asarray(p.resvec).copy())
        mg.clear_(p)
    assert np.array_equal(runs["tiles256"][0], runs["lane"][0])
    assert np.abs(runs["tiles256"][1] - runs["lane"][1]).max() <= 1e-14 * runs["lane"][1][0]


@pytest.mark.gpu
def test_march_with_wrong_grid_hint_and_exception_rows(mg, built, monkeypatch):
    """The marching kernel with a hint that describes the wrong grid (unstaged shifts gather from global memory) and
    with a few perturbed rows (exception rows computed from the CSR arrays): the solve is still the oracle's."""
    import scipy.sparse as sp
    from multigrid_jl_amd import device as D
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_ROWCLASS_KEEP_SINGLETONS", "0")
    monkeypatch.setenv("MG_MARCH_MIN_WG", "0")
    A, p, b = _setup(mg, [40, 30, 5], 2)                    # nodes 41 x 31 x 6
    h = mg.to_device(p)
    assert h.operator_kernel_variant(1, D.MG_OP_A) == 3
    for wrong in ((6, 31, 41), (31, 41, 6), (41 * 31, 3, 2)):
        assert h.lib.mg_set_grid_hint(h.handle, 1, *wrong) == 0
        assert h.lib.mg_finalize(h.handle) == 0
        _compare_solve(mg, p, b)
    mg.clear_(p)
    rng = np.random.default_rng(3)
    A, mesh = mg.poisson_shifted([24, 20, 18])
    A = A.tolil()
    for i in rng.choice(A.shape[0], size=6, replace=False):
        A[i, i] = A[i, i] * (1.0 + rng.random())
    A = sp.csr_matrix(A)
    A.sort_indices()
    p = mg.getMGparam(np.float64, np.int64, 2, 8, 6, 1e-10, "Jac", 0.8, 2, 1)
    mg.MGsetup(A, mesh, p, 1)
    bb = mg.seeded_rhs(A, 1)
    _compare_solve(mg, p, bb)
    var, nexc = p.device.operator_kernel_info(1, D.MG_OP_A)
    assert var == 3 and 6 <= nexc <= 256, (var, nexc)        # the perturbed rows (+ the singleton corner classes) in-kernel
    mg.clear_(p)


@pytest.mark.gpu
@pytest.mark.parametrize("nrhs", [2, 3, 16])
def test_rowclass_lane_spmm_block_rhs(mg, built, monkeypatch, nrhs):
    """csr_rowclass_lane_spmm (block right-hand sides on row-class operators: per-lane class walk, no matrix stream)
    against the oracle, and bit-identical to itself under the L2-tiled block order; csr_stream_spmm (MG_NO_LANE_MM=1)
    gives the same iterates to fp64 reassociation."""
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_ROWCLASS_MIN_COVER", "0.05")
    runs = {}
    for name, no_mm, budget in (("lane", "0", None), ("lane-tiled", "0", "2000"), ("stream", "1", None)):
        monkeypatch.setenv("MG_NO_LANE_MM", no_mm)
        if budget:
            monkeypatch.setenv("MG_SCHED_BUDGET", budget)
        else:
            monkeypatch.delenv("MG_SCHED_BUDGET", raising=False)
        A, p, b = _setup(mg, [20, 18, 16], 3, "Jac", 0.8, 2, 1, "W", 5, nrhs)
        x, hist = _compare_solve(mg, p, b)
        runs[name] = x.copy()
        mg.clear_(p)
    assert np.array_equal(runs["lane"], runs["lane-tiled"])           # scheduling changes nothing
    assert np.abs(runs["lane"] - runs["stream"]).max() <= 1e-12 * np.abs(runs["stream"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("cycle", ["V", "W", "F"])
def test_coarse_subcycle_graph_replay(mg, built, cycle, monkeypatch):
    """The launch-bound coarse sub-cycle is captured into a HIP graph on first use and replayed afterwards: the solve
    must give the same iterates, bit for bit, as the same hierarchy with graphs switched off (same kernels, same
    order), and the graph path must actually have run."""
    res = {}
    for no_graph in ("0", "1"):
        monkeypatch.setenv("MG_NO_GRAPH", no_graph)
        A, p, b = _setup(mg, [32, 32, 32], 4, cyc=cycle, maxIter=6)
        x = np.zeros_like(b)
        x, p, it = mg.solveMG(p, b, x)
        launches, graphs = p.device.graph_launches()
        if no_graph == "0":
            assert launches >= it and graphs >= 1
        else:
            assert launches == 0 and graphs == 0
        res[no_graph] = (x.copy(), list(p.resvec[: it + 1]))
        mg.clear_(p)
    assert np.array_equal(res["0"][0], res["1"][0])
    assert res["0"][1] == res["1"][1]


def test_periodic_in_x_operator_stays_off_the_tile_forms(mg, built, monkeypatch):
    """A coupling that wraps around a grid line (periodic in x: row (0, y, z) <-> row (n1-1, y, z)) decomposes as
    (dy, dx) = (+1, -1) / (-1, +1), keeps the in-plane entries within +-1 and factors as a product map - every check of the
    2-D tile forms passes, but at y = 0 / y = n2-1 the entry points at a line outside the grid, which the tile kernels hold as
    zeros.  Such classes must keep the tile forms (march3, band, four-stage) off the operator; the forms that index linearly
    compute it: fused sweep, residual and the sweep + residual pair against numpy, the solve against the oracle."""
    import scipy.sparse as sp
    import torch
    from multigrid_jl_amd import device as D
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_ROWCLASS_MIN_COVER", "0.05")
    monkeypatch.setenv("MG_MARCH_MIN_WG", "0")
    monkeypatch.setenv("MG_MARCH_MAX_LEN", "64")
    cells = [24, 20, 12]
    A, mesh = mg.poisson_shifted(cells)
    n1, n2, n3 = [c + 1 for c in cells]
    w = abs(A[1, 0])
    first = (np.arange(n2 * n3) * n1).astype(np.int64)
    last = first + n1 - 1
    Wm = sp.coo_matrix((np.full(first.size, -w), (first, last)), shape=A.shape)
    Dg = sp.coo_matrix((np.full(2 * first.size, w), (np.concatenate([first, last]), np.concatenate([first, last]))), shape=A.shape)
    A = (A + Wm + Wm.T + Dg).tocsr()
    A.sort_indices()
    p = mg.getMGparam(np.float64, np.int64, 2, 8, 6, 1e-10, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p, 1)
    b = mg.seeded_rhs(A, 1)
    h = mg.to_device(p)
    form, _ = h.sweep_residual_form(1)
    assert form != 3 and not h.four_stage_form(1)[0], form
    Al, dl = p.As[0], p.relaxPrecs[0]
    rng = np.random.default_rng(3)
    xh, bh = rng.standard_normal(Al.shape[0]), rng.standard_normal(Al.shape[0])
    x, bb = torch.from_numpy(xh).cuda(), torch.from_numpy(bh).cuda()
    t1, r1 = torch.zeros_like(x), torch.zeros_like(x)
    h.fused_dev(1, D.MG_K_SMOOTH, bb, x, t1)
    h.fused_dev(1, D.MG_K_RESIDUAL, bb, t1, r1)
    t_want = xh + dl * (bh - Al @ xh)
    r_want = bh - Al @ t_want
    assert np.abs(t1.cpu().numpy() - t_want).max() / np.abs(t_want).max() < KERNEL_TOL
    assert np.abs(r1.cpu().numpy() - r_want).max() / np.abs(r_want).max() < 10 * KERNEL_TOL
    if form == 2:
        t, r = torch.zeros_like(x), torch.zeros_like(x)
        h.sweep_residual_dev(1, bb, x, t, r)
        assert torch.equal(t, t1) and torch.equal(r, r1)
    _compare_solve(mg, p, b)
    mg.clear_(p)


@pytest.mark.gpu
def test_operators_with_64_bit_row_pointers(mg, built):
    """Round 6 (VERDICT r5 Missing 4; the reference is Int64 throughout, Multigrid.jl:19): operators of >= 2^31 - 4096 non-zeros are uploaded
    with 64-bit row pointers and served by the two streaming kernels (csr_stream_spmv / csr_longrow_spmv instantiated on long long positions).
    A 26 GB operator cannot sit in a test, so the option force_rowptr64 sends ORDINARY operators down that path: every product, residual and
    sweep bit-identical to the int32 instantiation of the same kernels (no_rowclass / no_pattern / no_small: the streaming formats), a general-CSR
    (SA-AMG) solve against the oracle, and the entry points that need int32 operators refusing loudly."""
    import scipy.sparse as sp
    import torch
    from multigrid_jl_amd import device as D
    from oracle import mg_oracle as orc
    rng = np.random.default_rng(23)
    # (a) single operators: short rows (row blocks), one row longer than a chunk, empty rows; long rows (csr_longrow_spmv)
    for n, m, avg in ((4000, 3500, 9), (600, 5000, 1300)):
        lens = np.clip(rng.poisson(avg, n), 0, m)
        lens[0] = 0
        lens[n // 2] = min(m, 6000)
        rows = np.repeat(np.arange(n), lens)
        cols = np.concatenate([np.sort(rng.choice(m, size=int(k), replace=False)) for k in lens])
        M = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(n, m))
        M.sort_indices()
        x, y0 = rng.standard_normal(m), rng.standard_normal(n)
        outs = []
        for env in ({"MG_FORCE_ROWPTR64": "1"}, {"MG_NO_ROWCLASS": "1", "MG_NO_PATTERN": "1", "MG_NO_CI16": "1"}):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                op = D.DeviceOperator(M, 0)
            finally:
                for k, v_ in old.items():
                    os.environ.pop(k, None) if v_ is None else os.environ.__setitem__(k, v_)
            y = torch.from_numpy(y0.copy()).cuda()
            op.apply(D.MG_K_SPMV, torch.from_numpy(x).cuda(), y, alpha=-1.5, beta=0.25)
            ref = -1.5 * (M @ x) + 0.25 * y0
            assert np.abs(y.cpu().numpy() - ref).max() <= 1e-13 * max(1.0, np.abs(ref).max())
            outs.append(y)
            op.close()
        assert torch.equal(outs[0], outs[1])
    # (b) a whole hierarchy (general CSR: SA-AMG) with every operator on 64-bit row pointers
    A, _ = mg.anisotropic_divsiggrad([20, 20, 20], weights=(1.0, 0.5, 0.25))
    p = mg.getMGparam(np.float64, np.int64, 6, 8, 5, 1e-12, "SPAI", 1.0, 1, 1, "V", "Julia", 0.4, 0.0)
    mg.SA_AMGsetup(A, p, True, 1)
    h = D.DeviceHierarchy(p, 0, 1, options={"force_rowptr64": 1})
    h0 = D.DeviceHierarchy(p, 0, 1, options={"no_rowclass": 1, "no_pattern": 1, "no_small": 1, "no_longrow": 1})
    try:
        b = torch.from_numpy(mg.seeded_rhs(A)).cuda()
        x, x0 = torch.zeros_like(b), torch.zeros_like(b)
        it, rv = h.solve_dev(b, x, 0.0, 5)
        it0, rv0 = h0.solve_dev(b, x0, 0.0, 5)
        assert np.array_equal(rv, rv0) and torch.equal(x, x0)
        hist = {}
        xo = np.zeros(A.shape[0])
        p.maxOuterIter, p.relativeTol = 5, 0.0
        orc.solveMG(p, mg.seeded_rhs(A), xo, False, hist)
        assert np.abs(rv - hist["resvec"]).max() <= 1e-10 * hist["resvec"][0]
        assert np.abs(x.cpu().numpy() - xo).max() <= 1e-10 * np.abs(xo).max()
        flag, itk, rk = h.pcg_dev(b, torch.zeros_like(b), 1e-9, 20)        # Krylov drivers run on it too
        assert flag == 0
        with pytest.raises(D.MGDeviceError, match="64-bit row pointers"):
            h.transpose_hierarchy()
    finally:
        h.close()
        h0.close()
