"""CPU: the host-side mirror of the reference interface and the C-ABI library (load + symbols only)."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_getMGparam_defaults_and_positional(mg):
    p = mg.getMGparam()                                    # MGdef.jl:156-157 defaults
    assert (p.levels, p.numCores, p.maxOuterIter, p.relativeTol) == (3, 8, 20, 1e-6)
    assert (p.relaxType, p.relaxParam, p.cycleType, p.coarseSolveType) == ("SPAI", 1.0, "V", "NoMUMPS")
    assert p.relaxPre(1) == 2 and p.relaxPost(3) == 2 and p.strongConnParam == 0.4
    q = mg.getMGparam(np.float64, np.int64, 4, 2, 5, 1e-2, "Jac", 0.8, lambda l: l, lambda l: 1, "W", "NoMUMPS", 0.5, 0.0)
    assert q.relaxPre(3) == 3 and q.cycleType == "W" and not mg.hierarchyExists(q)
    with pytest.raises(TypeError):
        mg.getMGparam(np.float32)
    with pytest.raises(ValueError):
        mg.getMGparam(cycleType="Z")


def test_mgsetup_structure_c1(mg):
    """C1 sizes from SURVEY.md 8: 33^3 rows / 245025 nnz, next level 17^3 / 117649 nnz."""
    A, mesh = mg.poisson_shifted([32, 32, 32])
    p = mg.getMGparam(levels=3, relaxType="Jac", relaxParam=0.8)
    mg.MGsetup(A, mesh, p)
    assert [a.shape[0] for a in p.As] == [35937, 4913, 729]
    assert [a.nnz for a in p.As] == [245025, 117649, 15625]
    assert p.Ps[0].nnz == 117649 and p.Rs[0].nnz == 117649           # P, R share nnz with the coarser A
    assert p.Ps[0].shape == (35937, 4913) and p.Rs[0].shape == (4913, 35937)
    assert np.allclose((p.Rs[0] - 0.125 * p.Ps[0].T).data, 0)           # RT = 0.5^dim * P (MGsetup.jl:60)
    assert mg.hierarchyExists(p) and p.LU is not None and p.nrhs == 1


def test_levels_shrink_when_coarsening_stops(mg):
    """P square -> param.levels = l (MGsetup.jl:84-92)."""
    A, mesh = mg.poisson_shifted([4, 4])
    p = mg.getMGparam(levels=6, relaxType="Jac", relaxParam=0.8)
    mg.MGsetup(A, mesh, p)
    assert p.levels == len(p.As) == 3 and p.As[-1].shape[0] == 4 and len(p.Ps) == 2
    assert len(p.relaxPrecs) == 3                          # relaxPrecs[l] was computed before the break (l.76)


def test_adjust_memory_and_errors(mg):
    p = mg.getMGparam()
    with pytest.raises(RuntimeError, match="Hierarchy is empty"):
        mg.adjustMemoryForNumRHS(p, 2)                     # MGsetup.jl:167-169
    A, mesh = mg.poisson_shifted([4, 4])
    mg.MGsetup(A, mesh, p, 3)
    assert p.nrhs == 3
    mg.adjustMemoryForNumRHS(p, 5)
    assert p.nrhs == 5


def test_replace_and_transpose_hierarchy(mg):
    A, mesh = mg.poisson_shifted([8, 8])
    p = mg.getMGparam(levels=3, relaxType="SPAI", relaxParam=1.0)
    mg.MGsetup(A, mesh, p)
    A2 = (A + sp.identity(A.shape[0]) * 3.0).tocsr()
    mg.replaceMatrixInHierarchy(p, A2)                     # MGsetup.jl:226-270
    assert np.allclose(p.As[1].toarray(), (p.Rs[0] @ A2 @ p.Ps[0]).toarray())
    assert np.allclose(p.relaxPrecs[0], mg.getRelaxPrec(A2, "SPAI", 1.0))
    # non-symmetric fine operator: transposeHierarchy transposes every level (MGsetup.jl:274-318)
    N = sp.diags([np.ones(A.shape[0] - 1)], [1], format="csr") * 0.1
    mg.replaceMatrixInHierarchy(p, (A + N).tocsr())
    As_before = [a.copy() for a in p.As]
    R_before = p.Rs[0].copy()
    mg.transposeHierarchy(p)
    assert p.doTranspose == 1
    for a, b in zip(p.As, As_before):
        assert abs(a - b.T).max() == 0
    assert abs(p.Ps[0] - R_before.T).max() == 0 and abs(p.Rs[0] - R_before).max() == 0   # literal l.298-299


def test_copy_clear(mg):
    A, mesh = mg.poisson_shifted([4, 4])
    p = mg.getMGparam(levels=2, relaxType="Jac", relaxParam=0.7, maxIter=7)
    mg.MGsetup(A, mesh, p)
    q = mg.copySolver(p)
    assert q.maxOuterIter == 7 and q.relaxParam == 0.7 and not mg.hierarchyExists(q)    # MGdef.jl:138-145
    mg.clear_(p)
    assert not mg.hierarchyExists(p) and p.LU is None and p.device is None


def test_unsupported_options_are_loud(mg):
    A, mesh = mg.poisson_shifted([4, 4])
    with pytest.raises(NotImplementedError):
        mg.MGsetup(A, mesh, mg.getMGparam(coarseSolveType="MUMPS", relaxType="Jac"))
    with pytest.raises(ValueError):                                   # relaxParam ./ diag(AT) needs a scalar (MGsetup.jl:334)
        mg.MGsetup(A, mesh, mg.getMGparam(coarseSolveType="GMRES", relaxType="Jac", relaxParam=[0.8, 0.8, 0.8]))
    p = mg.MGsetup(A, mesh, mg.getMGparam(coarseSolveType="GMRES", relaxType="Jac", relaxParam=0.8))
    assert np.allclose(p.LU, 0.8 / p.As[-1].diagonal())               # what defineCoarsestAinv keeps in param.LU
    with pytest.raises(NotImplementedError):
        mg.MGsetup(A, mesh, mg.getMGparam(transferOperatorType="SystemsFacesLinear"))
    with pytest.raises(ValueError):
        mg.MGsetup(A, mesh, mg.getMGparam(relaxType="hybridKaczmarzNodal"))


# ---- the C-ABI library ---------------------------------------------------------------------------------------
def _header_functions():
    txt = open(os.path.join(ROOT, "include", "mgvcycle.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mg_[A-Za-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(mg, built):
    names = _header_functions()
    assert len(names) >= 20
    lib = ctypes.CDLL(mg.device.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mgvcycle.h but not exported"
    assert sorted(mg.device.SIGNATURES) == names            # the binding covers exactly the header
    assert b"gfx950" in mg.device.load_library().mg_version()


def test_no_gpu_means_loud_failure_not_fallback(mg, built):
    """On a box without a GPU the product path must raise, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    A, mesh = mg.poisson_shifted([4, 4])
    p = mg.getMGparam(levels=2, relaxType="Jac", relaxParam=0.8)
    mg.MGsetup(A, mesh, p)
    b = mg.seeded_rhs(A)
    with pytest.raises(mg.device.MGDeviceError):
        mg.solveMG(p, b, np.zeros_like(b))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "multigrid.jl_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.lower(), f"{f} mentions the oracle"
