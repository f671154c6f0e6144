"""The oracle's hybrid-Kaczmarz restatement against the REFERENCE'S OWN compiled code: oracle/_ref/parRelax.so is
deps/src/parRelax.c built in place by oracle/Makefile (never copied into the repo).  With numCores = 1 the reference
binary is deterministic (domains in order, rows in order), and the restatement must reproduce it bit for bit on the
problem of test/Multigrid/testHybridKaczmarz.jl:8-31 (2-D div-sigma-grad, 64^2 cells, 4x4 domains, omega 0.8, 5 inner
sweeps, 2 right-hand sides) and on a 3-D case."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import mg_oracle as orc

REF = os.path.join(os.path.dirname(os.path.abspath(orc.__file__)), "_ref", "parRelax.so")
_i64p = C.POINTER(C.c_longlong)
_f64p = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint)


def ref_apply(A, Arr, x, b, invD, numit, numCores=1):
    lib = C.CDLL(REF)
    f = lib.applyHybridKaczmarz_FP64_INT64
    f.restype = None
    f.argtypes = [_i64p, _f64p, _i64p, C.c_longlong, C.c_longlong, _u32p, _f64p, _f64p, C.c_longlong, C.c_longlong, _f64p,
                  C.c_longlong, C.c_longlong]
    A = sp.csr_matrix(A)
    cp = np.ascontiguousarray(A.indptr, dtype=np.int64) + 1
    rv = np.ascontiguousarray(A.indices, dtype=np.int64) + 1
    nz = np.ascontiguousarray(A.data, dtype=np.float64)
    Arr = np.asfortranarray(Arr, dtype=np.uint32)
    n = A.shape[0]
    nrhs = 1 if x.ndim == 1 else x.shape[1]
    f(cp.ctypes.data_as(_i64p), nz.ctypes.data_as(_f64p), rv.ctypes.data_as(_i64p), Arr.shape[1], Arr.shape[0],
      Arr.ctypes.data_as(_u32p), x.ctypes.data_as(_f64p), b.ctypes.data_as(_f64p), nrhs, n, invD.ctypes.data_as(_f64p),
      int(numit), int(numCores))
    return x


def _problem(mg, cells, nrhs, seed):
    rng = np.random.default_rng(seed)
    mesh = mg.getRegularMesh([0.0, 1.0] * len(cells), cells)
    A = mg.getNodalDivSigGradMatrix(mesh, np.exp(rng.standard_normal(int(np.prod(cells)))))
    A = (A + 2e-1 * abs(A).sum(axis=0).max() * sp.identity(A.shape[0])).tocsr()      # testHybridKaczmarz.jl:22
    A.sort_indices()
    b = np.asfortranarray(A @ rng.random((A.shape[0], nrhs)))
    b /= np.linalg.norm(b)
    return A, mesh, b


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/parRelax.so not built (reference tree absent at build time)")
@pytest.mark.parametrize("cells,domains,nrhs,numit", [([64, 64], [4, 4], 2, 5), ([10, 10, 10], [2, 2, 2], 1, 3),
                                                       ([30, 20], [3, 2], 3, 2)])
def test_oracle_kaczmarz_equals_reference_binary(mg, cells, domains, nrhs, numit):
    A, mesh, b = _problem(mg, cells, nrhs, 11)
    Arr = orc.getIndicesOfCellsArray(cells, [0] * len(cells), domains)
    # every node is listed, nodes on sub-domain faces more than once (the boxes share their boundary nodes)
    listed = Arr[Arr > 0].astype(np.int64) - 1
    assert set(listed.tolist()) == set(range(A.shape[0])) and listed.size > A.shape[0]
    invD = orc.hybrid_kaczmarz_invdiag(A, 0.8)
    x_ref = np.zeros_like(b, order="F")
    ref_apply(A, Arr, x_ref, b, invD, numit, 1)
    x_orc = np.zeros_like(b, order="F")
    orc.applyHybridKaczmarz(A, Arr, x_orc, b, invD, numit)
    assert np.array_equal(x_ref, x_orc)                 # same operations in the same order: bit for bit
    assert np.linalg.norm(A @ x_orc - b) < np.linalg.norm(b)


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/parRelax.so not built")
def test_reference_test_problem_as_preconditioner(mg):
    """testHybridKaczmarz.jl:29-32: FGMRES_relaxation(Ar, b, x, 5, prec) with the hybrid-Kaczmarz preconditioner
    (x = 0, 5 inner sweeps) - the restated FGMRES with the reference binary as preconditioner reduces the residual."""
    A, mesh, b = _problem(mg, [64, 64], 2, 3)
    Arr = orc.getIndicesOfCellsArray([64, 64], [0, 0], [4, 4])
    invD = orc.hybrid_kaczmarz_invdiag(A, 0.8)

    def prec(r):
        z = np.zeros_like(r, order="F")
        return ref_apply(A, Arr, z, np.asfortranarray(r), invD, 5, 1)

    x = orc.FGMRES_relaxation(lambda z: A @ z, b.copy(), np.zeros_like(b), 5, prec, 1e-5 * np.linalg.norm(b))
    x = x[0] if isinstance(x, tuple) else x
    assert np.linalg.norm(A @ x - b) < 0.05 * np.linalg.norm(b)
