"""Hybrid Kaczmarz relaxation: the DomainDecomposition index functions on the CPU against the oracle's
literal restatement, and the HIP kernel through the C ABI against the reference's own binary (oracle/_ref/parRelax.so)
and the oracle."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import mg_oracle as orc


@pytest.mark.parametrize("cells,domains", [([64, 64], [4, 4]), ([10, 10, 10], [2, 2, 2]), ([30, 20], [3, 2]),
                                           ([16, 16, 16], [4, 2, 1]), ([9, 9], [3, 3])])
def test_dd_index_functions_match_restatement(mg, cells, domains):
    mesh = mg.getRegularMesh([0.0, 1.0] * len(cells), cells)
    zero = [0] * len(cells)
    assert np.array_equal(mg.getIndicesOfCellsArray(mesh, zero, domains), orc.getIndicesOfCellsArray(cells, zero, domains))
    for ic in range(1, int(np.prod(domains)) + 1):
        loc = mg.cs2loc(ic, domains)
        assert list(loc) == orc.cs2loc(ic, domains) and mg.loc2cs(loc, domains) == ic
        for ov in (zero, [1] * len(cells)):
            assert np.array_equal(mg.getNodalIndicesOfCell(domains, ov, loc, cells),
                                  orc.getNodalIndicesOfCell(domains, ov, np.asarray(loc), np.asarray(cells)))


def _problem(mg, cells, nrhs, seed):
    rng = np.random.default_rng(seed)
    mesh = mg.getRegularMesh([0.0, 1.0] * len(cells), cells)
    A = mg.getNodalDivSigGradMatrix(mesh, np.exp(rng.standard_normal(int(np.prod(cells)))))
    A = (A + 2e-1 * abs(A).sum(axis=0).max() * sp.identity(A.shape[0])).tocsr()      # testHybridKaczmarz.jl:22
    A.sort_indices()
    b = A @ rng.random((A.shape[0], nrhs))
    b = np.asfortranarray(b / np.linalg.norm(b))
    return A, mesh, (b[:, 0].copy() if nrhs == 1 else b)


@pytest.mark.gpu
@pytest.mark.parametrize("cells,domains,nrhs,numit", [([64, 64], [4, 4], 2, 5), ([10, 10, 10], [2, 2, 2], 1, 3),
                                                       ([30, 20], [3, 2], 3, 2)])
def test_device_kaczmarz_sequential_equals_reference_binary(mg, built, cells, domains, nrhs, numit):
    """One wavefront walking the sub-domains in order performs the reference's operations in the reference's order with
    separately rounded products and sums: bit-identical to oracle/_ref/parRelax.so run with numCores = 1."""
    from test_reference_parrelax import REF, ref_apply
    A, mesh, b = _problem(mg, cells, nrhs, 11)
    hk = mg.getHybridKaczmarz(np.float64, np.int64, A, mesh, domains, mg.getNodalIndicesOfCell, 0.8, 4, numit)
    hk.sequential = True
    x = np.zeros_like(b, order="F")
    mg.applyHybridKaczmarz(hk, A, b, x)
    xo = np.zeros_like(b, order="F")
    orc.applyHybridKaczmarz(A, hk.ArrIdxs, xo, b, hk.invDiag, numit)
    assert np.array_equal(x, xo)
    if os.path.exists(REF):
        xr = np.zeros_like(b, order="F")
        ref_apply(A, hk.ArrIdxs, xr, b, hk.invDiag, numit, 1)
        assert np.array_equal(x, xr)
    hk.close()


@pytest.mark.gpu
def test_device_kaczmarz_parallel_domains(mg, built):
    """One wavefront per sub-domain (the schedule of the reference's OpenMP threads): exact on an operator whose
    sub-domains do not couple, and on the reference's test problem (testHybridKaczmarz.jl:8-32) a preconditioner with
    which FGMRES_relaxation reduces the residual as it does with the reference binary."""
    from test_reference_parrelax import REF, ref_apply
    # block-diagonal: 4 independent 2-D problems, one per sub-domain list
    A1, mesh1, _ = _problem(mg, [12, 12], 1, 5)
    n1 = A1.shape[0]
    A = sp.block_diag([A1] * 4, format="csr")
    A.sort_indices()
    arr = np.zeros((n1, 4), dtype=np.uint32, order="F")
    rng = np.random.default_rng(2)
    for d in range(4):
        arr[:, d] = rng.permutation(n1) + 1 + d * n1
    hk = mg.hybridKaczmarz([4, 1], 0.8 / np.asarray(A.multiply(A).sum(axis=1)).ravel(), 4, 0.8, arr, None, 3,
                           mg.getNodalIndicesOfCell)
    b = rng.standard_normal(A.shape[0])
    x = np.zeros_like(b)
    mg.applyHybridKaczmarz(hk, A, b, x)
    xo = np.zeros_like(b)
    orc.applyHybridKaczmarz(A, arr, xo, b, hk.invDiag, 3)
    assert np.array_equal(x, xo)
    hk.close()
    # the reference's own test problem, as a preconditioner inside FGMRES_relaxation
    A, mesh, b = _problem(mg, [64, 64], 2, 3)
    hk = mg.getHybridKaczmarz(np.float64, np.int64, A, mesh, [4, 4], mg.getNodalIndicesOfCell, 0.8, 4, 5)
    prec = mg.getHybridKaczmarzPrecond(hk, A, 2)
    out = orc.FGMRES_relaxation(lambda z: A @ z, b.copy(), np.zeros_like(b), 5, lambda r: prec(r).copy(), 1e-5 * np.linalg.norm(b))
    x = out[0] if isinstance(out, tuple) else out
    res_dev = np.linalg.norm(A @ x - b)
    assert res_dev < 0.05 * np.linalg.norm(b)
    if os.path.exists(REF):
        def prec_ref(r):
            z = np.zeros_like(r, order="F")
            return ref_apply(A, hk.ArrIdxs, z, np.asfortranarray(r), hk.invDiag, 5, 4)
        out = orc.FGMRES_relaxation(lambda z: A @ z, b.copy(), np.zeros_like(b), 5, prec_ref, 1e-5 * np.linalg.norm(b))
        xr = out[0] if isinstance(out, tuple) else out
        # Sub-domains relaxed concurrently race on the nodes they share, on the device as in the reference's OpenMP
        # run: a (slightly) different preconditioner at every application.  Measured on this problem
        # (scripts/diag_kaczmarz.py): one thread / sequential device 2.3e-5, reference with 4 threads 3.5e-3, device
        # with one wavefront per sub-domain 1.8e-3.  Same order of magnitude as the 4-thread reference:
        assert res_dev < 10.0 * np.linalg.norm(A @ xr - b) + 1e-12
    hk.close()
