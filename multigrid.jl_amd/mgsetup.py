"""Geometric multigrid setup on the host (CPU), feeding the device cycle library.

Mirrors reference src/Multigrid/MGsetup.jl: ``MGsetup`` (l.7-138), ``getRelaxPrec`` (l.142-160),
``adjustMemoryForNumRHS`` (l.166-223), ``replaceMatrixInHierarchy`` (l.226-270),
``transposeHierarchy`` (l.274-318), ``defineCoarsestAinv`` (l.323-355), ``getSPAIprec`` (l.359-362).

The hierarchy is built on the CPU "exactly as the reference does" (BASELINE.json north_star); the
cycle itself never runs here - it is owned by the HIP library (``device.py``).
"""
from __future__ import annotations

import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .mgdef import MGparam, destroyCoarsestLU, _release_device
from .transfer_operators import getFWInterp


class multilevelOperatorConstructor:
    """Mirror of MGdef.jl:31-46: rediscretisation on every level instead of Galerkin."""

    def __init__(self, param, getOperator, restrictParams):
        self.param = param
        self.getOperator = getOperator
        self.restrictParams = restrictParams


def getMultilevelOperatorConstructor(param, getOperator, restrictParams):
    if restrictParams is None or (isinstance(restrictParams, (list, tuple)) and len(restrictParams) == 0):
        return multilevelOperatorConstructor(None, lambda mesh, p: getOperator(mesh),
                                             lambda mf, mc, p, level: None)
    return multilevelOperatorConstructor(param, getOperator, restrictParams)


def _as_csr(A):
    A = sp.csr_matrix(A, dtype=np.float64)
    A.sort_indices()
    return A


def getSPAIprec(A):
    """Q_i = conj(diag)_i / s_i with s_i = sum_j |AT[i,j]|^2 (MGsetup.jl:359-362).

    Row i of the reference's AT is COLUMN i of A, so s is the column-wise sum of squares of A
    (equal to the row norm only for symmetric A).
    """
    A = _as_csr(A)
    from .hostlib import col_sumsq
    s = col_sumsq(A)                                # (thread-parallel on the host; the scipy / numpy line below otherwise)
    if s is None:
        s = np.bincount(A.indices, weights=A.data * A.data, minlength=A.shape[1])
    return A.diagonal() / s


def getRelaxPrec(A, relaxType: str, relaxParam=1.0):
    """Jac: d = omega/diag (MGsetup.jl:145-147).  SPAI: d = omega*diag/colnorm^2 (l.148-149)."""
    if relaxType in ("Jac", "Jac-GMRES"):
        return np.ascontiguousarray(float(relaxParam) / _as_csr(A).diagonal(), dtype=np.float64)
    if relaxType == "SPAI":
        return np.ascontiguousarray(float(relaxParam) * getSPAIprec(A), dtype=np.float64)
    raise ValueError("Unknown relaxation type !!!!")


def _relax_param_arr(param: MGparam):
    if isinstance(param.relaxParam, (list, tuple, np.ndarray)):
        return list(param.relaxParam)
    return [param.relaxParam] * param.levels


def galerkin(R, A, P):
    """A_c = R*(A*P): the CSR view of ``Act = Ps[l]*AT*Rs[l]`` evaluated left to right (MGsetup.jl:102).
    The reference's serial Julia SpGEMM is the bulk of its setup time; here it is row-parallel on the host
    (csrc/mg_host.cpp) - still CPU, still before the device ever sees the hierarchy."""
    from .hostlib import galerkin_dense_gpu, galerkin_dense_gpu_ok, galerkin_sparse_gpu, galerkin_sparse_gpu_ok, spgemm
    # (opt-in, MG_SETUP_GPU=1: the largest products of an SA-AMG setup on the GPU - same pattern, values to rounding; hostlib.py)
    if galerkin_dense_gpu_ok(A, P):        # nearly dense levels: dense GEMMs
        Ac = galerkin_dense_gpu(R, A, P)
        if Ac is not None:
            return Ac
    if galerkin_sparse_gpu_ok(A, P):       # 10^10 products and more: rocSPARSE's SpGEMM
        Ac = galerkin_sparse_gpu(R, A, P)
        if Ac is not None:
            return Ac
    return spgemm(R, spgemm(A, P))


def defineCoarsestAinv(param: MGparam, Ac) -> None:
    """Coarsest-level factorisation (MGsetup.jl:323-355).  Default branch: ``lu(sparse(AT'))`` (l.350)."""
    if param.coarseSolveType == "MUMPS":
        raise NotImplementedError("MUMPS coarse solve is dead code in the reference (Multigrid.jl:29-40)")
    if param.coarseSolveType == "GMRES":
        # Jacobi-preconditioned FGMRES coarse solve (MGcycle.jl:152-168): param.LU = relaxParam ./ diag(AT) (l.334),
        # which only broadcasts for a scalar relaxParam
        if isinstance(param.relaxParam, (list, tuple, np.ndarray)):
            raise ValueError("coarseSolveType='GMRES' needs a scalar relaxParam (MGsetup.jl:334 broadcasts it over diag(AT))")
        param.LU = np.ascontiguousarray(float(param.relaxParam) / _as_csr(Ac).diagonal(), dtype=np.float64)
        return
    param.LU = coarse_lu(Ac)


def coarse_lu(Ac):
    """``lu(sparse(AT'))`` (MGsetup.jl:350).  Julia's lu is UMFPACK, which orders symmetric-pattern matrices by AMD on
    A+A'; SuperLU's closest ordering is MMD on A'+A (half the fill of its COLAMD default on these operators: 18M vs
    40M nonzeros per factor on a 33^3 27-point level)."""
    return spla.splu(sp.csc_matrix(Ac), permc_spec="MMD_AT_PLUS_A")


def MGsetup(ATf, Mesh, param: MGparam, nrhs: int = 1, verbose: bool = False) -> MGparam:
    """Build As/Ps/Rs/relaxPrecs level by level (MGsetup.jl:7-138).

    ``ATf`` is either the fine operator A (any scipy sparse; held as CSR = the reference's transposed CSC)
    or a ``multilevelOperatorConstructor`` (rediscretisation; then ``geometric=True``, l.53).
    """
    if param.transferOperatorType != "FullWeighting":
        raise NotImplementedError("only transferOperatorType='FullWeighting' (Systems.jl operators are out of scope)")
    _release_device(param)
    levels = param.levels
    relaxParamArr = _relax_param_arr(param)
    geometric = isinstance(ATf, multilevelOperatorConstructor)
    PDEparam = None
    if geometric:
        As = [_as_csr(ATf.getOperator(Mesh, ATf.param))]
        PDEparam = ATf.param
    else:
        As = [_as_csr(ATf)]
    from .operators import getRegularMesh
    Meshes = [Mesh]
    Ps, Rs, relaxPrecs = [], [], []
    n = np.asarray(Mesh.n, dtype=np.int64)
    Cop = As[0].nnz
    for l in range(1, levels):                      # l is the reference's 1-based level
        t0 = time.perf_counter()
        A = As[l - 1]
        P, nc_nodes = getFWInterp(n + 1, geometric)
        nc = nc_nodes - 1
        R = (P.T * (0.5 ** Meshes[l - 1].dim)).tocsr()      # RT = P*0.5^dim always (MGsetup.jl:56-60)
        R.sort_indices()
        relaxPrecs.append(getRelaxPrec(A, param.relaxType, relaxParamArr[l - 1]))
        if P.shape[0] == P.shape[1]:
            if verbose:
                print(f"Stopped Coarsening at level {l}")
            param.levels = l                                  # MGsetup.jl:84-92
            break
        Ps.append(P)
        Rs.append(R)
        Meshes.append(getRegularMesh(Meshes[l - 1].domain, nc))
        if geometric:
            PDEparam = ATf.restrictParams(Meshes[l - 1], Meshes[l], PDEparam, l)
            Ac = _as_csr(ATf.getOperator(Meshes[l], PDEparam))
        else:
            Ac = galerkin(R, A, P)
        As.append(Ac)
        Cop += Ac.nnz
        if verbose:
            print(f"MG setup: {n} cells took:{time.perf_counter() - t0:.3f}")
        n = nc
    if verbose:
        print("MG setup: Operator complexity = ", Cop / As[0].nnz)
    param.As = As
    param.Meshes = Meshes
    defineCoarsestAinv(param, As[-1])
    param.Ps = Ps
    param.Rs = Rs
    param.relaxPrecs = relaxPrecs
    adjustMemoryForNumRHS(param, nrhs, verbose)
    param.doTranspose = 0
    return param


def adjustMemoryForNumRHS(param: MGparam, nrhs: int = 1, verbose: bool = False) -> MGparam:
    """Size the per-level b/r/x scratch for ``nrhs`` columns (MGsetup.jl:166-223).

    On the device the scratch lives in HBM; this records the width and, when a handle exists,
    re-sizes it (``mg_set_nrhs``) only if the width changed, as the reference does (l.171-188).
    """
    if len(param.As) == 0:
        raise RuntimeError("The Hierarchy is empty - run a setup first.")
    nrhs = int(nrhs)
    if nrhs < 1:
        raise ValueError("nrhs must be >= 1")
    if param.nrhs != nrhs:
        param.nrhs = nrhs
        if param.device is not None:
            param.device.set_nrhs(nrhs)
    return param


def replaceMatrixInHierarchy(param: MGparam, A, verbose: bool = False) -> None:
    """New fine matrix, same P/R: recompute relaxPrecs, Galerkin products and the coarse LU (MGsetup.jl:226-270)."""
    relaxParamArr = _relax_param_arr(param)
    A = _as_csr(A)
    if param.device is not None and param.relaxType in ("Jac", "Jac-GMRES", "SPAI"):
        old = param.As[0]
        same = (A.shape == old.shape and A.nnz == old.nnz and np.array_equal(A.indptr, old.indptr)
                and np.array_equal(A.indices, old.indices))
        if same:
            from .device import MGDeviceError
            try:   # numeric-only Galerkin products on the device (fixed P, R and patterns: SURVEY 8f-2)
                param.device.replace_matrix(param, A)
                param.doTranspose = 0
                return
            except MGDeviceError:
                pass   # e.g. SA-AMG rows beyond the kernel's cap: the host path below applies
    param.As[0] = A
    for l in range(1, param.levels):
        Al = param.As[l - 1]
        param.relaxPrecs[l - 1] = getRelaxPrec(Al, param.relaxType, relaxParamArr[l - 1])
        param.As[l] = galerkin(param.Rs[l - 1], Al, param.Ps[l - 1])
    defineCoarsestAinv(param, param.As[-1])
    param.doTranspose = 0
    _release_device(param)            # re-uploaded lazily on the next cycle


def transposeHierarchy(param: MGparam, verbose: bool = False) -> None:
    """Transpose every operator, swap P<->R roles (MGsetup.jl:274-318).  Real VAL: conj is a no-op."""
    if param.relaxType not in ("Jac", "Jac-GMRES", "SPAI"):
        raise RuntimeError("Not supported")
    param.As[0] = _as_csr(param.As[0].T)
    param.doTranspose = (param.doTranspose + 1) % 2
    for l in range(1, param.levels):
        # reference: Ps[l] = sparse(Rs[l]'); Rs[l] = sparse(Ps[l]')  (the second line reads the NEW Ps[l],
        # l.298-299, so both end up holding the old R): reproduced literally.
        newP = _as_csr(param.Rs[l - 1].T)
        param.Ps[l - 1] = newP
        param.Rs[l - 1] = _as_csr(newP.T)
        param.As[l] = _as_csr(param.As[l].T)
    destroyCoarsestLU(param)
    defineCoarsestAinv(param, param.As[-1])
    # The resident hierarchy is transposed in HBM (mg_transpose_hierarchy: counting-sort CSR transposes, dense coarsest inverse
    # transposed in place, device formats rebuilt) instead of being dropped and uploaded again; what the library cannot do there
    # (sparse coarsest factors) falls back to the lazy re-upload.
    if param.device is not None:
        from .device import MGDeviceError
        try:
            param.device.transpose_hierarchy()
        except MGDeviceError:
            _release_device(param)
