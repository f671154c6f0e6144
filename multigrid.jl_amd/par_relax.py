"""Hybrid Kaczmarz relaxation behind the reference's interface (src/Multigrid/parRelax.jl): ``getHybridKaczmarz``,
``setupHybridKaczmarz``, ``getHybridKaczmarzPrecond``, ``applyHybridKaczmarz``.  The sweeps run on the device
(csrc: hybrid_kaczmarz <- deps/src/parRelax.h:7-43) through ``mg_kaczmarz_*``; there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, Optional

import numpy as np
import scipy.sparse as sp

from . import device as D
from .dd_indices import getIndicesOfCellsArray, getNodalIndicesOfCell

ArrIdxsType = np.uint32


@dataclass
class hybridKaczmarz:
    """Mirror of ``mutable struct hybridKaczmarz`` (parRelax.jl:8-17)."""
    numDomains: list
    invDiag: Optional[np.ndarray]
    numCores: int
    omega_damp: float
    ArrIdxs: np.ndarray
    precond: Optional[Callable]
    numit: int
    getIndicesOfCell: Callable
    sequential: bool = False            # device schedule: False = one wavefront per sub-domain (the OpenMP analogue)
    _handle: object = field(default=None, repr=False)
    _key: object = field(default=None, repr=False)

    def close(self):
        if self._handle:
            D.load_library().mg_kaczmarz_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def getHybridKaczmarz(VAL, IND, *args):
    """Both methods of parRelax.jl:24-29 and 39-47: (numDomains, getIndicesOfCell, omega, numCores, numit) leaves the
    setup to ``setupHybridKaczmarz``; (AT, Mesh, numDomains, getIndicesOfCell, omega, numCores, numit) performs it."""
    if len(args) == 5:
        numDomains, getIdx, omega, numCores, numit = args
        if int(np.prod(numDomains)) < numCores:
            print("*** WARNING: getHybridKaczmarz: numDomains < numCores. ***")
        return hybridKaczmarz(list(numDomains), None, int(numCores), float(omega), np.zeros((1, 1), ArrIdxsType), None,
                              int(numit), getIdx)
    AT, mesh, numDomains, getIdx, omega, numCores, numit = args
    p = getHybridKaczmarz(VAL, IND, numDomains, getIdx, omega, numCores, numit)
    return setupHybridKaczmarz(p, AT, mesh)


def setupHybridKaczmarz(param: hybridKaczmarz, AT, mesh):
    """invDiag = omega ./ sum(conj(AT).*AT, dims=1) and the index array of the sub-domains (parRelax.jl:31-36)."""
    A = sp.csr_matrix(AT)                     # this package holds the CSR of A where Julia holds the CSC of A' (MGdef.jl:75-77)
    param.invDiag = param.omega_damp / np.asarray(A.multiply(A).sum(axis=1)).ravel()
    param.ArrIdxs = getIndicesOfCellsArray(mesh, np.zeros(len(param.numDomains), dtype=np.int64), param.numDomains,
                                           param.getIndicesOfCell)
    param.close()
    return param


def _cheap_key(A):
    """What a call can check for nothing: the object, its buffers and its shape.  A matrix converted on every call (CSC in,
    CSR needed) never matches and pays the full key below - callers that sweep in a loop should hand over a CSR matrix."""
    if not sp.isspmatrix_csr(A):
        return None
    return (id(A), A.data.ctypes.data, A.indices.ctypes.data, A.indptr.ctypes.data, A.nnz, A.shape)


def _matrix_key(A):
    """Identity of the VALUES and of the PATTERN (row pointers included: equal data / indices with other row splits are
    another matrix): the reference passes AT.nzval on every call (parRelax.jl:61-64), so a matrix modified in place, or a
    new one allocated at a recycled id, must not hit the uploaded copy."""
    import zlib
    return (A.nnz, A.shape, zlib.crc32(np.ascontiguousarray(A.data).view(np.uint8)),
            zlib.crc32(np.ascontiguousarray(A.indices).view(np.uint8)), zlib.crc32(np.ascontiguousarray(A.indptr).view(np.uint8)))


def _device_handle(param: hybridKaczmarz, A, values_changed: bool = True):
    """The uploaded copy of A, re-used while A is the same matrix.  values_changed = False: the caller vouches that a CSR
    matrix with the same buffers as last time still holds the same values (skips the O(nnz) hash of every call)."""
    cheap = _cheap_key(A)
    if param._handle is not None and cheap is not None and cheap == getattr(param, "_cheap", None) and not values_changed:
        return param._handle
    Ac = A if sp.isspmatrix_csr(A) else sp.csr_matrix(A)      # converted ONCE: key and upload use the same CSR
    key = _matrix_key(Ac)
    if param._handle is not None and param._key == key:
        param._cheap = cheap
        return param._handle
    param.close()
    lib = D.load_library()
    A = Ac.copy() if not Ac.has_sorted_indices else Ac
    A.sort_indices()
    cp = np.ascontiguousarray(A.indptr, dtype=np.int64) + 1
    rv = np.ascontiguousarray(A.indices, dtype=np.int64) + 1
    nz = np.ascontiguousarray(A.data, dtype=np.float64)
    arr = np.asfortranarray(param.ArrIdxs, dtype=np.uint32)
    invd = np.ascontiguousarray(param.invDiag, dtype=np.float64)
    h = C.c_void_p()
    D._check(lib, lib.mg_kaczmarz_create_FP64_INT64(0, A.shape[0], D._i64(cp), D._f64(nz), D._i64(rv), arr.shape[1], arr.shape[0],
                                                    arr.ctypes.data_as(C.POINTER(C.c_uint)), D._f64(invd), C.byref(h)),
             "mg_kaczmarz_create")
    param._handle, param._key, param._cheap = h, key, cheap
    return h


def applyHybridKaczmarz(param: hybridKaczmarz, AT, r: np.ndarray, x: np.ndarray, numDomains: Optional[int] = None,
                        values_changed: bool = True):
    """``numit`` sweeps of x towards AT' x = r, in place (parRelax.jl:59-65).  values_changed = False (not in the reference's
    signature): the CSR matrix handed over is unchanged since the last call - the uploaded copy is reused without hashing it."""
    h = _device_handle(param, AT, values_changed)
    lib = D.load_library()
    if x.ndim == 2 and not x.flags.f_contiguous:
        raise ValueError("x must be column-major (Julia layout)")
    rr = np.asfortranarray(r, dtype=np.float64)
    nrhs = 1 if rr.ndim == 1 else rr.shape[1]
    D._check(lib, lib.mg_kaczmarz_apply_FP64(h, D._f64(x), D._f64(rr), nrhs, int(param.numit), 1 if param.sequential else 0),
             "mg_kaczmarz_apply")
    return x


def getHybridKaczmarzPrecond(param: hybridKaczmarz, AT, nrhs: int):
    """r -> x with x = 0 on entry (parRelax.jl:49-57); the returned closure reuses one buffer, as the reference does."""
    n = AT.shape[1]
    x = np.zeros(n) if nrhs == 1 else np.zeros((n, nrhs), order="F")

    Ac = AT if sp.isspmatrix_csr(AT) else sp.csr_matrix(AT)   # (converted once for every application of the closure)
    _device_handle(param, Ac, True)

    def precond(r):
        x[...] = 0.0
        # (the reference hands AT.nzval over on every call: values modified in place are honoured unless the caller set
        # param.static_matrix = True, which skips the O(nnz) hash of every application)
        applyHybridKaczmarz(param, Ac, r, x, values_changed=not getattr(param, "static_matrix", False))
        return x

    param.precond = precond
    return precond
