"""ctypes front end of libmghost.so (csrc/mg_host.cpp): native CPU helpers of the hierarchy SETUP
(never of the cycle): the SA aggregation sweep and a row-parallel SpGEMM for the Galerkin products."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmghost.so")
_lib = None
_i64p = C.POINTER(C.c_longlong)
_f64p = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError(f"{_PATH} is missing: run __graft_entry__.build()")
        _lib = C.CDLL(_PATH)
        _lib.mg_sa_aggregate_FP64_INT64.restype = C.c_int
        _lib.mg_sa_aggregate_FP64_INT64.argtypes = [C.c_longlong, _i64p, _i64p, _f64p, _i64p]
        _lib.mg_spgemm_count_INT64.restype = C.c_int
        _lib.mg_spgemm_count_INT64.argtypes = [C.c_longlong, C.c_longlong, _i64p, _i64p, _i64p, _i64p, _i64p, C.c_longlong]
        _lib.mg_spgemm_fill_FP64_INT64.restype = C.c_int
        _lib.mg_spgemm_fill_FP64_INT64.argtypes = [C.c_longlong, C.c_longlong, _i64p, _i64p, _f64p, _i64p, _i64p, _f64p,
                                                   _i64p, _i64p, _f64p, C.c_longlong]
        _lib.mg_host_max_threads.restype = C.c_longlong
    return _lib


def _p64(a):
    return a.ctypes.data_as(_i64p)


def _pf(a):
    return a.ctypes.data_as(_f64p)


def spgemm(A, B, nthreads: int = 0):
    """C = A*B (CSR, sorted column indices, numerically cancelled entries kept), row-parallel on the host."""
    A = sp.csr_matrix(A)
    B = sp.csr_matrix(B)
    if A.shape[1] != B.shape[0]:
        raise ValueError("dimension mismatch")
    Ap = np.ascontiguousarray(A.indptr, dtype=np.int64)
    Ai = np.ascontiguousarray(A.indices, dtype=np.int64)
    Av = np.ascontiguousarray(A.data, dtype=np.float64)
    Bp = np.ascontiguousarray(B.indptr, dtype=np.int64)
    Bi = np.ascontiguousarray(B.indices, dtype=np.int64)
    Bv = np.ascontiguousarray(B.data, dtype=np.float64)
    n = A.shape[0]
    Cp = np.zeros(n + 1, dtype=np.int64)
    L = lib()
    if nthreads <= 0:      # torchrun exports OMP_NUM_THREADS=1 to every rank: MG_HOST_THREADS overrides it here
        nthreads = int(os.environ.get("MG_HOST_THREADS", "0") or 0)
    rc = L.mg_spgemm_count_INT64(n, B.shape[1], _p64(Ap), _p64(Ai), _p64(Bp), _p64(Bi), _p64(Cp), int(nthreads))
    if rc != 0:
        raise RuntimeError(f"mg_spgemm_count_INT64 failed (status {rc})")
    np.cumsum(Cp, out=Cp)
    nnz = int(Cp[-1])
    Ci = np.empty(max(nnz, 1), dtype=np.int64)
    Cv = np.empty(max(nnz, 1), dtype=np.float64)
    rc = L.mg_spgemm_fill_FP64_INT64(n, B.shape[1], _p64(Ap), _p64(Ai), _pf(Av), _p64(Bp), _p64(Bi), _pf(Bv), _p64(Cp),
                                     _p64(Ci), _pf(Cv), int(nthreads))
    if rc != 0:
        raise RuntimeError(f"mg_spgemm_fill_FP64_INT64 failed (status {rc})")
    idx_t = np.int32 if max(nnz, B.shape[1]) < 2 ** 31 - 1 else np.int64
    Cm = sp.csr_matrix((Cv[:nnz], Ci[:nnz].astype(idx_t), Cp.astype(idx_t)), shape=(n, B.shape[1]))
    Cm.has_sorted_indices = True
    return Cm
