"""ctypes front end of libmghost.so (csrc/mg_host.cpp): native CPU helpers of the hierarchy SETUP
(never of the cycle): the SA aggregation sweep and a row-parallel SpGEMM for the Galerkin products."""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np
import scipy.sparse as sp

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmghost.so")
_lib = None
_i64p = C.POINTER(C.c_longlong)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int)


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
        _lib.mg_spgemm_count_INT32.restype = C.c_int
        _lib.mg_spgemm_count_INT32.argtypes = [C.c_longlong, C.c_longlong, _i32p, _i32p, _i32p, _i32p, _i64p, C.c_longlong]
        _lib.mg_spgemm_fill_FP64_INT32.restype = C.c_int
        _lib.mg_spgemm_fill_FP64_INT32.argtypes = [C.c_longlong, C.c_longlong, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p, _i64p, _i32p, _f64p,
                                                   C.c_longlong]
        _lib.mg_spgemm_symbolic_INT32.restype = C.c_void_p
        _lib.mg_spgemm_symbolic_INT32.argtypes = [C.c_longlong, C.c_longlong, _i32p, _i32p, _i32p, _i32p, _i64p, C.c_longlong]
        _lib.mg_spgemm_numeric_FP64_INT32.restype = C.c_int
        _lib.mg_spgemm_numeric_FP64_INT32.argtypes = [C.c_void_p, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p, _i64p, _i32p, _f64p, C.c_longlong]
        _lib.mg_spgemm_plan_free.restype = None
        _lib.mg_spgemm_plan_free.argtypes = [C.c_void_p]
        _lib.mg_csr_transpose_FP64_INT32.restype = C.c_int
        _lib.mg_csr_transpose_FP64_INT32.argtypes = [C.c_longlong, C.c_longlong, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p, C.c_longlong]
        _lib.mg_sa_strength_FP64_INT32.restype = C.c_int
        _lib.mg_sa_strength_FP64_INT32.argtypes = [C.c_longlong, _i32p, _i32p, _f64p, C.c_double, _f64p, C.c_longlong]
        _lib.mg_csr_add_transpose_symm_FP64_INT32.restype = C.c_int
        _lib.mg_csr_add_transpose_symm_FP64_INT32.argtypes = [C.c_longlong, _i32p, _i32p, _f64p, _f64p, C.c_longlong]
        _lib.mg_csr_compact_nonzero_FP64_INT32.restype = C.c_int
        _lib.mg_csr_compact_nonzero_FP64_INT32.argtypes = [C.c_longlong, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p, C.c_longlong]
        _lib.mg_csr_colsumsq_FP64_INT32.restype = C.c_int
        _lib.mg_csr_colsumsq_FP64_INT32.argtypes = [C.c_longlong, C.c_longlong, _i32p, _i32p, _f64p, _f64p, C.c_longlong]
        _lib.mg_sa_aggregate_FP64_INT32_BASE0.restype = C.c_int
        _lib.mg_sa_aggregate_FP64_INT32_BASE0.argtypes = [C.c_longlong, _i32p, _i32p, _f64p, _i64p]
    return _lib


def _p64(a):
    return a.ctypes.data_as(_i64p)


def _pf(a):
    return a.ctypes.data_as(_f64p)


def _threads(nthreads: int) -> int:
    if nthreads <= 0:      # torchrun exports OMP_NUM_THREADS=1 to every rank: MG_HOST_THREADS overrides it here
        nthreads = int(os.environ.get("MG_HOST_THREADS", "0") or 0)
    return int(nthreads)


def _p32(a):
    return a.ctypes.data_as(_i32p)


def spgemm(A, B, nthreads: int = 0):
    """C = A*B (CSR, sorted column indices, numerically cancelled entries kept), row-parallel on the host.
    Operands with 32-bit indices (scipy's default below 2^31 entries) are multiplied as they are - no widening copies."""
    A = sp.csr_matrix(A)
    B = sp.csr_matrix(B)
    if A.shape[1] != B.shape[0]:
        raise ValueError("dimension mismatch")
    n = A.shape[0]
    L = lib()
    nthreads = _threads(nthreads)
    Cp = np.zeros(n + 1, dtype=np.int64)
    Av = np.ascontiguousarray(A.data, dtype=np.float64)
    Bv = np.ascontiguousarray(B.data, dtype=np.float64)
    if A.indices.dtype == np.int32 and B.indices.dtype == np.int32 and A.indptr.dtype == np.int32 and B.indptr.dtype == np.int32:
        Ap, Ai, Bp, Bi = (np.ascontiguousarray(v) for v in (A.indptr, A.indices, B.indptr, B.indices))
        if not os.environ.get("MG_HOST_SPGEMM_TWO_PASS") and B.has_sorted_indices:
            # symbolic phase (sorted patterns, kept on the native side) + numeric phase: one walk over the products each, no sort / test in the second
            plan = L.mg_spgemm_symbolic_INT32(n, B.shape[1], _p32(Ap), _p32(Ai), _p32(Bp), _p32(Bi), _p64(Cp), nthreads)
            if plan:
                nnz = int(Cp[-1])
                if max(nnz, B.shape[1]) < 2 ** 31 - 1:
                    Ci = np.empty(max(nnz, 1), dtype=np.int32)
                    Cv = np.empty(max(nnz, 1), dtype=np.float64)
                    rc = L.mg_spgemm_numeric_FP64_INT32(plan, _p32(Ap), _p32(Ai), _pf(Av), _p32(Bp), _p32(Bi), _pf(Bv), _p64(Cp), _p32(Ci), _pf(Cv), nthreads)
                    if rc != 0:
                        raise RuntimeError(f"mg_spgemm_numeric_FP64_INT32 failed (status {rc})")
                    Cm = sp.csr_matrix((Cv[:nnz], Ci[:nnz], Cp.astype(np.int32)), shape=(n, B.shape[1]))
                    Cm.has_sorted_indices = True
                    return Cm
                L.mg_spgemm_plan_free(plan)
            Cp[:] = 0
        rc = L.mg_spgemm_count_INT32(n, B.shape[1], _p32(Ap), _p32(Ai), _p32(Bp), _p32(Bi), _p64(Cp), nthreads)
        if rc != 0:
            raise RuntimeError(f"mg_spgemm_count_INT32 failed (status {rc})")
        np.cumsum(Cp, out=Cp)
        nnz = int(Cp[-1])
        if max(nnz, B.shape[1]) < 2 ** 31 - 1:
            Ci = np.empty(max(nnz, 1), dtype=np.int32)
            Cv = np.empty(max(nnz, 1), dtype=np.float64)
            rc = L.mg_spgemm_fill_FP64_INT32(n, B.shape[1], _p32(Ap), _p32(Ai), _pf(Av), _p32(Bp), _p32(Bi), _pf(Bv), _p64(Cp), _p32(Ci),
                                             _pf(Cv), nthreads)
            if rc != 0:
                raise RuntimeError(f"mg_spgemm_fill_FP64_INT32 failed (status {rc})")
            Cm = sp.csr_matrix((Cv[:nnz], Ci[:nnz], Cp.astype(np.int32)), shape=(n, B.shape[1]))
            Cm.has_sorted_indices = True
            return Cm
        Cp[:] = 0                      # (the product needs 64-bit indices: the wide path below)
    Ap = np.ascontiguousarray(A.indptr, dtype=np.int64)
    Ai = np.ascontiguousarray(A.indices, dtype=np.int64)
    Bp = np.ascontiguousarray(B.indptr, dtype=np.int64)
    Bi = np.ascontiguousarray(B.indices, dtype=np.int64)
    rc = L.mg_spgemm_count_INT64(n, B.shape[1], _p64(Ap), _p64(Ai), _p64(Bp), _p64(Bi), _p64(Cp), nthreads)
    if rc != 0:
        raise RuntimeError(f"mg_spgemm_count_INT64 failed (status {rc})")
    np.cumsum(Cp, out=Cp)
    nnz = int(Cp[-1])
    Ci = np.empty(max(nnz, 1), dtype=np.int64)
    Cv = np.empty(max(nnz, 1), dtype=np.float64)
    rc = L.mg_spgemm_fill_FP64_INT64(n, B.shape[1], _p64(Ap), _p64(Ai), _pf(Av), _p64(Bp), _p64(Bi), _pf(Bv), _p64(Cp),
                                     _p64(Ci), _pf(Cv), nthreads)
    if rc != 0:
        raise RuntimeError(f"mg_spgemm_fill_FP64_INT64 failed (status {rc})")
    idx_t = np.int32 if max(nnz, B.shape[1]) < 2 ** 31 - 1 else np.int64
    Cm = sp.csr_matrix((Cv[:nnz], Ci[:nnz].astype(idx_t), Cp.astype(idx_t)), shape=(n, B.shape[1]))
    Cm.has_sorted_indices = True
    return Cm


def transpose_csr(M, nthreads: int = 0):
    """M' as CSR with sorted column indices (the reference's `sparse(P')`, SA-AMG.jl:47), thread-parallel on the host."""
    M = sp.csr_matrix(M)
    if M.indices.dtype != np.int32 or M.indptr.dtype != np.int32 or M.nnz == 0:
        T = sp.csr_matrix(M.T)
        T.sort_indices()
        return T
    n, m = M.shape
    ptr, idx, val = np.ascontiguousarray(M.indptr), np.ascontiguousarray(M.indices), np.ascontiguousarray(M.data, dtype=np.float64)
    tp = np.empty(m + 1, dtype=np.int32)
    ti = np.empty(M.nnz, dtype=np.int32)
    tv = np.empty(M.nnz, dtype=np.float64)
    rc = lib().mg_csr_transpose_FP64_INT32(n, m, _p32(ptr), _p32(idx), _pf(val), _p32(tp), _p32(ti), _pf(tv), _threads(nthreads))
    if rc != 0:
        raise RuntimeError("mg_csr_transpose_FP64_INT32 failed")
    T = sp.csr_matrix((tv, ti, tp), shape=(m, n))
    T.has_sorted_indices = True
    return T


def add_transpose(S, nthreads: int = 0):
    """S + S' (SA-AMG.jl:115) WITHOUT the entries that sum to zero (Julia's sparse `+` stores non-zero results only), rows sorted - for a
    structurally symmetric S with sorted rows thread-parallel on the host, into arrays of its own; scipy otherwise."""
    S = sp.csr_matrix(S)
    if S.shape[0] == S.shape[1] and S.indices.dtype == np.int32 and S.indptr.dtype == np.int32 and S.has_sorted_indices and S.nnz > 0:
        n = S.shape[0]
        ptr, idx, val = np.ascontiguousarray(S.indptr), np.ascontiguousarray(S.indices), np.ascontiguousarray(S.data, dtype=np.float64)
        out = np.empty(S.nnz, dtype=np.float64)
        if lib().mg_csr_add_transpose_symm_FP64_INT32(n, _p32(ptr), _p32(idx), _pf(val), _pf(out), _threads(nthreads)) == 0:
            nptr = np.empty(n + 1, dtype=np.int32)
            nidx = np.empty(S.nnz, dtype=np.int32)
            nval = np.empty(S.nnz, dtype=np.float64)
            if lib().mg_csr_compact_nonzero_FP64_INT32(n, _p32(ptr), _p32(idx), _pf(out), _p32(nptr), _p32(nidx), _pf(nval), _threads(nthreads)) == 0:
                nnz = int(nptr[-1])
                T = sp.csr_matrix((nval[:nnz], nidx[:nnz], nptr), shape=S.shape)
                T.has_sorted_indices = True
                return T
    T = (S + S.T).tocsr()
    T.eliminate_zeros()
    T.sort_indices()
    return T


def col_sumsq(A, nthreads: int = 0):
    """s[j] = sum_i A[i,j]^2 (getSPAIprec, MGsetup.jl:359-362), thread-parallel on the host; None when the operand does not fit the native path."""
    if not sp.isspmatrix_csr(A) or A.indices.dtype != np.int32 or A.indptr.dtype != np.int32 or A.nnz == 0 or A.data.dtype != np.float64:
        return None
    ptr, idx, val = np.ascontiguousarray(A.indptr), np.ascontiguousarray(A.indices), np.ascontiguousarray(A.data)
    out = np.empty(A.shape[1], dtype=np.float64)
    if lib().mg_csr_colsumsq_FP64_INT32(A.shape[0], A.shape[1], _p32(ptr), _p32(idx), _pf(val), _pf(out), _threads(nthreads)) != 0:
        return None
    return out


def sa_aggregate(S):
    """neighborhoodAggregationNew (SA-AMG.jl:119-211) on the CSR arrays of the symmetric strength matrix (= its CSC arrays): aggr[k] = 1-based
    root of node k's aggregate."""
    S = sp.csr_matrix(S)
    if not S.has_sorted_indices:
        S.sort_indices()
    n = S.shape[0]
    aggr = np.zeros(n, dtype=np.int64)
    val = np.ascontiguousarray(S.data, dtype=np.float64)
    if S.indices.dtype == np.int32 and S.indptr.dtype == np.int32:
        ptr, idx = np.ascontiguousarray(S.indptr), np.ascontiguousarray(S.indices)
        rc = lib().mg_sa_aggregate_FP64_INT32_BASE0(n, _p32(ptr), _p32(idx), _pf(val), _p64(aggr))
    else:
        colptr = np.ascontiguousarray(S.indptr, dtype=np.int64) + 1
        rowval = np.ascontiguousarray(S.indices, dtype=np.int64) + 1
        rc = lib().mg_sa_aggregate_FP64_INT64(n, _p64(colptr), _p64(rowval), _pf(val), _p64(aggr))
    if rc != 0:
        raise RuntimeError("aggregation failed")
    return aggr


def sa_strength(A, theta: float, nthreads: int = 0):
    """getStrengthMatrix up to (not including) the symmetrisation, row-parallel on the host: a CSR matrix with A's pattern, or None
    when the operands do not fit the native path (64-bit indices)."""
    A = sp.csr_matrix(A)
    if A.indices.dtype != np.int32 or A.indptr.dtype != np.int32 or A.nnz == 0:
        return None
    if not A.has_sorted_indices:
        A.sort_indices()
    ptr, idx, val = np.ascontiguousarray(A.indptr), np.ascontiguousarray(A.indices), np.ascontiguousarray(A.data, dtype=np.float64)
    out = np.empty(A.nnz, dtype=np.float64)
    if lib().mg_sa_strength_FP64_INT32(A.shape[0], _p32(ptr), _p32(idx), _pf(val), float(theta), _pf(out), _threads(nthreads)) != 0:
        return None
    S = sp.csr_matrix((out, idx, ptr), shape=A.shape)
    S.has_sorted_indices = True
    return S


# ------------------------------------------------------------------------------------------------------------------------
# Galerkin products of NEARLY DENSE levels on the GPU (setup time only; optional).  The SA-AMG hierarchy of anisotropic diffusion
# (BASELINE config C3) ends in levels whose operators are 10-100 % dense - 56 698 rows x 13 705 entries at 256^3 cells: 2.7 x 10^12
# products for R*(A*P) on the host, 300 of the setup's 540 s.  As dense fp64 GEMMs (rocBLAS through torch.matmul) the same product
# is seconds.  The PATTERN is the structural one (an entry whose terms cancel is kept, as Julia's and scipy's sparse products keep
# it): indicator matrices multiplied in fp32 (counts < 2^24: exact).  Values: the same terms summed in GEMM order instead of
# Gustavson order (differences at rounding level).  OPT-IN (round 6): MG_SETUP_GPU=1 - by default every product of the setup
# stays on the host, as the reference's does, and the hierarchy does not depend on whether a GPU is present.
# ------------------------------------------------------------------------------------------------------------------------
def setup_on_gpu_enabled() -> bool:
    return os.environ.get("MG_SETUP_GPU", "0") == "1"


def galerkin_dense_gpu_ok(A, P) -> bool:
    if not setup_on_gpu_enabled():
        return False
    n, nc = A.shape[0], P.shape[1]
    if A.shape[0] != A.shape[1] or n < int(os.environ.get("MG_SETUP_GPU_MIN_ROWS", "3000")) or A.nnz < float(os.environ.get("MG_SETUP_GPU_MIN_DENSITY", "0.04")) * n * n:
        return False
    try:
        import torch
        if not torch.cuda.is_available():
            return False
        free, _ = torch.cuda.mem_get_info()
    except Exception:
        return False
    # operands and products in fp64 + their indicators in fp32, the comparison / index temporaries of the pattern (1 B per entry of
    # A*P and R*(A*P), 16 B per stored entry of the result, its values), the index lists of the scatter
    need = 8.0 * (n * n + 2.0 * n * nc) + 4.0 * (n * n + 2.0 * n * nc) + 1.0 * (n * nc + nc * nc) + 28.0 * nc * nc + 16.0 * A.nnz + 2.0e9
    return need < 0.8 * free


def _dense_on_gpu(M, torch, dev):
    """(values, indicator) of a CSR matrix as dense fp64 / fp32 device tensors."""
    M = sp.csr_matrix(M)
    n, m = M.shape
    rows = torch.repeat_interleave(torch.arange(n, device=dev), torch.from_numpy(np.diff(M.indptr).astype(np.int64)).to(dev))
    cols = torch.from_numpy(M.indices.astype(np.int64)).to(dev)
    flat = rows * m + cols
    del rows, cols
    D = torch.zeros(n * m, dtype=torch.float64, device=dev)
    D.index_put_((flat,), torch.from_numpy(np.ascontiguousarray(M.data, dtype=np.float64)).to(dev), accumulate=True)   # (duplicates, if any, add up)
    S = torch.zeros(n * m, dtype=torch.float32, device=dev)
    S.index_fill_(0, flat, 1.0)
    return D.view(n, m), S.view(n, m)


def galerkin_dense_gpu(R, A, P):
    """A_c = R*(A*P) through dense GEMMs on the GPU; CSR (int32, sorted) with the structural pattern of the sparse product.
    None when it does not go through (out of memory, a rocBLAS error): the caller then multiplies on the host."""
    import torch
    try:
        return _galerkin_dense_gpu(R, A, P, torch)
    except Exception as e:
        print(f"[multigrid.jl_amd] dense GPU Galerkin product not used ({type(e).__name__}: {str(e)[:120]}): host product", file=sys.stderr, flush=True)
        try:
            torch.cuda.empty_cache()
        except Exception:
            pass
        return None


def _galerkin_dense_gpu(R, A, P, torch):
    dev = torch.device("cuda", torch.cuda.current_device())
    Ad, As_ = _dense_on_gpu(A, torch, dev)
    Pd, Ps_ = _dense_on_gpu(P, torch, dev)
    AP = Ad @ Pd
    del Ad
    APs = ((As_ @ Ps_) > 0.5).to(torch.float32)
    del As_, Pd, Ps_
    Rd, Rs_ = _dense_on_gpu(R, torch, dev)
    Ac = Rd @ AP
    del Rd, AP
    pat = (Rs_ @ APs) > 0.5
    del Rs_, APs
    nc = Ac.shape[0]
    counts = pat.sum(dim=1, dtype=torch.int64)
    idx = pat.nonzero(as_tuple=False)                     # row-major: sorted rows, ascending columns
    vals = Ac[pat]
    ptr = np.zeros(nc + 1, dtype=np.int64)
    np.cumsum(counts.cpu().numpy(), out=ptr[1:])
    ci = idx[:, 1].to(torch.int32).cpu().numpy()
    va = vals.cpu().numpy()
    del Ac, pat, idx, vals
    torch.cuda.empty_cache()
    Cm = sp.csr_matrix((va, ci, ptr.astype(np.int32 if ptr[-1] < 2 ** 31 - 1 else np.int64)), shape=(nc, P.shape[1]))
    Cm.has_sorted_indices = True
    return Cm


# Galerkin products too large for the host and too sparse for dense GEMMs: rocSPARSE's SpGEMM through torch.sparse.mm on CSR
# operands (structural pattern, checked against the host product in tests; values to rounding).  Taken from MG_SETUP_GPU_MIN_PRODUCTS
# estimated products on (the 241 k-row level of C3 at 256^3: 2.7 x 10^11); any failure (memory) falls back to the host product.
def galerkin_sparse_gpu_ok(A, P) -> bool:
    if not setup_on_gpu_enabled() or os.environ.get("MG_SETUP_GPU_SPARSE", "1") == "0":
        return False
    est = float(A.nnz) * float(P.nnz) / max(1, P.shape[0])
    if est < float(os.environ.get("MG_SETUP_GPU_MIN_PRODUCTS", "5e9")):
        return False
    try:
        import torch
        if not torch.cuda.is_available():
            return False
        free, _ = torch.cuda.mem_get_info()
    except Exception:
        return False
    return 16.0 * (A.nnz + 2 * P.nnz) * 8.0 < 0.8 * free      # (operands + a product several times their size)


def _csr_to_gpu(M, torch, dev):
    M = sp.csr_matrix(M)
    return torch.sparse_csr_tensor(torch.from_numpy(M.indptr.astype(np.int64)).to(dev), torch.from_numpy(M.indices.astype(np.int64)).to(dev),
                                   torch.from_numpy(np.ascontiguousarray(M.data, dtype=np.float64)).to(dev), size=M.shape)


def galerkin_sparse_gpu(R, A, P):
    """A_c = R*(A*P) through rocSPARSE's SpGEMM on the GPU; None when it does not go through (the caller then multiplies on the host)."""
    import torch
    import warnings
    dev = torch.device("cuda", torch.cuda.current_device())
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            At = _csr_to_gpu(A, torch, dev)
            Pt = _csr_to_gpu(P, torch, dev)
            AP = torch.sparse.mm(At, Pt)
            del At, Pt
            Rt = _csr_to_gpu(R, torch, dev)
            Ac = torch.sparse.mm(Rt, AP)
            del Rt, AP
            ptr = Ac.crow_indices().cpu().numpy()
            idx = Ac.col_indices().cpu().numpy()
            val = Ac.values().cpu().numpy()
            del Ac
    except Exception as e:                      # (out of memory, an unsupported size: the host product serves)
        print(f"[multigrid.jl_amd] GPU SpGEMM not used ({type(e).__name__}: {str(e)[:120]}): host product", file=sys.stderr, flush=True)
        try:
            torch.cuda.empty_cache()
        except Exception:
            pass
        return None
    torch.cuda.empty_cache()
    it = np.int32 if max(int(ptr[-1]), P.shape[1]) < 2 ** 31 - 1 else np.int64
    Cm = sp.csr_matrix((val, idx.astype(it), ptr.astype(it)), shape=(R.shape[0], P.shape[1]))
    Cm.sort_indices()
    return Cm
