"""Host mirror of src/ParallelJuliaSolver/parallelJuliaSolver.jl: "the host factors, the device applies".

The reference factors with UMFPACK in Julia (``setupLUFactor``, parallelJuliaSolver.jl:113-148) and applies the
triangular factors with its own native code (back end 3: ``applyLUsolve_FP64_INT64``, deps/src/parLU.cpp:52-63,
120-260; one OpenMP task per right-hand side).  Here the factorisation comes from SuperLU (scipy; no UMFPACK in this
image) in the same layout - L and U in CSR, 1-based Int64, L's diagonal last and U's diagonal first, A[p,q] = L*U -
and ``mg_lu_*`` applies them on the GPU, including the solve with the transposed matrix (``doTranspose``).
Same names and argument meaning as the reference; ``solveLinearSystem_`` is Julia's ``solveLinearSystem!``.
"""
import ctypes as C
import time
from dataclasses import dataclass, field
from typing import Any, Optional

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import device as D


@dataclass
class parallelJuliaSolver:
    """parallelJuliaSolver.jl:48-60.  ``backend`` is kept for signature parity (1 native Julia, 2 CSR Julia, 3 CSR C++):
    every value runs the device applier."""
    VAL: Any = np.float64
    IND: Any = np.int64
    L: Optional[sp.csr_matrix] = None
    U: Optional[sp.csr_matrix] = None
    p: Optional[np.ndarray] = None          # 1-based, A[p, q] = L*U
    q: Optional[np.ndarray] = None
    numCores: int = 1
    backend: int = 1
    doClear: int = 0
    nFac: int = 0
    facTime: float = 0.0
    nSolve: int = 0
    solveTime: float = 0.0
    _handle: Any = field(default=None, repr=False)

    def close(self):
        if self._handle is not None:
            D.load_library().mg_lu_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def getParallelJuliaSolver(VAL=np.float64, IND=np.int64, numCores: int = 1, backend: int = 1) -> parallelJuliaSolver:
    """parallelJuliaSolver.jl:63-70."""
    if np.dtype(VAL) != np.float64:
        raise TypeError("only Float64 factors are supported on the device path")
    return parallelJuliaSolver(VAL=VAL, IND=IND, numCores=numCores, backend=backend)


def clear_(param: parallelJuliaSolver):
    """``clear!(param)``: drop the factors and the device handle."""
    param.close()
    param.L = param.U = param.p = param.q = None
    param.doClear = 0
    return param


def _upload_factors(param: "parallelJuliaSolver") -> None:
    """(Re)create the device applier from param.L / U / p / q (the layout of setupLUFactor)."""
    lib = D.load_library()
    L, U = param.L, param.U
    a64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
    Lp, Lc, Lv = a64(L.indptr) + 1, a64(L.indices) + 1, np.ascontiguousarray(L.data, dtype=np.float64)
    Up, Uc, Uv = a64(U.indptr) + 1, a64(U.indices) + 1, np.ascontiguousarray(U.data, dtype=np.float64)
    param.close()
    h = C.c_void_p()
    D._check(lib, lib.mg_lu_create_FP64_INT64(0, L.shape[0], D._i64(Lp), D._i64(Lc), D._f64(Lv), D._i64(Up), D._i64(Uc),
                                              D._f64(Uv), D._i64(param.p), D._i64(param.q), C.byref(h)), "mg_lu_create")
    param._handle = h


def copySolver(param: parallelJuliaSolver) -> parallelJuliaSolver:
    """``copySolver`` (parallelJuliaSolver.jl:257-260): copies of L, U, p, q, the settings, counters reset to zero.  The
    copy gets a device applier of its own (created on its first solve) - a set-up solver stays set up."""
    new = getParallelJuliaSolver(param.VAL, param.IND, numCores=param.numCores, backend=param.backend)
    if param.L is not None:
        new.L, new.U = param.L.copy(), param.U.copy()
        new.p, new.q = param.p.copy(), param.q.copy()
    return new


def setupLUFactor(AI, param: parallelJuliaSolver) -> parallelJuliaSolver:
    """Factor and convert to the native applier's layout (parallelJuliaSolver.jl:113-148, convertCSC2MyCSR l.26-31)."""
    lu = spla.splu(sp.csc_matrix(AI), permc_spec="MMD_AT_PLUS_A")
    L = sp.csr_matrix(lu.L)
    U = sp.csr_matrix(lu.U)
    L.sort_indices()                              # lower: the diagonal is the last entry of every row
    U.sort_indices()                              # upper: the diagonal is the first
    param.L, param.U = L, U
    param.p = (np.argsort(lu.perm_r) + 1).astype(np.int64)
    param.q = (np.argsort(lu.perm_c) + 1).astype(np.int64)
    _upload_factors(param)
    return param


def setupSolver(AI, param: parallelJuliaSolver) -> parallelJuliaSolver:
    """jInv.LinearSolvers.setupSolver (parallelJuliaSolver.jl:107-110)."""
    return setupLUFactor(AI, param)


def solve(b: np.ndarray, x: np.ndarray, LU: parallelJuliaSolver, doTranspose: int = 0) -> np.ndarray:
    """x[q] = U \\ (L \\ b[p]), or with doTranspose x[p] = L' \\ (U' \\ b[q]) (parallelJuliaSolver.jl:151-207); x is
    written in place (column-major, as Julia holds it)."""
    if LU._handle is None:
        if LU.L is None:
            raise RuntimeError("the factors were not set up")
        _upload_factors(LU)          # a copySolver() copy: factors present, device applier not yet created
    lib = D.load_library()
    bb = np.asfortranarray(b, dtype=np.float64)
    if x.dtype != np.float64 or (x.ndim == 2 and not x.flags.f_contiguous) or not x.flags.writeable:
        raise ValueError("x must be a writable Float64 array in column-major (Julia) layout")
    if bb.shape != x.shape:
        raise ValueError("b and x differ in shape")
    n = bb.shape[0]
    nrhs = 1 if bb.ndim == 1 else bb.shape[1]
    D._check(lib, lib.mg_lu_solve_FP64(LU._handle, D._f64(bb), D._f64(x), n, nrhs, int(doTranspose)), "mg_lu_solve")
    return x


def solveLinearSystem_(A, B, X: np.ndarray, param: parallelJuliaSolver, doTranspose: int = 0):
    """``solveLinearSystem!(A,B,X,param,doTranspose)`` (parallelJuliaSolver.jl:86-103): factor on first use, then apply."""
    if param.doClear == 1:
        clear_(param)
    if param.L is None:
        t0 = time.perf_counter()
        setupLUFactor(A, param)
        param.facTime += time.perf_counter() - t0
        param.nFac += 1
    if B is not None and np.size(B) > 0:
        if sp.issparse(B):
            B = np.asfortranarray(B.toarray())
        t0 = time.perf_counter()
        X = solve(B, X, param, doTranspose)
        param.solveTime += time.perf_counter() - t0
        param.nSolve += 1
    return X, param


def solveLinearSystem(A, B, param: parallelJuliaSolver, doTranspose: int = 0):
    """``solveLinearSystem(A,B,param,doTranspose)`` (parallelJuliaSolver.jl:75-83): X is a fresh copy of B's shape."""
    Bd = np.asfortranarray(B.toarray() if sp.issparse(B) else B, dtype=np.float64)
    return solveLinearSystem_(A, Bd, Bd.copy(order="F"), param, doTranspose)
