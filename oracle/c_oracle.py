"""ORACLE - TEST INFRASTRUCTURE ONLY.  ctypes front end of liboracle_mg.so (oracle/mg_oracle.c):
the plain-C/OpenMP restatement of the reference CPU cycle with Int64 1-based indices and the unfused
operation sequence.  Used as a second checker in tests/ and as bench.py's timed cpu_baseline."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liboracle_mg.so")
_i64p = C.POINTER(C.c_longlong)
_f64p = C.POINTER(C.c_double)


class OracleLevel(C.Structure):
    _fields_ = [("n", C.c_longlong), ("nc", C.c_longlong),
                ("A_colptr", _i64p), ("A_rowval", _i64p), ("A_nzval", _f64p),
                ("P_colptr", _i64p), ("P_rowval", _i64p), ("P_nzval", _f64p),
                ("R_colptr", _i64p), ("R_rowval", _i64p), ("R_nzval", _f64p),
                ("d", _f64p), ("npre", C.c_longlong), ("npost", C.c_longlong),
                ("b", _f64p), ("r", _f64p), ("x", _f64p)]


def build():
    subprocess.check_call(["make", "-C", _HERE, "liboracle_mg.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.oracle_solveMG.restype = C.c_longlong
        _lib.oracle_solveMG.argtypes = [C.POINTER(OracleLevel), C.c_longlong, _f64p, _f64p, _f64p, C.c_longlong,
                                        C.c_double, C.c_longlong, C.c_longlong, C.c_longlong, _f64p]
        _lib.oracle_recursive_cycle.restype = None
        _lib.oracle_recursive_cycle.argtypes = [C.POINTER(OracleLevel), C.c_longlong, _f64p, C.c_longlong, _f64p,
                                                _f64p, C.c_longlong, C.c_longlong, C.c_longlong]
        _lib.oracle_spmatmul_FP64_INT64.restype = None
        _lib.oracle_spmatmul_FP64_INT64.argtypes = [C.c_double, _i64p, _i64p, _f64p, C.c_longlong, C.c_longlong,
                                                    _f64p, C.c_double, _f64p, C.c_longlong, C.c_longlong]
        _lib.oracle_max_threads.restype = C.c_longlong
        _lib.oracle_numa_clone_csr_FP64_INT64.restype = C.c_void_p
        _lib.oracle_numa_clone_csr_FP64_INT64.argtypes = [C.c_longlong, _i64p, _i64p, _f64p, C.POINTER(_i64p),
                                                          C.POINTER(_i64p), C.POINTER(_f64p), C.c_longlong]
        _lib.oracle_numa_clone_vec.restype = _f64p
        _lib.oracle_numa_clone_vec.argtypes = [_f64p, C.c_longlong, C.c_longlong, C.c_longlong]
        _lib.oracle_copy_vec.restype = None
        _lib.oracle_copy_vec.argtypes = [_f64p, _f64p, C.c_longlong, C.c_longlong, C.c_longlong]
        _lib.oracle_free.restype = None
        _lib.oracle_free.argtypes = [C.c_void_p]
    return _lib


def _p64(a):
    return a.ctypes.data_as(_i64p)


def _pf(a):
    return a.ctypes.data_as(_f64p)


class COracle:
    """Holds the Int64 1-based copies of a hierarchy (as Julia would) and runs the C cycle on it."""

    def __init__(self, param, nrhs=1, first_touch_threads=0):
        """first_touch_threads > 0 (the timed baseline): every array of the hierarchy and the scratch vectors are
        cloned into memory first touched by that many OpenMP threads with the static row partition of the kernels
        (NUMA placement); 0 (tests): the numpy arrays are used as they are."""
        self.keep = []
        self.owned = []
        self.ft = int(first_touch_threads)
        nl = len(param.As)
        self.nl = nl
        self.nrhs = nrhs
        self.levels = (OracleLevel * nl)()
        self.cycleType = ord(param.cycleType)
        for l in range(nl):
            L = self.levels[l]
            A = param.As[l]
            L.n = A.shape[0]
            L.A_colptr, L.A_rowval, L.A_nzval = self._jl(A)
            if l < nl - 1:
                L.nc = param.As[l + 1].shape[0]
                L.P_colptr, L.P_rowval, L.P_nzval = self._jl(param.Ps[l])
                L.R_colptr, L.R_rowval, L.R_nzval = self._jl(param.Rs[l])
                d = np.ascontiguousarray(param.relaxPrecs[l], dtype=np.float64)
                self.keep.append(d)
                L.d = _pf(d)
                L.npre = int(param.relaxPre(l + 1))
                L.npost = int(param.relaxPost(l + 1))
            for name in ("b", "r", "x"):
                if self.ft:
                    ptr = lib().oracle_numa_clone_vec(None, A.shape[0], nrhs, self.ft)
                    self.owned.append(C.cast(ptr, C.c_void_p))
                    setattr(L, name, ptr)
                    continue
                buf = np.zeros(A.shape[0] * nrhs)
                self.keep.append(buf)
                setattr(L, name, _pf(buf))
            if self.ft and l < nl - 1:
                dptr = lib().oracle_numa_clone_vec(L.d, A.shape[0], 1, self.ft)
                self.owned.append(C.cast(dptr, C.c_void_p))
                L.d = dptr
        nc = param.As[-1].shape[0]
        self.Ainv = np.asfortranarray(param.LU.solve(np.eye(nc)))

    def _jl(self, M):
        cp = np.ascontiguousarray(M.indptr, dtype=np.int64) + 1
        rv = np.ascontiguousarray(M.indices, dtype=np.int64) + 1
        nz = np.ascontiguousarray(M.data, dtype=np.float64)
        if self.ft:
            ocp, orv, onz = _i64p(), _i64p(), _f64p()
            if not lib().oracle_numa_clone_csr_FP64_INT64(M.shape[0], _p64(cp), _p64(rv), _pf(nz), C.byref(ocp),
                                                          C.byref(orv), C.byref(onz), self.ft):
                raise MemoryError("oracle_numa_clone_csr")
            self.owned += [C.cast(ocp, C.c_void_p), C.cast(orv, C.c_void_p), C.cast(onz, C.c_void_p)]
            return ocp, orv, onz
        self.keep += [cp, rv, nz]
        return _p64(cp), _p64(rv), _pf(nz)

    def close(self):
        for ptr in self.owned:
            lib().oracle_free(ptr)
        self.owned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solveMG_placed(self, b, tol, maxIter, numCores):
        """Timed-baseline form: b and x live in first-touched memory too; returns (iters, resvec, x as numpy copy)."""
        n = self.levels[0].n
        b = np.asfortranarray(b, dtype=np.float64)
        bp = lib().oracle_numa_clone_vec(_pf(b), n, self.nrhs, int(numCores))
        xp = lib().oracle_numa_clone_vec(None, n, self.nrhs, int(numCores))
        resvec = np.zeros(maxIter + 1)
        import time
        t0 = time.perf_counter()
        it = lib().oracle_solveMG(self.levels, self.nl, _pf(self.Ainv), bp, xp, self.nrhs, float(tol), int(maxIter),
                                  self.cycleType, int(numCores), _pf(resvec))
        dt = time.perf_counter() - t0
        x = np.ctypeslib.as_array(xp, shape=(n * self.nrhs,)).copy()
        lib().oracle_free(C.cast(bp, C.c_void_p))
        lib().oracle_free(C.cast(xp, C.c_void_p))
        return int(it), resvec[: it + 1], x, dt

    def solveMG(self, b, x, tol, maxIter, numCores):
        b = np.asfortranarray(b, dtype=np.float64)
        assert x.flags.f_contiguous or x.ndim == 1
        resvec = np.zeros(maxIter + 1)
        it = lib().oracle_solveMG(self.levels, self.nl, _pf(self.Ainv), _pf(b), _pf(x), self.nrhs, float(tol),
                                  int(maxIter), self.cycleType, int(numCores), _pf(resvec))
        return int(it), resvec[: it + 1]

    def cycle(self, b, x, numCores):
        b = np.asfortranarray(b, dtype=np.float64)
        lib().oracle_recursive_cycle(self.levels, self.nl, _pf(self.Ainv), 1, _pf(b), _pf(x), self.nrhs,
                                     self.cycleType, int(numCores))
        return x


def max_threads():
    return int(lib().oracle_max_threads())
