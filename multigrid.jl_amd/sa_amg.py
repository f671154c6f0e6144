"""Smoothed-aggregation AMG setup on the host (CPU), feeding the device cycle library.

Mirrors reference src/Multigrid/SA-AMG.jl: ``SA_AMGsetup`` (l.8-76), ``getAggregation`` (l.78-86),
``getStrengthMatrix`` (l.88-116), ``neighborhoodAggregationNew`` (l.119-211, native: csrc/mg_host.cpp),
``aggrArray2P`` (l.213-224).  Treister & Yavneh, SISC 37(1), 2015.

CSR-of-A view of the reference's transposed storage (MGdef.jl:75-77): column j of ``AT`` is row j of A.
With d = relaxPrecs[l] and P0 the tentative (piecewise-constant) prolongation,

    DAT = AT*diag(d) = (D A)'          rho = min(norm(DAT,1), norm(DAT,Inf))    (entry-wise norms)
    PT  = P0' - (1.33/rho) P0'*DAT  =>  P = (I - (1.33/rho) D A) P0,   R = P'   (l.44-48)
    Ac  = P' (A P)                                                              (l.50)
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np
import scipy.sparse as sp

from .mgdef import MGparam, _release_device
from .mgsetup import _as_csr, adjustMemoryForNumRHS, defineCoarsestAinv, galerkin, getRelaxPrec

from .hostlib import add_transpose, sa_aggregate, sa_strength, spgemm, transpose_csr


def getStrengthMatrix(A, strengthConnParam: float):
    """S = -A, every ROW scaled by its largest positive entry (>= 1e-16*max), diagonal := 1, entries < theta
    zeroed, then S + S' (SA-AMG.jl:88-116).  Julia's sparse ``+`` stores only non-zero results, so the
    pattern of the result is: pairs strong in at least one direction, plus the diagonal."""
    A = _as_csr(A)
    n = A.shape[0]
    S = sa_strength(A, strengthConnParam)           # (row-parallel native form of the lines below: same operations, same order)
    if S is not None:
        return add_transpose(S)                     # (S + S' without the entries that sum to zero, rows sorted)
    S = (-A).tocsr()
    S.sort_indices()
    mm = 1e-16 * S.data.max()
    rows = np.repeat(np.arange(n), np.diff(S.indptr))
    rowmax = np.full(n, mm)
    nonempty = np.diff(S.indptr) > 0
    rowmax[nonempty] = np.maximum(mm, np.maximum.reduceat(S.data, S.indptr[:-1][nonempty]))
    S.data = S.data * (1.0 / rowmax)[rows]          # scal_k = 1/maxVal_j; nzval *= scal_k (l.100-103)
    S.data[S.indices == rows] = 1.0
    S.data[S.data < strengthConnParam] = 0.0
    return add_transpose(S)                         # S + S' (l.115): thread-parallel on the host for the symmetric pattern A gives S


def neighborhoodAggregationNew(S):
    """Three-pass greedy aggregation (SA-AMG.jl:119-211), executed by the native helper."""
    return sa_aggregate(S)                                            # (S symmetric: CSR arrays == CSC arrays)


def aggrArray2P(aggr):
    """Tentative prolongation P0 (n x Nc), P0[i, id(aggr[i])] = 1, roots numbered in node order (l.213-224)."""
    aggr = np.asarray(aggr, dtype=np.int64)
    n = aggr.size
    roots = np.nonzero(aggr == np.arange(1, n + 1))[0]
    fine2coarse = np.zeros(n + 1, dtype=np.int64)
    fine2coarse[roots + 1] = np.arange(1, roots.size + 1)
    a = fine2coarse[aggr]
    if np.any(a == 0):
        raise RuntimeError("nodes without aggregates")
    P = sp.csr_matrix((np.ones(n), (np.arange(n), a - 1)), shape=(n, roots.size))
    P.sort_indices()
    return P


def getAggregation(A, strengthConnParam: float):
    """Identity (coarsening stops) when n <= 100 (SA-AMG.jl:78-86)."""
    n = A.shape[0]
    if n <= 100:
        return sp.identity(n, format="csr")
    S = getStrengthMatrix(A, strengthConnParam)
    return aggrArray2P(neighborhoodAggregationNew(S))


def SA_AMGsetup(A, param: MGparam, symm: bool = True, nrhs: int = 1, verbose: bool = False) -> None:
    """Build the SA-AMG hierarchy (SA-AMG.jl:8-76)."""
    if not symm:
        raise RuntimeError("not supported yet...")
    if param.relaxType not in ("Jac", "Jac-GMRES", "SPAI"):
        raise ValueError("Unknown relaxation type !!!!")
    _release_device(param)
    As = [_as_csr(A)]
    Ps, Rs, relaxPrecs = [], [], []
    Cop = As[0].nnz
    for l in range(1, param.levels):
        t0 = time.perf_counter()
        Al = As[l - 1]
        d = getRelaxPrec(Al, param.relaxType, param.relaxParam)
        P0 = getAggregation(Al, param.strongConnParam)
        if P0.shape[0] == P0.shape[1]:
            if verbose:
                print(f"Stopped Coarsening at level {l}")
            param.levels = l                                             # l.35-42: relaxPrecs[1:l-1]
            break
        relaxPrecs.append(d)
        DA = sp.csr_matrix((Al.data * np.repeat(d, np.diff(Al.indptr)), Al.indices, Al.indptr), shape=Al.shape)   # (AT*diag(d))' (l.44): row i times d[i]
        absDA = np.abs(DA.data)
        rho = min(float(absDA.sum()), float(absDA.max()))                # entry-wise norms (l.45)
        del absDA
        P = (P0 - (1.33 / rho) * spgemm(DA, P0)).tocsr()                 # l.46
        P.sort_indices()
        R = transpose_csr(P)                                             # l.47
        Ps.append(P)
        Rs.append(R)
        Ac = galerkin(R, Al, P)                                          # l.50
        As.append(Ac)
        Cop += Ac.nnz
        if verbose:
            print(f"MG setup: {Al.shape[0]} took:{time.perf_counter() - t0:.3f}")
    if verbose:
        print("MG Setup: Operator complexity = ", Cop / As[0].nnz)
    nc = As[-1].shape[0]
    As[-1] = _as_csr(As[-1] + 1e-8 * float(abs(As[-1]).sum()) * sp.identity(nc, format="csr"))   # l.63
    defineCoarsestAinv(param, As[-1])
    param.As, param.Ps, param.Rs, param.relaxPrecs = As, Ps, Rs, relaxPrecs
    param.Meshes = []
    adjustMemoryForNumRHS(param, nrhs, verbose)
