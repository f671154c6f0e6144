"""jInv ``AbstractSolver`` wrappers (host side): ``MGsolver`` / ``SA_AMGsolver``.

Mirrors reference src/Multigrid/MGWrapper.jl (l.6-103) and SAAMGWrapper.jl (l.5-94): lazy setup on the first
solve, transpose handling for non-symmetric operators, the Krylov switch, timing counters.  The solve itself
runs on the device (``solve_funcs``).  Krylov drivers available here: "PCG" (``solveCG_MG``), "BiCGSTAB"
(``solveBiCGSTAB_MG``), "GMRES" (``solveGMRES_MG``, FGMRES(5) as MGWrapper.jl:69-71) and plain cycles (anything else
for ``MGsolver``, ``solveMG``).
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Any

import numpy as np
import scipy.sparse as sp

from .mgdef import MGparam, clear_, copySolver as _copy_param, hierarchyExists
from .mgsetup import MGsetup, transposeHierarchy
from .sa_amg import SA_AMGsetup
from .solve_funcs import solveBiCGSTAB_MG, solveCG_MG, solveGMRES_MG, solveMG


@dataclass
class MGsolver:
    MG: MGparam
    Krylov: str = "GMRES"
    sym: int = 1              # 0 = unsymmetric, 1 = s.p.d., 2 = general symmetric
    out: int = -1
    isTranspose: bool = False
    doClear: int = 0
    tol: float = 1e-6
    nIter: int = 0
    timeSetup: float = 0.0
    timeSolve: float = 0.0
    kind: str = "GMG"         # "GMG" (MGsetup) or "SA" (SA_AMGsetup)


def getMGsolver(MG: MGparam, Mesh, sym, Krylov: str = "GMRES", out: int = -1) -> MGsolver:
    """MGWrapper.jl:22-25."""
    MG.Meshes = [Mesh]
    return MGsolver(MG, Krylov, int(sym), out, False, 0, MG.relativeTol, 0, 0.0, 0.0, "GMG")


def getSA_AMGsolver(MG: MGparam, Krylov: str = "BiCGSTAB", sym: int = 1, out: int = -1) -> MGsolver:
    """SAAMGWrapper.jl:19-24."""
    if sym != 1:
        print("Non-symmetric AMG version is not implemented yet...")
    return MGsolver(MG, Krylov, int(sym), out, False, 0, MG.relativeTol, 0, 0.0, 0.0, "SA")


def solveLinearSystem_(A, B: np.ndarray, X: np.ndarray, param: MGsolver, doTranspose: int = 0):
    """``solveLinearSystem!(A,B,X,param,doTranspose)`` (MGWrapper.jl:27-86 / SAAMGWrapper.jl:26-78): X in place."""
    if sp.issparse(B):
        B = np.asarray(B.todense())
    if B.ndim == 2 and B.shape[1] == 1:
        B = B[:, 0].copy()
        Xv = X.reshape(-1)
    else:
        Xv = X
    if param.doClear == 1:
        clear_(param.MG)
    if np.linalg.norm(B) == 0.0:
        X[...] = 0.0
        return X, param
    nrhs = 1 if B.ndim == 1 else B.shape[1]
    verbose = param.out > 0
    if not hierarchyExists(param.MG):
        doTi = (doTranspose + 1) % 2 if param.isTranspose else doTranspose
        # The reference hands MGsetup the TRANSPOSED matrix (it stores AT and applies AT'), so it transposes for
        # doTransposeIterative == 0 (MGWrapper.jl:54-56).  This package's MGsetup(M) sets up and applies M itself (CSR
        # of the operator, tests/test_host_api.py: As[1] == R*A*P), so the operator to hand over is A for
        # doTransposeIterative == 0 and A' for 1: the condition is the other way round.
        if param.sym != 1 and doTi == 1:
            A = sp.csr_matrix(A.T)
        t0 = time.perf_counter()
        if param.kind == "SA":
            SA_AMGsetup(A, param.MG, param.sym == 1, nrhs, verbose)
        else:
            MGsetup(A, param.MG.Meshes[0], param.MG, nrhs, verbose)
        param.timeSetup += time.perf_counter() - t0
        param.MG.doTranspose = doTranspose
    if param.sym != 1 and doTranspose != param.MG.doTranspose:
        t0 = time.perf_counter()
        transposeHierarchy(param.MG)
        param.timeSetup += time.perf_counter() - t0
    t0 = time.perf_counter()
    Bf = np.asfortranarray(B)
    if param.Krylov == "BiCGSTAB":
        _, _, num_iter, _ = solveBiCGSTAB_MG(param.MG.As[0], param.MG, Bf, Xv, verbose)
    elif param.Krylov == "PCG":
        _, _, num_iter = solveCG_MG(param.MG.As[0], param.MG, Bf, Xv, verbose)
    elif param.Krylov == "GMRES":
        if param.kind == "SA":
            raise ValueError("SA_AMGsolver supports Krylov 'BiCGSTAB' or 'PCG' (SAAMGWrapper.jl:61-65)")
        _, _, num_iter, _ = solveGMRES_MG(param.MG.As[0], param.MG, Bf, Xv, True, 5, verbose)     # MGWrapper.jl:69-71
    elif param.kind == "SA":
        raise ValueError("SA_AMGsolver supports Krylov 'BiCGSTAB' or 'PCG' (SAAMGWrapper.jl:61-65)")
    else:
        _, _, num_iter = solveMG(param.MG, Bf, Xv, verbose)
    param.nIter += num_iter * nrhs
    param.timeSolve += time.perf_counter() - t0
    return X, param


def setupSolver(A, s: MGsolver) -> MGsolver:
    """MGWrapper.jl:88-91."""
    s.MG = MGsetup(A, s.MG.Meshes[0], s.MG, 1, s.out > 0)
    return s


def copySolverWrapper(s: MGsolver) -> MGsolver:
    """copySolver(s) (MGWrapper.jl:93-97): copies what is necessary, not the hierarchy."""
    mg = _copy_param(s.MG)
    mg.Meshes = list(s.MG.Meshes[:1])
    return MGsolver(mg, s.Krylov, s.sym, s.out, s.isTranspose, s.doClear, s.tol, 0, 0.0, 0.0, s.kind)


def clearSolver_(s: MGsolver) -> None:
    """clear!(s) (MGWrapper.jl:100-103)."""
    clear_(s.MG)
    s.doClear = 0
