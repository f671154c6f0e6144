"""Host-side mirror of the reference's cycle API: every call goes to the HIP library.

Mirrors reference src/Multigrid/SolveFuncs.jl:3-39 (``solveMG``), MGcycle.jl:1-118 (``recursiveCycle``),
SpMatMul.jl:4-26 (``SpMatMul``) and SolveFuncs.jl:43-63 (``getMultigridPreconditioner``).
Same names, argument meaning and in-place contract; the arithmetic runs on the MI355X only.
"""
from __future__ import annotations

import numpy as np

from .device import DeviceHierarchy, MG_OP_A, MG_OP_P, MG_OP_R
from .mgdef import MGparam, hierarchyExists
from .mgsetup import adjustMemoryForNumRHS


def _ncols(b):
    return 1 if b.ndim == 1 else int(b.shape[1])


def to_device(param: MGparam, device_id: int = 0) -> DeviceHierarchy:
    """Upload the hierarchy (lifecycle hook at the end of MGsetup/SA_AMGsetup, MGsetup.jl:135-137)."""
    if not hierarchyExists(param):
        raise RuntimeError("The Hierarchy is empty - run a setup first.")
    if param.device is None:
        param.device = DeviceHierarchy(param, device_id=device_id, nrhs=max(1, param.nrhs))
    else:
        # the reference reads cycleType / relaxType / relaxPre / relaxPost from param on every cycle
        # (MGcycle.jl:44-45,72-85): follow changes made after the upload
        param.device.sync_schedule(param)
    return param.device


def solveMG(param: MGparam, b: np.ndarray, x: np.ndarray, verbose: bool = False):
    """``(x, param, iter) = solveMG(param,b,x,verbose)``: x is updated IN PLACE (testGMG.jl:54-55)."""
    adjustMemoryForNumRHS(param, _ncols(b))
    dev = to_device(param)
    _, iters, resvec = dev.solve(b, x, param.relativeTol, param.maxOuterIter)
    param.resvec = resvec
    if verbose:
        for c in range(1, iters + 1):
            print(f"Cycle {c} done with relres: {resvec[c] / resvec[0]}. Convergence factor: {resvec[c] / resvec[c - 1]}")
    return x, param, iters


def recursiveCycle(param: MGparam, b: np.ndarray, x: np.ndarray, level: int = 1):
    """One cycle from the finest level.  Only ``level == 1`` is an entry point of the device library."""
    if level != 1:
        raise ValueError("the device library owns the recursion: only level=1 can be entered from the host")
    adjustMemoryForNumRHS(param, _ncols(b))
    to_device(param).cycle(b, x, -1)
    return x


def getMultigridPreconditioner(param: MGparam, B: np.ndarray, verbose: bool = False):
    """``M(b) = (z .= 0; recursiveCycle(param,b,z,1); z)`` (SolveFuncs.jl:59): x = 0 on entry."""
    if not hierarchyExists(param):
        print("You have to do a setup first.")
    adjustMemoryForNumRHS(param, _ncols(B))
    dev = to_device(param)
    if B.dtype == np.float32:            # mixed precision (SolveFuncs.jl:52-58): bl .= b; cycle in Float64; z2 .= z
        z2 = np.zeros_like(B, order="F")

        def MMG32(b):
            dev.cycle_mixed_f32(np.asfortranarray(b, dtype=np.float32), z2)
            return z2

        return MMG32
    z = np.zeros_like(B, order="F")

    def MMG(b):
        z[...] = 0.0
        dev.cycle(np.asfortranarray(b), z, 1)
        return z

    return MMG


def solveCG_MG(A, param: MGparam, b: np.ndarray, x0: np.ndarray, verbose: bool = False):
    """``(x, param, iter) = solveCG_MG(AT,param,b,x0,verbose)`` (SolveFuncs.jl:104-116): KrylovMethods.cg with
    the multigrid cycle as preconditioner, vectors resident on the device across iterations.  ``A`` is accepted
    for signature parity; the operator applied is ``param.As[1]`` on the device (the reference passes the same
    matrix twice).  x0 is updated in place.  ``size(b,2) > 1`` takes the blockCG branch (l.113), also on the device."""
    adjustMemoryForNumRHS(param, _ncols(b))
    dev = to_device(param)
    x, flag, it, resvec = dev.pcg(b, x0, param.relativeTol, param.maxOuterIter)
    param.resvec = resvec
    param.flag = flag
    if verbose:
        for k, r in enumerate(resvec):
            print(f"{k + 1:3d}\t{r:1.2e}")
    return x, param, it


def solveBiCGSTAB_MG(A, param: MGparam, b: np.ndarray, x0: np.ndarray, verbose: bool = False):
    """``(x, param, iter, nprec) = solveBiCGSTAB_MG(AT,param,b,x0,verbose)`` (SolveFuncs.jl:87-101): KrylovMethods.bicgstb
    with M1 = the multigrid cycle, M2 = identity, on the device; blocks take the blockBiCGSTB branch (l.95)."""
    adjustMemoryForNumRHS(param, _ncols(b))
    dev = to_device(param)
    x, flag, it, resvec = dev.bicgstab(b, x0, param.relativeTol, param.maxOuterIter)
    param.resvec = resvec
    param.flag = flag
    nprec = 2 * it * _ncols(b) + (flag == -3) * _ncols(b)        # SolveFuncs.jl:99 as written
    return x, param, it, nprec


def solveGMRES_MG(A, param: MGparam, b: np.ndarray, x0: np.ndarray, flexible: bool, inner: int, verbose: bool = False):
    """``(x, param, iter, resvec) = solveGMRES_MG(AT,param,b,x0,flexible,inner,verbose)`` (SolveFuncs.jl:119-133):
    KrylovMethods.fgmres with the multigrid cycle as preconditioner on the device (always the flexible variant: the
    cycle is a fixed linear operator, so flexible and standard GMRES generate the same iterates); blocks take the
    blockFGMRES branch (l.130)."""
    adjustMemoryForNumRHS(param, _ncols(b))
    dev = to_device(param)
    x, flag, it, resvec = dev.fgmres(b, x0, inner, param.relativeTol, param.maxOuterIter)
    param.resvec = resvec
    param.flag = flag
    return x, param, it, resvec


_WHICH = {"A": MG_OP_A, "P": MG_OP_P, "R": MG_OP_R}


def SpMatMul(param: MGparam, level: int, which: str, x: np.ndarray, target: np.ndarray,
             alpha: float = 1.0, beta: float = 0.0):
    """``target = beta*target + alpha*Op*x`` (SpMatMul.jl:4-13) with Op = As/Ps/Rs[level] resident on device."""
    adjustMemoryForNumRHS(param, _ncols(x))
    to_device(param).spmv(level, _WHICH[which], alpha, x, beta, target)
    return target
