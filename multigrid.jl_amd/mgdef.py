"""Hierarchy container and constructors (host side).

Mirrors reference src/Multigrid/MGdef.jl: ``MGparam`` (l.91-116), ``getMGparam`` (l.149-161),
``clear!`` (l.179-189), ``destroyCoarsestLU`` (l.191-206), ``hierarchyExists`` (l.208),
``copySolver`` (l.138-145).

Storage convention.  The reference keeps every operator transposed as a ``SparseMatrixCSC``
so that ``A*x`` is ``AT'*x`` (MGdef.jl:75-77): CSC of A' *is* CSR of A.  Here the same arrays are held
as scipy CSR matrices named for what they apply:

    reference ``As[l]``  (AT, CSC)            <->  ``As[l]``  CSR of A_l            (n_l  x n_l)
    reference ``Ps[l]``  (PT, CSC n_c x n_f)  <->  ``Ps[l]``  CSR of P_l            (n_f  x n_c)
    reference ``Rs[l]``  (RT, CSC n_f x n_c)  <->  ``Rs[l]``  CSR of R_l            (n_c  x n_f)

so ``X.indptr`` is Julia's ``colptr-1`` and ``X.indices`` is ``rowval-1``.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Callable, List, Optional

import numpy as np


@dataclass
class MGparam:
    levels: int = 3
    numCores: int = 8
    maxOuterIter: int = 20
    relativeTol: float = 1e-6
    relaxType: str = "SPAI"
    relaxParam: Any = 1.0
    relaxPre: Callable[[int], int] = None
    relaxPost: Callable[[int], int] = None
    cycleType: str = "V"
    Ps: List[Any] = field(default_factory=list)
    Rs: List[Any] = field(default_factory=list)
    As: List[Any] = field(default_factory=list)
    relaxPrecs: List[Any] = field(default_factory=list)
    nrhs: int = 0                      # what memCycle is sized for (adjustMemoryForNumRHS)
    coarseSolveType: str = "NoMUMPS"
    LU: Any = None
    doTranspose: int = 0
    strongConnParam: float = 0.4
    FilteringParam: float = 0.0
    Meshes: List[Any] = field(default_factory=list)
    transferOperatorType: str = "FullWeighting"
    singlePrecision: bool = False
    VAL: Any = np.float64
    IND: Any = np.int64
    # device side (no reference counterpart): handle of the HIP cycle library + last residual history
    device: Any = None
    resvec: Optional[np.ndarray] = None
    flag: int = 0


def getMGparam(VAL=np.float64, IND=np.int64, levels=3, numCores=8, maxIter=20, relativeTol=1e-6,
               relaxType="SPAI", relaxParam=1.0, relaxPre=2, relaxPost=2, cycleType="V",
               coarseSolveType="NoMUMPS", strongConnParam=0.4, FilteringParam=0.0,
               transferOperatorType="FullWeighting") -> MGparam:
    """Positional constructor with the reference's defaults (MGdef.jl:149-161).

    ``relaxPre``/``relaxPost`` may be ints or functions of the (1-based) level, as in MGdef.jl:98-99,158-159.
    Only ``Float64``/``Int64`` are on the device path (fp32/complex deliberately off).
    """
    if np.dtype(VAL) != np.float64:
        raise TypeError("only VAL=Float64 is supported on the device path")
    if np.dtype(IND) != np.int64:
        raise TypeError("only IND=Int64 is supported")
    pre = relaxPre if callable(relaxPre) else (lambda level, _k=int(relaxPre): _k)
    post = relaxPost if callable(relaxPost) else (lambda level, _k=int(relaxPost): _k)
    if cycleType not in ("V", "W", "F", "K"):
        raise ValueError("cycleType must be one of 'V','W','F','K'")
    return MGparam(levels=int(levels), numCores=int(numCores), maxOuterIter=int(maxIter),
                   relativeTol=float(relativeTol), relaxType=str(relaxType), relaxParam=relaxParam,
                   relaxPre=pre, relaxPost=post, cycleType=cycleType, coarseSolveType=str(coarseSolveType),
                   strongConnParam=float(strongConnParam), FilteringParam=float(FilteringParam),
                   transferOperatorType=str(transferOperatorType), VAL=np.float64, IND=np.int64)


def hierarchyExists(param: MGparam) -> bool:
    return len(param.As) > 0


def destroyCoarsestLU(param: MGparam) -> None:
    param.LU = None


def _release_device(param: MGparam) -> None:
    if param.device is not None:
        param.device.close()
        param.device = None


def clear_(param: MGparam) -> None:
    """``clear!(param)``: drop the hierarchy, the scratch memory and the device handle."""
    param.Ps, param.Rs, param.As = [], [], []
    param.relaxPrecs = []
    param.Meshes = []
    param.nrhs = 0
    destroyCoarsestLU(param)
    _release_device(param)


def copySolver(MG: MGparam) -> MGparam:
    """Copies the solver parameters without the setup and allocated memory (MGdef.jl:138-145)."""
    return getMGparam(MG.VAL, MG.IND, MG.levels, MG.numCores, MG.maxOuterIter, MG.relativeTol, MG.relaxType,
                      MG.relaxParam, MG.relaxPre, MG.relaxPost, MG.cycleType, MG.coarseSolveType,
                      MG.strongConnParam, MG.FilteringParam, MG.transferOperatorType)
