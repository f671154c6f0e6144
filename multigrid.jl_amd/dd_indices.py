"""Index functions of the reference's DomainDecomposition sub-package that the hot path's callers use
(src/DomainDecomposition/DDIndices.jl, DDService.jl): the box rule, nodal index lists per sub-domain and the padded
index array the hybrid Kaczmarz smoother walks.  Host logic (numpy); 1-based indices as in Julia."""
from __future__ import annotations

import numpy as np


def cs2loc(cs: int, n):
    """Linear (1-based) -> per-dimension (1-based) sub-domain index, x fastest (DDService.jl:38-48)."""
    n = [int(k) for k in n]
    cs = int(cs) - 1
    loc = []
    for d in range(len(n)):
        loc.append(cs % n[d] + 1)
        cs //= n[d]
    return np.asarray(loc, dtype=np.int64)


def loc2cs(loc, n) -> int:
    """Per-dimension (1-based) -> linear (1-based) (DDService.jl:27-36)."""
    cs, stride = 1, 1
    for d in range(len(n)):
        cs += (int(loc[d]) - 1) * stride
        stride *= int(n[d])
    return cs


def getOriginalBoundingBoxCells(NumCells, overlap, i, nc):
    """Cells of sub-domain i: div(nc, NumCells) cells per box, the last box takes the remainder (DDIndices.jl:41-47)."""
    NumCells, i, nc = (np.asarray(a, dtype=np.int64) for a in (NumCells, i, nc))
    size = nc // NumCells
    upper_left = (i - 1) * size + 1
    bottom_right = np.where(i == NumCells, nc, upper_left + size - 1)
    return upper_left, bottom_right


def getBoxWithOverlap(upper_left, bottom_right, nc, overlap):
    """Grow the box by `overlap` on every side that is not a boundary of the mesh (DDIndices.jl:61-92)."""
    upper_left, bottom_right, nc, overlap = (np.asarray(a, dtype=np.int64) for a in (upper_left, bottom_right, nc, overlap))
    return (np.where(upper_left > 1, upper_left - overlap, upper_left),
            np.where(bottom_right < nc, bottom_right + overlap, bottom_right))


def getNodalIndicesOfCell(NumCells, overlap, i, nc):
    """1-based nodal indices (x fastest) of sub-domain i; neighbouring boxes share their face nodes
    (DDIndices.jl:141-162).  The reference's 3-D plane stride is (nc[1]+1)^2 (l.157): exact for nc[1] == nc[2], kept."""
    nc = np.asarray(nc, dtype=np.int64)
    ul, br = getOriginalBoundingBoxCells(NumCells, overlap, i, nc)
    ul, br = getBoxWithOverlap(ul, br + 1, nc + 1, overlap)
    strides = [1, int(nc[0]) + 1, (int(nc[0]) + 1) * (int(nc[0]) + 1)][: len(nc)]
    idx = np.zeros(1, dtype=np.int64)
    for d in reversed(range(len(nc))):                      # x varies fastest in the flattened result
        r = (np.arange(ul[d], br[d] + 1, dtype=np.int64) - (1 if d else 0)) * strides[d]
        idx = (r[None, :] + idx[:, None]).ravel() if d == len(nc) - 1 else (idx[:, None] + r[None, :]).ravel()
    return idx


def getIndicesOfCellsArray(mesh, overlap, numDomains, getIndicesOfCell=getNodalIndicesOfCell):
    """domainLength x prod(numDomains) UInt32 array, column ic = the indices of sub-domain ic, zero padded
    (DDService.jl:2-18; the column length is that of the box the reference probes with, l.9)."""
    ncells = np.asarray(mesh.n, dtype=np.int64)
    numDomains = [int(k) for k in numDomains]
    probe = getIndicesOfCell(numDomains, overlap, ncells // 2 + 1, ncells)
    arr = np.zeros((len(probe), int(np.prod(numDomains))), dtype=np.uint32, order="F")
    for ic in range(1, arr.shape[1] + 1):
        ii = getIndicesOfCell(numDomains, overlap, cs2loc(ic, numDomains), ncells)
        if len(ii) > arr.shape[0]:
            raise IndexError("BoundsError: sub-domain %d lists %d indices, the array holds %d" % (ic, len(ii), arr.shape[0]))
        arr[: len(ii), ic - 1] = ii
    return arr
