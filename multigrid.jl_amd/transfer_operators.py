"""Full-weighting transfer operators on nodal grids (host side).

Mirrors reference src/Multigrid/GeometricTransferOperators.jl:5-46
(``getFWInterp`` / ``get1DFWInterp``).  ``n_nodes`` is the number of NODES per dimension.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def get1DFWInterp(n_nodes: int, geometric: bool = False):
    """1-D linear interpolation P (n_nodes x nc).  Cases follow GeometricTransferOperators.jl:22-46:

    * odd node count: tridiag(1/2, 1, 1/2) sampled at every other column (l.27-29);
    * even, geometric: identity - coarsening stops (l.31-33);
    * even, algebraic: keep the last node as an extra coarse point and overwrite the
      trailing 2x2 block with the identity (l.35-36);
    * n_nodes <= 2: identity (l.41-43).
    """
    n_nodes = int(n_nodes)
    if n_nodes > 2:
        half = 0.5 * np.ones(n_nodes - 1)
        T = sp.diags([half, np.ones(n_nodes), half], [-1, 0, 1], format="csc")
        if n_nodes % 2 == 1:
            P = T[:, 0::2]
        elif geometric:
            P = sp.identity(n_nodes, format="csc")
        else:
            cols = list(range(0, n_nodes, 2)) + [n_nodes - 1]
            P = T[:, cols].tolil()
            P[n_nodes - 2:, P.shape[1] - 2:] = np.eye(2)
            P = P.tocsc()
            P.eliminate_zeros()
    else:
        P = sp.identity(n_nodes, format="csc")
    P = sp.csr_matrix(P)
    P.sort_indices()
    return P, int(P.shape[1])


def getFWInterp(n_nodes, geometric: bool = False):
    """P = P3 (x) P2 (x) P1 (x-fastest ordering), GeometricTransferOperators.jl:5-20.  Returns (P, nc_nodes)."""
    n_nodes = [int(k) for k in n_nodes]
    Ps, ncs = zip(*[get1DFWInterp(k, geometric) for k in n_nodes])
    if len(n_nodes) == 2:
        P = sp.kron(Ps[1], Ps[0], format="csr")
    elif len(n_nodes) == 3:
        P = sp.kron(Ps[2], sp.kron(Ps[1], Ps[0], format="csr"), format="csr")
    else:
        raise ValueError("getFWInterp: 2-D or 3-D only")
    P.sort_indices()
    return P, np.asarray(ncs, dtype=np.int64)


# ---- coefficient restriction for rediscretisation (GeometricTransferOperators.jl:52-82; Systems.jl:134-183) ----
def get1DRestrictionCells(n: int):
    """2:1 aggregation of cells, rows (1, 1); identity below 8 cells (Systems.jl:134-148)."""
    n = int(n)
    if n < 8:
        return sp.identity(n, format="csr"), n
    nc = n // 2
    if 2 * nc != n:
        raise ValueError("Err: get1DRestrictionCells(): size should be a multiplication of 2")
    rows = np.repeat(np.arange(nc), 2)
    cols = np.arange(n)
    return sp.csr_matrix((np.ones(n), (rows, cols)), shape=(nc, n)), nc


def getRestrictionCellCentered(n):
    """kron(R3, kron(R2, R1)) of the 1-D cell aggregations (Systems.jl:167-183); n = cells per dimension."""
    n = [int(k) for k in n]
    Rs, ncs = zip(*[get1DRestrictionCells(k) for k in n])
    if len(n) == 2:
        R = sp.kron(Rs[1], Rs[0], format="csr")
    elif len(n) == 3:
        R = sp.kron(Rs[2], sp.kron(Rs[1], Rs[0], format="csr"), format="csr")
    else:
        raise ValueError("getRestrictionCells() : Dimension not supported!")
    return R, np.asarray(ncs, dtype=np.int64)


def restrictCellCenteredVariables(rho, n):
    """rho_c = 0.5^dim * (R*rho): the mean over each 2^dim block of cells (GeometricTransferOperators.jl:52-58)."""
    R, nc = getRestrictionCellCentered(n)
    scale = 0.5 ** len(n)
    rho_c = scale * (R @ np.asarray(rho, dtype=np.float64).ravel())
    return rho_c, (R * scale).tocsr()


def restrictNodalVariables(rho, n_nodes):
    """rho_c = 0.5^dim * (P'*rho) with the geometric full-weighting P (GeometricTransferOperators.jl:60-67)."""
    P, nc = getFWInterp(n_nodes, True)
    return (0.5 ** len(n_nodes)) * (P.T @ np.asarray(rho, dtype=np.float64).ravel())
