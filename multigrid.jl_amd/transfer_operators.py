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
