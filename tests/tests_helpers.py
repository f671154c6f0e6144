import numpy as np
import scipy.sparse as sp


def divsiggrad(n, shift, seed=42):
    from multigrid_jl_amd.operators import getRegularMesh, getNodalDivSigGradMatrix, entrynorm1
    rng = np.random.default_rng(seed)
    mesh = getRegularMesh([0, 1] * len(n), n)
    m = np.exp(rng.standard_normal(mesh.nc))
    A = getNodalDivSigGradMatrix(mesh, m)
    A = (A + shift * entrynorm1(A) * sp.identity(A.shape[0])).tocsr()
    A.sort_indices()
    return A
