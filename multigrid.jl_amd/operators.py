"""Synthetic test operators on regular meshes (host side, scipy.sparse).

The reference takes its operators from jInv.Mesh (un-vendored, jInv 1.0.0,
reference Manifest.toml:149-155): ``getRegularMesh``, ``getNodalGradientMatrix``,
``getNodalLaplacianMatrix``, ``getNodalDivSigGradMatrix``.  Their published
definitions are restated here so that the reference's own test problems
(test/Multigrid/testGMGRAPforPoisson.jl:8-13,59-64; testSAforDivSigGrad.jl:9-14,96-100;
testGMG.jl:21-23,48,65) can be rebuilt without Julia.

Node ordering is x-fastest (Julia column-major), as jInv and
DomainDecomposition/DDIndices.jl:141-162 assume.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


class RegularMesh:
    """Mirror of jInv.Mesh.RegularMesh: ``domain`` [x0,x1,y0,y1(,z0,z1)], ``n`` cells per dim."""

    def __init__(self, domain, n):
        self.domain = np.asarray(domain, dtype=np.float64)
        self.n = np.asarray(n, dtype=np.int64)
        self.dim = int(self.n.size)
        if self.domain.size != 2 * self.dim:
            raise ValueError("domain must hold 2*dim entries")
        self.h = (self.domain[1::2] - self.domain[0::2]) / self.n

    @property
    def nc(self):
        return int(np.prod(self.n))

    @property
    def nn(self):
        return int(np.prod(self.n + 1))


def getRegularMesh(domain, n):
    return RegularMesh(domain, n)


def _ddx(n, h):
    """1-D nodal difference: n x (n+1), rows (-1, 1)/h."""
    return sp.diags([-np.ones(n), np.ones(n)], [0, 1], shape=(n, n + 1), format="csr") / h


def _av(n):
    """1-D node->cell average: n x (n+1), rows (1/2, 1/2)."""
    return sp.diags([0.5 * np.ones(n), 0.5 * np.ones(n)], [0, 1], shape=(n, n + 1), format="csr")


def _kron3(a3, a2, a1):
    return sp.kron(a3, sp.kron(a2, a1, format="csr"), format="csr")


def getNodalGradientMatrix(mesh: RegularMesh):
    """G: nodes -> edges, blocks stacked x-edges, y-edges(, z-edges)."""
    n, h = mesh.n, mesh.h
    I = [sp.identity(int(k) + 1, format="csr") for k in n]
    D = [_ddx(int(k), float(hk)) for k, hk in zip(n, h)]
    if mesh.dim == 2:
        return sp.vstack([sp.kron(I[1], D[0], format="csr"),
                          sp.kron(D[1], I[0], format="csr")], format="csr")
    return sp.vstack([_kron3(I[2], I[1], D[0]),
                      _kron3(I[2], D[1], I[0]),
                      _kron3(D[2], I[1], I[0])], format="csr")


def getNodalLaplacianMatrix(mesh: RegularMesh):
    """G'G assembled as the Kronecker sum of the 1-D operators D_k'D_k (same matrix, much cheaper to build)."""
    n, h = mesh.n, mesh.h
    I = [sp.identity(int(k) + 1, format="csr") for k in n]
    L = []
    for k, hk in zip(n, h):
        D = _ddx(int(k), float(hk))
        L.append((D.T @ D).tocsr())
    if mesh.dim == 2:
        A = sp.kron(I[1], L[0], format="csr") + sp.kron(L[1], I[0], format="csr")
    else:
        A = _kron3(I[2], I[1], L[0]) + _kron3(I[2], L[1], I[0]) + _kron3(L[2], I[1], I[0])
    A = A.tocsr()
    A.sort_indices()
    return A


def getEdgeAverageMatrix(mesh: RegularMesh):
    """Ae: edges -> cell centres (average of the edges of each direction, summed over directions / dim)."""
    n = mesh.n
    I = [sp.identity(int(k), format="csr") for k in n]
    A = [_av(int(k)) for k in n]
    if mesh.dim == 2:
        blocks = [sp.kron(A[1], I[0], format="csr"), sp.kron(I[1], A[0], format="csr")]
    else:
        blocks = [_kron3(A[2], A[1], I[0]), _kron3(A[2], I[1], A[0]), _kron3(I[2], A[1], A[0])]
    return sp.hstack(blocks, format="csr") / mesh.dim


def getNodalDivSigGradMatrix(mesh: RegularMesh, sigma):
    """A = G' diag(Ae' (sigma * dim)) G : cell coefficient averaged to the edges."""
    G = getNodalGradientMatrix(mesh)
    Ae = getEdgeAverageMatrix(mesh)
    sig_e = Ae.T @ (np.asarray(sigma, dtype=np.float64).ravel() * mesh.dim)
    return (G.T @ sp.diags(sig_e) @ G).tocsr()


def opnorm1(A):
    """Julia ``opnorm(A,1)``: max column sum of |a_ij|."""
    return float(abs(A).sum(axis=0).max())


def entrynorm1(A):
    """Julia ``norm(A,1)`` on a sparse matrix: entry-wise sum |a_ij| (SURVEY note N1)."""
    return float(abs(A).sum())


def poisson_shifted(n_cells, domain=None):
    """A = G'G + 1e-4*opnorm(G'G,1)*I  (testGMGRAPforPoisson.jl:59-64).  Returns (A csr, mesh)."""
    n_cells = [int(k) for k in np.atleast_1d(n_cells)]
    if domain is None:
        domain = [0.0, 1.0] * len(n_cells)
    mesh = getRegularMesh(domain, n_cells)
    A = getNodalLaplacianMatrix(mesh)
    A = (A + 1e-4 * opnorm1(A) * sp.identity(A.shape[0], format="csr")).tocsr()
    A.sort_indices()
    return A, mesh


def anisotropic_divsiggrad(n_cells, weights=(1.0, 1e-2, 1e-4), seed=7, shift=1e-6, domain=None):
    """C3 operator: G' diag(w_dir * sigma_edge) G + shift*sum|a_ij|*I, general CSR.

    sigma is a seeded log-normal cell field (``exp.(randn)`` idiom of testSAforDivSigGrad.jl:98-100)
    averaged to the edges; per-direction weights give the anisotropy.
    """
    n_cells = [int(k) for k in np.atleast_1d(n_cells)]
    if domain is None:
        domain = [0.0, 1.0] * len(n_cells)
    mesh = getRegularMesh(domain, n_cells)
    rng = np.random.default_rng(seed)
    sigma = np.exp(rng.standard_normal(mesh.nc))
    G = getNodalGradientMatrix(mesh)
    Ae = getEdgeAverageMatrix(mesh)
    sig_e = Ae.T @ (sigma * mesh.dim)
    n = mesh.n
    if mesh.dim == 2:
        ne = [int(n[0] * (n[1] + 1)), int((n[0] + 1) * n[1])]
    else:
        ne = [int(n[0] * (n[1] + 1) * (n[2] + 1)), int((n[0] + 1) * n[1] * (n[2] + 1)),
              int((n[0] + 1) * (n[1] + 1) * n[2])]
    w = np.concatenate([np.full(k, float(wd)) for k, wd in zip(ne, weights)])
    A = (G.T @ sp.diags(sig_e * w) @ G).tocsr()
    A = (A + shift * entrynorm1(A) * sp.identity(A.shape[0], format="csr")).tocsr()
    A.sort_indices()
    return A, mesh


def seeded_rhs(A, nrhs=1, seed=1234):
    """b = A*u, u ~ U[0,1) from default_rng(seed); b /= ||b||_F  (testGMGRAPforPoisson.jl:30-31)."""
    rng = np.random.default_rng(seed)
    n = A.shape[0]
    u = rng.random((n, nrhs)) if nrhs > 1 else rng.random(n)
    b = A @ u
    b = b / np.linalg.norm(b)
    return np.asfortranarray(b)
