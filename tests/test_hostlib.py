"""-m "not gpu": the native helpers of the hierarchy SETUP (csrc/mg_host.cpp through multigrid.jl_amd/hostlib.py) against scipy / numpy
restatements of the reference lines they serve: Galerkin products (MGsetup.jl:102, SA-AMG.jl:50), `sparse(P')` (SA-AMG.jl:47), the
strength matrix and `S + S'` (SA-AMG.jl:88-116), the aggregation sweep (SA-AMG.jl:119-211: 32-bit 0-based arrays against the 64-bit
1-based entry point), the column sums of squares of getSPAIprec (MGsetup.jl:359-362)."""
import numpy as np
import pytest
import scipy.sparse as sp

from multigrid_jl_amd import hostlib as H


def _rand(n, m, density, seed, dtype_idx=np.int32):
    M = sp.random(n, m, density=density, random_state=seed, format="csr", dtype=np.float64)
    M.sort_indices()
    M.indices = M.indices.astype(dtype_idx)
    M.indptr = M.indptr.astype(dtype_idx)
    return M


def _structural(A, B):
    """Pattern of A*B with the entries whose terms cancel kept (Julia's and scipy's sparse products keep them)."""
    Ai, Bi = A.copy(), B.copy()
    Ai.data[:] = 1.0
    Bi.data[:] = 1.0
    S = (Ai @ Bi).tocsr()
    S.sort_indices()
    return S


@pytest.mark.parametrize("n,k,m,da,db,idx", [(300, 200, 150, 0.05, 0.08, np.int32), (64, 3000, 40, 0.3, 0.02, np.int32), (500, 500, 4000, 0.004, 0.001, np.int32),
                                              (120, 80, 60, 0.1, 0.1, np.int64)])
def test_spgemm_pattern_and_values(n, k, m, da, db, idx, monkeypatch):
    A, B = _rand(n, k, da, 1, idx), _rand(k, m, db, 2, idx)
    # a product whose terms cancel exactly stays an entry
    A = A.tolil(); B = B.tolil()
    A.rows[0], A.data[0] = [0, 1], [1.0, -1.0]               # row 0 of A: two entries only
    B[0, 0], B[1, 0] = 2.0, 2.0
    A, B = sp.csr_matrix(A), sp.csr_matrix(B)
    A.sort_indices(); B.sort_indices()
    A.indices, A.indptr, B.indices, B.indptr = (v.astype(idx) for v in (A.indices, A.indptr, B.indices, B.indptr))
    S = _structural(A, B)
    want = (A @ B).tocsr()
    for two_pass in ("", "1"):          # symbolic + numeric phases / the count + fill pair
        monkeypatch.setenv("MG_HOST_SPGEMM_TWO_PASS", two_pass)
        C = H.spgemm(A, B, nthreads=3)
        assert C.has_sorted_indices and np.array_equal(C.indptr, S.indptr) and np.array_equal(C.indices, S.indices)
        assert C[0, 0] == 0.0 and 0 in C.indices[C.indptr[0]:C.indptr[1]]
        assert abs(C - want).max() <= 1e-14 * max(1.0, abs(want).max())


def test_transpose_add_transpose_and_column_norms():
    A = _rand(400, 300, 0.05, 5)
    T = H.transpose_csr(A, nthreads=3)
    W = sp.csr_matrix(A.T)
    W.sort_indices()
    assert T.has_sorted_indices and np.array_equal(T.indptr, W.indptr) and np.array_equal(T.indices, W.indices) and np.array_equal(T.data, W.data)
    # column sums of squares: the bits of the serial scatter, whatever the thread count
    want = np.bincount(A.indices, weights=A.data * A.data, minlength=A.shape[1])
    for nt in (1, 3, 8):
        assert np.array_equal(H.col_sumsq(A, nthreads=nt), want)
    # S + S' on a structurally symmetric S, zero sums dropped, own arrays
    B = _rand(200, 200, 0.05, 6)
    S = (B + B.T).tocsr()
    S.sort_indices()
    S.indices, S.indptr = S.indices.astype(np.int32), S.indptr.astype(np.int32)
    S.data = np.random.default_rng(7).standard_normal(S.nnz)
    i, j = 3, int(S.indices[S.indptr[3]])                     # make one pair cancel: S[i,j] = -S[j,i]
    if i != j:
        S[j, i] = -S[i, j]
    before = (S.indptr.copy(), S.indices.copy(), S.data.copy())
    got = H.add_transpose(S, nthreads=3)
    want = (S + S.T).tocsr()
    want.eliminate_zeros()
    want.sort_indices()
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices) and np.allclose(got.data, want.data, rtol=0, atol=0)
    assert got[i, j] == 0.0 and j not in got.indices[got.indptr[i]:got.indptr[i + 1]] or i == j
    assert all(np.array_equal(a, b) for a, b in zip(before, (S.indptr, S.indices, S.data)))      # the operand is left alone
    assert not np.shares_memory(got.indices, S.indices) and not np.shares_memory(got.indptr, S.indptr)


def test_strength_matrix_and_aggregation_paths_agree():
    import multigrid_jl_amd as mg
    from multigrid_jl_amd import sa_amg
    A, _ = mg.anisotropic_divsiggrad([12, 10, 8], weights=(1.0, 0.25, 0.0625))
    A = sp.csr_matrix(A)
    A.sort_indices()
    theta = 0.4
    S = sa_amg.getStrengthMatrix(A, theta)                     # native: strength, S + S', zero-free
    # the vectorised restatement of SA-AMG.jl:88-116
    n = A.shape[0]
    W = (-A).tocsr()
    W.sort_indices()
    mm = 1e-16 * W.data.max()
    rows = np.repeat(np.arange(n), np.diff(W.indptr))
    rowmax = np.maximum(mm, np.maximum.reduceat(W.data, W.indptr[:-1]))
    W.data = W.data * (1.0 / rowmax)[rows]
    W.data[W.indices == rows] = 1.0
    W.data[W.data < theta] = 0.0
    W = (W + W.T).tocsr()
    W.eliminate_zeros()
    W.sort_indices()
    assert np.array_equal(S.indptr, W.indptr) and np.array_equal(S.indices, W.indices) and np.array_equal(S.data, W.data)
    # aggregation: scipy's 0-based int32 arrays as they are against the reference-convention entry point (1-based Int64)
    a32 = H.sa_aggregate(S)
    S64 = S.copy()
    S64.indices, S64.indptr = S64.indices.astype(np.int64), S64.indptr.astype(np.int64)
    a64 = H.sa_aggregate(S64)
    assert np.array_equal(a32, a64) and a32.min() >= 1 and a32.max() <= n
