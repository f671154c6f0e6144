"""ORACLE - TEST INFRASTRUCTURE ONLY.  CPU restatement (numpy/scipy, fp64) of the multigrid cycle of
JuliaInv/Multigrid.jl v0.8.0.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this; the product (multigrid.jl_amd/) never does.

PARITY UNPINNED BY VALUE.  The reference path is Julia + the un-vendored Fortran package
ParSpMatVec 0.1.1 (Manifest.toml:87-91); neither Julia nor gfortran exists in the build image and the
reference's tests hold no golden vectors (every input is an unseeded rand, every assertion a residual
threshold: test/Multigrid/testGMG.jl:49,55; testGMGRAPforPoisson.jl:30,40,78; testSAforDivSigGrad.jl:31,38,112).
What pins this oracle instead (tests/test_oracle.py):
  (i)   the reference's known-answer thresholds re-expressed with seeded inputs;
  (ii)  independent formulations: dense two-grid error-propagation algebra on tiny grids;
  (iii) SpMatMul's in-tree definition ``target = beta*target + alpha*A*x`` (SpMatMul.jl:5,9).

Operators are scipy CSR matrices = the reference's transposed CSC arrays (MGdef.jl:75-77).
Every function cites the reference lines it restates and keeps their operation order.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


# --------------------------------------------------------------------------------------------------
# SpMatMul.jl
# --------------------------------------------------------------------------------------------------
def SpMatMul(alpha, A, x, beta, target):
    """target = beta*target + alpha*A*x  (SpMatMul.jl:4-13; fallback mul!(target,adjoint(AT),x,alpha,beta), l.9).

    scipy's csr_matvec accumulates each row sequentially in stored (sorted-column) order - the same
    in-row order as a row-parallel CSR kernel.  Blocks (N x nrhs) are handled column by column.
    """
    Ax = A @ x
    if beta == 0.0:
        target[...] = alpha * Ax
    else:
        target[...] = beta * target + alpha * Ax
    return target


def SpMatMul2(A, x, target):
    """alpha = 1, beta = 0 (SpMatMul.jl:16-26)."""
    return SpMatMul(1.0, A, x, 0.0, target)


def addVectors(alpha, x, target):
    """target += alpha*x  (SpMatMul.jl:29-37, BLAS.axpy!)."""
    target += alpha * x


# --------------------------------------------------------------------------------------------------
# MGcycle.jl
# --------------------------------------------------------------------------------------------------
def _dmul(d, r):
    return d * r if r.ndim == 1 else d[:, None] * r


def relax(A, r, x, b, d, numit):
    """Damped Jacobi / SPAI-0 sweeps (MGcycle.jl:122-136).  r is left stale after the last sweep (l.124).
    Note the loop runs numit-1 times and the final update is unconditional: numit=0 still updates once."""
    for _ in range(1, numit):
        x += _dmul(d, r)                         # x .+= d.*r            (l.129)
        SpMatMul(-1.0, A, x, 0.0, r)             # r = -A x              (l.130)
        addVectors(1.0, b, r)                    # r = r + b             (l.131)
    x += _dmul(d, r)                             # (l.134)
    return x


def FGMRES_relaxation(Afun, r0, x0, inner, prec, TOL):
    """FGMRES.jl:48-126: x0 += Z*t, t = pinv((AZ)'(AZ)) (AZ)'r0 over the directions z_1 = prec(r0),
    z_j = prec(A z_{j-1}); blocks are treated as one long vector (l.51)."""
    shape = r0.shape
    nm = r0.size
    rnorm0 = np.linalg.norm(r0)
    H = np.zeros((inner, inner))
    xi = np.zeros(inner)
    t = np.zeros(inner)
    Z = np.zeros((nm, inner))
    AZ = np.zeros((nm, inner))
    w = None
    rnorms = np.zeros(inner)
    for j in range(inner):
        z = prec(r0) if j == 0 else prec(w)                  # l.83-87
        Z[:, j] = np.asarray(z).reshape(nm, order="F")
        w = Afun(z)                                          # l.91
        AZ[:, j] = np.asarray(w).reshape(nm, order="F")
        t = AZ.T @ AZ[:, j]                                  # gemv 'C' (l.95)
        xi[j] = np.dot(AZ[:, j], np.asarray(r0).reshape(nm, order="F"))   # l.97
        H[:, j] = t
        H[j, :] = t
        H = 0.5 * H + 0.5 * H.T
        t = np.linalg.pinv(H) @ xi                           # l.102
        rnorms[j] = np.sqrt(abs(t @ (H @ t) - 2.0 * (t @ xi) + rnorm0 ** 2))   # l.104
        if rnorms[j] < TOL:                                  # l.114-117
            break
    if inner > 0:
        x0 += (Z @ t).reshape(shape, order="F")              # l.121-123
    return x0


def solveCoarsest(param, b, x):
    """Default branch: z = param.LU \\ b ; x[:] = z  (MGcycle.jl:177-178).  coarseSolveType "GMRES" (l.152-168): x = 0, one
    restart of KrylovMethods.fgmres(Afun, b, 10, tol=0.01, maxIter=1) preconditioned by M2(v) = d .* v with
    d = param.LU = relaxParam ./ diag(AT) (MGsetup.jl:334); blocks of right-hand sides go to blockFGMRES (l.164-166)."""
    if getattr(param, "coarseSolveType", "") == "GMRES":
        b = np.asarray(b)
        A = param.As[-1]
        d = np.asarray(param.LU, dtype=np.float64)
        if b.ndim == 2 and b.shape[1] > 1:     # MGcycle.jl:164-166: blockFGMRES(Afun, b, 10, tol = 0.01, maxIter = 1, M = M2, X = x)
            Z = blockFGMRES(lambda V: A @ V, b, 10, tol=0.01, maxIter=1, M=lambda V: d[:, None] * V,
                            X=np.zeros((A.shape[0], b.shape[1])))[0]
            x[...] = Z.reshape(x.shape)
            return x
        z = fgmres(lambda v: A @ v, b.reshape(-1), 10, tol=0.01, maxIter=1, M=lambda v: d * v, x=np.zeros(A.shape[0]))[0]
        x[...] = z.reshape(x.shape)
        return x
    x[...] = param.LU.solve(np.asarray(b))
    return x


class _Mem:
    """CYCLEmem per level (MGdef.jl:56-60), sized by adjustMemoryForNumRHS (MGsetup.jl:166-223)."""

    def __init__(self, param, nrhs):
        self.b, self.r, self.x = [], [], []
        for l, A in enumerate(param.As):
            n = A.shape[0]
            shp = (n,) if nrhs == 1 else (n, nrhs)
            self.r.append(np.zeros(shp))
            self.x.append(np.zeros(shp))
            self.b.append(np.zeros(shp))
        self.b[-1] = self.r[-1]                  # coarsest .b aliases .r (MGsetup.jl:217-218)


def recursiveCycle(param, b, x, level, mem=None, cycleType=None):
    """One cycle from `level` (1-based), operation order of MGcycle.jl:1-118."""
    nrhs = 1 if b.ndim == 1 else b.shape[1]
    if mem is None:
        mem = _Mem(param, nrhs)
    if cycleType is None:
        cycleType = param.cycleType
    As = param.As
    nlevels = len(As)
    if level == nlevels:                         # l.13-18
        r = mem.r[level - 1]
        r[...] = b
        return solveCoarsest(param, r, x)
    A = As[level - 1]
    r = mem.r[level - 1]
    r[...] = b                                   # l.26-28
    if np.linalg.norm(x) > 0.0:                  # l.29-31
        SpMatMul(-1.0, A, x, 1.0, r)
    D = param.relaxPrecs[level - 1]
    P = param.Ps[level - 1]
    R = param.Rs[level - 1]
    npresmth = param.relaxPre(level)
    npostsmth = param.relaxPost(level)
    gmres_relax = getattr(param, "relaxType", "Jac") == "Jac-GMRES"
    gmresTol = 1e-5                              # l.5
    MM = lambda xx: _dmul(D, xx)                 # l.36-38
    Afun = lambda z: A @ z                       # getAfun (SolveFuncs.jl:65-71)
    if gmres_relax:
        x = FGMRES_relaxation(Afun, r, x, npresmth, MM, gmresTol)        # l.48-50
    else:
        x = relax(A, r, x, b, D, npresmth)       # l.54
    SpMatMul(-1.0, A, x, 0.0, r)                 # l.58
    addVectors(1.0, b, r)                        # l.60
    xc = mem.x[level]
    xc[...] = 0.0                                # l.63-64
    bc = mem.b[level]
    bc = SpMatMul2(R, r, bc)                     # l.66
    if level == nlevels - 1:
        xc = solveCoarsest(param, bc, xc)        # l.67-69
    else:
        if cycleType == "K":                     # l.72-76
            Ac = As[level]

            def MMG(v):
                yz = np.zeros_like(v)
                return recursiveCycle(param, v, yz, level + 1, mem, "K").copy()

            xc = FGMRES_relaxation(lambda z: Ac @ z, bc.copy(), xc, 2, MMG, gmresTol)
        else:
            xc = recursiveCycle(param, bc, xc, level + 1, mem, cycleType)      # l.78
        if cycleType == "W":
            xc = recursiveCycle(param, bc, xc, level + 1, mem, "W")            # l.79-80
        elif cycleType == "F":
            xc = recursiveCycle(param, bc, xc, level + 1, mem, "V")            # l.81-84
    SpMatMul(1.0, P, xc, 1.0, x)                 # x += P xc             (l.90)
    r[...] = b                                   # l.92
    SpMatMul(-1.0, A, x, 1.0, r)                 # l.93
    if gmres_relax:
        x = FGMRES_relaxation(Afun, r, x, npostsmth, MM, gmresTol)       # l.96-98
    else:
        x = relax(A, r, x, b, D, npostsmth)      # l.102
    return x


# --------------------------------------------------------------------------------------------------
# SolveFuncs.jl
# --------------------------------------------------------------------------------------------------
def solveMG(param, b, x, verbose=False, history=None):
    """solveMG (SolveFuncs.jl:3-39).  Returns (x, param, iter); x updated in place.
    `history` (list) receives [res_init, res after cycle 1, ...] and, if it is a dict, also x per cycle."""
    nrhs = 1 if b.ndim == 1 else b.shape[1]
    mem = _Mem(param, nrhs)
    tol = param.relativeTol
    maxIter = param.maxOuterIter
    A = param.As[0]
    r = mem.r[0]
    r[...] = b
    if np.linalg.norm(x) == 0:
        res = np.linalg.norm(b)
    else:
        SpMatMul(-1.0, A, x, 1.0, r)
        res = np.linalg.norm(r)
    res_init = res
    resvec = [res_init]
    xs = []
    it = 0
    for count in range(1, maxIter + 1):
        x = recursiveCycle(param, b, x, 1, mem)
        SpMatMul(-1.0, A, x, 0.0, r)
        addVectors(1.0, b, r)
        it += 1
        res_prev = res
        res = np.linalg.norm(r)
        resvec.append(res)
        xs.append(x.copy())
        if verbose:
            print(f"Cycle {count} done with relres: {res / res_init}. Convergence factor: {res / res_prev}")
        if res / res_init < tol:
            break
    if isinstance(history, list):
        history.extend(resvec)
    elif isinstance(history, dict):
        history["resvec"] = np.array(resvec)
        history["xs"] = xs
    return x, param, it


# --------------------------------------------------------------------------------------------------
# Setup restated with explicit loops / dense algebra (small grids only): independent of the
# vectorised host code in multigrid.jl_amd/mgsetup.py that it checks.
# --------------------------------------------------------------------------------------------------
def get1DFWInterp_dense(n_nodes, geometric=False):
    """GeometricTransferOperators.jl:22-46, entry by entry."""
    if n_nodes > 2:
        T = np.zeros((n_nodes, n_nodes))
        for i in range(n_nodes):
            T[i, i] = 1.0
            if i > 0:
                T[i, i - 1] = 0.5
            if i < n_nodes - 1:
                T[i, i + 1] = 0.5
        if n_nodes % 2 == 1:
            P = T[:, 0::2]                                       # l.27-29
        elif geometric:
            P = np.eye(n_nodes)                                  # l.31-33
        else:
            cols = list(range(0, n_nodes, 2)) + [n_nodes - 1]    # l.35
            P = T[:, cols].copy()
            P[-2:, -2:] = np.eye(2)                              # l.36
    else:
        P = np.eye(n_nodes)                                      # l.41-43
    return P, P.shape[1]


def getFWInterp_dense(n_nodes, geometric=False):
    """kron(P3, kron(P2, P1)) (GeometricTransferOperators.jl:5-20)."""
    Ps = [get1DFWInterp_dense(int(k), geometric)[0] for k in n_nodes]
    P = Ps[0]
    for Pk in Ps[1:]:
        P = np.kron(Pk, P)
    return P, np.array([p.shape[1] for p in Ps])


def getSPAIprec_dense(Adense):
    """Q_i = diag_i / sum_j |AT[i,j]|^2, AT = A' (MGsetup.jl:359-362): column norms of A."""
    AT = Adense.T
    s = (AT ** 2).sum(axis=1)
    return np.diag(AT) / s


def getRelaxPrec_dense(Adense, relaxType, relaxParam):
    """MGsetup.jl:142-149."""
    if relaxType in ("Jac", "Jac-GMRES"):
        return relaxParam / np.diag(Adense)
    if relaxType == "SPAI":
        return relaxParam * getSPAIprec_dense(Adense)
    raise ValueError("Unknown relaxation type !!!!")


class Hierarchy:
    """Minimal MGparam stand-in the oracle cycle runs on."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def MGsetup_dense(Adense, n_cells, levels, relaxType, relaxParam, relaxPre, relaxPost, cycleType="V",
                  relativeTol=1e-6, maxOuterIter=20):
    """MGsetup (MGsetup.jl:7-138), FullWeighting + Galerkin, with dense matrices (tiny grids only)."""
    n = np.asarray(n_cells, dtype=np.int64)
    dim = n.size
    As = [np.array(Adense, dtype=np.float64)]
    Ps, Rs, relaxPrecs = [], [], []
    for l in range(1, levels):
        A = As[-1]
        P, nc_nodes = getFWInterp_dense(n + 1, False)           # l.54
        R = (0.5 ** dim) * P.T                                  # l.56-60
        relaxPrecs.append(getRelaxPrec_dense(A, relaxType, relaxParam))   # l.76
        if P.shape[0] == P.shape[1]:                            # l.84-92
            break
        Ps.append(P)
        Rs.append(R)
        As.append(R @ (A @ P))                                  # l.102
        n = nc_nodes - 1
    csr = lambda M: sp.csr_matrix(M)
    pre = relaxPre if callable(relaxPre) else (lambda level: relaxPre)
    post = relaxPost if callable(relaxPost) else (lambda level: relaxPost)
    return Hierarchy(As=[csr(a) for a in As], Ps=[csr(p) for p in Ps], Rs=[csr(r) for r in Rs],
                     relaxPrecs=relaxPrecs, LU=spla.splu(sp.csc_matrix(As[-1])), relaxPre=pre, relaxPost=post,
                     cycleType=cycleType, relativeTol=relativeTol, maxOuterIter=maxOuterIter,
                     dense_As=As, dense_Ps=Ps, dense_Rs=Rs)


def two_grid_error_matrix(h: Hierarchy, nu1, nu2):
    """Textbook two-grid error propagation  E = S^nu2 (I - P Ac^-1 R A) S^nu1,  S = I - diag(d) A.
    Independent formulation used to check recursiveCycle on 2-level hierarchies."""
    A = h.dense_As[0]
    P = h.dense_Ps[0]
    R = h.dense_Rs[0]
    Ac = h.dense_As[1]
    n = A.shape[0]
    S = np.eye(n) - np.diag(h.relaxPrecs[0]) @ A
    K = np.eye(n) - P @ np.linalg.solve(Ac, R @ A)
    return np.linalg.matrix_power(S, nu2) @ K @ np.linalg.matrix_power(S, nu1)


# --------------------------------------------------------------------------------------------------
# SA-AMG.jl restated with literal (1-based, pure-Python) loops - small cases only.
# --------------------------------------------------------------------------------------------------
def getStrengthMatrix_loops(A, theta):
    """SA-AMG.jl:88-116 on the CSC arrays of AT (= CSR arrays of A), loop for loop.  Returns dense S + S'."""
    A = sp.csr_matrix(A)
    A.sort_indices()
    n = A.shape[0]
    colptr = A.indptr + 1
    rowval = A.indices + 1
    nzval = -A.data.copy()                                         # S = -AT
    mm = 1e-16 * nzval.max()
    for j in range(1, n + 1):
        maxVal_j = mm
        for g in range(colptr[j - 1], colptr[j]):
            if nzval[g - 1] > maxVal_j:
                maxVal_j = nzval[g - 1]
        scal_k = 1.0 / maxVal_j
        for g in range(colptr[j - 1], colptr[j]):
            nzval[g - 1] *= scal_k
        for g in range(colptr[j - 1], colptr[j]):
            if rowval[g - 1] == j:
                nzval[g - 1] = 1.0
        for g in range(colptr[j - 1], colptr[j]):
            if nzval[g - 1] < theta:
                nzval[g - 1] = 0.0
    S = np.zeros((n, n))
    for j in range(1, n + 1):
        for g in range(colptr[j - 1], colptr[j]):
            S[rowval[g - 1] - 1, j - 1] = nzval[g - 1]
    return S + S.T                                                 # Julia drops the zeros of the sum (note N3)


def neighborhoodAggregationNew_loops(Sdense):
    """SA-AMG.jl:119-211, literal, including the nested second loop of pass 3 (note N2).
    Sdense: symmetric; its non-zeros are the stored entries."""
    n = Sdense.shape[0]
    cols = [np.nonzero(Sdense[:, k])[0] + 1 for k in range(n)]     # sorted row indices of column k (1-based)
    tau = 3.0
    aggr = [0] * (n + 1)
    aux = [0.0] * (n + 1)
    aux_count = [0] * (n + 1)
    avg = sum(len(c) for c in cols) / n
    for k in range(1, n + 1):
        if len(cols[k - 1]) > tau * avg:
            aux_count[k] = -1
    for k in range(1, n + 1):
        flag = False
        if aux_count[k] == -1:
            continue
        for j in cols[k - 1]:
            if aggr[j] != 0:
                flag = True
                break
        if not flag:
            for j in cols[k - 1]:
                if aux_count[j] != -1:
                    aggr[j] = k
                    aux_count[k] += 1
    for k in range(1, n + 1):
        flag = False
        if aux_count[k] != -1:
            continue
        aux_count[k] = 0
        for j in cols[k - 1]:
            if aggr[j] != 0:
                flag = True
                break
        if not flag:
            for j in cols[k - 1]:
                aggr[j] = k
                aux_count[k] += 1
    for k in range(1, n + 1):
        chosen_score = 0.0
        chosen = 0
        if aggr[k] == 0:
            for j in cols[k - 1]:
                if aggr[j] > 0:
                    a = aggr[j]
                    aux[a] += Sdense[j - 1, k - 1]
                for j2 in cols[k - 1]:
                    if aggr[j2] > 0:
                        a = aggr[j2]
                        if chosen_score < aux[a] / aux_count[a]:
                            chosen_score = aux[a] / aux_count[a]
                            chosen = a
                            aux[a] = 0
                aggr[k] = -chosen
    for k in range(1, n + 1):
        if aggr[k] < 0:
            aggr[k] = -aggr[k]
    return np.array(aggr[1:], dtype=np.int64)


def aggrArray2P_loops(aggr):
    """SA-AMG.jl:213-224."""
    n = len(aggr)
    fine2coarse = [0] * (n + 1)
    cnt = 0
    for i in range(1, n + 1):
        if aggr[i - 1] == i:
            cnt += 1
            fine2coarse[i] = cnt
    P = np.zeros((n, cnt))
    for i in range(1, n + 1):
        c = fine2coarse[aggr[i - 1]]
        if c == 0:
            raise RuntimeError("nodes without aggregates")
        P[i - 1, c - 1] = 1.0
    return P


def SA_AMGsetup_dense(Adense, levels, relaxType, relaxParam, theta, relaxPre, relaxPost, cycleType="V",
                      relativeTol=1e-6, maxOuterIter=20):
    """SA_AMGsetup (SA-AMG.jl:8-76) in the reference's own transposed variables, dense."""
    As = [np.array(Adense, dtype=np.float64)]
    Ps, Rs, relaxPrecs = [], [], []
    for l in range(1, levels):
        A = As[-1]
        AT = A.T
        d = getRelaxPrec_dense(A, relaxType, relaxParam)               # l.28
        n = A.shape[0]
        if n <= 100:                                                   # l.79-81
            break
        S = getStrengthMatrix_loops(sp.csr_matrix(A), theta)
        P0 = aggrArray2P_loops(neighborhoodAggregationNew_loops(S))    # n x Nc
        P0T = P0.T                                                     # P0 = sparse(P0') (l.34)
        if P0T.shape[0] == P0T.shape[1]:
            break
        relaxPrecs.append(d)
        DAT = AT @ np.diag(d)                                          # l.44
        rho = min(np.abs(DAT).sum(), np.abs(DAT).max())                # l.45, entry-wise norms (note N1)
        PT = P0T - (1.33 / rho) * (P0T @ DAT)                          # l.46
        RT = PT.T                                                      # l.47
        Ps.append(PT.T)                                                # applied operator P  = PT'
        Rs.append(RT.T)                                                # applied operator R  = RT'
        Act = PT @ AT @ RT                                             # l.50
        As.append(Act.T)
    nc = As[-1].shape[0]
    As[-1] = As[-1] + 1e-8 * np.abs(As[-1]).sum() * np.eye(nc)         # l.63
    pre = relaxPre if callable(relaxPre) else (lambda level: relaxPre)
    post = relaxPost if callable(relaxPost) else (lambda level: relaxPost)
    csr = lambda M: sp.csr_matrix(M)
    return Hierarchy(As=[csr(a) for a in As], Ps=[csr(p) for p in Ps], Rs=[csr(r) for r in Rs],
                     relaxPrecs=relaxPrecs, LU=spla.splu(sp.csc_matrix(As[-1])), relaxPre=pre, relaxPost=post,
                     cycleType=cycleType, relativeTol=relativeTol, maxOuterIter=maxOuterIter,
                     dense_As=As, dense_Ps=Ps, dense_Rs=Rs)


# --------------------------------------------------------------------------------------------------
# KrylovMethods.cg (v0.6.0 @master, un-vendored: reference Manifest.toml:35-41) as called by solveCG_MG
# (SolveFuncs.jl:104-116).  The package source is not under /root/reference; this restates its published
# algorithm (preconditioned CG with ||r||/||b|| <= tol stopping, flag -2 on alpha = Inf or < 0, -9 on b = 0).
# --------------------------------------------------------------------------------------------------
def cg(Afun, b, tol=1e-2, maxIter=100, M=None, x=None):
    n = b.size
    if np.linalg.norm(b) == 0:
        return np.zeros(n), -9, 0.0, 0, np.array([0.0])
    if x is None:
        x = np.zeros(n)
        r = b.copy()
    else:
        r = b - Afun(x)
    z = M(r) if M is not None else r.copy()
    p = z.copy()
    nr0 = np.linalg.norm(b)
    resvec = np.zeros(maxIter)
    flag = -1
    last = 0
    for it in range(1, maxIter + 1):
        last = it
        Ap = Afun(p)
        gamma = np.dot(r, z)
        alpha = gamma / np.dot(p, Ap)
        if np.isinf(alpha) or alpha < 0:
            flag = -2
            break
        x += alpha * p
        r -= alpha * Ap
        resvec[it - 1] = np.linalg.norm(r) / nr0
        if resvec[it - 1] <= tol:
            flag = 0
            break
        z = M(r) if M is not None else r.copy()
        beta = np.dot(z, r) / gamma
        p = z + beta * p
    return x, flag, resvec[last - 1], last, resvec[:last]


def getMultigridPreconditioner(param, B):
    """M(b) = (z .= 0; recursiveCycle(param,b,z,1); z)  (SolveFuncs.jl:59)."""
    nrhs = 1 if B.ndim == 1 else B.shape[1]
    mem = _Mem(param, nrhs)
    z = np.zeros_like(B)

    def MMG(b):
        z[...] = 0.0
        recursiveCycle(param, b, z, 1, mem)
        return z.copy()

    return MMG


def solveCG_MG(param, b, x0):
    """solveCG_MG (SolveFuncs.jl:104-116), one right-hand side."""
    A = param.As[0]
    x, flag, rn, it, resvec = cg(lambda v: A @ v, b, tol=param.relativeTol, maxIter=param.maxOuterIter,
                                 M=getMultigridPreconditioner(param, b), x=x0)
    return x, flag, it, resvec


def bicgstb(Afun, b, tol=1e-6, maxIter=100, M1=None, x=None):
    """KrylovMethods.bicgstb (v0.6.0, un-vendored) as called by solveBiCGSTAB_MG (SolveFuncs.jl:87-101), restated from
    the published algorithm (van der Vorst; Barrett et al. 'Templates').  M2 = identity.  Flags as documented in
    include/mgvcycle.h (mg_bicgstab_FP64)."""
    n = b.size
    bn = np.linalg.norm(b)
    if bn == 0:
        return np.zeros(n), -9, 0, np.array([0.0])
    if x is None:
        x = np.zeros(n)
    r = b - Afun(x)
    M = M1 if M1 is not None else (lambda v: v.copy())
    resvec = [np.linalg.norm(r) / bn]
    if resvec[0] < tol:
        return x, 0, 0, np.array(resvec)
    rtld = r.copy()
    omega, alpha, rho1 = 1.0, 0.0, 0.0
    p = np.zeros(n)
    v = np.zeros(n)
    flag, it = -1, 0
    for k in range(1, maxIter + 1):
        it = k
        rho = np.dot(rtld, r)
        if rho == 0.0:
            flag = -2
            break
        if k > 1:
            beta = (rho / rho1) * (alpha / omega)
            p = r + beta * (p - omega * v)
        else:
            p = r.copy()
        phat = M(p)
        v = Afun(phat)
        alpha = rho / np.dot(rtld, v)
        s = r - alpha * v
        sn = np.linalg.norm(s) / bn
        resvec.append(sn)
        if sn < tol:
            x += alpha * phat
            flag = -3
            break
        shat = M(s)
        t = Afun(shat)
        omega = np.dot(t, s) / np.dot(t, t)
        x += alpha * phat + omega * shat
        r = s - omega * t
        err = np.linalg.norm(r) / bn
        resvec.append(err)
        if err <= tol:
            flag = 0
            break
        if omega == 0.0:
            flag = -2
            break
        rho1 = rho
    return x, flag, it, np.array(resvec)


def solveBiCGSTAB_MG(param, b, x0):
    A = param.As[0]
    return bicgstb(lambda v: A @ v, b, tol=param.relativeTol, maxIter=param.maxOuterIter,
                   M1=getMultigridPreconditioner(param, b), x=x0)


def fgmres(Afun, b, restrt, tol=1e-2, maxIter=100, M=None, x=None):
    """KrylovMethods.fgmres (v0.6.0, un-vendored) as called by solveGMRES_MG (SolveFuncs.jl:119-133), flexible variant,
    restated from the published algorithm (Saad, FGMRES(m)): MGS Arnoldi, Givens rotations, residual estimate per
    inner step.  Returns (x, flag, total inner steps, resvec)."""
    n = b.size
    bn = np.linalg.norm(b)
    if bn == 0:
        return np.zeros(n), -9, 0, np.zeros(0)
    if x is None:
        x = np.zeros(n)
    Mf = M if M is not None else (lambda v: v.copy())
    r = b - Afun(x)
    rn = np.linalg.norm(r)
    if rn / bn < tol:
        return x, 0, 0, np.zeros(0)
    m = restrt
    resvec, flag, total = [], -1, 0
    for it in range(1, maxIter + 1):
        V = np.zeros((n, m + 1))
        Z = np.zeros((n, m))
        H = np.zeros((m + 1, m))
        cs = np.zeros(m)
        sn = np.zeros(m)
        s = np.zeros(m + 1)
        V[:, 0] = r / rn
        s[0] = rn
        used = 0
        for i in range(m):
            Z[:, i] = Mf(V[:, i])
            w = Afun(Z[:, i])
            for k in range(i + 1):
                H[k, i] = np.dot(w, V[:, k])
                w = w - H[k, i] * V[:, k]
            H[i + 1, i] = np.linalg.norm(w)
            if H[i + 1, i] != 0:
                V[:, i + 1] = w / H[i + 1, i]
            for k in range(i):
                t = cs[k] * H[k, i] + sn[k] * H[k + 1, i]
                H[k + 1, i] = -sn[k] * H[k, i] + cs[k] * H[k + 1, i]
                H[k, i] = t
            rr = np.hypot(H[i, i], H[i + 1, i])
            cs[i], sn[i] = (1.0, 0.0) if rr == 0 else (H[i, i] / rr, H[i + 1, i] / rr)
            H[i, i], H[i + 1, i] = rr, 0.0
            s[i + 1] = -sn[i] * s[i]
            s[i] = cs[i] * s[i]
            err = abs(s[i + 1]) / bn
            resvec.append(err)
            total += 1
            used = i + 1
            if err <= tol:
                flag = 0
                break
        y = np.linalg.solve(np.triu(H[:used, :used]), s[:used])
        x = x + Z[:, :used] @ y
        if flag == 0:
            break
        r = b - Afun(x)
        rn = np.linalg.norm(r)
        if rn / bn <= tol:
            flag = 0
            break
    return x, flag, total, np.array(resvec)


def solveGMRES_MG(param, b, x0, inner):
    A = param.As[0]
    return fgmres(lambda v: A @ v, b, inner, tol=param.relativeTol, maxIter=param.maxOuterIter,
                  M=getMultigridPreconditioner(param, b), x=x0)


# ---------------------------------------------------------------------------------------------------------------------
# Hybrid Kaczmarz (deps/src/parRelax.h:7-43, src/Multigrid/parRelax.jl:31-65; src/DomainDecomposition/DDService.jl:2-18,
# DDIndices.jl:41-92,141-162).  PINNED against the reference's own compiled code: tests/test_reference_parrelax.py runs
# oracle/_ref/parRelax.so (built from deps/src/parRelax.c by oracle/Makefile) with numCores = 1 on the same arrays.
# ---------------------------------------------------------------------------------------------------------------------
def cs2loc(cs, n):
    """DDService.jl:38-48 (1-based linear -> 1-based per-dimension box index, x fastest)."""
    n = [int(k) for k in n]
    if len(n) == 3:
        return [(cs - 1) % n[0] + 1, ((cs - 1) % (n[0] * n[1])) // n[0] + 1, (cs - 1) // (n[0] * n[1]) + 1]
    return [(cs - 1) % n[0] + 1, (cs - 1) // n[0] + 1]


def getOriginalBoundingBoxCells(NumCells, overlap, i, nc):
    """DDIndices.jl:41-47."""
    NumCells, i, nc = np.asarray(NumCells), np.asarray(i), np.asarray(nc)
    cellSize = nc // NumCells
    ul = (i - 1) * cellSize + 1
    br = ul + (cellSize - 1)
    br[i == NumCells] = nc[i == NumCells]
    return ul, br


def getBoxWithOverlap(ul, br, nc, overlap):
    """DDIndices.jl:61-92: extend the box by the overlap where it does not touch the boundary."""
    ul, br = np.array(ul).copy(), np.array(br).copy()
    nul, nbr = ul.copy(), br.copy()
    for d in range(len(nc)):
        if ul[d] > 1:
            nul[d] -= overlap[d]
        if br[d] < nc[d]:
            nbr[d] += overlap[d]
    return nul, nbr


def getNodalIndicesOfCell(NumCells, overlap, i, nc):
    """DDIndices.jl:141-162 (1-based nodal indices of sub-domain i, x fastest; its 3-D stride is (nc[1]+1)*(nc[1]+1),
    l.157 - reproduced literally)."""
    nc = np.asarray(nc)
    dim = len(nc)
    ul, br = getOriginalBoundingBoxCells(NumCells, overlap, i, nc)
    ul, br = getBoxWithOverlap(ul, br + 1, nc + 1, overlap)
    r = [np.arange(ul[d], br[d] + 1) for d in range(dim)]
    if dim == 2:
        I1, I2 = np.meshgrid(r[0], r[1], indexing="ij")
        return (I1.ravel(order="F") + (I2.ravel(order="F") - 1) * (nc[0] + 1)).astype(np.int64)
    I1, I2, I3 = np.meshgrid(r[0], r[1], r[2], indexing="ij")
    return (I1.ravel(order="F") + (I2.ravel(order="F") - 1) * (nc[0] + 1)
            + (I3.ravel(order="F") - 1) * ((nc[0] + 1) * (nc[0] + 1))).astype(np.int64)


def getIndicesOfCellsArray(n_cells, overlap, numDomains, getIndicesOfCell=getNodalIndicesOfCell):
    """DDService.jl:2-18: column ic = the (1-based) indices of sub-domain ic, zero padded to the length of the middle one."""
    n_cells = np.asarray(n_cells)
    numDomains = [int(k) for k in numDomains]
    mid = getIndicesOfCell(numDomains, overlap, n_cells // 2 + 1, n_cells)
    Arr = np.zeros((len(mid), int(np.prod(numDomains))), dtype=np.uint32, order="F")
    for ic in range(1, Arr.shape[1] + 1):
        II = getIndicesOfCell(numDomains, overlap, np.asarray(cs2loc(ic, numDomains)), n_cells)
        Arr[: len(II), ic - 1] = II
    return Arr


def hybrid_kaczmarz_invdiag(A, omega):
    """parRelax.jl:44: omega ./ sum(conj(AT).*AT, dims=1)  = omega / ||row i of A||^2."""
    A = sp.csr_matrix(A)
    return omega / np.asarray(A.multiply(A).sum(axis=1)).ravel()


def applyHybridKaczmarz(A, ArrIdxs, x, b, invD, numit):
    """parRelax.h:7-43 with ONE thread (domains in order, rows of a domain in order): for every listed row,
    inner = (b_i - a_i.x) * invD_i ; x[cols] += inner * a_i.  x (n x nrhs, column-major or 1-D) is updated in place."""
    A = sp.csr_matrix(A)
    rp, ci, va = A.indptr, A.indices, A.data
    X = x.reshape(A.shape[0], -1, order="F") if x.ndim == 1 else x
    B = b.reshape(A.shape[0], -1, order="F") if b.ndim == 1 else b
    for _ in range(int(numit)):
        for dom in range(ArrIdxs.shape[1]):
            for row1 in ArrIdxs[:, dom]:
                if row1 == 0:
                    continue
                i = int(row1) - 1
                s, e = rp[i], rp[i + 1]
                for c in range(X.shape[1]):
                    inner = B[i, c]
                    for k in range(s, e):                 # sequential, stored order (l.24-27)
                        inner -= va[k] * X[ci[k], c]
                    inner *= invD[i]
                    for k in range(s, e):                 # l.29-32
                        X[ci[k], c] += inner * va[k]
    return x


# ---------------------------------------------------------------------------------------------------------------------
# Block Krylov methods (KrylovMethods v0.6.0, un-vendored: blockCG, blockBiCGSTB, blockFGMRES; call sites
# SolveFuncs.jl:95,113,130).  The package source is not in the reference tree, so these restate the PUBLISHED
# algorithms the package implements (O'Leary 1980 block CG with a pseudo-inverse of P'AP; El Guennouni, Jbilou and
# Sadok 2003 block BiCGSTAB; block flexible GMRES with block modified Gram-Schmidt).  Stopping rules follow the
# package's documented convention: per-column relative residuals, all columns <= tol (blockCG / blockBiCGSTB), Frobenius
# norm for blockFGMRES.  Exact flag values cannot be checked offline; they are documented in include/mgvcycle.h.
# ---------------------------------------------------------------------------------------------------------------------
def _colnorms(M_):
    return np.sqrt((M_ * M_).sum(axis=0))


def blockCG(Afun, B, tol=1e-2, maxIter=100, M=None, X=None):
    """O'Leary's block CG: Alpha = pinv(P'Q) P'R ; X += P Alpha ; R -= Q Alpha ; Beta = -pinv(P'Q) Q'Z ; P = Z + P Beta.
    Returns (X, flag, resmat[iter, nrhs] of ||r_j||/||b_j||, iterations)."""
    B = np.asarray(B, dtype=np.float64)
    n, k = B.shape
    nb = _colnorms(B)
    if not np.any(nb > 0):
        return np.zeros((n, k)), -9, np.zeros((0, k)), 0
    nb = np.where(nb > 0, nb, 1.0)
    if X is None:
        X = np.zeros((n, k))
        R = B.copy()
    else:
        R = B - Afun(X)
    Mf = M if M is not None else (lambda V: V.copy())
    Z = Mf(R).copy()
    P = Z.copy()
    res = []
    flag, it = -1, 0
    for it in range(1, maxIter + 1):
        Q = Afun(P).copy()
        PTQ = P.T @ Q
        pinvPTQ = np.linalg.pinv(0.5 * (PTQ + PTQ.T))            # P'AP is symmetric for the s.p.d. operators CG is for
        Alpha = pinvPTQ @ (P.T @ R)
        X += P @ Alpha
        R -= Q @ Alpha
        res.append(_colnorms(R) / nb)
        if res[-1].max() <= tol:
            flag = 0
            break
        Z = Mf(R).copy()
        Beta = -pinvPTQ @ (Q.T @ Z)
        P = Z + P @ Beta
    return X, flag, np.array(res), it


def blockBiCGSTB(Afun, B, tol=1e-6, maxIter=100, M1=None, X=None):
    """Block BiCGSTAB (El Guennouni / Jbilou / Sadok) with right preconditioning by M1 (M2 = identity):
    Phat = M(P); V = A Phat; alpha = (R0'V) \ (R0'R); S = R - V alpha; Shat = M(S); T = A Shat;
    omega = <T,S>_F / <T,T>_F; X += Phat alpha + omega Shat; R = S - omega T; beta = -(R0'V) \ (R0'T);
    P = R + (P - omega V) beta.  Residual entries: max_j ||s_j||/||b_j|| after the half step, max_j ||r_j||/||b_j||
    after the full one.  Returns (X, flag, iterations, resvec)."""
    B = np.asarray(B, dtype=np.float64)
    n, k = B.shape
    nb = _colnorms(B)
    if not np.any(nb > 0):
        return np.zeros((n, k)), -9, 0, np.zeros(0)
    nb = np.where(nb > 0, nb, 1.0)
    if X is None:
        X = np.zeros((n, k))
    Mf = M1 if M1 is not None else (lambda V: V.copy())
    R = B - Afun(X)
    resvec = [(_colnorms(R) / nb).max()]
    if resvec[0] < tol:
        return X, 0, 0, np.array(resvec)
    R0 = R.copy()
    P = R.copy()
    flag, it = -1, 0
    for it in range(1, maxIter + 1):
        Phat = Mf(P).copy()
        V = Afun(Phat).copy()
        RtV = R0.T @ V
        alpha = np.linalg.solve(RtV, R0.T @ R)
        S = R - V @ alpha
        sn = (_colnorms(S) / nb).max()
        resvec.append(sn)
        if sn < tol:
            X += Phat @ alpha
            flag = -3
            break
        Shat = Mf(S).copy()
        T = Afun(Shat).copy()
        tt = (T * T).sum()
        if tt == 0.0:
            flag = -2
            break
        omega = (T * S).sum() / tt
        X += Phat @ alpha + omega * Shat
        R = S - omega * T
        err = (_colnorms(R) / nb).max()
        resvec.append(err)
        if err <= tol:
            flag = 0
            break
        if omega == 0.0:
            flag = -2
            break
        beta = -np.linalg.solve(RtV, R0.T @ T)
        P = R + (P - omega * V) @ beta
    return X, flag, it, np.array(resvec)


def _chol_semidefinite(G, rtol=1e-14):
    """Upper triangular Rf with G = Rf' Rf for a positive SEMI-definite Gram matrix: a column whose pivot falls below
    rtol * G[c,c] is linearly dependent on the earlier ones and gets a zero row / zero diagonal (its direction is dropped)."""
    k = G.shape[0]
    Rf = np.zeros((k, k))
    for c in range(k):
        d = G[c, c] - np.dot(Rf[:c, c], Rf[:c, c])
        if G[c, c] <= 0.0 or d <= rtol * G[c, c]:
            continue
        Rf[c, c] = np.sqrt(d)
        for j in range(c + 1, k):
            Rf[c, j] = (G[c, j] - np.dot(Rf[:c, c], Rf[:c, j])) / Rf[c, c]
    return Rf


def _tri_pinv_apply(W, Rf):
    """Q = W Rf^+ for the upper triangular Rf of _chol_semidefinite (zero pivots give zero columns of Q)."""
    k = Rf.shape[0]
    Q = np.zeros_like(W)
    for c in range(k):
        if Rf[c, c] == 0.0:
            continue
        Q[:, c] = (W[:, c] - Q[:, :c] @ Rf[:c, c]) / Rf[c, c]
    return Q


def _cholqr(W):
    """Orthonormal basis of the columns of W by Cholesky QR applied twice (Gram matrix -> triangular factor -> Q = W Rf^+),
    rank deficiency tolerated: (Q, Rf) with W = Q Rf, Rf upper triangular."""
    R1 = _chol_semidefinite(W.T @ W)
    Q = _tri_pinv_apply(W, R1)
    R2 = _chol_semidefinite(Q.T @ Q)
    Q = _tri_pinv_apply(Q, R2)
    return Q, R2 @ R1


def blockFGMRES(Afun, B, restrt, tol=1e-2, maxIter=100, M=None, X=None):
    """Block flexible GMRES(restrt): block modified Gram-Schmidt Arnoldi on n x k blocks (orthonormalised by Cholesky QR
    applied twice - the package uses Julia's qr; any orthonormal basis of the same block gives the same iterates), the
    small block least-squares problem solved exactly after every inner step, residual estimate ||xi - H Y||_F/||B||_F.
    Returns (X, flag, total inner steps, resvec)."""
    B = np.asarray(B, dtype=np.float64)
    n, k = B.shape
    bn = np.linalg.norm(B)
    if bn == 0:
        return np.zeros((n, k)), -9, 0, np.zeros(0)
    if X is None:
        X = np.zeros((n, k))
    Mf = M if M is not None else (lambda V: V.copy())
    R = B - Afun(X)
    if np.linalg.norm(R) / bn < tol:
        return X, 0, 0, np.zeros(0)
    m = restrt
    resvec, flag, total = [], -1, 0
    for it in range(1, maxIter + 1):
        V = [None] * (m + 1)
        Z = [None] * m
        H = np.zeros(((m + 1) * k, m * k))
        V[0], Rf = _cholqr(R)
        xi = np.zeros(((m + 1) * k, k))
        xi[:k] = Rf
        used, Y = 0, None
        for j in range(m):
            Z[j] = Mf(V[j]).copy()
            W = Afun(Z[j]).copy()
            for i in range(j + 1):
                Hij = V[i].T @ W
                H[i * k:(i + 1) * k, j * k:(j + 1) * k] = Hij
                W = W - V[i] @ Hij
            V[j + 1], Hn = _cholqr(W)
            H[(j + 1) * k:(j + 2) * k, j * k:(j + 1) * k] = Hn
            Hb = H[:(j + 2) * k, :(j + 1) * k]
            Y = np.linalg.lstsq(Hb, xi[:(j + 2) * k], rcond=None)[0]
            err = np.linalg.norm(xi[:(j + 2) * k] - Hb @ Y) / bn
            resvec.append(err)
            total += 1
            used = j + 1
            if err <= tol:
                flag = 0
                break
        for j in range(used):
            X = X + Z[j] @ Y[j * k:(j + 1) * k]
        if flag == 0:
            break
        R = B - Afun(X)
        if np.linalg.norm(R) / bn <= tol:
            flag = 0
            break
    return X, flag, total, np.array(resvec)
