"""Sharded geometric-multigrid SETUP: every rank builds only its own part of the hierarchy.

``MGsetup`` (reference src/Multigrid/MGsetup.jl:7-138) needs the global fine matrix; at 512^3 (943 M
non-zeros) that cannot be held - let alone multiplied - on every rank's host.  For an operator
given by a LOCAL generator on a regular mesh (the synthetic Poisson/diffusion operators of the benchmarks)
the Galerkin hierarchy is local too: row i of R*A*P only involves nodes within a few grid layers of i.  So
each rank runs the reference's own setup steps (``getFWInterp``, ``RT = P*0.5^dim``, ``Ps*AT*Rs``,
``getRelaxPrec``; MGsetup.jl:54-60,76,102) on its box EXTENDED by a margin: rows closer than the margin to an
artificial cut are wrong and are thrown away, the rows the rank owns are exactly the rows of the global
hierarchy (``tests/test_distributed.py::test_structured_setup_equals_global``).  The first replicated level
is assembled from every rank's rows (one ``all_gather_object``), below it the ordinary ``MGsetup`` runs.

The partition is the DomainDecomposition box rule (DDIndices.jl:41-47), as in ``distributed.box_owner``.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .distributed import DistributedHierarchy, HaloPlan, Partition
from .hostlib import spgemm
from .mgdef import MGparam, getMGparam
from .mgsetup import MGsetup, galerkin, getRelaxPrec
from .operators import getRegularMesh
from .transfer_operators import getFWInterp


def _ids(origin, ext_nodes, glob_nodes):
    """Global x-fastest ids of all nodes of a local box (origin + ext_nodes per dim) in local x-fastest order."""
    dim = len(ext_nodes)
    ax = [np.arange(ext_nodes[k], dtype=np.int64) + int(origin[k]) for k in range(dim)]
    if dim == 2:
        g = ax[0][None, :] + glob_nodes[0] * ax[1][:, None]
    else:
        g = ax[0][None, None, :] + glob_nodes[0] * (ax[1][None, :, None] + glob_nodes[1] * ax[2][:, None, None])
    return g.reshape(-1)


def _outer(masks):
    """x-fastest flattening of the tensor product of per-dimension arrays (bool -> and, int -> combined rank)."""
    if len(masks) == 2:
        return (masks[0][None, :] & masks[1][:, None]).reshape(-1)
    return (masks[0][None, None, :] & masks[1][None, :, None] & masks[2][:, None, None]).reshape(-1)


class _LevelGeom:
    """Geometry of one level on this rank: global node counts, extended box, ownership."""

    def __init__(self, glob_nodes, origin, ext_nodes, own1d, box, numDomains):
        self.glob_nodes = [int(v) for v in glob_nodes]
        self.origin = [int(v) for v in origin]
        self.ext_nodes = [int(v) for v in ext_nodes]
        self.own1d = own1d                      # per dim: owner box index of every GLOBAL coordinate
        dim = len(glob_nodes)
        self.nd = [int(v) for v in numDomains]
        self.gid = _ids(self.origin, self.ext_nodes, self.glob_nodes)            # ext-local -> global id
        masks = [own1d[k][self.origin[k]: self.origin[k] + self.ext_nodes[k]] == box[k] for k in range(dim)]
        self.own_mask = _outer(masks)
        self.own_loc = np.nonzero(self.own_mask)[0]                              # ext-local ids of owned nodes
        self.own_gid = self.gid[self.own_loc]                                    # ascending
        self.own_pos = np.full(self.gid.size, -1, dtype=np.int64)
        self.own_pos[self.own_loc] = np.arange(self.own_loc.size)
        # owner rank of every ext node (loc2cs numbering: x fastest)
        o1 = [own1d[k][self.origin[k]: self.origin[k] + self.ext_nodes[k]].astype(np.int64) for k in range(dim)]
        if dim == 2:
            self.owner_ext = (o1[0][None, :] + self.nd[0] * o1[1][:, None]).reshape(-1)
        else:
            self.owner_ext = (o1[0][None, None, :] + self.nd[0] * (o1[1][None, :, None] + self.nd[1] * o1[2][:, None, None])).reshape(-1)

    def owner_of_all(self):
        """Owner rank of every GLOBAL node of this level (only used for the small replicated level)."""
        o = [self.own1d[k].astype(np.int64) for k in range(len(self.nd))]
        if len(o) == 2:
            return (o[0][None, :] + self.nd[0] * o[1][:, None]).reshape(-1)
        return (o[0][None, None, :] + self.nd[0] * (o[1][None, :, None] + self.nd[1] * o[2][:, None, None])).reshape(-1)


def _localize_ext(M_rows, src: _LevelGeom, rank: int, nranks: int):
    """Rows (already cut to the owned rows) with ext-local column ids of the SOURCE level -> local numbering
    [owned | halo by (owner, global id)], the halo's global ids per owner (= what to request) and recv splits."""
    M_rows = sp.csr_matrix(M_rows)
    cols = M_rows.indices
    pos = src.own_pos[cols]
    mine = pos >= 0
    hcols = np.unique(cols[~mine])
    hgid = src.gid[hcols]
    hown = src.owner_ext[hcols]
    if np.any(hown == rank):
        raise RuntimeError("structured setup: margin too small (an owned column lies outside the owned box)")
    order = np.lexsort((hgid, hown))
    hcols, hgid, hown = hcols[order], hgid[order], hown[order]
    n_own = int(src.own_loc.size)
    lut = np.full(src.gid.size, -1, dtype=np.int64)
    lut[hcols] = n_own + np.arange(hcols.size)
    newcol = np.where(mine, pos, lut[cols])
    Mloc = sp.csr_matrix((M_rows.data, newcol, M_rows.indptr), shape=(M_rows.shape[0], n_own + hcols.size))
    Mloc.sort_indices()
    requests = {int(q): hgid[hown == q] for q in np.unique(hown)}
    recv_splits = np.bincount(hown, minlength=nranks).astype(int).tolist()
    return Mloc, requests, recv_splits, int(hcols.size)


def _make_plan(requests_all, src: _LevelGeom, rank: int, nranks: int, recv_splits, n_halo):
    """Send side of a halo plan from everybody's request lists (gathered once at setup)."""
    send_idx, send_splits = [], []
    for q in range(nranks):
        want = requests_all[q].get(rank) if q != rank else None
        if want is None or len(want) == 0:
            send_splits.append(0)
            continue
        p = np.searchsorted(src.own_gid, want)
        if np.any(p >= src.own_gid.size) or np.any(src.own_gid[np.minimum(p, src.own_gid.size - 1)] != want):
            raise RuntimeError("structured setup: a peer requested a node this rank does not own")
        send_idx.append(p)
        send_splits.append(int(p.size))
    send_idx = np.concatenate(send_idx) if send_idx else np.zeros(0, dtype=np.int64)
    return HaloPlan(int(src.own_loc.size), int(n_halo), send_idx.astype(np.int64), send_splits, list(recv_splits))


def setup_on_margin_box(global_cells, numDomains, rank, size, param: MGparam, operator, domain=None,
                        replicate_below: int = 300_000):
    """The reference's setup steps (``getFWInterp``, ``RT = P*0.5^dim``, ``Ps*AT*Rs``, ``getRelaxPrec``;
    MGsetup.jl:54-60,76,102) on this rank's box EXTENDED by a margin of 3*2^a nodes (a = number of sharded levels),
    aligned to 2^a so that all a coarsenings stay nested.  Rows in the outermost layer next to an artificial cut are
    wrong on every level; every other row equals the global hierarchy's row.  Shared by the two sharded layouts
    (``structured_gmg``: owned rows + halo columns; ``ghost_dist.ghost_gmg``: extended boxes with ghost layers)."""
    cells = np.asarray(global_cells, dtype=np.int64)
    nd = np.asarray(numDomains, dtype=np.int64)
    dim = cells.size
    if int(np.prod(nd)) != size:
        raise ValueError("numDomains does not match the number of ranks")
    nl = int(param.levels)
    if np.any(cells % (1 << (nl - 1))):
        raise ValueError("structured setup needs cells divisible by 2^(levels-1) (odd node counts on every level)")
    if domain is None:
        domain = [0.0, 1.0] * dim
    domain = np.asarray(domain, dtype=np.float64)
    h = (domain[1::2] - domain[0::2]) / cells
    # sharded levels: the finest always, then while the level is large
    nglob = [int(np.prod((cells >> l) + 1)) for l in range(nl)]
    a = 1
    while a < nl - 1 and nglob[a] > replicate_below:
        a += 1
    # box coordinates of this rank (loc2cs: x fastest)
    box = [int(rank % nd[0]), int((rank // nd[0]) % nd[1])] + ([int(rank // (nd[0] * nd[1]))] if dim == 3 else [])
    own1d = [[np.minimum(np.arange(cells[k] + 1) // max(int(cells[k] // nd[k]), 1), nd[k] - 1) for k in range(dim)]]
    for l in range(1, a + 1):
        own1d.append([o[::2] for o in own1d[-1]])
    # extended box on the fine level: margin 3*2^a nodes, aligned to 2^a so that all a coarsenings stay nested
    m1, al = 3 * (1 << a), 1 << a
    lo_e, hi_e = [], []
    for k in range(dim):
        idx = np.nonzero(own1d[0][k] == box[k])[0]
        lo, hi = int(idx[0]), int(idx[-1])
        lo_e.append(max(0, (lo - m1) // al * al))
        hi_e.append(min(int(cells[k]), -((-(hi + m1)) // al) * al))
    cells_loc = np.array([hi_e[k] - lo_e[k] for k in range(dim)], dtype=np.int64)
    dom_loc = np.ravel([[domain[2 * k] + lo_e[k] * h[k], domain[2 * k] + hi_e[k] * h[k]] for k in range(dim)])
    mesh_loc = getRegularMesh(dom_loc, cells_loc)
    # ---- the reference's setup steps on the extended box (MGsetup.jl:54-60,76,102) ----------------------
    As = [sp.csr_matrix(operator(mesh_loc))]
    As[0].sort_indices()
    Ps, Rs, ds = [], [], []
    n = cells_loc.copy()
    for l in range(a):
        P, nc_nodes = getFWInterp(n + 1, False)
        R = (P.T * (0.5 ** dim)).tocsr()
        R.sort_indices()
        ds.append(getRelaxPrec(As[l], param.relaxType, param.relaxParam if not isinstance(param.relaxParam, (list, tuple, np.ndarray)) else param.relaxParam[l]))
        Ps.append(P)
        Rs.append(R)
        As.append(galerkin(R, As[l], P))
        n = nc_nodes - 1
    geoms = []
    for l in range(a + 1):
        geoms.append(_LevelGeom((cells >> l) + 1, [v >> l for v in lo_e], (cells_loc >> l) + 1, own1d[l], box, nd))
    return dict(cells=cells, nd=nd, dim=dim, nl=nl, a=a, domain=domain, h=h, nglob=nglob, box=box, own1d=own1d, lo_e=lo_e,
                hi_e=hi_e, cells_loc=cells_loc, As=As, Ps=Ps, Rs=Rs, ds=ds, geoms=geoms)


def structured_gmg(global_cells, numDomains, comm, backend, param: MGparam, operator, domain=None, nrhs: int = 1,
                   replicate_below: int = 300_000, gather_objects=None):
    """Build this rank's ``DistributedHierarchy`` of a FullWeighting/Galerkin GMG hierarchy without any
    global matrix.

    ``operator(mesh_loc) -> csr`` generates the fine operator on a sub-mesh (rows next to an artificial
    cut may be anything: they are discarded).  ``param`` carries levels / smoother / cycle settings as for
    ``MGsetup``.  ``gather_objects(obj) -> list`` defaults to ``torch.distributed.all_gather_object``.
    Returns (hierarchy, info) where info holds the level geometry (for building right-hand sides).
    """
    DistributedHierarchy.check_supported(param)
    S = setup_on_margin_box(global_cells, numDomains, comm.rank, comm.size, param, operator, domain, replicate_below)
    cells, nd, dim, rank, size, nl, a = S["cells"], S["nd"], S["dim"], comm.rank, comm.size, S["nl"], S["a"]
    domain, h, nglob, box, own1d = S["domain"], S["h"], S["nglob"], S["box"], S["own1d"]
    cells_loc, As, Ps, Rs, ds, geoms = S["cells_loc"], S["As"], S["Ps"], S["Rs"], S["ds"], S["geoms"]
    # ---- cut out the owned rows, renumber columns, collect halo requests ---------------------------------
    pieces, requests = [], {}
    for l in range(a):
        g, gc = geoms[l], geoms[l + 1]
        A_rows = As[l][g.own_loc, :]
        R_rows = Rs[l][gc.own_loc, :]
        P_rows = Ps[l][g.own_loc, :]
        A_loc, requests[("A", l)], rsA, nhA = _localize_ext(A_rows, g, rank, size)
        R_loc, requests[("R", l)], rsR, nhR = _localize_ext(R_rows, g, rank, size)
        if l + 1 < a:
            P_loc, requests[("P", l)], rsP, nhP = _localize_ext(P_rows, gc, rank, size)
        else:   # the prolongation from the replicated level reads the full vector: global column ids
            Pr = sp.csr_matrix(P_rows)
            P_loc = sp.csr_matrix((Pr.data, gc.gid[Pr.indices], Pr.indptr), shape=(Pr.shape[0], nglob[a]))
            P_loc.sort_indices()
            rsP, nhP = None, 0
        pieces.append((A_loc, rsA, nhA, R_loc, rsR, nhR, P_loc, rsP, nhP))
    # this rank's rows of the first replicated level, global column ids
    gt = geoms[a]
    T_rows = sp.csr_matrix(As[a][gt.own_loc, :])
    tail_piece = (gt.own_gid, T_rows.indptr, gt.gid[T_rows.indices], T_rows.data)
    if gather_objects is None:
        import torch.distributed as dist

        def gather_objects(obj):
            out = [None] * size
            dist.all_gather_object(out, obj)
            return out
    gathered = gather_objects((requests, tail_piece)) if size > 1 else [(requests, tail_piece)]
    req_all = [g_[0] for g_ in gathered]
    # ---- halo plans -------------------------------------------------------------------------------------
    local_levels = []
    for l in range(a):
        g, gc = geoms[l], geoms[l + 1]
        A_loc, rsA, nhA, R_loc, rsR, nhR, P_loc, rsP, nhP = pieces[l]
        planA = _make_plan([r_.get(("A", l), {}) for r_ in req_all], g, rank, size, rsA, nhA)
        planR = _make_plan([r_.get(("R", l), {}) for r_ in req_all], g, rank, size, rsR, nhR)
        planP = _make_plan([r_.get(("P", l), {}) for r_ in req_all], gc, rank, size, rsP, nhP) if rsP is not None else None
        from .distributed import _box_of_rows
        if l + 1 < a:       # columns of P: [this rank's box of the next sharded level | halo]
            cbox, cbox_cols = _box_of_rows(gc.own_gid, gc.glob_nodes), int(gc.own_loc.size)
        else:               # the replicated tail: the whole coarse grid
            cbox, cbox_cols = tuple(int(v) for v in gc.glob_nodes), int(P_loc.shape[1])
            if int(np.prod(cbox)) != cbox_cols:
                cbox = None
        local_levels.append(dict(n_own=int(g.own_loc.size), A=A_loc, planA=planA, R=R_loc, planR=planR, P=P_loc,
                                 planP=planP, d=np.asarray(ds[l])[g.own_loc], npre=param.relaxPre(l + 1),
                                 npost=param.relaxPost(l + 1), box=_box_of_rows(g.own_gid, g.glob_nodes),
                                 cbox=cbox, cbox_cols=cbox_cols))
    # ---- the replicated tail: assemble its finest operator from everybody's rows, then plain MGsetup -----
    nt = nglob[a]
    rows_i, cols_i, vals_i = [], [], []
    for _, (gids, indptr, gcols, data) in gathered:
        rows_i.append(np.repeat(gids, np.diff(indptr)))
        cols_i.append(gcols)
        vals_i.append(data)
    A_tail = sp.csr_matrix((np.concatenate(vals_i), (np.concatenate(rows_i), np.concatenate(cols_i))), shape=(nt, nt))
    A_tail.sort_indices()
    p_tail = getMGparam(np.float64, np.int64, nl - a, param.numCores, param.maxOuterIter, param.relativeTol,
                        param.relaxType, param.relaxParam, lambda level, _s=a: param.relaxPre(level + _s),
                        lambda level, _s=a: param.relaxPost(level + _s), param.cycleType, param.coarseSolveType,
                        param.strongConnParam, param.FilteringParam, param.transferOperatorType)
    MGsetup(A_tail, getRegularMesh(domain, cells >> a), p_tail, nrhs)
    tail_part = Partition(gt.owner_of_all().astype(np.int32), size)
    H = DistributedHierarchy(comm, backend, local_levels, p_tail, tail_part.owner, tail_part.local_index,
                             tail_part.counts, geoms[0].own_gid, param.cycleType, nl, nrhs)
    info = dict(geoms=geoms, A_ext=As[0], sharded_levels=a, ext_cells=cells_loc, h=h)
    return H, info


def poisson_operator(global_cells, domain=None):
    """Local generator of ``poisson_shifted`` (G'G + 1e-4*opnorm(G'G,1)*I, testGMGRAPforPoisson.jl:59-64):
    the shift uses the GLOBAL operator's 1-norm, sum_k 4/h_k^2 (an interior column: |-1|+|2|+|-1| per dimension)."""
    from .operators import getNodalLaplacianMatrix
    cells = np.asarray(global_cells, dtype=np.float64)
    dim = cells.size
    dom = np.asarray(domain if domain is not None else [0.0, 1.0] * dim, dtype=np.float64)
    h = (dom[1::2] - dom[0::2]) / cells
    shift = 1e-4 * float(np.sum(4.0 / h ** 2))

    def op(mesh_loc):
        L = getNodalLaplacianMatrix(mesh_loc)
        return (L + shift * sp.identity(L.shape[0], format="csr")).tocsr()

    return op


def local_rhs(info, nrhs: int = 1, seed: int = 1234):
    """This rank's rows of b = A*u, u ~ U[0,1) seeded over the GLOBAL node set (same u as ``seeded_rhs``);
    returns (b_own unnormalised, sum of squares of b_own): divide by the all-reduced norm."""
    g = info["geoms"][0]
    n = int(np.prod(g.glob_nodes))
    rng = np.random.default_rng(seed)
    u = rng.random((n, nrhs)) if nrhs > 1 else rng.random(n)
    b = info["A_ext"][g.own_loc, :] @ u[g.gid]
    return np.ascontiguousarray(b), float(np.sum(b * b))
