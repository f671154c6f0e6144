"""Sharded multigrid cycle with DEEP GHOST LAYERS: every rank runs the single-GPU cycle on its box EXTENDED by ghost layers.

The reference's DomainDecomposition boxes carry an ``overlap`` (``getBoxWithOverlap``, src/DomainDecomposition/DDIndices.jl:61-92;
box rule l.41-47; fan-out DDParallel.jl:87-105,133-139).  Here the overlap is what makes the sharded cycle communication-avoiding:

* on every sharded level a rank holds an EXTENDED box = its owned box + g ghost layers on the sides it shares with a
  neighbour (none on a side that is a boundary of the global grid); the extended boxes of consecutive levels are nested
  (fine = 2 * coarse - 1 nodes per dimension), so the local operators A, P, R of a level are ordinary grid operators
  and the rank's part of the hierarchy is an ordinary single-GPU GMG hierarchy: ``libmgvcycle.so`` runs ALL its
  single-GPU kernels on it - the four-stage pass of the solve loop, the 27-point marching form, the marching
  restriction, the staged prolongation, the pipelined stopping test;
* a product with A costs one ghost layer of validity: a vector that is valid on the owned box + v layers gives a result
  valid on v - 1 layers (the outermost layer of an extended box holds rows cut off at the artificial boundary: they
  compute bounded garbage nobody reads with a non-zero weight).  The library keeps the validity depth of every level
  buffer and refreshes ALL ghost layers of a vector in ONE exchange where the next operation needs more than is left
  (``mg_ghost_*``, csrc/mg_dist.inc): per V(2,1) step one exchange on the fine level - started right behind the
  four-stage pass and overlapped with the restriction and the whole coarse cycle - and two small ones per coarser
  sharded level (the restricted right-hand side on the way down, the correction on the way up), where the halo form
  (``distributed.py``) needs five to six per level;
* the first replicated level is reached through a restriction whose rows are the coarse nodes this rank owns, followed
  by one all-reduce (every entry has exactly one non-zero contribution: the sum is exact); from there down every rank
  runs the same replicated levels, graphs included, inside the same hierarchy handle;
* norms count the OWNED rows only (box mask in the kernels that fuse ``||r||^2``) and are all-reduced.

Ghost widths: 3 layers on the last sharded level (x1 = d.*b valid on 3, the sweep on 2, the residual on 1, the
restriction of the owned coarse rows needs 1), doubled towards the fine level by the nesting (fine: 11-12 layers for
three sharded levels, 5-6 for two), which is also what the fine level's four-stage pass (4 stages + 1 layer for the
restriction behind it) asks for.  The redundant work is the price of one exchange per pass.

This module is the HOST side: geometry, local hierarchy, exchange plans (pure numpy / scipy - checked on CPU with gloo
by tests/test_ghost_dist.py) and the ctypes binding of the device side.
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import scipy.sparse as sp

from .mgdef import MGparam, getMGparam
from .mgsetup import MGsetup
from .operators import getRegularMesh

G_LAST = 3          # ghost layers of the coarser sharded levels (see the module docstring)
G_FINE = 5          # ghost layers of the fine level: the four-stage pass (4 products) + the layer the restriction behind it reads
FULL = 1 << 20      # validity depth of a vector whose every ghost layer is up to date


def default_box_of(rank: int, nd) -> List[int]:
    """Box coordinates of a rank (``loc2cs``, DDService.jl:27-36: x fastest)."""
    nd = [int(v) for v in nd]
    b = [rank % nd[0], (rank // nd[0]) % nd[1]]
    if len(nd) == 3:
        b.append(rank // (nd[0] * nd[1]))
    return b


def _own_ranges(cells, nd, box, level):
    """Inclusive global node range [lo, hi] per dimension of the box's owned nodes on `level` (0 = fine).
    Box rule DDIndices.jl:41-47 (cellSize = div(nc, NumCells), the last box takes the rest); interface nodes belong to
    the upper box; a coarse node belongs to the owner of the coincident fine node."""
    out = []
    for k in range(len(cells)):
        nc, d = int(cells[k]), int(nd[k])
        cs = max(nc // d, 1)
        lo = box[k] * cs
        hi = (box[k] + 1) * cs - 1 if box[k] < d - 1 else nc
        s = 1 << level
        out.append((-(-lo // s), hi // s))
    return out


def ghost_boxes(cells, nd, box, a: int, g_last: int = G_LAST, g_fine: int = G_FINE, nested: bool = False):
    """Owned and extended boxes of one rank on the sharded levels 0..a-1.
    Returns a list (per level) of dicts: own (inclusive global ranges), ext (inclusive global ranges), gmin.

    Every level gets the ghost width ITS OWN passes consume (round 6): ``g_fine`` layers on the fine level (the four-stage pass
    of the solve loop: four products + the layer the restriction behind it reads), ``g_last`` on every coarser sharded level
    (x1 = d.*b, sweep, residual, restriction).  The extended box of a level ends on nodes of the next level (even coordinates)
    and the next level's box holds at least the parents of all its nodes: the local P and R are then the grid transfer
    operators between the fine box and a SUB-BOX of the coarse box (``grid_cell_prolong`` / ``grid_wave_restrict`` take the
    sub-box's offset), the coarse rows outside that sub-box are ghost rows every exchange overwrites.
    ``nested=True``: round 5's rule - ``g_last`` layers on the last sharded level, every finer box = 2 * coarse - 1 nodes
    (ghost widths double per level: 11-12 fine layers for three sharded levels)."""
    dim = len(cells)
    lv = [dict(own=_own_ranges(cells, nd, box, l)) for l in range(a)]
    if nested:
        last = lv[a - 1]
        n_last = [(int(cells[k]) >> (a - 1)) + 1 for k in range(dim)]
        last["ext"] = [(max(0, last["own"][k][0] - g_last), min(n_last[k] - 1, last["own"][k][1] + g_last)) for k in range(dim)]
        for l in range(a - 2, -1, -1):
            lv[l]["ext"] = [(2 * e[0], 2 * e[1]) for e in lv[l + 1]["ext"]]
    else:
        prev = None
        for l in range(a):
            n_l = [(int(cells[k]) >> l) + 1 for k in range(dim)]
            want = g_fine if l == 0 else g_last
            ext = []
            for k in range(dim):
                olo, ohi = lv[l]["own"][k]
                lo = max(0, olo - want) if olo > 0 else 0
                hi = min(n_l[k] - 1, ohi + want) if ohi < n_l[k] - 1 else n_l[k] - 1
                if prev is not None:                       # the parents of every node of the finer extended box
                    lo, hi = min(lo, prev[k][0] // 2), max(hi, -(-prev[k][1] // 2))
                if l + 1 < a:                              # the box ends on nodes of the next level
                    lo, hi = lo - (lo & 1), min(n_l[k] - 1, hi + (hi & 1))
                    if hi & 1:
                        raise RuntimeError("ghost boxes: a sharded level with an even node count")
                ext.append((int(lo), int(hi)))
            lv[l]["ext"] = ext
            prev = ext
    for l in range(a):
        n_l = [(int(cells[k]) >> l) + 1 for k in range(dim)]
        g = FULL
        for k in range(dim):
            (olo, ohi), (elo, ehi) = lv[l]["own"][k], lv[l]["ext"][k]
            if not (elo <= olo and ohi <= ehi):
                raise RuntimeError("ghost box does not contain the owned box")
            if olo > 0:
                g = min(g, olo - elo)
            if ohi < n_l[k] - 1:
                g = min(g, ehi - ohi)
        lv[l]["gmin"] = int(g)
        lv[l]["nodes"] = n_l
    return lv


def _box_ids(lo, n, origin, n_host):
    """x-fastest ids, inside a host box (origin, n_host nodes per dim), of the sub-box starting at global `lo` with `n` nodes."""
    dim = len(n)
    ax = [np.arange(n[k], dtype=np.int64) + (int(lo[k]) - int(origin[k])) for k in range(dim)]
    for k in range(dim):
        if ax[k][0] < 0 or ax[k][-1] >= n_host[k]:
            raise RuntimeError("sub-box leaves its host box")
    if dim == 2:
        g = ax[0][None, :] + n_host[0] * ax[1][:, None]
    else:
        g = ax[0][None, None, :] + n_host[0] * (ax[1][None, :, None] + n_host[1] * ax[2][:, None, None])
    return g.reshape(-1)


def _submatrix(M, row_ids, col_lut, ncols):
    """M[row_ids, :] with the columns renumbered through col_lut (-1: dropped), stored order kept."""
    R = sp.csr_matrix(M)[row_ids, :].tocsr()
    R.sort_indices()
    newcol = col_lut[R.indices]
    keep = newcol >= 0
    csum = np.concatenate([[0], np.cumsum(keep)])
    indptr = csum[R.indptr]
    out = sp.csr_matrix((R.data[keep], newcol[keep], indptr), shape=(len(row_ids), ncols))
    out.sort_indices()
    return out


class _Mesh:
    def __init__(self, n):
        self.n = np.asarray(n, dtype=np.int64)


class GhostLevel:
    """One sharded level of this rank: extended box, owned box inside it, exchange plan."""

    def __init__(self):
        self.ext_lo: List[int] = []       # global coordinates of the extended box's first node
        self.ext_n: List[int] = []        # nodes per dimension of the extended box
        self.own_lo: List[int] = []       # owned box in extended-box coordinates [lo, hi)
        self.own_hi: List[int] = []
        self.gmin = FULL
        self.send_idx = np.zeros(0, dtype=np.int64)     # extended-box ids of owned nodes, grouped by peer
        self.recv_idx = np.zeros(0, dtype=np.int64)     # extended-box ids of ghost nodes, grouped by owner
        self.send_splits: List[int] = []
        self.recv_splits: List[int] = []

    @property
    def n(self):
        return int(np.prod(self.ext_n))

    def own_mask(self):
        dim = len(self.ext_n)
        m = [(np.arange(self.ext_n[k]) >= self.own_lo[k]) & (np.arange(self.ext_n[k]) < self.own_hi[k]) for k in range(dim)]
        if dim == 2:
            return (m[0][None, :] & m[1][:, None]).reshape(-1)
        return (m[0][None, None, :] & m[1][None, :, None] & m[2][:, None, None]).reshape(-1)

    def depth_mask(self, depth):
        """Nodes of the extended box within `depth` layers of the owned box (what a vector of that validity depth holds)."""
        dim = len(self.ext_n)
        if depth >= FULL:
            return np.ones(self.n, dtype=bool)
        m = [(np.arange(self.ext_n[k]) >= self.own_lo[k] - depth) & (np.arange(self.ext_n[k]) < self.own_hi[k] + depth) for k in range(dim)]
        if dim == 2:
            return (m[0][None, :] & m[1][:, None]).reshape(-1)
        return (m[0][None, None, :] & m[1][None, :, None] & m[2][:, None, None]).reshape(-1)


class GhostSetup:
    """Everything a rank needs for the ghost-layer form: the local hierarchy (an ordinary MGparam), the sharded levels'
    geometry and exchange plans."""

    def __init__(self):
        self.param: Optional[MGparam] = None
        self.levels: List[GhostLevel] = []
        self.a = 0
        self.rank = 0
        self.size = 1
        self.gid_fine = None            # global id of every node of the fine extended box
        self.n_tail = 0
        self.info = {}


def ghost_gmg(global_cells, numDomains, rank: int, size: int, param: MGparam, operator, domain=None, nrhs: int = 1,
              replicate_below: int = 300_000, gather_objects=None, g_last: int = G_LAST, dry_tail: bool = False,
              g_fine: int = G_FINE, nested: bool = False) -> GhostSetup:
    """Build this rank's part of a FullWeighting / Galerkin GMG hierarchy in the ghost-layer form (no global matrix).

    ``operator(mesh_loc) -> csr`` generates the fine operator on a sub-mesh (rows next to an artificial cut may be
    anything).  ``gather_objects(obj) -> list`` (one entry per rank) defaults to ``torch.distributed.all_gather_object``;
    it is called once, for the rows of the first replicated level."""
    from .structured_setup import setup_on_margin_box
    if param.relaxType not in ("Jac", "SPAI", "Jac-GMRES") or param.cycleType not in ("V", "W", "F", "K"):
        raise NotImplementedError(f"the ghost-layer form does not know relaxType {param.relaxType!r} / cycleType {param.cycleType!r}")
    S = setup_on_margin_box(global_cells, numDomains, rank, size, param, operator, domain, replicate_below)
    cells, nd, dim, nl, a = S["cells"], S["nd"], S["dim"], S["nl"], S["a"]
    As, Ps, Rs, ds, geoms = S["As"], S["Ps"], S["Rs"], S["ds"], S["geoms"]
    box = S["box"]
    boxes = ghost_boxes(cells, nd, box, a, g_last, g_fine, nested)
    G = GhostSetup()
    G.a, G.rank, G.size = a, rank, size
    # ---- sub-boxes of the setup boxes -------------------------------------------------------------------------------
    ids, luts = [], []
    for l in range(a):
        g = geoms[l]
        lo = [e[0] for e in boxes[l]["ext"]]
        n = [e[1] - e[0] + 1 for e in boxes[l]["ext"]]
        i = _box_ids(lo, n, g.origin, g.ext_nodes)
        lut = np.full(g.gid.size, -1, dtype=np.int64)
        lut[i] = np.arange(i.size)
        ids.append(i)
        luts.append(lut)
        L = GhostLevel()
        L.ext_lo, L.ext_n = lo, n
        L.own_lo = [boxes[l]["own"][k][0] - lo[k] for k in range(dim)]
        L.own_hi = [boxes[l]["own"][k][1] - lo[k] + 1 for k in range(dim)]
        L.gmin = boxes[l]["gmin"]
        G.levels.append(L)
    G.gid_fine = geoms[0].gid[ids[0]]
    A_loc, P_loc, R_loc, d_loc = [], [], [], []
    nt = S["nglob"][a]
    gt = geoms[a]
    for l in range(a):
        A_loc.append(_submatrix(As[l], ids[l], luts[l], ids[l].size))
        d_loc.append(np.asarray(ds[l])[ids[l]])
        if l + 1 < a:
            # the coarse box holds the parents of every fine node (ghost_boxes): P loses no entry; R's rows outside the sub-box
            # of those parents are empty (the fine box ends on coarse nodes)
            P_loc.append(_submatrix(Ps[l], ids[l], luts[l + 1], ids[l + 1].size))
            if P_loc[-1].nnz != sp.csr_matrix(Ps[l])[ids[l], :].nnz:
                raise RuntimeError("ghost boxes: a fine node interpolates from a coarse node outside the coarse box")
            R_loc.append(_submatrix(Rs[l], ids[l + 1], luts[l], ids[l].size))
        else:
            # into / out of the first replicated level: P reads the replicated vector (global column ids); R has one row per
            # node of the replicated level, non-empty for the nodes this rank owns (the all-reduce adds the ranks' parts)
            P_loc.append(_submatrix(Ps[l], ids[l], gt.gid, nt))
            Rown = _submatrix(Rs[l], gt.own_loc, luts[l], ids[l].size)
            Rt = sp.csr_matrix((nt, ids[l].size))
            indptr = np.zeros(nt + 1, dtype=np.int64)
            indptr[gt.own_gid + 1] = np.diff(Rown.indptr)
            indptr = np.cumsum(indptr)
            Rt = sp.csr_matrix((Rown.data, Rown.indices, indptr), shape=(nt, ids[l].size))   # (own_gid ascending: row order kept)
            R_loc.append(Rt)
    # ---- the replicated levels: assemble the first one from everybody's rows, plain MGsetup below ---------------------
    T_rows = sp.csr_matrix(As[a][gt.own_loc, :])
    tail_piece = (gt.own_gid, T_rows.indptr, gt.gid[T_rows.indices], T_rows.data)
    if dry_tail:
        gathered = None
    elif size > 1:
        if gather_objects is None:
            import torch.distributed as dist

            def gather_objects(obj):
                out = [None] * size
                dist.all_gather_object(out, obj)
                return out
        gathered = gather_objects(tail_piece)
    else:
        gathered = [tail_piece]
    if dry_tail:
        # TIMING AID (one rank of a larger world alone on its GPU, NativeGhostHierarchy(transport="dry")): the other ranks' rows of
        # the first replicated level do not exist here; a hierarchy of the same sizes and stencils stands in - the operator
        # re-discretised one level above and coarsened once (27-point Galerkin levels, as the real tail has)
        from .distributed import _sub_hierarchy
        mesh_up = getRegularMesh(S["domain"], cells >> (a - 1))
        p_up = getMGparam(np.float64, np.int64, nl - a + 1, param.numCores, param.maxOuterIter, param.relativeTol,
                          param.relaxType, param.relaxParam, lambda level, _s=a - 1: param.relaxPre(level + _s),
                          lambda level, _s=a - 1: param.relaxPost(level + _s), param.cycleType, param.coarseSolveType,
                          param.strongConnParam, param.FilteringParam, param.transferOperatorType)
        MGsetup(sp.csr_matrix(operator(mesh_up)), mesh_up, p_up, nrhs)
        p_tail = _sub_hierarchy(p_up, 1)
    else:
        rows_i, cols_i, vals_i = [], [], []
        for gids, indptr, gcols, data in gathered:
            rows_i.append(np.repeat(gids, np.diff(indptr)))
            cols_i.append(gcols)
            vals_i.append(data)
        A_tail = sp.csr_matrix((np.concatenate(vals_i), (np.concatenate(rows_i), np.concatenate(cols_i))), shape=(nt, nt))
        A_tail.sort_indices()
        p_tail = getMGparam(np.float64, np.int64, nl - a, param.numCores, param.maxOuterIter, param.relativeTol,
                            param.relaxType, param.relaxParam, lambda level, _s=a: param.relaxPre(level + _s),
                            lambda level, _s=a: param.relaxPost(level + _s), param.cycleType, param.coarseSolveType,
                            param.strongConnParam, param.FilteringParam, param.transferOperatorType)
        MGsetup(A_tail, getRegularMesh(S["domain"], cells >> a), p_tail, nrhs)
    # ---- the local hierarchy as one MGparam (levels 0..a-1 on extended boxes, a.. replicated) ---------------------------
    p = getMGparam(np.float64, np.int64, a + len(p_tail.As), param.numCores, param.maxOuterIter, param.relativeTol,
                   param.relaxType, param.relaxParam, param.relaxPre, param.relaxPost, param.cycleType,
                   param.coarseSolveType, param.strongConnParam, param.FilteringParam, param.transferOperatorType)
    p.As = A_loc + list(p_tail.As)
    p.Ps = P_loc + list(p_tail.Ps)
    p.Rs = R_loc + list(p_tail.Rs)
    p.relaxPrecs = d_loc + list(p_tail.relaxPrecs)
    p.LU = p_tail.LU
    p.levels = len(p.As)
    p.nrhs = nrhs
    p.Meshes = [_Mesh(np.asarray(L.ext_n) - 1) for L in G.levels] + [_Mesh((cells >> (a + j))) for j in range(len(p_tail.As))]
    G.param = p
    G.n_tail = nt
    # ---- exchange plans: every rank can compute every other rank's boxes -----------------------------------------------
    all_boxes = [ghost_boxes(cells, nd, default_box_of(q, nd), a, g_last, g_fine, nested) for q in range(size)]
    for l in range(a):
        L = G.levels[l]
        me = boxes[l]
        # every rank must take the same exchange decisions: the validity bookkeeping counts with the smallest ghost width
        # over the cut sides of ALL ranks
        L.gmin = int(min(bq[l]["gmin"] for bq in all_boxes))
        send, recv, ss, rs = [], [], [], []
        for q in range(size):
            if q == rank:
                ss.append(0)
                rs.append(0)
                continue
            other = all_boxes[q][l]
            s_ids = _intersect_ids(me["own"], other["ext"], L.ext_lo, L.ext_n)     # what q's ghost layers hold of mine
            r_ids = _intersect_ids(other["own"], me["ext"], L.ext_lo, L.ext_n)     # what my ghost layers hold of q's
            send.append(s_ids)
            recv.append(r_ids)
            ss.append(int(s_ids.size))
            rs.append(int(r_ids.size))
        L.send_idx = np.concatenate(send) if send else np.zeros(0, dtype=np.int64)
        L.recv_idx = np.concatenate(recv) if recv else np.zeros(0, dtype=np.int64)
        L.send_splits, L.recv_splits = ss, rs
        n_ghost = L.n - int(np.prod([L.own_hi[k] - L.own_lo[k] for k in range(dim)]))
        if L.recv_idx.size != n_ghost or np.unique(L.recv_idx).size != n_ghost:
            raise RuntimeError("ghost plan: the peers' owned boxes do not tile this rank's ghost layers")
    G.info = dict(geoms=geoms, A_setup=As[0], ids_fine=ids[0], sharded_levels=a, h=S["h"], boxes=boxes)
    return G


def _intersect_ids(box_a, box_b, ext_lo, ext_n):
    """Extended-box ids (x fastest = ascending global id inside the intersection) of the nodes in both inclusive boxes."""
    dim = len(ext_n)
    lo = [max(box_a[k][0], box_b[k][0]) for k in range(dim)]
    hi = [min(box_a[k][1], box_b[k][1]) for k in range(dim)]
    if any(hi[k] < lo[k] for k in range(dim)):
        return np.zeros(0, dtype=np.int64)
    n = [hi[k] - lo[k] + 1 for k in range(dim)]
    return _box_ids(lo, n, ext_lo, ext_n)


def local_rhs(G: GhostSetup, nrhs: int = 1, seed: int = 1234):
    """b = A*u on this rank's extended fine box (u ~ U[0,1) seeded over the GLOBAL node set, as ``seeded_rhs``), valid on
    every row but the outermost cut layer, and the sum of squares over the OWNED rows: divide by the all-reduced norm."""
    g = G.info["geoms"][0]
    n = int(np.prod(g.glob_nodes))
    rng = np.random.default_rng(seed)
    u = rng.random((n, nrhs)) if nrhs > 1 else rng.random(n)
    b = G.info["A_setup"][G.info["ids_fine"], :] @ u[g.gid]
    own = G.levels[0].own_mask()
    return np.ascontiguousarray(b), float(np.sum(b[own] ** 2))


# ======================================================================================================
# device side: the local hierarchy is an ordinary DeviceHierarchy; mg_ghost_* attaches geometry, plans and transport
# ======================================================================================================
class NativeGhostHierarchy:
    """This rank's hierarchy on the GPU + the ghost-layer exchange behind the C ABI (``mg_ghost_*``).

    transport="rccl": the library's own RCCL communicator (unique id from rank 0, broadcast with ``torch.distributed``);
    transport="plugin": every exchange goes through ``torch.distributed`` on host buffers (tests, ranks sharing one GPU);
    transport="dry": timing aid - one rank of a larger world alone on its GPU, nothing travels (``ghost_gmg(dry_tail=True)``);
    a world of one rank needs none of them."""

    def __init__(self, G: GhostSetup, device_id: int = 0, transport: str = "rccl", group=None, options=None, collectives=None):
        import ctypes as C
        from . import device as D
        self.G = G
        self.D = D
        self.group = group
        # plug-in transport only: an object with all_to_all(send, send_splits, recv_splits) -> recv and all_reduce(values) -> sums on
        # numpy arrays (default: torch.distributed on `group`); e.g. ranks that are THREADS of one process (tests/test_ghost_dist.py)
        self.collectives = collectives
        self.nrhs = int(getattr(G.param, "nrhs", 1) or 1)      # (a block is solved column by column: mg_solve_dev_FP64, ghost-layer form)
        self.dev = D.DeviceHierarchy(G.param, device_id, self.nrhs, options=options)
        self.lib = lib = self.dev.lib
        h = self.dev.handle
        rank, size = G.rank, G.size
        uid = uid2 = None
        if transport == "rccl":      # two communicators: collectives on the compute stream, ghost-layer send / recv on the side stream
            buf, buf2 = C.create_string_buffer(128), C.create_string_buffer(128)
            if rank == 0:
                D._check(lib, lib.mg_dist_unique_id(buf), "mg_dist_unique_id")
                D._check(lib, lib.mg_dist_unique_id(buf2), "mg_dist_unique_id")
            box = [buf.raw if rank == 0 else None, buf2.raw if rank == 0 else None]
            if size > 1:
                import torch.distributed as dist
                dist.broadcast_object_list(box, src=0, group=group)
            uid, uid2 = C.create_string_buffer(box[0], 128), C.create_string_buffer(box[1], 128)
        D._check(lib, lib.mg_ghost_attach(h, rank, size, G.a, uid), "mg_ghost_attach")
        if uid2 is not None and os.environ.get("MG_GHOST_ONE_COMM", "0") != "1":      # (=1: send / receive share the first communicator)
            D._check(lib, lib.mg_ghost_set_side_comm(h, uid2), "mg_ghost_set_side_comm")
        self._cb = None
        if transport == "dry":
            D._check(lib, lib.mg_ghost_set_dry(h, 1), "mg_ghost_set_dry")
        elif uid is None and size > 1:
            self._install_plugin()
        i64 = lambda v: np.ascontiguousarray(v, dtype=np.int64)
        for l, L in enumerate(G.levels, start=1):
            pad = lambda v: i64(list(v) + [1] * (3 - len(v)))
            lo, hi = i64(list(L.own_lo) + [0] * (3 - len(L.own_lo))), pad(L.own_hi)
            ext = pad(L.ext_n)
            si, ri, ss, rs = i64(L.send_idx), i64(L.recv_idx), i64(L.send_splits), i64(L.recv_splits)
            D._check(lib, lib.mg_ghost_set_level_INT64(h, l, D._i64(ext), D._i64(lo), D._i64(hi), int(min(L.gmin, FULL)),
                                                       si.size, D._i64(si) if si.size else None, D._i64(ss),
                                                       ri.size, D._i64(ri) if ri.size else None, D._i64(rs)),
                     "mg_ghost_set_level")
        D._check(lib, lib.mg_ghost_finalize(h), "mg_ghost_finalize")

    def _install_plugin(self):
        import ctypes as C
        import torch
        import torch.distributed as dist
        size, group = self.G.size, self.group
        dp, lp = C.POINTER(C.c_double), C.POINTER(C.c_longlong)
        FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_longlong, dp, lp, dp, lp, C.c_longlong)

        class _TorchCollectives:
            def all_to_all(self, send, ss, rs):
                r_t = torch.zeros(sum(rs), dtype=torch.float64)
                dist.all_to_all_single(r_t, torch.from_numpy(send), rs, ss, group=group)
                return r_t.numpy()

            def all_reduce(self, values):
                t = torch.from_numpy(values)
                dist.all_reduce(t, group=group)
                return t.numpy()

        coll = self.collectives if self.collectives is not None else _TorchCollectives()

        def cb(_user, op, send, send_splits, recv, recv_splits, count):
            try:
                if op == 0:
                    ss = [int(send_splits[i]) for i in range(size)]
                    rs = [int(recv_splits[i]) for i in range(size)]
                    out = coll.all_to_all(np.ctypeslib.as_array(send, shape=(max(sum(ss), 1),))[: sum(ss)].copy(), ss, rs)
                    if sum(rs):
                        np.ctypeslib.as_array(recv, shape=(sum(rs),))[:] = out
                elif op == 1:
                    np.ctypeslib.as_array(recv, shape=(int(count),))[:] = coll.all_reduce(np.ctypeslib.as_array(send, shape=(int(count),)).copy())
                else:
                    return 1
                return 0
            except Exception as e:          # never unwind through the C frame
                print("exchange plug-in error:", repr(e), flush=True)
                return 1

        self._cb = FN(cb)
        self.D._check(self.lib, self.lib.mg_ghost_set_exchange_plugin(self.dev.handle, C.cast(self._cb, C.c_void_p), None),
                      "mg_ghost_set_exchange_plugin")

    @property
    def n_ext(self):
        return self.G.levels[0].n

    def solve(self, b_ext, x_ext, tol: float, maxIter: int):
        """solveMG on this rank's extended fine box (device tensors of n_ext doubles - n_ext x nrhs, row-major, for a block; the
        owned rows of b must be valid, the library fills its ghost layers; x: in/out, owned rows valid on return)."""
        return self.dev.solve_dev(b_ext, x_ext, tol, maxIter)

    def cycle(self, b_ext, x_ext, x_is_zero: bool):
        return self.dev.cycle_dev(b_ext, x_ext, 1 if x_is_zero else 0)

    def exchanges(self):
        """(exchanges started, doubles sent) since attach: the communication the schedule really issues."""
        import ctypes as C
        a, b = C.c_longlong(0), C.c_longlong(0)
        self.D._check(self.lib, self.lib.mg_ghost_stats(self.dev.handle, C.byref(a), C.byref(b)), "mg_ghost_stats")
        return int(a.value), int(b.value)

    def allreduces(self) -> int:
        """all-reduces this rank entered since attach (norms, dots, rows of the first replicated level)."""
        import ctypes as C
        c = C.c_longlong(0)
        self.D._check(self.lib, self.lib.mg_ghost_allreduce_count(self.dev.handle, C.byref(c)), "mg_ghost_allreduce_count")
        return int(c.value)

    # -- MG-preconditioned Krylov on this rank's extended fine box (solveCG_MG / solveBiCGSTAB_MG / solveGMRES_MG, SolveFuncs.jl:74-133):
    #    b, x device tensors of n_ext doubles, owned rows of b valid, owned rows of x valid on return; sums over the owned rows of all ranks
    #    (a block of nrhs > 1 columns, n_ext x nrhs row-major: KrylovMethods.blockCG / blockBiCGSTB / blockFGMRES, as the reference's wrappers call them)
    def pcg(self, b_ext, x_ext, tol: float, maxIter: int):
        return self.dev.pcg_dev(b_ext, x_ext, tol, maxIter) if self.nrhs == 1 else self.dev.block_pcg_dev(b_ext, x_ext, tol, maxIter)

    def bicgstab(self, b_ext, x_ext, tol: float, maxIter: int):
        return self.dev.bicgstab_dev(b_ext, x_ext, tol, maxIter) if self.nrhs == 1 else self.dev.block_bicgstab_dev(b_ext, x_ext, tol, maxIter)

    def fgmres(self, b_ext, x_ext, inner: int, tol: float, maxIter: int):
        return self.dev.fgmres_dev(b_ext, x_ext, inner, tol, maxIter) if self.nrhs == 1 else self.dev.block_fgmres_dev(b_ext, x_ext, inner, tol, maxIter)

    def comm_count(self) -> int:
        import ctypes as C
        c = C.c_longlong(0)
        self.D._check(self.lib, self.lib.mg_ghost_comm_count(self.dev.handle, C.byref(c)), "mg_ghost_comm_count")
        return int(c.value)

    def close(self):
        if self.dev is not None:
            self.dev.close()
            self.dev = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
