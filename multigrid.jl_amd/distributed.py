"""Multi-GPU multigrid cycle: one process per GPU, the fine levels sharded by the reference's
DomainDecomposition box rule, halo exchange per SpMV, coarse tail replicated.

What is taken from the reference (src/DomainDecomposition/): only the PARTITION -
``getOriginalBoundingBoxCells`` (DDIndices.jl:41-47: cellSize = div(nc,NumCells), the last box absorbs
the remainder), the x-fastest nodal enumeration (DDIndices.jl:141-162) and the subdomain numbering
``loc2cs`` (DDService.jl:27-36).  The Schwarz solver itself is a different algorithm and is out of
scope; here every rank runs the SAME multigrid cycle as the single-GPU library (MGcycle.jl:1-118,
SolveFuncs.jl:3-39) on its rows, so the iterates equal the single-GPU ones up to fp64 reassociation.

Layout.  On level l a rank owns a set of rows (a box of nodes for GMG, a contiguous block for general
CSR), renumbered locally 0..n_own-1.  Each local operator keeps its owned rows; column indices are
renumbered [owned | halo grouped by owner rank, ascending global index], so the local source vector
is [owned values | received halo] and ONE all_to_all_single per SpMV fills the halo tail in place.
Levels at or below ``replicate_below`` rows are all-gathered and run replicated on every GPU by an
ordinary single-GPU hierarchy (no further communication), re-entering the sharded path on the way up.

PyTorch is plumbing only: device buffers, streams and ``torch.distributed`` (backend "nccl" = RCCL over
xGMI; "gloo" for the CPU tests).  All arithmetic runs in the HIP kernels of libmgvcycle.so through
``HipBackend``; there is no CPU fallback in this module - the CPU tests inject their own checker backend.
"""
from __future__ import annotations

import os as _os
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import scipy.sparse as sp

from . import device as D
from .mgdef import MGparam


# ======================================================================================================
# partition
# ======================================================================================================
def box_owner(n_nodes, numDomains) -> np.ndarray:
    """Owner subdomain (0-based linear id, x-fastest as ``loc2cs``) of every node of a regular nodal grid.

    Cells are split by ``getOriginalBoundingBoxCells`` (DDIndices.jl:41-47): cellSize = div(nc, NumCells),
    box i holds cells (i-1)*cellSize+1 .. i*cellSize and the last box takes the remainder.  A node belongs to
    the box of the cell on its upper side, the last node of a dimension to the last box - so every node has
    exactly one owner and the interface nodes go to the upper box.
    """
    n_nodes = np.asarray(n_nodes, dtype=np.int64)
    numDomains = np.asarray(numDomains, dtype=np.int64)
    if n_nodes.size != numDomains.size:
        raise ValueError("n_nodes and numDomains must have the same dimension")
    idx1d = []
    for k in range(n_nodes.size):
        nc = int(n_nodes[k]) - 1
        nd = int(numDomains[k])
        if nd < 1 or nd > max(nc, 1):
            raise ValueError(f"cannot split {nc} cells into {nd} boxes")
        cs = max(nc // nd, 1)
        idx1d.append(np.minimum(np.arange(n_nodes[k]) // cs, nd - 1))
    if n_nodes.size == 2:
        own = idx1d[0][None, :] + numDomains[0] * idx1d[1][:, None]
    else:
        own = (idx1d[0][None, None, :] + numDomains[0] * idx1d[1][None, :, None]
               + numDomains[0] * numDomains[1] * idx1d[2][:, None, None])
    return own.reshape(-1).astype(np.int32)          # C-order of (z,y,x) == x-fastest linear index


def block_owner(n: int, nranks: int) -> np.ndarray:
    """Contiguous row blocks (general CSR / SA-AMG: SURVEY.md 8e 'unstructured')."""
    return np.minimum(np.arange(n, dtype=np.int64) * nranks // n, nranks - 1).astype(np.int32)


def default_domains(nranks: int, dim: int):
    """[2,2,2] for 8 GPUs (7 neighbours = the 7 xGMI peers), slabs [1,1,k] otherwise."""
    if dim == 3:
        table = {1: [1, 1, 1], 2: [1, 1, 2], 4: [1, 2, 2], 8: [2, 2, 2]}
        return table.get(nranks, [1, 1, nranks])
    table = {1: [1, 1], 2: [1, 2], 4: [2, 2]}
    return table.get(nranks, [1, nranks])


class Partition:
    """Row ownership of one level: ``rows[r]`` = ascending global ids owned by rank r (= its local order)."""

    def __init__(self, owner: np.ndarray, nranks: int):
        self.owner = np.ascontiguousarray(owner, dtype=np.int32)
        self.nranks = int(nranks)
        n = self.owner.size
        order = np.argsort(self.owner, kind="stable")
        counts = np.bincount(self.owner, minlength=nranks)
        if np.any(counts == 0):
            raise ValueError("a rank owns no rows on a distributed level")
        off = np.concatenate([[0], np.cumsum(counts)])
        self.counts = counts
        self.rows = [order[off[r]:off[r + 1]] for r in range(nranks)]
        self.local_index = np.empty(n, dtype=np.int64)
        self.local_index[order] = np.arange(n) - off[self.owner[order]]

    def coarsen(self, P) -> "Partition":
        """Owner of a coarse point = owner of the fine row carrying its largest interpolation weight
        (for full weighting: the coincident fine node; for aggregation: a member of the aggregate)."""
        Pc = sp.csc_matrix(P)
        nc = Pc.shape[1]
        owner_c = np.empty(nc, dtype=np.int32)
        absd = np.abs(Pc.data)
        # argmax per column (first maximum)
        colmax = np.maximum.reduceat(absd, Pc.indptr[:-1])
        col_of = np.repeat(np.arange(nc), np.diff(Pc.indptr))
        ismax = absd == colmax[col_of]
        first = np.full(nc, -1, dtype=np.int64)
        pos = np.nonzero(ismax)[0][::-1]
        first[col_of[pos]] = pos                       # the smallest position wins (reverse assignment)
        owner_c[:] = self.owner[Pc.indices[first]]
        return Partition(owner_c, self.nranks)


@dataclass
class HaloPlan:
    n_own_src: int
    n_halo: int
    send_idx: np.ndarray          # local indices (into the owned part of the source vector) to send, grouped by peer
    send_splits: List[int]
    recv_splits: List[int]


def localize(M, row_part: Partition, col_part: Optional[Partition], rank: int):
    """Rows of M owned by `rank`, columns renumbered [owned | halo by (owner, global id)] + the halo plan.
    ``col_part=None``: the source vector is replicated (coarse tail) - global column ids are kept."""
    M = sp.csr_matrix(M)
    rows = row_part.rows[rank]
    Ml = M[rows, :].tocsr()
    Ml.sort_indices()
    if col_part is None:
        return Ml, None
    nr = row_part.nranks
    cols = Ml.indices
    cown = col_part.owner[cols]
    mine = cown == rank
    halo = np.unique(cols[~mine])
    ho = col_part.owner[halo]
    order = np.lexsort((halo, ho))
    halo, ho = halo[order], ho[order]
    n_own = int(col_part.counts[rank])
    newcol = np.empty(cols.size, dtype=np.int64)
    newcol[mine] = col_part.local_index[cols[mine]]
    # position of each off-rank column in the (owner, id)-ordered halo list
    key = ho.astype(np.int64) * M.shape[1] + halo
    q = cown[~mine].astype(np.int64) * M.shape[1] + cols[~mine]
    newcol[~mine] = n_own + np.searchsorted(key, q)
    Mloc = sp.csr_matrix((Ml.data, newcol, Ml.indptr), shape=(rows.size, n_own + halo.size))
    Mloc.sort_indices()
    recv_splits = np.bincount(ho, minlength=nr).astype(int).tolist()
    # what the peers need from me: columns I own that appear in THEIR rows (computed locally: every
    # rank holds the global pattern in this round's host-setup design)
    ro = np.repeat(row_part.owner, np.diff(M.indptr))
    co = col_part.owner[M.indices]
    sel = (co == rank) & (ro != rank)
    dest = ro[sel].astype(np.int64)
    gcol = M.indices[sel].astype(np.int64)
    pairs = np.unique(dest * M.shape[1] + gcol)
    dest_u = pairs // M.shape[1]
    gcol_u = pairs - dest_u * M.shape[1]
    send_idx = col_part.local_index[gcol_u]
    send_splits = np.bincount(dest_u, minlength=nr).astype(int).tolist()
    return Mloc, HaloPlan(n_own, int(halo.size), send_idx.astype(np.int64), send_splits, recv_splits)


# ======================================================================================================
# communication (torch.distributed plumbing)
# ======================================================================================================
class TorchComm:
    """all_to_all_single / all_reduce / all_gather on a torch.distributed group.
    ``stage_through_host`` moves device tensors through pinned host memory for backends that cannot take
    device tensors (gloo): used to exercise the HIP path with 2 processes on ONE GPU."""

    def __init__(self, group=None, stage_through_host: bool = False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.stage = stage_through_host

    def all_to_all(self, out, inp, out_splits, in_splits):
        if self.stage and out.is_cuda:
            o = out.cpu()
            self.dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(o)
        else:
            self.dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    def all_to_all_start(self, out, inp, out_splits, in_splits):
        """Begin the exchange and return a handle for ``finish`` (None if it already completed): with RCCL the
        collective runs on torch's communication stream while the caller enqueues the interior kernel."""
        if self.stage or not out.is_cuda:
            self.all_to_all(out, inp, out_splits, in_splits)
            return None
        return self.dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group, async_op=True)

    @staticmethod
    def finish(handle):
        if handle is not None:
            handle.wait()            # the CURRENT stream waits for the collective; the host does not block

    def all_reduce_sum(self, t):
        if self.stage and t.is_cuda:
            c = t.cpu()
            self.dist.all_reduce(c, group=self.group)
            t.copy_(c)
        else:
            self.dist.all_reduce(t, group=self.group)
        return t

    def all_gather(self, out, inp):
        if self.stage and out.is_cuda:
            o = out.cpu()
            self.dist.all_gather_into_tensor(o, inp.cpu(), group=self.group)
            out.copy_(o)
        else:
            self.dist.all_gather_into_tensor(out, inp, group=self.group)


class SingleComm:
    """World of one process (no communication)."""
    rank, size = 0, 1

    def all_to_all(self, out, inp, out_splits, in_splits):
        out.copy_(inp)

    def all_to_all_start(self, out, inp, out_splits, in_splits):
        return None

    @staticmethod
    def finish(handle):
        pass

    def all_reduce_sum(self, t):
        return t

    def all_gather(self, out, inp):
        out.copy_(inp)


# ======================================================================================================
# local compute backend: the HIP library
# ======================================================================================================
class HipBackend:
    """Local arithmetic of the distributed cycle = the single-GPU kernels of libmgvcycle.so, enqueued on
    torch's current stream so that they order with the RCCL collectives torch issues."""

    def __init__(self, device_id: int):
        import torch
        if not torch.cuda.is_available():
            raise D.MGDeviceError("no GPU visible: the multigrid cycle has no CPU fallback")
        self.torch = torch
        self.device_id = int(device_id)
        self.dev = torch.device("cuda", self.device_id)
        torch.cuda.set_device(self.dev)
        D.load_library()
        self.ws = torch.zeros(1024, dtype=torch.float64, device=self.dev)
        self.scalar = torch.zeros(1, dtype=torch.float64, device=self.dev)

    def stream(self) -> int:
        return int(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def zeros(self, *shape):
        return self.torch.zeros(*shape, dtype=self.torch.float64, device=self.dev)

    def from_numpy(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)

    def index_tensor(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(self.dev)

    supports_box = True          # box-form local operators (mg_op_create_box_FP64_INT64) with phase-split launches

    def operator(self, M, box=None, regular_cols=None, coarse_box=None):
        return D.DeviceOperator(M, self.device_id, box=box, regular_cols=regular_cols, coarse_box=coarse_box)

    def bind_relax(self, op, d, n):
        self.synchronize()
        op.bind_relax(d, n)

    def apply(self, op, kernel, x, y, b=None, d=None, alpha=1.0, beta=0.0, nrhs=1, row_offset=0, phase=0):
        op.apply(kernel, x, y, b, d, alpha, beta, nrhs, self.stream(), row_offset, phase)

    def dscale(self, d, b, x, n, nrhs):
        D.vec_dscale(d, b, x, n, nrhs, self.stream())

    def xpdr(self, x, d, r, xout, n, nrhs):
        D.vec_xpdr(x, d, r, xout, n, nrhs, self.stream())

    def sumsq(self, x, length):
        D.vec_sumsq(x, length, self.ws, self.scalar, self.stream())
        return self.scalar

    def index_select(self, src, idx, out):
        self.torch.index_select(src, 0, idx, out=out)

    def tail(self, sub: MGparam, nrhs: int):
        h = D.DeviceHierarchy(sub, self.device_id, nrhs)
        h.set_stream(self.stream())
        return _HipTail(h)

    def synchronize(self):
        self.torch.cuda.synchronize(self.dev)


class _HipTail:
    def __init__(self, h):
        self.h = h

    def cycle(self, b, x, x_zero: bool, ctype: str):
        D._check(self.h.lib, self.h.lib.mg_set_cycle_type(self.h.handle, ord(ctype)), "mg_set_cycle_type")
        self.h.cycle_async_dev(b, x, 1 if x_zero else 0)


# ======================================================================================================
# the distributed hierarchy and cycle
# ======================================================================================================
class _Level:
    pass


def _sub_hierarchy(param: MGparam, start: int) -> MGparam:
    """Levels start.. of `param` as a hierarchy of their own (the replicated coarse tail)."""
    from .mgdef import getMGparam
    sub = getMGparam(np.float64, np.int64, len(param.As) - start, param.numCores, param.maxOuterIter,
                     param.relativeTol, param.relaxType, param.relaxParam,
                     lambda level, _s=start: param.relaxPre(level + _s), lambda level, _s=start: param.relaxPost(level + _s),
                     param.cycleType, param.coarseSolveType, param.strongConnParam, param.FilteringParam,
                     param.transferOperatorType)
    sub.As = param.As[start:]
    sub.Ps = param.Ps[start:]
    sub.Rs = param.Rs[start:]
    sub.relaxPrecs = param.relaxPrecs[start:]
    sub.LU = param.LU
    sub.levels = len(sub.As)
    return sub


def _reorder_for_overlap(local_levels, rows_fine, keep_order: bool = False):
    """Renumber every sharded level's owned rows as [interior | boundary], interior = rows of A whose columns
    are all owned.  The interior part of an SpMV with A can then run while the halo is in flight, the boundary
    part after it arrived.  A pure local permutation: applied consistently to the rows/owned columns of every
    local operator, to d, to the send lists and to the global ids of the owned fine rows.
    keep_order (box form): the rows stay in the natural order of the owned box and A is ONE operator whose
    rows that read the halo are exception rows of the device format (n_int = n_own)."""
    perms, invs, nints = [], [], []
    for ld in local_levels:
        A = ld["A"]
        n = ld["n_own"]
        touches_halo = np.zeros(n, dtype=bool)
        rows_of = np.repeat(np.arange(n), np.diff(A.indptr))
        touches_halo[rows_of[A.indices >= n]] = True
        if keep_order:
            touches_halo[:] = False
        perm = np.concatenate([np.nonzero(~touches_halo)[0], np.nonzero(touches_halo)[0]])
        inv = np.empty(n, dtype=np.int64)
        inv[perm] = np.arange(n)
        perms.append(perm)
        invs.append(inv)
        nints.append(int((~touches_halo).sum()))

    def remap_cols(M, inv):
        M = M.tocsr(copy=True)
        own = M.indices < inv.size
        M.indices = np.where(own, inv[np.minimum(M.indices, inv.size - 1)], M.indices).astype(M.indices.dtype)
        M.sort_indices()
        return M

    a = len(local_levels)
    for l, ld in enumerate(local_levels):
        perm, inv = perms[l], invs[l]
        ld["A"] = remap_cols(ld["A"][perm, :], inv)
        ld["d"] = np.asarray(ld["d"])[perm]
        ld["planA"].send_idx = inv[ld["planA"].send_idx]
        ld["planR"].send_idx = inv[ld["planR"].send_idx]
        R = ld["R"] if l + 1 >= a else ld["R"][perms[l + 1], :]          # rows live on level l+1
        ld["R"] = remap_cols(R, inv)
        P = ld["P"][perm, :]
        if l + 1 < a:
            P = remap_cols(P, invs[l + 1])
            ld["planP"].send_idx = invs[l + 1][ld["planP"].send_idx]
        else:
            P = P.tocsr()
            P.sort_indices()
        ld["P"] = P
        ld["n_int"] = nints[l]
    return np.asarray(rows_fine)[perms[0]], perms[0]


def _box_of_rows(rows, nodes):
    """(b1,b2,b3) if the ascending global ids `rows` are exactly a box of the x-fastest grid `nodes`, else None."""
    nodes = [int(v) for v in nodes] + [1] * (3 - len(nodes))
    rows = np.asarray(rows, dtype=np.int64)
    i, j, k = rows % nodes[0], (rows // nodes[0]) % nodes[1], rows // (nodes[0] * nodes[1])
    ext = [int(v.max() - v.min() + 1) for v in (i, j, k)]
    if ext[0] * ext[1] * ext[2] != rows.size:
        return None
    want = ((k - k.min()) * ext[1] + (j - j.min())) * ext[0] + (i - i.min())
    return tuple(ext) if np.array_equal(want, np.arange(rows.size)) else None


class DistributedHierarchy:
    """The multi-GPU counterpart of ``DeviceHierarchy``: this rank's rows of the sharded levels plus the
    replicated coarse tail.  Two builders:

    * ``from_global``     - every rank holds the global host hierarchy ``param`` (any CSR hierarchy, GMG or
                            SA-AMG) and cuts its rows out of it;
    * ``from_structured`` - ``structured_setup.py``: every rank builds only its own part of a GMG hierarchy
                            on an overlapping local box (no global matrix ever exists) - what makes the
                            512^3 / 8-GPU configuration feasible.
    """

    def __init__(self, comm, backend, local_levels, tail_param: MGparam, tail_owner, tail_local_index,
                 tail_counts, rows_fine, cycleType: str, nl_total: int, nrhs: int = 1):
        self.comm = comm
        self.be = be = backend
        self.nrhs = k = int(nrhs)
        self.cycleType = cycleType
        self.relaxType = getattr(tail_param, "relaxType", "Jac")
        rank, size = comm.rank, comm.size
        self.first_tail = len(local_levels)
        self.nl = int(nl_total)
        self.levels: List[_Level] = []
        # BOX form: every sharded level's owned rows are a box of a regular grid (ld["box"]) and the backend can hold A as
        # one square grid operator with the halo appended - the staged single-GPU kernels then serve the sharded levels
        self.box_form = (k == 1 and getattr(be, "supports_box", False) and all(ld.get("box") is not None for ld in local_levels)
                         and not _os.environ.get("MG_DIST_NO_BOX"))
        rows_fine, self.fine_perm = _reorder_for_overlap(local_levels, rows_fine, keep_order=self.box_form)
        for ld in local_levels:
            L = _Level()
            L.n_own = int(ld["n_own"])
            L.n_int = int(ld["n_int"])
            L.planA, L.planR, L.planP = ld["planA"], ld["planR"], ld["planP"]
            A = ld["A"]
            L.box = None
            if self.box_form:
                # square [owned box | halo] with empty halo rows; rows reading the halo become exception rows on the device
                n_tot = A.shape[1]
                Asq = sp.csr_matrix((A.data, A.indices, np.concatenate([A.indptr, np.full(n_tot - A.shape[0], A.indptr[-1],
                                                                                           dtype=A.indptr.dtype)])),
                                    shape=(n_tot, n_tot))
                L.box = tuple(int(v) for v in ld["box"])
                L.A_int = be.operator(Asq, box=L.box, regular_cols=L.n_own)
                L.A_bnd = None
            else:
                # A is held as two operators: interior rows (no halo columns) and boundary rows
                L.A_int = be.operator(A[: L.n_int, :]) if L.n_int > 0 else None
                L.A_bnd = be.operator(A[L.n_int:, :]) if L.n_int < L.n_own else None
            if self.box_form and not _os.environ.get("MG_DIST_NO_GRID_P") and L.n_own < ld["R"].shape[1]:
                # R with the owned | halo split of its columns: the coarse rows that read owned residuals only run beside
                # the exchange of r (phase 1), the others behind it (phase 2)
                L.R = be.operator(ld["R"], regular_cols=L.n_own)
            else:
                L.R = be.operator(ld["R"])
            if self.box_form and ld.get("cbox") is not None and not _os.environ.get("MG_DIST_NO_GRID_P"):
                # grid form of P: rows = the owned fine box, columns = [owned coarse box | halo] (or the whole replicated
                # coarse grid): the LDS-staged prolongation kernel serves the rows that read no halo column
                L.P = be.operator(ld["P"], box=L.box, regular_cols=int(ld["cbox_cols"]), coarse_box=tuple(int(v) for v in ld["cbox"]))
            else:
                L.P = be.operator(ld["P"])
            L.nnzA, L.nnzR, L.nnzP = ld["A"].nnz, ld["R"].nnz, ld["P"].nnz
            L.d = be.from_numpy(np.asarray(ld["d"], dtype=np.float64))
            if hasattr(be, "bind_relax") and L.A_int is not None:      # relaxPrec from the class dictionary where possible
                be.bind_relax(L.A_int, L.d, L.n_own if L.box else L.n_int)
            L.npre = max(1, int(ld["npre"]))            # relax() always updates once (MGcycle.jl:127-134)
            L.npost = max(1, int(ld["npost"]))
            self.levels.append(L)
        # vectors: capacity = owned + the largest halo any operator appends to that vector
        for l, L in enumerate(self.levels):
            halo_x = L.planA.n_halo
            if l > 0 and self.levels[l - 1].planP is not None:
                halo_x = max(halo_x, self.levels[l - 1].planP.n_halo)
            L.cap_x = L.n_own + halo_x
            L.cap_r = L.n_own + L.planR.n_halo
            L.x0 = be.zeros(L.cap_x, k) if k > 1 else be.zeros(L.cap_x)
            L.x1 = be.zeros(L.cap_x, k) if k > 1 else be.zeros(L.cap_x)
            L.r = be.zeros(L.cap_r, k) if k > 1 else be.zeros(L.cap_r)
            L.b = (be.zeros(L.n_own, k) if k > 1 else be.zeros(L.n_own)) if l > 0 else None
            for plan in (L.planA, L.planR, L.planP):
                if plan is not None:
                    plan.send_idx_t = be.index_tensor(plan.send_idx)
                    ns = int(plan.send_idx.size)
                    plan.send_buf = be.zeros(max(ns, 1), k) if k > 1 else be.zeros(max(ns, 1))
                    # a collective must be entered by every rank or by none: agree once, at setup
                    flag = be.zeros(1)
                    flag += float(ns + plan.n_halo)
                    plan.active = size > 1 and float(comm.all_reduce_sum(flag).item()) > 0.0
        # replicated tail
        nt = int(tail_param.As[0].shape[0])
        self.n_tail = nt
        self.tail = be.tail(tail_param, k)
        tail_counts = np.asarray(tail_counts)
        self.own_tail = int(tail_counts[rank])
        self.max_tail = int(tail_counts.max())
        self.bc_pad = be.zeros(self.max_tail, k) if k > 1 else be.zeros(self.max_tail)
        self.bc_all = be.zeros(self.max_tail * size, k) if k > 1 else be.zeros(self.max_tail * size)
        gather_index = np.asarray(tail_owner, dtype=np.int64) * self.max_tail + np.asarray(tail_local_index, dtype=np.int64)
        self.gather_index = be.index_tensor(gather_index)
        self.b_tail = be.zeros(nt, k) if k > 1 else be.zeros(nt)
        self.x_tail = be.zeros(nt, k) if k > 1 else be.zeros(nt)
        self.rows_fine = np.asarray(rows_fine)

    @staticmethod
    def check_supported(param: MGparam, native: bool = False):
        """The Python-sequenced schedule implements the pointwise smoothers and V/W/F cycles; the native sequencer
        (mg_dist_*) also the Jac-GMRES smoother and the K-cycle (all-reduced dots).  Anything else must fail loudly rather
        than silently run a different method."""
        relax_ok = ("Jac", "SPAI", "Jac-GMRES") if native else ("Jac", "SPAI")
        cycle_ok = ("V", "W", "F", "K") if native else ("V", "W", "F")
        if param.relaxType not in relax_ok:
            raise NotImplementedError(f"relaxType={param.relaxType!r} is not implemented in the multi-GPU cycle"
                                      + ("" if native else " of the Python sequencer (the native one, mg_dist_*, has Jac-GMRES)"))
        if param.cycleType not in cycle_ok:
            raise NotImplementedError(f"cycleType={param.cycleType!r} is not implemented in the multi-GPU cycle"
                                      + ("" if native else " of the Python sequencer (the native one, mg_dist_*, has the K-cycle)"))

    @classmethod
    def from_global(cls, param: MGparam, comm, backend, fine_owner: np.ndarray, nrhs: int = 1,
                    replicate_below: int = 300_000, level_nodes=None, native_only: bool = False):
        """level_nodes (optional): nodes per dimension of every level's regular grid ([n1,n2(,n3)] per level, x fastest).
        When a rank's rows of a sharded level are a box of that grid, the level is held in BOX form.
        native_only: the hierarchy will be driven by NativeDistributedHierarchy only (Jac-GMRES / K-cycle allowed)."""
        cls.check_supported(param, native=native_only)
        rank, size = comm.rank, comm.size
        nl = len(param.As)
        if nl < 2:
            raise ValueError("a distributed hierarchy needs at least two levels")
        # which levels are sharded: the finest one always; then while the level is large enough
        parts = [Partition(fine_owner, size)]
        a = 1
        while a < nl - 1 and param.As[a].shape[0] > replicate_below:
            try:
                parts.append(parts[-1].coarsen(param.Ps[a - 1]))
            except ValueError:
                break
            a += 1
        # the partition of the first replicated level is still needed (rows of R, gather of bc)
        part_tail = parts[-1].coarsen(param.Ps[a - 1])
        local_levels = []
        for l in range(a):
            part = parts[l]
            cpart = parts[l + 1] if l + 1 < a else part_tail
            A_loc, planA = localize(param.As[l], part, part, rank)
            R_loc, planR = localize(param.Rs[l], cpart, part, rank)
            # the prolongation gathers the sharded x_{l+1}, or the replicated tail solution (no halo)
            P_loc, planP = localize(param.Ps[l], part, cpart if l + 1 < a else None, rank)
            cbox, cbox_cols = None, 0
            if level_nodes is not None and l + 1 < len(level_nodes):
                if l + 1 < a:       # columns of P: [this rank's box of the next sharded level | halo]
                    cbox, cbox_cols = _box_of_rows(cpart.rows[rank], level_nodes[l + 1]), int(cpart.counts[rank])
                else:               # the replicated tail: the whole coarse grid
                    cbox, cbox_cols = tuple(int(v) for v in level_nodes[l + 1]), int(P_loc.shape[1])
                    if int(np.prod(cbox)) != cbox_cols:
                        cbox = None
            local_levels.append(dict(n_own=int(part.counts[rank]), A=A_loc, planA=planA, R=R_loc, planR=planR,
                                     P=P_loc, planP=planP, d=np.asarray(param.relaxPrecs[l])[part.rows[rank]],
                                     npre=param.relaxPre(l + 1), npost=param.relaxPost(l + 1),
                                     box=_box_of_rows(part.rows[rank], level_nodes[l]) if level_nodes is not None else None,
                                     cbox=cbox, cbox_cols=cbox_cols))
        H = cls(comm, backend, local_levels, _sub_hierarchy(param, a), part_tail.owner, part_tail.local_index,
                part_tail.counts, parts[0].rows[rank], param.cycleType, nl, nrhs)
        H.parts, H.part_tail = parts, part_tail
        return H

    # ---- helpers ---------------------------------------------------------------------------------------
    def scatter_fine(self, v_global: np.ndarray):
        """This rank's rows of a global fine-level vector/block (host) as a device tensor."""
        return self.be.from_numpy(np.asarray(v_global)[self.rows_fine])

    def order_fine(self, v_own):
        """A vector given on this rank's fine rows in ascending-global-id order -> this hierarchy's local
        order ([interior | boundary], the order of ``rows_fine``)."""
        return np.asarray(v_own)[self.fine_perm]

    def exchange_start(self, plan: HaloPlan, buf):
        """Pack the values the peers need and start filling the halo tail buf[n_own : n_own+n_halo]
        (one all_to_all_single); returns a handle for ``comm.finish``."""
        if plan is None or not plan.active:
            return None
        ns = int(plan.send_idx.size)
        send = plan.send_buf[:ns]
        if ns:
            self.be.index_select(buf, plan.send_idx_t, send)
        recv = buf[plan.n_own_src: plan.n_own_src + plan.n_halo]
        return self.comm.all_to_all_start(recv, send, plan.recv_splits, plan.send_splits)

    def exchange(self, plan: HaloPlan, buf):
        self.comm.finish(self.exchange_start(plan, buf))

    def apply_A(self, L, kernel, x, out, b):
        """out = b - A x  /  out = x + d.*(b - A x) on this rank's rows with the halo exchange OVERLAPPED: the
        interior rows (no halo column) are computed while the all_to_all is in flight on the communication
        stream, the boundary rows once it has landed."""
        be, k = self.be, self.nrhs
        h = self.exchange_start(L.planA, x)
        if L.box is not None:       # box form: rows in dictionary classes now, rows that read the halo after it landed
            be.apply(L.A_int, kernel, x, out, b=b, d=L.d, nrhs=k, row_offset=0, phase=1)
            self.comm.finish(h)
            be.apply(L.A_int, kernel, x, out, b=b, d=L.d, nrhs=k, row_offset=0, phase=2)
            return
        if L.A_int is not None:
            be.apply(L.A_int, kernel, x, out, b=b, d=L.d, nrhs=k, row_offset=0)
        self.comm.finish(h)
        if L.A_bnd is not None:
            be.apply(L.A_bnd, kernel, x, out, b=b, d=L.d, nrhs=k, row_offset=L.n_int)

    def norm(self, v, n_own):
        """Global Frobenius norm of a sharded vector (SolveFuncs.jl:15,20,30): local sum of squares + all-reduce."""
        s = self.be.sumsq(v, n_own * self.nrhs)
        self.comm.all_reduce_sum(s)
        return float(s.item()) ** 0.5

    # ---- the cycle (mirror of csrc/mgvcycle.hip cycle_level; MGcycle.jl:1-118) ---------------------------
    def _cycle(self, l, b, xa, xb, x_zero, ctype, r_valid=False):
        be, k = self.be, self.nrhs
        L = self.levels[l]
        cur, alt = xa, xb
        npre, npost = L.npre, L.npost
        if x_zero:
            be.dscale(L.d, b, cur, L.n_own, k)
            npre -= 1
        elif r_valid:
            be.xpdr(cur, L.d, L.r, alt, L.n_own, k)
            cur, alt = alt, cur
            npre -= 1
        for _ in range(npre):
            self.apply_A(L, D.MG_K_SMOOTH, cur, alt, b)
            cur, alt = alt, cur
        self.apply_A(L, D.MG_K_RESIDUAL, cur, L.r, b)
        self.exchange(L.planR, L.r)
        if l + 1 < len(self.levels):
            C = self.levels[l + 1]
            be.apply(L.R, D.MG_K_RESTRICT, L.r, C.b, nrhs=k)
            xc = self._cycle(l + 1, C.b, C.x0, C.x1, True, ctype)
            if ctype in ("W", "F"):
                other = C.x1 if xc is C.x0 else C.x0
                xc = self._cycle(l + 1, C.b, xc, other, False, "W" if ctype == "W" else "V")
            self.exchange(L.planP, xc)
            be.apply(L.P, D.MG_K_PROLONG, xc, cur, alpha=1.0, beta=1.0, nrhs=k)
        else:
            # restrict into this rank's rows of the first replicated level, all-gather, run the tail replicated
            be.apply(L.R, D.MG_K_RESTRICT, L.r, self.bc_pad, nrhs=k)
            self.comm.all_gather(self.bc_all, self.bc_pad)
            be.index_select(self.bc_all, self.gather_index, self.b_tail)
            self.tail.cycle(self.b_tail, self.x_tail, True, ctype)
            if self.first_tail < self.nl - 1 and ctype in ("W", "F"):       # second visit (MGcycle.jl:79-84)
                self.tail.cycle(self.b_tail, self.x_tail, False, "W" if ctype == "W" else "V")
            be.apply(L.P, D.MG_K_PROLONG, self.x_tail, cur, alpha=1.0, beta=1.0, nrhs=k)
        for _ in range(npost):
            self.apply_A(L, D.MG_K_SMOOTH, cur, alt, b)
            cur, alt = alt, cur
        return cur

    # ---- public: cycle / solve on this rank's rows ---------------------------------------------------------
    def cycle(self, b_loc, x_loc, x_is_zero: bool):
        """One cycle; b_loc/x_loc hold this rank's fine rows (length n_own [x nrhs]); x_loc updated in place."""
        L = self.levels[0]
        L.x0[: L.n_own].copy_(x_loc)
        self._python_sequencer_supported()
        res = self._cycle(0, b_loc, L.x0, L.x1, bool(x_is_zero), self.cycleType)
        x_loc.copy_(res[: L.n_own])
        return x_loc

    def _python_sequencer_supported(self):
        if self.relaxType not in ("Jac", "SPAI") or self.cycleType not in ("V", "W", "F"):
            raise NotImplementedError("Jac-GMRES smoothing and the K-cycle run in the native sequencer only (NativeDistributedHierarchy)")

    def solve(self, b_loc, x_loc, tol: float, maxIter: int):
        """solveMG (SolveFuncs.jl:3-39) on sharded vectors; returns (iters, resvec)."""
        self._python_sequencer_supported()
        be, k = self.be, self.nrhs
        L = self.levels[0]
        cur, alt = L.x0, L.x1
        cur[: L.n_own].copy_(x_loc)
        xn = self.norm(cur, L.n_own)
        x_zero = xn == 0.0
        if x_zero:
            res0 = self.norm(b_loc, L.n_own)
        else:
            self.apply_A(L, D.MG_K_RESIDUAL, cur, L.r, b_loc)
            res0 = self.norm(L.r, L.n_own)
        resvec = [res0]
        it = 0
        for count in range(1, maxIter + 1):
            out = self._cycle(0, b_loc, cur, alt, x_zero, self.cycleType, r_valid=(count > 1 or not x_zero))
            if out is not cur:
                cur, alt = alt, cur
            x_zero = False
            self.apply_A(L, D.MG_K_RESIDUAL, cur, L.r, b_loc)
            res = self.norm(L.r, L.n_own)
            it += 1
            resvec.append(res)
            if res / res0 < tol:
                break
        x_loc.copy_(cur[: L.n_own])
        return it, np.array(resvec)

    def local_algorithmic_bytes(self):
        """Per-rank algorithmic bytes of the sharded levels of one V-cycle from x=0."""
        k = self.nrhs
        t = 0.0
        for L in self.levels:
            n = L.n_own
            sweep = 12.0 * L.nnzA + 4.0 * (n + 1) + 8.0 * k * (3 * n) + 8.0 * n
            resid = 12.0 * L.nnzA + 4.0 * (n + 1) + 8.0 * k * (3 * n)
            t += 8.0 * n * (1 + 2 * k) + (L.npre - 1 + L.npost) * sweep + resid
            t += 12.0 * (L.nnzR + L.nnzP) + 8.0 * k * (3 * n)
        return t


# ======================================================================================================
# the native sequencer (csrc: mg_dist_*): same local operators and plans, hot loop in C++ / RCCL
# ======================================================================================================
class NativeDistributedHierarchy:
    """The sharded cycle behind the C ABI (``mg_dist_*``, include/mgvcycle.h): the level schedule runs in C++, the halo
    exchange is RCCL send/recv on a side stream overlapped with the interior rows.  Built FROM a ``DistributedHierarchy``
    (which stays the setup helper and the checker): its device operators, relaxPrecs, halo plans and replicated tail are
    handed over by handle; the vectors and the loop belong to the library.  One right-hand side.

    transport="rccl": a communicator of its own (unique id from rank 0, broadcast with ``torch.distributed``);
    transport="plugin": every exchange goes through ``H.comm`` on host buffers (tests, ranks sharing one GPU)."""

    def __init__(self, H: DistributedHierarchy, transport: str = "rccl"):
        import ctypes as C
        if H.nrhs != 1 and (H.cycleType == "K" or H.relaxType == "Jac-GMRES"):
            raise NotImplementedError("blocks of right-hand sides: V, W, F cycles with the pointwise smoothers")
        self.H = H
        self.lib = lib = D.load_library()
        comm = H.comm
        rank, size = comm.rank, comm.size
        self.handle = C.c_void_p()
        uid = None
        if transport == "rccl":
            buf = C.create_string_buffer(128)
            if rank == 0:
                D._check(lib, lib.mg_dist_unique_id(buf), "mg_dist_unique_id")
            box = [buf.raw if rank == 0 else None]
            if size > 1:
                import torch.distributed as dist
                dist.broadcast_object_list(box, src=0, group=getattr(comm, "group", None))
            uid = C.create_string_buffer(box[0], 128)
        nl_sh = len(H.levels)
        D._check(lib, lib.mg_dist_create(H.be.device_id, rank, size, uid, nl_sh, H.nl, ord(H.cycleType), C.byref(self.handle)),
                 "mg_dist_create")
        D._check(lib, lib.mg_dist_set_relax_type(self.handle, 1 if H.relaxType == "Jac-GMRES" else 0), "mg_dist_set_relax_type")
        if H.nrhs != 1:
            D._check(lib, lib.mg_dist_set_nrhs(self.handle, int(H.nrhs)), "mg_dist_set_nrhs")
        self._cb = None
        if uid is None and size > 1:
            self._install_plugin(comm)
        i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
        for l, L in enumerate(H.levels, start=1):
            hA_int = L.A_int.handle if L.A_int is not None else None
            hA_bnd = L.A_bnd.handle if L.A_bnd is not None else None
            D._check(lib, lib.mg_dist_set_level(self.handle, l, L.n_own, L.n_int, hA_int, hA_bnd, L.P.handle, L.R.handle,
                                                D._ptr(L.d), L.npre, L.npost), "mg_dist_set_level")
            if L.box is not None:
                D._check(lib, lib.mg_dist_set_level_box(self.handle, l, 1), "mg_dist_set_level_box")
            for which, plan in ((D.MG_OP_A, L.planA), (D.MG_OP_R, L.planR), (D.MG_OP_P, L.planP)):
                if plan is None:
                    continue
                si, ss, rs = i64(plan.send_idx), i64(plan.send_splits), i64(plan.recv_splits)
                D._check(lib, lib.mg_dist_set_plan_INT64(self.handle, l, which, plan.n_own_src, plan.n_halo, si.size,
                                                         D._i64(si) if si.size else None, D._i64(ss), D._i64(rs),
                                                         1 if plan.active else 0), "mg_dist_set_plan")
        gi = i64(H.gather_index.cpu().numpy())
        D._check(lib, lib.mg_dist_set_tail_INT64(self.handle, H.tail.h.handle, H.n_tail, H.own_tail, H.max_tail, D._i64(gi)),
                 "mg_dist_set_tail")
        D._check(lib, lib.mg_dist_finalize(self.handle), "mg_dist_finalize")

    def _install_plugin(self, comm):
        import ctypes as C
        import torch
        size = comm.size
        dp, lp = C.POINTER(C.c_double), C.POINTER(C.c_longlong)
        FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_longlong, dp, lp, dp, lp, C.c_longlong)
        dist = comm.dist

        def cb(_user, op, send, send_splits, recv, recv_splits, count):
            try:
                if op == 0:
                    ss = [int(send_splits[i]) for i in range(size)]
                    rs = [int(recv_splits[i]) for i in range(size)]
                    s_t = torch.from_numpy(np.ctypeslib.as_array(send, shape=(max(sum(ss), 1),))[: sum(ss)].copy())
                    r_t = torch.zeros(sum(rs), dtype=torch.float64)
                    dist.all_to_all_single(r_t, s_t, rs, ss, group=comm.group)
                    if sum(rs):
                        np.ctypeslib.as_array(recv, shape=(sum(rs),))[:] = r_t.numpy()
                elif op == 1:
                    t = torch.from_numpy(np.ctypeslib.as_array(send, shape=(int(count),)).copy())
                    dist.all_reduce(t, group=comm.group)
                    np.ctypeslib.as_array(recv, shape=(int(count),))[:] = t.numpy()
                else:
                    t = torch.from_numpy(np.ctypeslib.as_array(send, shape=(int(count),)).copy())
                    o = torch.zeros(int(count) * size, dtype=torch.float64)
                    dist.all_gather_into_tensor(o, t, group=comm.group)
                    np.ctypeslib.as_array(recv, shape=(int(count) * size,))[:] = o.numpy()
                return 0
            except Exception as e:          # never unwind through the C frame
                print("exchange plug-in error:", repr(e), flush=True)
                return 1

        self._cb = FN(cb)
        D._check(self.lib, self.lib.mg_dist_set_exchange_plugin(self.handle, C.cast(self._cb, C.c_void_p), None),
                 "mg_dist_set_exchange_plugin")

    def cycle(self, b_loc, x_loc, x_is_zero: bool):
        self.H.be.synchronize()       # the library enqueues on its own streams: whatever torch still has in flight for
        D._check(self.lib, self.lib.mg_dist_cycle_dev_FP64(self.handle, D._ptr(b_loc), D._ptr(x_loc), self.H.levels[0].n_own,
                                                           1 if x_is_zero else 0), "mg_dist_cycle_dev")
        return x_loc

    def solve(self, b_loc, x_loc, tol: float, maxIter: int):
        import ctypes as C
        iters = C.c_longlong(0)
        resvec = np.zeros(int(maxIter) + 1)
        self.H.be.synchronize()       # b_loc / x_loc (a fill, a copy) must have landed before the library's streams read them
        D._check(self.lib, self.lib.mg_dist_solve_dev_FP64(self.handle, D._ptr(b_loc), D._ptr(x_loc), self.H.levels[0].n_own,
                                                           float(tol), int(maxIter), C.byref(iters), D._f64(resvec)),
                 "mg_dist_solve_dev")
        return int(iters.value), resvec[: iters.value + 1]

    def comm_count(self) -> int:
        """Ranks of the sequencer's RCCL communicator as the library reports them (0: plug-in transport)."""
        import ctypes as C
        c = C.c_longlong(0)
        D._check(self.lib, self.lib.mg_dist_comm_count(self.handle, C.byref(c)), "mg_dist_comm_count")
        return int(c.value)

    def close(self):
        """Destroy the native sequencer; the tail hierarchy goes back to the Python sequencer's stream first."""
        if self.handle:
            self.lib.mg_dist_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
