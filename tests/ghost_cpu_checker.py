"""TEST-ONLY numpy sequencer of the ghost-layer form of the sharded cycle (multigrid.jl_amd/ghost_dist.py).

It runs one rank's part of the hierarchy - extended boxes, exchange plans - with the schedule and the validity
rules of the library (csrc/mg_schedule.inc cycle_level / solve_dev with the gh_* calls of csrc/mg_ghost.inc): where an
operation asks for ghost layers the bookkeeping says are gone, all layers are exchanged (``torch.distributed``, gloo).
Every result is POISONED with NaN beyond the depth the rules claim for it, so a rule that claims too much shows up as a
NaN in an owned row.  The replicated levels run the oracle's recursiveCycle.  The product never imports this file.
"""
import numpy as np
import torch
import torch.distributed as dist

from multigrid_jl_amd.ghost_dist import FULL
from oracle import mg_oracle as orc


class GhostCpuSequencer:
    def __init__(self, G, group=None, poison=True):
        self.G, self.p, self.a = G, G.param, G.a
        self.group = group
        self.poison = poison
        self.size = G.size
        self.depth = [dict() for _ in range(self.a)]
        self.exchanges = 0
        from multigrid_jl_amd.distributed import _sub_hierarchy
        self.tail = _sub_hierarchy(self.p, self.a)
        self.r = [np.zeros(L.n) for L in G.levels]

    # ---- bookkeeping (gh_depth / gh_set / gh_dec / gh_need) ---------------------------------------------------------
    def active(self, l):
        return l < self.a and self.size > 1

    def dep(self, l, v):
        if not self.active(l):
            return FULL
        return self.depth[l].get(id(v), 0)

    def setd(self, l, v, d):
        if not self.active(l):
            return
        g = self.G.levels[l].gmin
        assert d >= 0, f"level {l}: an operation consumed more ghost layers than its input had (depth {d})"   # (gh_set: MG_ERR_STATE)
        d = FULL if d >= g else d
        self.depth[l][id(v)] = d
        if self.poison and d < FULL:
            v[~self.G.levels[l].depth_mask(d)] = np.nan

    def dec(self, l, d):
        if not self.active(l):
            return FULL
        return self.G.levels[l].gmin - 1 if d >= FULL else d - 1

    def exchange(self, l, v):
        L = self.G.levels[l]
        own = L.own_mask()
        assert not np.isnan(v[own]).any(), f"level {l}: an owned row is invalid before an exchange (a validity rule claims too much)"
        send = torch.from_numpy(np.ascontiguousarray(v[L.send_idx]))
        recv = torch.zeros(int(L.recv_idx.size), dtype=torch.float64)
        dist.all_to_all_single(recv, send, L.recv_splits, L.send_splits, group=self.group)
        v[L.recv_idx] = recv.numpy()
        self.exchanges += 1
        self.depth[l][id(v)] = FULL
        assert not np.isnan(v).any()

    def need(self, l, v, want):
        if not self.active(l):
            return
        if self.dep(l, v) < min(want, self.G.levels[l].gmin):
            self.exchange(l, v)

    # ---- operations -------------------------------------------------------------------------------------------------------
    def sweep(self, l, b, x):
        self.need(l, x, 1)
        self.need(l, b, 0)
        out = x + self.p.relaxPrecs[l] * (b - self.p.As[l] @ x)
        self.setd(l, out, min(self.dec(l, self.dep(l, x)), self.dep(l, b)))
        return out

    def residual(self, l, b, x):
        self.need(l, x, 1)
        self.need(l, b, 0)
        out = b - self.p.As[l] @ x
        self.setd(l, out, min(self.dec(l, self.dep(l, x)), self.dep(l, b)))
        return out

    def norm_own(self, v):
        L = self.G.levels[0]
        own = L.own_mask()
        s = torch.tensor([float(np.dot(v[own], v[own]))], dtype=torch.float64)
        if self.size > 1:
            dist.all_reduce(s, group=self.group)
        return float(s.item()) ** 0.5

    # ---- cycle_level ------------------------------------------------------------------------------------------------------
    def cycle_level(self, l, b, cur, x_zero, ctype, r_valid=False, xnext=None, defer_post=False, pre_done=False):
        p, a = self.p, self.a
        if l >= a:       # replicated levels: the oracle's cycle on the sub-hierarchy (identical on every rank)
            sub = self.tail
            if l > a:
                from multigrid_jl_amd.distributed import _sub_hierarchy
                sub = _sub_hierarchy(p, l)
            x = np.zeros_like(b) if x_zero else cur.copy()
            res = orc.recursiveCycle(sub, b, x, 1, None, ctype)
            return res
        d = p.relaxPrecs[l]
        npre, npost = max(1, int(p.relaxPre(l + 1))), max(1, int(p.relaxPost(l + 1)))
        from_zero = False
        if pre_done:
            npre = 0
        elif x_zero:
            self.need(l, b, 0)
            from_zero = npre == 2
            if not from_zero:
                cur = d * b
                self.setd(l, cur, self.dep(l, b))
            npre -= 1
        elif r_valid:
            if xnext is None:
                r = self.r[l]
                nxt = cur + d * r
                self.setd(l, nxt, min(self.dep(l, cur), self.dep(l, r)))
                cur = nxt
            else:
                cur = xnext
            npre -= 1
        fuse_pre = npre >= 1
        for _ in range(npre - (1 if fuse_pre else 0)):
            cur = self.sweep(l, b, cur)
        if pre_done:
            r = self.r[l]
        elif fuse_pre:
            if from_zero:
                self.need(l, b, 2)
                x1 = d * b
                dx1 = self.dep(l, b)
            else:
                self.need(l, cur, 2)
                self.need(l, b, 1)
                x1, dx1 = cur, self.dep(l, cur)
            t = x1 + d * (b - p.As[l] @ x1)
            dt = min(self.dec(l, dx1), self.dep(l, b))
            self.setd(l, t, dt)
            r = b - p.As[l] @ t
            self.setd(l, r, min(self.dec(l, dt), self.dep(l, b)))
            cur = t
        else:
            r = self.residual(l, b, cur)
        self.r[l] = r
        if self.active(l):
            inside = npost - (1 if defer_post else 0)
            self.need(l, cur, inside + (5 if defer_post else 0))      # (gh_prefetch: same decision, landing later)
            self.need(l, r, 1)
        bc = p.Rs[l] @ r
        if l + 1 < a:
            if self.active(l + 1):
                dr = min(self.dep(l, r), self.G.levels[l].gmin)
                self.setd(l + 1, bc, (dr - 1) // 2)
                self.need(l + 1, bc, FULL)
        elif self.size > 1:
            own_rows = np.diff(p.Rs[l].indptr) > 0
            assert not np.isnan(bc[own_rows]).any(), "a row of the first replicated level saw an invalid residual"
            bc = np.where(own_rows, bc, 0.0)
            t_ = torch.from_numpy(bc.copy())
            dist.all_reduce(t_, group=self.group)
            bc = t_.numpy()
        xc = self.cycle_level(l + 1, bc, None, True, ctype)
        if l + 1 < len(p.As) - 1 and ctype in ("W", "F"):
            xc = self.cycle_level(l + 1, bc, xc, False, "W" if ctype == "W" else "V")
        if l + 1 < a and self.active(l + 1):
            self.need(l + 1, xc, FULL)
        dcur = self.dep(l, cur)
        cur = cur + p.Ps[l] @ xc
        self.setd(l, cur, dcur)
        nlast = npost - (1 if defer_post else 0)
        for _ in range(nlast):
            cur = self.sweep(l, b, cur)
        return cur

    # ---- public -------------------------------------------------------------------------------------------------------------
    def _begin(self, b):
        for dct in self.depth:
            dct.clear()
        b = np.array(b, dtype=np.float64)
        if self.active(0):
            self.depth[0][id(b)] = 0
            if self.poison:
                b[~self.G.levels[0].own_mask()] = np.nan
            self.need(0, b, FULL)
        return b

    def cycle(self, b, x, x_zero):
        b = self._begin(b)
        x = np.array(x, dtype=np.float64)
        if self.active(0) and not x_zero:
            self.setd(0, x, 0)
        return self.cycle_level(0, b, x, x_zero, self.p.cycleType)

    def solve(self, b, x, tol, maxIter, fused4=True):
        """solveMG as solve_dev runs it on a sharded hierarchy: the last post-smoothing sweep deferred into the fused pass
        behind the cycle - the four-stage pass while the loop goes on by count (fused4), the sweep + residual pair on the
        last step.  Returns (iters, resvec, x)."""
        p = self.p
        b = self._begin(b)
        cur = np.array(x, dtype=np.float64)
        if self.active(0):
            self.setd(0, cur, 0)
        d, A = p.relaxPrecs[0], p.As[0]
        xn = self.norm_own(cur)
        x_zero = xn == 0.0
        if x_zero:
            res0 = self.norm_own(b)
        else:
            self.r[0] = self.residual(0, b, cur)
            res0 = self.norm_own(self.r[0])
        resvec = [res0]
        npre = max(1, int(p.relaxPre(1)))
        # (four_stage_serves: the pass consumes four ghost layers of x in one kernel - the fine level must have them)
        can4 = fused4 and npre == 2 and (not self.active(0) or self.G.levels[0].gmin >= 4)
        pre_done, xnext, it = False, None, 0
        for count in range(1, maxIter + 1):
            cur = self.cycle_level(0, b, cur, x_zero, p.cycleType, r_valid=(count > 1 or not x_zero), xnext=xnext,
                                   defer_post=True, pre_done=pre_done)
            x_zero, pre_done, xnext = False, False, None
            dep, dec = (lambda v: self.dep(0, v)), (lambda v: self.dec(0, v))
            if can4 and count < maxIter:
                self.need(0, cur, 5)
                self.need(0, b, 4)
                db = dep(b)
                t = cur + d * (b - A @ cur)
                dt = min(dec(dep(cur)), db)
                self.setd(0, t, dt)
                r = b - A @ t
                dr = min(dec(dt), db)
                self.setd(0, r, dr)
                res = self.norm_own(r)
                xn_ = t + d * r
                dn = min(dt, dr)
                self.setd(0, xn_, dn)
                tp = xn_ + d * (b - A @ xn_)
                dp = min(dec(dn), db)
                self.setd(0, tp, dp)
                rp = b - A @ tp
                self.setd(0, rp, min(dec(dp), db))
                it += 1
                resvec.append(res)
                if res / res0 < tol:
                    cur = t                      # (the library re-creates it from the pass's input: same values)
                    break
                cur, self.r[0], pre_done = tp, rp, True
                continue
            self.need(0, cur, 2)
            self.need(0, b, 1)
            t = cur + d * (b - A @ cur)
            dt = min(dec(dep(cur)), dep(b))
            self.setd(0, t, dt)
            r = b - A @ t
            dr = min(dec(dt), dep(b))
            self.setd(0, r, dr)
            res = self.norm_own(r)
            it += 1
            resvec.append(res)
            self.r[0] = r
            if count < maxIter:
                xnext = t + d * r
                self.setd(0, xnext, min(dt, dr))
            cur = t
            if res / res0 < tol:
                break
        return it, np.array(resvec), cur
