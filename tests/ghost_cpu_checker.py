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
        got = FULL
        if l + 1 < a and self.active(l + 1):
            # a correction valid on k >= 1 layers gives a fine vector valid on 2k - 1; it travels only where that falls short of
            # what the post-smoothing (and the fused pass behind a deferred last sweep) consume
            want = npost - (1 if defer_post else 0) + (5 if defer_post else 0)
            reach = lambda: FULL if self.dep(l + 1, xc) >= FULL else 2 * self.dep(l + 1, xc) - 1
            if reach() < min(want, self.G.levels[l].gmin):
                self.need(l + 1, xc, FULL)
            got = reach()
        dcur = self.dep(l, cur)
        cur = cur + p.Ps[l] @ xc
        self.setd(l, cur, min(dcur, got))
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


# ---- sharded MG-preconditioned Krylov (mg_krylov.inc in the ghost-layer form): the oracle's drivers with the three sharded pieces -------
class GhostKrylovCpu:
    """KrylovMethods' cg / bicgstb / fgmres as oracle/mg_oracle.py restates them, on ONE rank's extended fine box the way the library
    runs them on a ghost-attached handle: a dot is the sum over the owned rows of all ranks, a product with A follows one exchange
    of its input's ghost layers, M is the sharded cycle from x = 0.  Counts exchanges (through the sequencer) and all-reduces."""

    def __init__(self, S: GhostCpuSequencer):
        self.S, self.G = S, S.G
        self.own = self.G.levels[0].own_mask()
        self.A = S.p.As[0]
        self.allreduces = 0

    def dot(self, x, y):
        s = torch.tensor([float(np.dot(x[self.own], y[self.own]))], dtype=torch.float64)
        if self.S.size > 1:
            dist.all_reduce(s, group=self.S.group)
        self.allreduces += 1
        return float(s.item())

    def norm(self, x):
        return self.dot(x, x) ** 0.5

    def Afun(self, v):
        v = np.array(v, dtype=np.float64)
        if self.S.active(0):
            if self.S.poison:
                v[~self.own] = np.nan               # (only the owned rows of a Krylov vector are meaningful)
            self.S.depth[0][id(v)] = 0
            self.S.need(0, v, 1)
        return self.A @ v

    def M(self, v):
        return self.S.cycle(v, np.zeros_like(v), True)

    def cg(self, b, x, tol, maxIter):
        nr0 = self.norm(b)
        r = b - self.Afun(x)
        z = self.M(r)
        p = z.copy()
        resvec, flag, it = [], -1, 0
        gamma = self.dot(r, z)
        for it in range(1, maxIter + 1):
            Ap = self.Afun(p)
            alpha = gamma / self.dot(p, Ap)
            if np.isinf(alpha) or alpha < 0:
                flag = -2
                break
            x = x + alpha * p
            r = r - alpha * Ap
            rn = self.norm(r) / nr0
            resvec.append(rn)
            if rn <= tol:
                flag = 0
                break
            z = self.M(r)
            zr = self.dot(z, r)
            beta, gamma = zr / gamma, zr
            p = z + beta * p
        return x, flag, it, np.array(resvec)

    def bicgstb(self, b, x, tol, maxIter):
        bn = self.norm(b)
        r = b - self.Afun(x)
        resvec = [self.norm(r) / bn]
        rtld = r.copy()
        omega, alpha, rho1, flag, it = 1.0, 0.0, 0.0, -1, 0
        p = v = None
        for it in range(1, maxIter + 1):
            rho = self.dot(rtld, r)
            if rho == 0:
                flag = -2
                break
            p = r.copy() if it == 1 else r + ((rho / rho1) * (alpha / omega)) * (p - omega * v)
            phat = self.M(p)
            v = self.Afun(phat)
            alpha = rho / self.dot(rtld, v)
            s = r - alpha * v
            sn = self.norm(s) / bn
            resvec.append(sn)
            if sn < tol:
                x = x + alpha * phat
                flag = -3
                break
            shat = self.M(s)
            t = self.Afun(shat)
            ts, tt = self.dot(t, s), self.dot(t, t)
            self.allreduces -= 1                       # (the library sends the two scalars of omega in ONE all-reduce)
            omega = ts / tt
            x = x + alpha * phat + omega * shat
            r = s - omega * t
            err = self.norm(r) / bn
            resvec.append(err)
            if err <= tol:
                flag = 0
                break
            if omega == 0:
                flag = -2
                break
            rho1 = rho
        return x, flag, it, np.array(resvec)

    def fgmres(self, b, x, m, tol, maxIter):
        n = b.size
        bn = self.norm(b)
        r = b - self.Afun(x)
        rn = self.norm(r)
        resvec, flag, total = [], -1, 0
        for it in range(1, maxIter + 1):
            V, Z, H = np.zeros((n, m + 1)), np.zeros((n, m)), np.zeros((m + 1, m))
            cs, sn, s = np.zeros(m), np.zeros(m), np.zeros(m + 1)
            V[:, 0] = r / rn
            s[0] = rn
            used = 0
            for i in range(m):
                Z[:, i] = self.M(V[:, i])
                w = self.Afun(Z[:, i])
                for k in range(i + 1):
                    H[k, i] = self.dot(w, V[:, k])
                    w = w - H[k, i] * V[:, k]
                H[i + 1, i] = self.norm(w)
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
            r = b - self.Afun(x)
            rn = self.norm(r)
            if rn / bn <= tol:
                flag = 0
                break
        return x, flag, total, np.array(resvec)
