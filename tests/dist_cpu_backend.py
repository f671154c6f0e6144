"""TEST-ONLY local-compute backend for multigrid.jl_amd/distributed.py: the oracle's arithmetic on CPU
tensors, so that the partition / halo / schedule logic of the multi-GPU path can be exercised with
``gloo`` on a machine without GPUs.  The product's backend is ``HipBackend`` (HIP kernels only)."""
import numpy as np
import torch

from multigrid_jl_amd import device as D
from oracle import mg_oracle as orc


class CpuCheckerBackend:
    def __init__(self):
        self.torch = torch

    def zeros(self, *shape):
        return torch.zeros(*shape, dtype=torch.float64)

    def from_numpy(self, a):
        return torch.from_numpy(np.array(a, dtype=np.float64, order="C"))

    def index_tensor(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64))

    def index_select(self, src, idx, out):
        torch.index_select(src, 0, idx, out=out)

    supports_box = True

    def operator(self, M, box=None, regular_cols=None, coarse_box=None):
        M = M.tocsr()
        if coarse_box is not None or box is None:   # grid form of P / split R: every row is computed, the hints only pick kernels
            return M
        if box is not None:            # box form: square [owned box | halo], only the owned rows are ever computed
            M = M[: int(regular_cols), :].tocsr()
            M._mg_box = True
        return M

    def apply(self, op, kernel, x, y, b=None, d=None, alpha=1.0, beta=0.0, nrhs=1, row_offset=0, phase=0):
        if getattr(op, "_mg_box", False) and phase == 1:
            return                     # (the checker computes every row once the halo has landed: phase 2)
        nr, nc = op.shape
        xn = x.numpy()[:nc]
        Ax = op @ xn
        yn = y.numpy()
        o = row_offset
        if kernel in (D.MG_K_SPMV, D.MG_K_RESTRICT, D.MG_K_PROLONG):
            yn[o:o + nr] = alpha * Ax + (beta * yn[o:o + nr] if beta != 0.0 else 0.0)
        elif kernel == D.MG_K_RESIDUAL:
            yn[o:o + nr] = b.numpy()[o:o + nr] - Ax
        elif kernel == D.MG_K_SMOOTH:
            dn = d.numpy()[o:o + nr]
            dd = dn if xn.ndim == 1 else dn[:, None]
            yn[o:o + nr] = x.numpy()[o:o + nr] + dd * (b.numpy()[o:o + nr] - Ax)
        else:
            raise ValueError(kernel)

    def dscale(self, d, b, x, n, nrhs):
        dn = d.numpy()
        x.numpy()[:n] = (dn if nrhs == 1 else dn[:, None]) * b.numpy()[:n]

    def xpdr(self, x, d, r, xout, n, nrhs):
        dn = d.numpy()
        xout.numpy()[:n] = x.numpy()[:n] + (dn if nrhs == 1 else dn[:, None]) * r.numpy()[:n]

    def sumsq(self, x, length):
        v = x.numpy().reshape(-1)[:length]
        return torch.tensor([float(np.dot(v, v))], dtype=torch.float64)

    def tail(self, sub, nrhs):
        return _OracleTail(sub)

    def synchronize(self):
        pass


class _OracleTail:
    def __init__(self, sub):
        self.sub = sub

    def cycle(self, b, x, x_zero, ctype):
        xn = x.numpy()
        if x_zero:
            xn[...] = 0.0
        res = orc.recursiveCycle(self.sub, b.numpy(), xn, 1, None, ctype)
        if res is not xn:
            xn[...] = res
