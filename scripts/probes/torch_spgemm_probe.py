"""Does torch.sparse.mm(CSR, CSR) run on this ROCm build, how fast, and is its pattern the structural one?
usage: python3 scripts/probes/torch_spgemm_probe.py [n] [density]"""
import sys, time
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, ".")
from multigrid_jl_amd import hostlib as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dens = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
A = sp.random(n, n, density=dens, random_state=1, format="csr", dtype=np.float64)
P = sp.random(n, n // 5, density=dens / 2, random_state=2, format="csr", dtype=np.float64)
t0 = time.perf_counter(); C0 = H.spgemm(A, P); t_cpu = time.perf_counter() - t0
prod = int(np.diff(P.indptr).astype(np.int64)[A.indices].sum())
print(f"host: {t_cpu:.2f} s, nnzC {C0.nnz}, products {prod/1e9:.2f} G", flush=True)
def tocsr(M):
    return torch.sparse_csr_tensor(torch.from_numpy(M.indptr.astype(np.int64)), torch.from_numpy(M.indices.astype(np.int64)), torch.from_numpy(M.data), size=M.shape).cuda()
try:
    At, Pt = tocsr(A), tocsr(P)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter(); Ct = torch.sparse.mm(At, Pt); torch.cuda.synchronize(); t_gpu = time.perf_counter() - t0
        print(f"torch.sparse.mm CSR x CSR: {t_gpu:.3f} s, layout {Ct.layout}, nnz {Ct._nnz()}", flush=True)
    ci = Ct.col_indices().cpu().numpy(); cp = Ct.crow_indices().cpu().numpy(); cv = Ct.values().cpu().numpy()
    C1 = sp.csr_matrix((cv, ci, cp), shape=C0.shape); C1.sort_indices()
    print("pattern equal:", C1.nnz == C0.nnz and np.array_equal(C1.indices, C0.indices), "max rel diff", float(np.abs(C1.data - C0.data).max() / np.abs(C0.data).max()) if C1.nnz == C0.nnz else None)
except Exception as e:
    print("torch.sparse.mm CSR x CSR failed:", type(e).__name__, e)
