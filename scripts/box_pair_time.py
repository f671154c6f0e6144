"""Cost of the sharded fine level's sweep + residual pair on ONE rank's box with real faces (no exchange: kernels only):
the two-stage pass over the box (t away from the rows that read the halo, r two layers in) and the three list kernels that
finish the face layers.  usage: python3 scripts/box_pair_time.py [cells]"""
import os, sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multigrid_jl_amd as mg
from multigrid_jl_amd import device as D, distributed as dd

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A, mesh = mg.poisson_shifted([cells] * 3)
nodes = mesh.n + 1
for doms in ([1, 1, 2], [1, 2, 1], [2, 1, 1], [2, 2, 2]):
    nr = int(np.prod(doms))
    part = dd.Partition(dd.box_owner(nodes, doms), nr)
    Ml, plan = dd.localize(A, part, part, 0)
    n_own, n_tot = int(part.counts[0]), Ml.shape[1]
    Asq = sp.csr_matrix((Ml.data, Ml.indices, np.concatenate([Ml.indptr, np.full(n_tot - Ml.shape[0], Ml.indptr[-1], dtype=Ml.indptr.dtype)])),
                        shape=(n_tot, n_tot))
    own = dd.box_owner(nodes, doms).reshape(nodes[2], nodes[1], nodes[0]) == 0
    box = (int(own.any(axis=(0, 1)).sum()), int(own.any(axis=(0, 2)).sum()), int(own.any(axis=(1, 2)).sum()))
    op = D.DeviceOperator(Asq, 0, box=box, regular_cols=n_own)
    rng = np.random.default_rng(0)
    d = torch.from_numpy(0.8 / Ml.diagonal()[:n_own] if Ml.shape[0] == n_own else 0.8 / Asq.diagonal()[:n_own]).cuda()
    op.bind_relax(d, n_own)
    x, b = torch.from_numpy(rng.standard_normal(n_tot)).cuda(), torch.from_numpy(rng.standard_normal(n_tot)).cuda()
    t, r = torch.zeros(n_tot, dtype=torch.float64).cuda(), torch.zeros(n_tot, dtype=torch.float64).cuda()
    yes, l1, l2 = op.can_sweep_residual(x, d)
    s = torch.cuda.current_stream().cuda_stream

    def timed(fn, reps=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    out = [f"boxes {doms}: box {box} = {n_own} rows, halo {n_tot - n_own}, fused pass available {yes}, list 1 {l1} rows, list 2 {l2} rows"]
    if yes:
        out.append(f"pass {timed(lambda: op.sweep_residual(x, b, d, t, r, stream=s)):.1f} us")
        out.append(f"t on list 1 {timed(lambda: op.apply_list(1, D.MG_K_SMOOTH, x, t, b, d, stream=s)):.1f} us")
        out.append(f"r on list 2 {timed(lambda: op.apply_list(2, D.MG_K_RESIDUAL, t, r, b, d, stream=s)):.1f} us")
        out.append(f"r on list 1 {timed(lambda: op.apply_list(1, D.MG_K_RESIDUAL, t, r, b, d, stream=s)):.1f} us")
    out.append(f"two launches: sweep phase 1 {timed(lambda: op.apply(D.MG_K_SMOOTH, x, t, b=b, d=d, stream=s, phase=1)):.1f} + phase 2 "
               f"{timed(lambda: op.apply(D.MG_K_SMOOTH, x, t, b=b, d=d, stream=s, phase=2)):.1f} us, residual phase 1 "
               f"{timed(lambda: op.apply(D.MG_K_RESIDUAL, t, r, b=b, stream=s, phase=1)):.1f} + phase 2 "
               f"{timed(lambda: op.apply(D.MG_K_RESIDUAL, t, r, b=b, stream=s, phase=2)):.1f} us")
    print("  ".join(out), flush=True)
    op.close()
