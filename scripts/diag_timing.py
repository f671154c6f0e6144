import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multigrid_jl_amd as mg
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A, mesh = mg.poisson_shifted([cells] * 3)
lv = {32: 3, 64: 4, 128: 5, 256: 6}[cells]
p = mg.getMGparam(np.float64, np.int64, lv, 8, 10, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
mg.MGsetup(A, mesh, p, 1)
h = mg.to_device(p)
b = torch.from_numpy(mg.seeded_rhs(A)).cuda()
x = torch.zeros_like(b)
torch.cuda.synchronize()
def t_solve(K, tag):
    x.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter(); h.solve_dev(b, x, 0.0, K); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{tag}: solve K={K}: {dt/K*1e3:.3f} ms/step", flush=True)
for i in range(3): t_solve(10, f"plain{i}")
h.profile_enable(True); t_solve(10, "events"); h.profile_enable(False)
t_solve(10, "plain-after")
x.zero_(); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10): h.cycle_dev(b, x, 1 if i == 0 else 0)
print(f"cycle_dev x10: {(time.perf_counter()-t0)/10*1e3:.3f} ms/cycle", flush=True)
for lvl in (1, 2, 3):
    for k in (mg.device.MG_K_SMOOTH, mg.device.MG_K_RESIDUAL, mg.device.MG_K_PROLONG, mg.device.MG_K_RESTRICT):
        ms, bts = h.time_op(lvl, k, 20)
        print(f"time_op L{lvl} {mg.device.KERNEL_NAMES[k]}: {ms:.4f} ms  {bts/ms/1e6:.0f} GB/s", flush=True)
