#!/usr/bin/env python3
"""bench.py - V-cycle DoF-updates/s + achieved HBM GB/s (BASELINE.json metric) on MI355X.

A "step" is one iteration of the reference's solveMG loop (src/Multigrid/SolveFuncs.jl:24-37): one
recursiveCycle from the finest level + the residual SpMV + the Frobenius norm.  The timed region is
exactly K steps (mg_solve_dev_FP64 with maxIter=K, tol=0, x0=0 as SURVEY.md 8d prescribes) with the
hierarchy, b and x resident in HBM.

Workloads (BASELINE.json configs):
  c2 (default)  3-D 7-pt Poisson 256^3 cells, GMG V(2,1) damped Jacobi w=0.8, 6 levels, fp64, nrhs=1
  c5            same operator, 16 right-hand sides (block SpMM path)
  c3            SA-AMG on anisotropic diffusion (edge weights 16:4:1 x log-normal sigma), general CSR, V(1,1) SPAI
  c1            32^3 cells (CPU-plumbing size; parity case, not a bench line)
Use --cells N to shrink the grid for quick checks (the JSON then names the reduced workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c5"])
    ap.add_argument("--cells", type=int, default=0, help="override cells per dimension")
    ap.add_argument("--levels", type=int, default=0)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = 256^3 cells PER GPU (N=8 is BASELINE configs[3], 512^3), sharded host setup; "
                         "strong = the same 256^3 grid cut into N boxes (every rank builds the global hierarchy)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sharded-path", action="store_true",
                    help="diagnostic: run the N>1 (Python-sequenced, sharded) weak-scaling path with world size 1")
    ap.add_argument("--prewarm", type=float, default=0.5, help="seconds of untimed kernel launches before warm-up")
    ap.add_argument("--cpu-cycles", type=int, default=0, help="cycles of the CPU baseline sample (0 = auto)")
    return ap.parse_args()


def levels_for(cells):
    """Coarsest grid 9^3 nodes: 32 -> 3, 256 -> 6, 512 -> 7."""
    lv = 1
    c = cells
    while c > 8 and c % 2 == 0:
        c //= 2
        lv += 1
    return lv


def main():
    args = parse()
    import torch
    import multigrid_jl_amd as mg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # MG_BENCH_SHARE_GPU=1: functional check of the N>1 path on a ONE-GPU box (all ranks on cuda:0, gloo with
    # host staging instead of RCCL) - the numbers of such a run mean nothing.
    share = os.environ.get("MG_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    if world == 1 and args.force_sharded_path:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the multigrid cycle has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    cells = args.cells or {"c1": 32, "c2": 256, "c3": 256, "c5": 256}[args.workload]
    nrhs = 16 if args.workload == "c5" else 1
    levels = args.levels or levels_for(cells)
    K, W = args.steps, args.warmup

    if world > 1:
        os.environ.setdefault("MG_HOST_THREADS", str(max(1, (os.cpu_count() or 8) // world)))
    if (world > 1 or args.force_sharded_path) and args.scaling == "weak":
        if args.workload != "c2":
            raise SystemExit("multi-GPU weak scaling is defined for the c2/c4 Poisson workload")
        return bench_weak(args, mg, torch, dist, cells, K, W, rank, world, local_rank)

    # ---- host setup (CPU, as in the reference) ---------------------------------------------------
    t0 = time.perf_counter()
    if args.workload == "c3":
        A, mesh = mg.anisotropic_divsiggrad([cells] * 3, weights=(1.0, 0.25, 0.0625))
        t_op = time.perf_counter() - t0
        p = mg.getMGparam(np.float64, np.int64, args.levels or 14, os.cpu_count() or 8, K, 0.0, "SPAI", 1.0, 1, 1, "V",
                          "Julia", 0.4, 0.0)
        t0 = time.perf_counter()
        mg.SA_AMGsetup(A, p, True, nrhs)
        desc = (f"SA-AMG (theta=0.4, V(1,1) SPAI w=1) on anisotropic diffusion {cells}^3 cells, edge weights 16:4:1 x "
                f"log-normal sigma, general CSR")
    else:
        A, mesh = mg.poisson_shifted([cells] * 3)
        t_op = time.perf_counter() - t0
        p = mg.getMGparam(np.float64, np.int64, levels, os.cpu_count() or 8, K, 0.0, "Jac", 0.8, 2, 1, "V",
                          "NoMUMPS", 0.5, 0.0, "FullWeighting")
        t0 = time.perf_counter()
        mg.MGsetup(A, mesh, p, nrhs)
        desc = f"3D 7-pt Poisson {cells}^3 cells, GMG V(2,1) damped-Jacobi w=0.8"
    t_setup = time.perf_counter() - t0
    b_host = mg.seeded_rhs(A, nrhs)
    if world > 1:
        if args.workload == "c3":
            raise SystemExit("c3 is a single-GPU workload (BASELINE.json configs[2])")
        return bench_distributed(args, mg, torch, dist, A, mesh, p, b_host, cells, nrhs, K, W, rank, world,
                                 local_rank, t_op, t_setup)
    t0 = time.perf_counter()
    h = mg.to_device(p, device_id=local_rank)
    t_upload = time.perf_counter() - t0
    n = A.shape[0]
    log(f"[rank {rank}] {cells}^3 cells, N={n}, nnz={A.nnz}, levels={p.levels}, nrhs={nrhs}: operator {t_op:.1f}s, "
        f"MGsetup {t_setup:.1f}s, upload {t_upload:.1f}s, HBM {h.device_bytes() / 1e9:.2f} GB")

    # device-resident b / x (row-major [n][nrhs] is the library's block layout)
    b = torch.from_numpy(np.ascontiguousarray(b_host)).to(dev)
    x = torch.zeros_like(b)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- pre-warm (untimed): ~0.5 s of the fine-level kernels so that clocks/power state and the HIP
    # runtime's lazily grown pools settle before anything is timed (a one-off 70-80 ms stall was observed
    # at a random point in the first tens of ms of GPU activity of a fresh process) ------------------
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.prewarm:
        h.time_op(1, mg.device.MG_K_RESIDUAL, 50)
    # ---- W untimed warm-up steps, then exactly K timed steps ----------------------------------------
    if W > 0:
        h.solve_dev(b, x, 0.0, W)
    x.zero_()
    barrier()
    t0 = time.perf_counter()
    iters, resvec = h.solve_dev(b, x, 0.0, K)
    barrier()
    dt = time.perf_counter() - t0
    assert iters == K
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    relres = float(resvec[-1] / resvec[0])
    dof_per_s = world * n * nrhs * K / dt          # replicas when world > 1

    # ---- per-kernel HIP-event accounting over the same K steps (separate, instrumented pass) -----
    h.profile_reset()
    h.profile_enable(True)
    x.zero_()
    torch.cuda.synchronize()
    h.solve_dev(b, x, 0.0, K)
    h.profile_enable(False)
    prof = h.profile()
    tot_ms = sum(v[0] for v in prof.values())
    # dominant kernel: the fine-level fused smoother/residual SpMV (csr_stream_spmv/spmm on As[1])
    dom = {}
    for name in ("smooth", "residual"):
        ms, cnt, bts = prof[(1, name)]
        dom[name] = {"avg_ms": ms / cnt, "launches": cnt, "bytes": bts, "gbs": bts / (ms / cnt) / 1e6}
    ms_s, cnt_s, bts_s = prof[(1, "smooth")]
    achieved = bts_s / (ms_s / cnt_s) / 1e6        # GB/s, fine level
    step_bytes = sum(v[2] * v[1] for v in prof.values()) / K            # algorithmic bytes per step (all launches)
    # ---- what the kernel serving the fine sweep really is, and what it really streams -----------------
    # The hierarchy chooses a device format per operator at upload (lossless):
    #   row classes   csr_rowclass_spmv<MODE>          6 B/row + dictionary; NO matrix stream (constant-coefficient
    #                                                  stencils, their Galerkin operators, full-weighting P/R)
    #   pattern-coded csr_pattern_spmv<MODE,NT,DLDS>   8 B/nnz + row descriptors
    #   plain CSR     csr_stream_spmv<MODE,NT>         12 B/nnz + row pointers   (csr_stream_spmm at nrhs > 1)
    # `achieved`/`frac` keep the prescribed definition (ALGORITHMIC CSR bytes / kernel time); `streamed_*` is
    # what the kernel in use moves.  With row classes frac can exceed 1: that is traffic AVOIDED, not bandwidth.
    # The roofline object aggregates the fine level's kernel SYMBOL over every level that symbol serves (what
    # `rocprofv3 --stats` averages per kernel name).
    rcs, fmts, sym, fmt_of = {}, {}, {}, {}
    for l in range(1, len(p.As)):
        rcs[l] = h.operator_rowclasses(l, mg.device.MG_OP_A)
        fmts[l] = h.operator_format(l, mg.device.MG_OP_A)
        ntl = "true" if 12.0 * p.As[l - 1].nnz > 128.0e6 else "false"
        if nrhs > 1:
            sym[l], fmt_of[l] = f"mgk::csr_stream_spmm<2, {ntl}>", "plain CSR (block right-hand sides)"
        elif rcs[l][0] > 0:
            var, nexc = h.operator_kernel_info(l, mg.device.MG_OP_A)
            inl = "true" if 0 < nexc <= 256 else "false"   # a short list of exception rows is handled in-kernel
            sym[l] = {0: "mgk::csr_rowclass_spmv<2, %s>", 1: "mgk::csr_rowclass_window_spmv<2, %s>",
                      2: "mgk::csr_rowclass_tile_spmv<2, %s>"}[var] % inl
            fmt_of[l] = "row classes"
        elif fmts[l][0] > 0:
            dl = "true" if (fmts[l][1] <= 1024 and fmts[l][0] < 1024) else "false"
            sym[l], fmt_of[l] = f"mgk::csr_pattern_spmv<2, {ntl}, {dl}>", "pattern-coded"
        else:
            sym[l], fmt_of[l] = f"mgk::csr_stream_spmv<2, {ntl}>", "plain CSR"
    kname, fmt_name = sym[1], fmt_of[1]
    lv = [l for l in sym if sym[l] == kname and (l, "smooth") in prof]
    sm_all = [prof[(l, "smooth")] for l in lv]
    all_ms = sum(v[0] for v in sm_all)
    all_cnt = sum(v[1] for v in sm_all)
    all_bytes = sum(v[2] * v[1] for v in sm_all)
    n1, nnz1 = p.As[0].shape[0], p.As[0].nnz
    streamed = None
    if nrhs == 1:
        # matrix-side bytes of the kernel in use + the vectors of a SMOOTH launch (x gathered once, b, d, x' written)
        # vectors of a SMOOTH launch: x gathered once, b, x' written (+ d unless it comes from the class dictionary)
        per_level = {l: rcs[l][2] + (24.0 if (rcs[l][0] > 0 and h.operator_rowclass_flags(l, mg.device.MG_OP_A)[1]) else 32.0) * p.As[l - 1].shape[0] for l in lv}
        cnts = {l: prof[(l, "smooth")][1] for l in lv if (l, "smooth") in prof}
        tot = sum(cnts.values())
        streamed = sum(per_level[l] * cnts[l] for l in cnts) / tot if tot else None
        fine_streamed = per_level.get(1)
    ach_sym = all_bytes / all_ms / 1e6
    avg_ms = all_ms / all_cnt
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tfile):
        try:   # only a PMC figure measured for THIS kernel symbol counts
            ent = json.load(open(tfile)).get(f"{args.workload}_{cells}", {})
            if ent.get("kernel") == kname:
                traffic = ent.get("smooth_symbol_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": kname + " (fused damped-Jacobi sweep x' = x + d.*(b - A x)), levels " + str(lv),
                "achieved": round(ach_sym, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach_sym / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": round(all_bytes / all_cnt, 1),
                "avg_launch_ms": round(avg_ms, 5), "launches": all_cnt,
                "device_format": fmt_name,
                "streamed_bytes_per_launch": (round(streamed, 1) if streamed else None),
                "streamed_achieved": (round(streamed / avg_ms / 1e6, 1) if streamed else None),
                "streamed_frac": (round(streamed / avg_ms / 1e6 / HBM_PEAK_GBS, 4) if streamed else None),
                "note": ("`achieved`/`frac` price the launch at the CSR algorithmic bytes (12 B/nnz + vectors), as the "
                         "metric is defined; the kernel in use (`device_format`) moves `streamed_bytes_per_launch`. "
                         + ("Row classes remove the matrix stream for operators made of a few distinct rows (this "
                            "constant-coefficient workload): frac > 1 is traffic avoided, NOT bandwidth above peak - "
                            "`streamed_frac` is the bandwidth figure. Operators without that redundancy (e.g. workload "
                            "c3) run the streaming kernels." if fmt_name in ("row classes", "mixed") else "")),
                "row_classes": {f"L{l}": {"classes": rcs[l][0], "dictionary_entries": rcs[l][1],
                                          "implicit_first_column": h.operator_rowclass_flags(l, mg.device.MG_OP_A)[0],
                                          "relaxPrec_from_dictionary": h.operator_rowclass_flags(l, mg.device.MG_OP_A)[1]}
                                for l in lv if rcs[l][0] > 0},
                "fine_level_only": {"launches": cnt_s, "avg_launch_ms": round(ms_s / cnt_s, 5),
                                    "algorithmic_bytes_per_launch": bts_s, "achieved": round(achieved, 1),
                                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                                    "streamed_bytes_per_launch": (fine_streamed if nrhs == 1 else None),
                                    "streamed_achieved": (round(fine_streamed / (ms_s / cnt_s) / 1e6, 1) if nrhs == 1 else None)},
                "residual_level1": {k: (round(v, 5) if isinstance(v, float) else v) for k, v in dom["residual"].items()},
                "kernel_ms_per_step": round(tot_ms / K, 4),
                "step_algorithmic_GB": round(step_bytes / 1e9, 4),
                "step_hbm_gbs": round(step_bytes / (dt / K) / 1e9, 1),
                "kernel_time_share": {f"L{l}:{k}": round(v[0] / tot_ms, 4) for (l, k), v in sorted(prof.items()) if v[0] / tot_ms > 0.01}}

    # ---- CPU baseline: the C/OpenMP oracle ("port") on a bounded sample of the same workload ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import c_oracle
        cores = c_oracle.max_threads()
        co = c_oracle.COracle(p, nrhs)
        xc = np.zeros_like(b_host)
        ncyc = args.cpu_cycles or (12 if cells >= 200 else 50)   # about 10 s of host work (128 threads)
        if nrhs > 1:
            ncyc = args.cpu_cycles or 1
        co.solveMG(b_host, xc, 0.0, 1, cores)          # warm-up cycle
        xc[...] = 0.0
        t0 = time.perf_counter()
        it, rv = co.solveMG(b_host, xc, 0.0, ncyc, cores)
        tc = time.perf_counter() - t0
        cpu = {"value": round(n * nrhs * it / tc, 1), "unit": "DoF-updates/s", "cores": cores, "kind": "port",
               "sample": f"{it} solveMG steps of the same workload ({cells}^3 cells, nrhs={nrhs}) on the C/OpenMP oracle, "
                         f"Int64 indices, unfused op sequence, {tc:.2f}s",
               "relres_after_sample": float(rv[-1] / rv[0])}
        # parity spot-check at full size: the oracle's residual history on its sample vs the device's
        k = min(len(rv), len(resvec))
        cpu["resvec_rel_diff_vs_gpu"] = float(np.abs(rv[:k] - resvec[:k]).max() / resvec[0])

    if rank == 0:
        out = {
            "metric": "V-cycle DoF-updates/s", "value": round(dof_per_s, 1), "unit": "DoF-updates/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{desc} ({n} nodal DoF), {p.levels} levels, nrhs={nrhs}, fp64, "
                                   f"solveMG step = cycle + residual + norm",
                       "cells": cells, "levels": p.levels, "nrhs": nrhs, "N": n, "nnz": int(A.nnz),
                       "level_rows": [int(a.shape[0]) for a in p.As], "level_nnz": [int(a.nnz) for a in p.As],
                       "parallelism": "1 process per GPU"},
            "relres_after_steps": relres,
            "setup_s": {"operator": round(t_op, 2), "MGsetup": round(t_setup, 2), "upload": round(t_upload, 2)},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    mg.clear_(p)
    if world > 1:
        dist.destroy_process_group()


def bench_weak(args, mg, torch, dist, cells, K, W, rank, world, local_rank):
    """N > 1, WEAK scaling: `cells`^3 per GPU; the global grid is cells x (boxes per dimension) - 512^3 on 8
    GPUs (BASELINE.json configs[3]).  No global matrix exists anywhere: every rank builds its part of the
    hierarchy on an overlapping box (multigrid.jl_amd/structured_setup.py)."""
    from multigrid_jl_amd import distributed as dd, structured_setup as ss
    share = os.environ.get("MG_BENCH_SHARE_GPU") == "1"
    dev = torch.device("cuda", local_rank)
    domains = dd.default_domains(world, 3)
    gcells = [cells * d for d in domains]
    domain = np.ravel([[0.0, float(d)] for d in domains])          # h = 1/cells in every direction, as on one GPU
    levels = args.levels or levels_for(min(gcells))
    p = mg.getMGparam(np.float64, np.int64, levels, os.cpu_count() or 8, K, 0.0, "Jac", 0.8, 2, 1, "V",
                      "NoMUMPS", 0.5, 0.0, "FullWeighting")
    t0 = time.perf_counter()
    be = dd.HipBackend(local_rank)
    comm = dd.TorchComm(stage_through_host=share)
    H, info = ss.structured_gmg(gcells, domains, comm, be, p, ss.poisson_operator(gcells, domain), domain=domain, nrhs=1)
    t_setup = time.perf_counter() - t0
    n = int(np.prod(np.asarray(gcells) + 1))
    b_own, ssq = ss.local_rhs(info, 1)
    red_dev = torch.device("cpu") if share else dev
    tot = torch.tensor([ssq], device=red_dev, dtype=torch.float64)
    dist.all_reduce(tot)
    b = be.from_numpy(H.order_fine(b_own) / float(tot.item()) ** 0.5)
    x = torch.zeros_like(b)
    log(f"[rank {rank}] global {gcells} cells over boxes {domains}: own {H.levels[0].n_own} of {n} rows, "
        f"{len(H.levels)} sharded levels + replicated tail of {H.n_tail} rows; halo A1 {H.levels[0].planA.n_halo}; "
        f"sharded setup {t_setup:.1f}s")

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(20 if args.prewarm > 0 else 0):      # fixed count: every rank enters the same collectives
        H.cycle(b, x, False)
    if W > 0:
        x.zero_()
        H.solve(b, x, 0.0, W)
    x.zero_()
    barrier()
    t0 = time.perf_counter()
    iters, resvec = H.solve(b, x, 0.0, K)
    barrier()
    dt = time.perf_counter() - t0
    assert iters == K
    tt = torch.tensor([dt], device=red_dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    lb = torch.tensor([H.local_algorithmic_bytes()], device=red_dev, dtype=torch.float64)
    dist.all_reduce(lb, op=dist.ReduceOp.MAX)
    if rank == 0:
        ach = float(lb.item()) / (dt / K) / 1e9
        out = {
            "metric": "V-cycle DoF-updates/s", "value": round(n * K / dt, 1), "unit": "DoF-updates/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"3D 7-pt Poisson, {cells}^3 cells per GPU = {gcells} cells global ({n} nodal DoF), "
                                   f"GMG V(2,1) damped-Jacobi w=0.8, {levels} levels, nrhs=1, fp64, "
                                   f"solveMG step = cycle + residual + norm",
                       "cells_per_gpu": cells, "global_cells": gcells, "levels": levels, "nrhs": 1, "N": n,
                       "parallelism": f"DomainDecomposition boxes {domains}, {len(H.levels)} sharded levels, replicated "
                                      f"tail from {H.n_tail} rows, one all_to_all_single halo exchange per SpMV (RCCL), "
                                      f"sharded host setup"},
            "relres_after_steps": float(resvec[-1] / resvec[0]),
            "setup_s": {"sharded_setup_incl_upload": round(t_setup, 2)},
            "roofline": {"bound": "hbm", "kernel": "sharded levels of one V-cycle, per GPU (max over ranks)",
                         "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                         "note": "ALGORITHMIC (CSR-priced) bytes of the rank's sharded levels / step time; the local "
                                 "operators of this constant-coefficient workload are stored as row classes (no matrix "
                                 "stream), so this is not a bandwidth figure - see the N=1 line's `streamed_*` fields"},
            "cpu_baseline": None,
        }
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def bench_distributed(args, mg, torch, dist, A, mesh, p, b_host, cells, nrhs, K, W, rank, world, local_rank,
                      t_op, t_setup):
    """N > 1: STRONG scaling of the same workload - the fine grid is sharded over the ranks by the
    DomainDecomposition box rule, one all_to_all_single halo exchange per SpMV over RCCL/xGMI, coarse tail
    replicated (multigrid.jl_amd/distributed.py).  Every rank builds the global hierarchy on its host."""
    from multigrid_jl_amd import distributed as dd
    dev = torch.device("cuda", local_rank)
    t0 = time.perf_counter()
    domains = dd.default_domains(world, 3)
    owner = dd.box_owner(mesh.n + 1, domains)
    be = dd.HipBackend(local_rank)
    share = os.environ.get("MG_BENCH_SHARE_GPU") == "1"
    comm = dd.TorchComm(stage_through_host=share)
    H = dd.DistributedHierarchy.from_global(p, comm, be, owner, nrhs)
    t_upload = time.perf_counter() - t0
    n = A.shape[0]
    log(f"[rank {rank}] sharded {cells}^3 cells over {domains}: own {H.levels[0].n_own} of {n} rows, "
        f"{len(H.levels)} sharded levels + replicated tail of {H.n_tail} rows; halo A1 {H.levels[0].planA.n_halo}; "
        f"operator {t_op:.1f}s, MGsetup {t_setup:.1f}s, localize+upload {t_upload:.1f}s")
    b = H.scatter_fine(b_host)
    x = torch.zeros_like(b)

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    # pre-warm: a FIXED number of untimed cycles (a time-based loop would give the ranks different
    # collective counts and deadlock)
    for _ in range(20 if args.prewarm > 0 else 0):
        H.cycle(b, x, False)
    if W > 0:
        x.zero_()
        H.solve(b, x, 0.0, W)
    x.zero_()
    barrier()
    t0 = time.perf_counter()
    iters, resvec = H.solve(b, x, 0.0, K)
    barrier()
    dt = time.perf_counter() - t0
    assert iters == K
    red_dev = torch.device("cpu") if share else dev
    tt = torch.tensor([dt], device=red_dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    # per-rank algorithmic bytes of the sharded levels (max over ranks) -> achieved GB/s per GPU
    lb = torch.tensor([H.local_algorithmic_bytes()], device=red_dev, dtype=torch.float64)
    dist.all_reduce(lb, op=dist.ReduceOp.MAX)
    if rank == 0:
        out = {
            "metric": "V-cycle DoF-updates/s", "value": round(n * nrhs * K / dt, 1), "unit": "DoF-updates/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"3D 7-pt Poisson {cells}^3 cells ({n} nodal DoF), GMG V(2,1) damped-Jacobi w=0.8, "
                                   f"{p.levels} levels, nrhs={nrhs}, fp64, solveMG step = cycle + residual + norm",
                       "cells": cells, "levels": p.levels, "nrhs": nrhs, "N": n, "nnz": int(A.nnz),
                       "parallelism": f"DomainDecomposition boxes {domains}, {len(H.levels)} sharded levels, "
                                      f"replicated tail from {H.n_tail} rows, all_to_all_single halo per SpMV (RCCL)"},
            "relres_after_steps": float(resvec[-1] / resvec[0]),
            "roofline": {"bound": "hbm", "kernel": "sharded levels of one V-cycle, per GPU (max over ranks)",
                         "achieved": round(float(lb.item()) / (dt / K) / 1e9, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(float(lb.item()) / (dt / K) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": None},
            "cpu_baseline": None,
        }
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
