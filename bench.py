#!/usr/bin/env python3
"""bench.py - V-cycle DoF-updates/s + achieved HBM GB/s (BASELINE.json metric) on MI355X.

A "step" is one iteration of the reference's solveMG loop (src/Multigrid/SolveFuncs.jl:24-37): one
recursiveCycle from the finest level + the residual SpMV + the Frobenius norm.  The timed region is
exactly K steps (mg_solve_dev_FP64 with maxIter=K, tol=0, x0=0 as SURVEY.md 8d prescribes) with the
hierarchy, b and x resident in HBM.

Workloads (BASELINE.json configs):
  c2 (default)  3-D 7-pt Poisson 256^3 cells, GMG V(2,1) damped Jacobi w=0.8, 6 levels, fp64, nrhs=1
  c5            same operator, 16 right-hand sides (solved column by column on the single-vector kernels where the fine
                level has the four-stage pass - MG_NO_COLUMNS=1: the block SpMM path)
  c3            SA-AMG on anisotropic diffusion (edge weights 16:4:1 x log-normal sigma), general CSR, V(1,1) SPAI
  c1            32^3 cells (CPU-plumbing size; parity case, not a bench line)
Use --cells N to shrink the grid for quick checks (the JSON then names the reduced workload).
"""
import argparse
import json
import os
import sys
import time

# the CPU baseline's OpenMP threads are pinned one per core, spread over the sockets (must be set before any
# OpenMP runtime is loaded, i.e. before torch / the oracle library are imported)
if os.environ.get("WORLD_SIZE", "1") == "1":      # (one rank only: N ranks pinning to the same cores would collide)
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def emit(out):
    """Print THE json line as the last line of stdout: native libraries (RCCL prints a version banner) write to the C
    stdio buffer, which would otherwise be flushed after Python's line at exit."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


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
    ap.add_argument("--python-sequencer", action="store_true",
                    help="N>1: sequence the sharded cycle from Python (torch.distributed) instead of the native mg_dist_* path")
    ap.add_argument("--sharded-form", default="ghost", choices=["ghost", "halo"],
                    help="N>1 / --force-sharded-path: ghost = every rank runs the single-GPU kernels on its box extended by ghost layers "
                         "(mg_ghost_*, one exchange per fused pass); halo = round 2-4's form (mg_dist_*: owned rows + halo columns, one exchange per product)")
    ap.add_argument("--ghost-dry", default="", help="R/N: timing aid - this ONE process is rank R of a world of N in the ghost-layer form, alone on "
                                                    "its GPU (exchanges pack and unpack, nothing travels): one GPU's compute share of the N-GPU step")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of exactly --steps steps each; the median is reported")
    ap.add_argument("--c3-weights", default="1,0.25,0.0625", help="c3: edge weights of the three directions (SURVEY 8d states 1,1e-2,1e-4; "
                    "the default 16:4:1 is what fits at 256^3 - profiles/HISTORY.md section 10)")
    ap.add_argument("--no-generic-pass", action="store_true", help="skip the second pass with the streaming formats forced")
    ap.add_argument("--no-c5-leg", action="store_true", help="skip the 16-right-hand-side leg (BASELINE configs[4]) on the same handle")
    ap.add_argument("--no-divsiggrad", action="store_true", help="skip the variable-coefficient (div sigma grad) leg")
    ap.add_argument("--no-c3-leg", action="store_true", help="skip the SA-AMG / general-CSR leg (BASELINE configs[2] at 128^3 cells) of the default line")
    ap.add_argument("--global-cells", default="", help="single-GPU path: a,b,c cells of a non-cubic grid with h = 1/--cells in every "
                                                       "direction (the GLOBAL grid of an N-GPU weak-scaling run, on one GPU)")
    ap.add_argument("--strong-reference", default="auto", choices=["auto", "on", "off"],
                    help="N > 1: also run the GLOBAL grid on ONE GPU in the same job (when it fits) and print strong_speedup_vs_n1")
    ap.add_argument("--n1-reference", default="auto", choices=["auto", "on", "off"],
                    help="--gpus N > 1 started without a launcher: also run the per-GPU workload on one GPU first and report "
                         "parallel_efficiency_vs_n1 (auto: weak scaling only - a strong-scaling N = 1 run of 512^3 is its own job)")
    return ap.parse_args()


def visible_gpus():
    """GPUs this process tree may use, WITHOUT initialising HIP (the launcher must stay GPU-free: it starts the ranks)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            return len([t for t in v.split(",") if t.strip() != ""])
    n = 0
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        for node in os.listdir(base):
            for line in open(os.path.join(base, node, "properties")):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
    except Exception:
        n = 0
    if n == 0:
        try:
            import torch
            n = torch.cuda.device_count()      # (counting devices does not initialise the GPU on this image)
        except Exception:
            n = 0
    return n


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def host_mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return float(line.split()[1]) / 1e6
    except Exception:
        pass
    return 0.0


def run_ranks(argv, n, extra_env=None, tag="", deadline_s=None):
    """Start n fresh rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set - the fan-out the
    reference does with its worker map, DDParallel.jl:87-105,133-139), wait for all of them and return
    (exit code, parsed JSON line of rank 0 or None).  The parent never touches the GPU and never execs."""
    import subprocess
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "MG_BENCH_CHILD": "1",
                    "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
    # Watch ALL ranks: rank 0's pipe is drained by a thread; the first rank that exits non-zero (or a global deadline,
    # MG_BENCH_LAUNCH_TIMEOUT seconds, default 3600) ends the others - a rank left alone in an RCCL / gloo collective would
    # otherwise wait forever (the raw communicator of the native sequencer has no watchdog).
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + min(float(os.environ.get("MG_BENCH_LAUNCH_TIMEOUT", "3600")), deadline_s or 1e9)
    failed = None
    while True:
        rcs = [q.poll() for q in procs]
        if all(c is not None for c in rcs):
            break
        bad = next((i for i, c in enumerate(rcs) if c not in (None, 0)), None)
        if bad is not None or time.time() > deadline:
            failed = f"rank {bad} exited with {rcs[bad]}" if bad is not None else "deadline reached"
            for q in procs:
                if q.poll() is None:
                    q.terminate()
            t_kill = time.time() + 10.0
            while time.time() < t_kill and any(q.poll() is None for q in procs):
                time.sleep(0.1)
            for q in procs:
                if q.poll() is None:
                    q.kill()
            break
        time.sleep(0.2)
    rcs = [q.wait() for q in procs]
    reader.join(timeout=10.0)
    out0 = b"".join(c for c in chunks if c)
    if failed:
        log(f"[launcher{tag}] {failed}: the remaining ranks were stopped")
        if all(c == 0 for c in rcs):
            rcs[0] = 1
    line = None
    for ln in (out0 or b"").decode(errors="replace").splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                line = json.loads(ln)
            except Exception:
                pass
    rc = next((c for c in rcs if c != 0), 0)
    if rc != 0:
        log(f"[launcher{tag}] rank exit codes {rcs}")
    return rc, line


def strong_reference_run(args, n):
    """The north_star's scaling claim is STRONG: the same global grid on 1 and on N GPUs.  A weak-scaling run (cells^3 per GPU)
    has the global grid cells x boxes; when that fits one GPU (512^3 = 135 M rows: 40 GB of HBM, ~170 GB of host memory during
    setup) it is also run through the single-GPU path in this job, in a fresh process of its own, BEFORE the N-GPU run.
    Returns its JSON line or None."""
    if n <= 1 or args.workload != "c2" or args.strong_reference == "off" or os.environ.get("MG_BENCH_SHARE_GPU") == "1":
        return None
    from multigrid_jl_amd.distributed import default_domains
    cells = args.cells or 256
    gcells = [cells] * 3 if args.scaling == "strong" else [cells * d for d in default_domains(n, 3)]
    rows = 1
    for c in gcells:
        rows *= c + 1
    if args.strong_reference != "on" and rows > 140_000_000:
        return None
    # 512^3 on one GPU needs ~170 GB of host memory during its setup: 'auto' only where the host has it (else the failure would
    # come after minutes of setup, inside the other ranks' rendezvous window)
    need_gb = 1.3e-6 * rows
    if args.strong_reference != "on" and host_mem_available_gb() < need_gb:
        log(f"[launcher] strong reference skipped: {host_mem_available_gb():.0f} GB of host memory available, ~{need_gb:.0f} GB needed")
        return None
    a1 = ["--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline", "--no-generic-pass",
          "--workload", "c2", "--cells", str(cells), "--global-cells", ",".join(str(c) for c in gcells)]
    # (under torchrun the other ranks wait in the rendezvous meanwhile: the child gets less than the process group's timeout)
    rc1, sref = run_ranks(a1, 1, {"MG_BENCH_N1_REFERENCE": "1"}, " n1-global", deadline_s=PG_TIMEOUT.total_seconds() - 300.0)
    if rc1 != 0 or sref is None:
        log("[launcher] the N = 1 run of the global grid failed (memory?); strong_speedup_vs_n1 stays null")
        return None
    return sref


def attach_strong_reference(line, sref):
    """strong_speedup_vs_n1 = time per step of the global grid on ONE GPU / time per step of the N-GPU run (same box, same job)."""
    if sref is not None:
        line["strong_n1_reference"] = {"value": float(sref["value"]), "ms_per_step": sref.get("ms_per_step"),
                                       "workload": sref["config"]["workload"], "path": "single-GPU path, same box, same job"}
        line["strong_speedup_vs_n1"] = round(float(sref["ms_per_step"]) / float(line["ms_per_step"]), 4)
    else:
        line["strong_speedup_vs_n1"] = None


import datetime
PG_TIMEOUT = datetime.timedelta(minutes=40)      # (rank 0 may run the N = 1 reference of the global grid before it joins)
STRONG_REF = None      # (torchrun launch: rank 0 runs the global grid on one GPU before it touches its own)


def launch(args):
    """`python bench.py --gpus N` with no launcher around it: be the launcher.  Parses, starts N ranks, relays rank 0's
    single JSON line (plus what only the launcher knows: the N = 1 reference of the same per-GPU workload and the
    parallel efficiency against it), exits non-zero if any rank did."""
    n = args.gpus
    argv = [a for a in sys.argv[1:]]
    have = visible_gpus()
    extra = {}
    if have < n:
        if os.environ.get("MG_BENCH_SHARE_GPU") == "1" or os.environ.get("MG_BENCH_ALLOW_SHARE") == "1":
            extra["MG_BENCH_SHARE_GPU"] = "1"
            log(f"[launcher] {n} ranks requested, {have} GPU(s) visible: the ranks SHARE cuda:0 through the host-staged "
                f"transport (functional run; its numbers mean nothing)")
        else:
            raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible (set MG_BENCH_ALLOW_SHARE=1 for a functional "
                             f"run with all ranks on one GPU)")
    ref = None
    if args.n1_reference == "on" or (args.n1_reference == "auto" and args.scaling == "weak"):
        # the same per-GPU workload on ONE GPU through the single-GPU path (== the default `--gpus 1` line, minus the
        # CPU baseline and the generic-CSR pass): the denominator of parallel_efficiency_vs_n1
        a1 = ["--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline",
              "--no-generic-pass", "--workload", args.workload]
        if args.cells:
            a1 += ["--cells", str(args.cells)]
        if args.levels:
            a1 += ["--levels", str(args.levels)]
        rc1, ref = run_ranks(a1, 1, {"MG_BENCH_N1_REFERENCE": "1"}, " n1")
        if rc1 != 0 or ref is None:
            log("[launcher] the N = 1 reference run failed; parallel_efficiency_vs_n1 stays null")
            ref = None
    sref = strong_reference_run(args, n)
    rc, line = run_ranks(argv, n, extra)
    if rc != 0 or line is None:
        raise SystemExit(rc or 1)
    attach_strong_reference(line, sref)
    line["launcher"] = {"ranks_started": n, "gpus_visible": have, "shared_gpu": bool(extra),
                        "how": "bench.py started the ranks itself (fresh processes, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set)"}
    if ref is not None:
        v1 = float(ref["value"])
        line["n1_reference"] = {"value": v1, "ms_per_step": ref.get("ms_per_step"), "workload": ref["config"]["workload"],
                                "path": "single-GPU path, same box, same run"}
        # weak: N x the work in the same time; strong: the same work N x faster - value(N) / (N * value(1)) either way,
        # given that the N = 1 run holds the per-GPU workload (weak) or the whole problem (strong)
        line["parallel_efficiency_vs_n1"] = round(float(line["value"]) / (n * v1), 4)
    else:
        line["parallel_efficiency_vs_n1"] = None
    emit(line)


def levels_for(cells):
    """Levels down to a coarsest grid of at most 9^3 nodes: 32 -> 3, 256 -> 6, 512 -> 7.  Node counts that become
    even on the way (400 cells: 401 -> 201 -> 101 -> 51 -> 26 -> 14 -> 8) keep coarsening - full weighting handles them by
    keeping the last node (GeometricTransferOperators.jl:35-36); stopping at the first even count left a 26^3 coarsest level
    on the sparse-factor path (0.55 ms of a 2.4 ms step at 400^3, round 3)."""
    lv, n = 1, cells + 1
    while n > 9:
        n = (n + 1) // 2 if n % 2 == 1 else n // 2 + 1
        lv += 1
    return lv


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch(args)          # no torchrun around us: start the ranks ourselves (this process stays GPU-free)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus and not args.force_sharded_path:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={os.environ['WORLD_SIZE']}")
    global STRONG_REF
    if ("WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1 and int(os.environ.get("RANK", "0")) == 0
            and os.environ.get("MG_BENCH_CHILD") != "1" and not args.force_sharded_path):
        # started by torchrun (the driver's N > 1 launch): no launcher of ours ran the strong reference - rank 0 does, in a child
        # process, before anything here touches a GPU; the other ranks wait in the rendezvous meanwhile
        try:
            STRONG_REF = strong_reference_run(args, int(os.environ["WORLD_SIZE"]))
        except Exception as e:      # (never let the reference take the scaling run down)
            log(f"[rank 0] strong reference run failed: {type(e).__name__}: {e}")
            STRONG_REF = None
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
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=PG_TIMEOUT)
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if share:
            dist.init_process_group("gloo", timeout=PG_TIMEOUT)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=PG_TIMEOUT)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the multigrid cycle has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    cells = args.cells or {"c1": 32, "c2": 256, "c3": 256, "c5": 256}[args.workload]
    nrhs = 16 if args.workload == "c5" else 1
    levels = args.levels or levels_for(min(int(c) for c in args.global_cells.split(",")) if args.global_cells else cells)
    K, W = args.steps, args.warmup

    if world > 1:
        os.environ.setdefault("MG_HOST_THREADS", str(max(1, (os.cpu_count() or 8) // world)))
    if world > 1 or args.force_sharded_path or args.ghost_dry:
        if args.workload != "c2":
            raise SystemExit("the multi-GPU bench is defined for the c2/c4 Poisson workload")
        if args.sharded_form == "ghost" and not args.python_sequencer:
            return bench_ghost(args, mg, torch, dist, cells, K, W, rank, world, local_rank)
        return bench_weak(args, mg, torch, dist, cells, K, W, rank, world, local_rank)

    # ---- host setup (CPU, as in the reference) ---------------------------------------------------
    t0 = time.perf_counter()
    if args.workload == "c3":
        import threading
        hb_stop = threading.Event()

        def _heartbeat():      # (the SA setup at 256^3 takes minutes on the host: a line a minute says the run is alive)
            t_hb = time.perf_counter()
            while not hb_stop.wait(60.0):
                print(f"[bench] host setup running, {time.perf_counter() - t_hb:.0f} s", file=sys.stderr, flush=True)
        threading.Thread(target=_heartbeat, daemon=True).start()
        os.environ.setdefault("MG_SETUP_GPU", "1")     # (opt-in since round 6: the largest Galerkin products of this setup on the GPU; =0: host only)
        c3w = tuple(float(w) for w in args.c3_weights.split(","))
        A, mesh = mg.anisotropic_divsiggrad([cells] * 3, weights=c3w)
        t_op = time.perf_counter() - t0
        p = mg.getMGparam(np.float64, np.int64, args.levels or 14, os.cpu_count() or 8, K, 0.0, "SPAI", 1.0, 1, 1, "V",
                          "Julia", 0.4, 0.0)
        t0 = time.perf_counter()
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):     # (per-level setup times go to stderr: stdout carries the one JSON line)
            mg.SA_AMGsetup(A, p, True, nrhs, verbose=True)
        hb_stop.set()
        desc = f"SA-AMG th=0.4 V(1,1) SPAI, anisotropic diffusion {cells}^3 cells, weights {':'.join(f'{w:g}' for w in c3w)} x lognormal, CSR"
    else:
        gc = [int(c) for c in args.global_cells.split(",")] if args.global_cells else [cells] * 3
        A, mesh = mg.poisson_shifted(gc, [v for g in gc for v in (0.0, g / float(cells))] if args.global_cells else None)
        t_op = time.perf_counter() - t0
        p = mg.getMGparam(np.float64, np.int64, levels, os.cpu_count() or 8, K, 0.0, "Jac", 0.8, 2, 1, "V",
                          "NoMUMPS", 0.5, 0.0, "FullWeighting")
        t0 = time.perf_counter()
        mg.MGsetup(A, mesh, p, nrhs)
        desc = (f"3D 7-pt Poisson {cells}^3 cells, GMG V(2,1) damped-Jacobi w=0.8" if not args.global_cells else
                f"3D 7-pt Poisson {'x'.join(str(c) for c in gc)} cells (h = 1/{cells}), GMG V(2,1) damped-Jacobi w=0.8")
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
    # ---- W untimed warm-up steps, then exactly K timed steps; the region is repeated `--repeats` times (each one
    # bracketed by barrier + synchronize, x reset to 0 outside it) and the MEDIAN region is reported ---------------
    if W > 0:
        h.solve_dev(b, x, 0.0, W)
    region = []
    for _ in range(max(1, args.repeats)):
        x.zero_()
        barrier()
        t0 = time.perf_counter()
        iters, resvec = h.solve_dev(b, x, 0.0, K)
        barrier()
        region.append(time.perf_counter() - t0)
        assert iters == K
    dt = float(np.median(region))
    relres = float(resvec[-1] / resvec[0])
    dof_per_s = world * n * nrhs * K / dt

    # ---- per-kernel HIP-event accounting over the same K steps (separate, instrumented pass) -----
    prof, moved, tot_ms = profiled_pass(h, b, x, K, torch)
    kname, fmt_name = kernel_symbol(h, mg, p, 1, nrhs)
    # the dominant fine-level kernel: the single-stage fused sweep, or - where the two-stage marching kernel serves the
    # level - the sweep + residual pass (csr_rowclass_march2_spmv), whichever takes more of the step
    dom = max((k for k in ((1, "smooth"), (1, "smooth+residual"), (1, "four-stage")) if k in prof), key=lambda k: prof[k][0])
    kdesc = "fine-level fused damped-Jacobi sweep x' = x + d.*(b - A x), level 1"
    tile_geo = None
    if dom[1] == "four-stage":
        ok4, geo = h.four_stage_form(1)
        kname = f"mgk::csr_rowclass_march4_spmv<{geo[8]}, {geo[4]}, NPM, PITCH>"
        tile_geo = {"tiles_per_line": geo[0], "tiles_per_column": geo[1], "TX": geo[2], "TY": geo[3], "rows_per_lane": geo[4],
                    "threads_per_workgroup": geo[8], "workgroups": geo[5], "lds_bytes": geo[6],
                    "schedule": f"lockstep: {geo[9]} segments of {geo[10]} planes per tile",
                    "class_table_entries": geo[11], "estimated_fill_bytes_per_row": geo[7] / 100.0}
        kdesc = ("the solve loop's two fine-level passes across the stopping test as ONE four-stage pass, level 1: last post-smoothing "
                 "sweep t = x + d.*(b - A x), stopping-test residual r = b - A t with ||r||^2, first pre-smoothing update xn = t + d.*r, "
                 "second pre-smoothing sweep t' = xn + d.*(b - A xn) and the restriction's residual r' = b - A t' (x, b in; t', r' out)")
    if dom[1] == "smooth+residual":
        form, geo = h.sweep_residual_form(1)
        kname = "mgk::csr_rowclass_march2_spmv<false>"
        if form == 3:
            kname = f"mgk::csr_rowclass_march3_spmv<ZERO, OUT, {geo[8]}, {geo[4]}, 2>"
            tile_geo = {"tiles_per_line": geo[0], "tiles_per_column": geo[1], "TX": geo[2], "TY": geo[3], "rows_per_lane": geo[4],
                        "threads_per_workgroup": geo[8], "workgroups": geo[5], "lds_bytes": geo[6],
                        "schedule": (f"lockstep: {geo[9]} segments of {geo[10]} planes per tile" if geo[9] else "balanced ranges"),
                        "class_table_entries": geo[11], "estimated_fill_bytes_per_row": geo[7] / 100.0}
        kdesc = ("fine-level damped-Jacobi sweep t = x + d.*(b - A x) AND the residual r = b - A t in one pass, level 1: the "
                 "last pre-smoothing sweep with the residual the restriction needs (x, b in; t, r out)")
    ms_s, cnt_s, bts_s = prof[dom]
    avg_s = ms_s / cnt_s
    mv_s = moved[dom]
    step_bytes = sum(v[2] * v[1] for v in prof.values()) / K            # algorithmic (CSR-priced) bytes per step
    step_moved = sum(moved[k] * prof[k][1] for k in prof) / K           # bytes the kernels in use have to move per step
    traffic_prof = None
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tfile):
        try:   # an OFFLINE rocprofv3 --pmc figure for this kernel symbol (never measured inside this run)
            ent = json.load(open(tfile)).get(f"{args.workload}_{cells}", {})
            if ent.get("kernel") == kname:
                traffic_prof = {"bytes_per_launch": ent.get("smooth_symbol_bytes_per_launch"),
                                "measured_at_commit": ent.get("commit"), "file": "profiles/pmc_traffic.json"}
        except Exception:
            traffic_prof = None
    kern_table = {}
    for (l, k), v in sorted(prof.items()):
        if v[0] / tot_ms > 0.01:
            a = v[0] / v[1]
            kern_table[f"L{l}:{k}"] = {"avg_ms": round(a, 5), "launches_per_step": round(v[1] / K, 2),
                                       "share": round(v[0] / tot_ms, 4), "moved_MB": round(moved[(l, k)] / 1e6, 2),
                                       "frac": round(moved[(l, k)] / a / 1e6 / HBM_PEAK_GBS, 4)}
    roofline = {"bound": "hbm",
                "kernel": kname + " (" + kdesc + ")",
                "achieved": round(mv_s / avg_s / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(mv_s / avg_s / 1e6 / HBM_PEAK_GBS, 4),
                # HBM bytes per launch of this kernel symbol from the committed PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, each counter in a
                # run of its own: scripts/round5_evidence.sh part 2) - an offline figure of the same symbol on the same workload, never
                # collected inside a timed run; null when the run uses another kernel
                "traffic": (traffic_prof or {}).get("bytes_per_launch"), "traffic_from_profile": traffic_prof,
                "bytes_per_launch": mv_s, "avg_launch_ms": round(avg_s, 5), "launches": cnt_s,
                "device_format": fmt_name,
                "definition": "achieved = bytes the kernel IN USE has to move for one launch (device format's matrix side "
                              "+ every vector element once; mg_profile_get_moved) / its average HIP-event duration in this "
                              "run; always <= HBM peak. `csr_equivalent` prices the same launch at SURVEY 8d's CSR bytes "
                              "(12 B/nnz + 4 B/row + vectors): with row classes that exceeds the peak - traffic avoided, "
                              "not bandwidth.",
                "csr_equivalent": {"bytes_per_launch": bts_s, "achieved": round(bts_s / avg_s / 1e6, 1),
                                   "ratio_to_peak": round(bts_s / avg_s / 1e6 / HBM_PEAK_GBS, 4)},
                "kernel_ms_per_step": round(tot_ms / K, 4),
                "step_moved_GB": round(step_moved / 1e9, 4), "step_moved_gbs": round(step_moved / (dt / K) / 1e9, 1),
                "step_csr_equivalent_GB": round(step_bytes / 1e9, 4),
                "kernels": kern_table}
    if tile_geo:
        roofline["inplane_tiles"] = tile_geo
    if (1, "smooth+residual+norm") in prof:
        # the same kernel as the solve loop runs it: last post-smoothing sweep + the residual of the stopping test; x and b
        # in, t + d.*r (the next cycle's first update) and ||r||^2 partials out - the iterate itself is a dead store unless
        # the loop stops (26 B/row instead of 34)
        ms2, cnt2, _ = prof[(1, "smooth+residual+norm")]
        mv2 = moved[(1, "smooth+residual+norm")]
        roofline["solve_loop_variant"] = {"avg_launch_ms": round(ms2 / cnt2, 5), "launches": cnt2, "bytes_per_launch": mv2,
                                          "achieved": round(mv2 / (ms2 / cnt2) / 1e6, 1),
                                          "frac": round(mv2 / (ms2 / cnt2) / 1e6 / HBM_PEAK_GBS, 4)}
    if dom[1] == "four-stage":
        # Four products with A per 32 B/row: the pass is bound by vector-ALU issue (fp64 FMAs + LDS traffic of four stencil
        # stages on tile + rings), not by HBM - its fraction of the HBM peak is LOWER than the two-stage pass's although it
        # replaces two of them in less time than 1.5.  Reported next to it: the same time priced at the bytes of the two
        # two-stage passes it replaces (x, b in, t + d.*r out; then x, b in, t, r out = 56 B/row) and of the four single-stage
        # launches those replaced (4 x 24 B/row).
        n8 = 8.0 * n
        roofline["temporal_blocking"] = {
            "stages_per_pass": 4, "moved_bytes_per_launch": mv_s,
            "two_two_stage_passes_would_move": 7.0 * n8, "equivalent_vs_two_stage_passes": round(7.0 * n8 / avg_s / 1e6, 1),
            "equivalent_vs_two_stage_passes_ratio_to_peak": round(7.0 * n8 / avg_s / 1e6 / HBM_PEAK_GBS, 4),
            "four_single_stage_launches_would_move": 12.0 * n8, "equivalent_vs_single_stage": round(12.0 * n8 / avg_s / 1e6, 1),
            "equivalent_vs_single_stage_ratio_to_peak": round(12.0 * n8 / avg_s / 1e6 / HBM_PEAK_GBS, 4),
            "fp64_flops_per_launch": 2.0 * 36.0 * n, "fp64_tflops": round(2.0 * 36.0 * n / avg_s / 1e9, 2),
            "note": "frac above = bytes THIS kernel has to move / its time / HBM peak; `equivalent_*` price the same time at the bytes "
                    "of the launches it replaces - traffic avoided, not bandwidth.  fp64_flops: 4 stages x 9 fused multiply-adds per row "
                    "(useful rows only; the rings add 17 %)"}
        for key in ((1, "smooth+residual"), (1, "smooth+residual+norm")):
            if key in prof:   # the two-stage passes still in the step (first cycle of a solve: from x = 0; last step by count)
                ms2, cnt2, _ = prof[key]
                roofline.setdefault("two_stage_passes_in_this_run", {})[key[1]] = {
                    "avg_launch_ms": round(ms2 / cnt2, 5), "launches": cnt2, "bytes_per_launch": moved[key],
                    "frac": round(moved[key] / (ms2 / cnt2) / 1e6 / HBM_PEAK_GBS, 4)}
        # time-weighted fraction of the HBM peak over every fine-level sweep / residual launch of the run
        tw_b = sum(moved[k] * prof[k][1] for k in prof if k[0] == 1 and k[1] in ("four-stage", "smooth+residual", "smooth+residual+norm", "smooth", "residual"))
        tw_t = sum(prof[k][0] for k in prof if k[0] == 1 and k[1] in ("four-stage", "smooth+residual", "smooth+residual+norm", "smooth", "residual"))
        roofline["fine_level_sweeps_time_weighted_frac"] = round(tw_b / tw_t / 1e6 / HBM_PEAK_GBS, 4) if tw_t > 0 else None
    if dom[1] == "smooth+residual":
        # Temporal blocking lowers the COMPULSORY bytes (34 B/row instead of 2 x 26 for the two launches it replaces), so
        # its fraction of the peak is not comparable with a single-stage kernel's: report, next to it, (a) the same time
        # priced at the bytes the two single-stage launches would have to move and (b) the single-stage sweep itself
        # (the kernel of the sharded path and of operators the two-stage kernel does not serve), timed in this run.
        Dm = mg.device
        try:
            ms1, _ = h.time_op(1, Dm.MG_K_SMOOTH, 30)
        except Exception:
            ms1 = None
        n8 = 8.0 * n
        fmt_b = float(h.operator_rowclasses(1, Dm.MG_OP_A)[2])    # class ids + dictionary: matrix side of one launch
        two_launch = 2.0 * fmt_b + 6.0 * n8                       # (x, b in; one vector out) twice
        roofline["temporal_blocking"] = {
            "stages_per_pass": 2, "moved_bytes_per_launch": mv_s,
            "two_single_stage_launches_would_move": two_launch,
            "equivalent_achieved": round(two_launch / avg_s / 1e6, 1),
            "equivalent_ratio_to_peak": round(two_launch / avg_s / 1e6 / HBM_PEAK_GBS, 4),
            "note": "frac above = bytes THIS kernel has to move / its time / peak; `equivalent_*` prices the same time at the "
                    "bytes of the two single-stage launches (sweep, then residual) it replaces - traffic avoided, not bandwidth"}
        if ms1:
            k1, _ = kernel_symbol(h, mg, p, 1, nrhs)
            mv1 = fmt_b + 3.0 * n8
            roofline["single_stage_sweep"] = {
                "kernel": k1, "avg_launch_ms": round(ms1, 5), "bytes_per_launch": mv1,
                "achieved": round(mv1 / ms1 / 1e6, 1), "frac": round(mv1 / ms1 / 1e6 / HBM_PEAK_GBS, 4),
                "timing": "mg_time_op_dev_FP64: HIP-event average of 30 back-to-back launches in this run (back-to-back launches "
                          "of one kernel run 10-15 % slower than the same launch inside the cycle)"}
    if nrhs == 1:
        rc = h.operator_rowclasses(1, mg.device.MG_OP_A)
        if rc[0] > 0:
            fl = h.operator_rowclass_flags(1, mg.device.MG_OP_A)
            roofline["row_classes_L1"] = {"classes": rc[0], "dictionary_entries": rc[1], "implicit_first_column": fl[0],
                                          "relaxPrec_from_dictionary": fl[1]}

    # ---- BASELINE configs[4] (C5) on the SAME handle: 16 right-hand sides (adjustMemoryForNumRHS = mg_set_nrhs), a few steps ----
    c5_leg = None
    if args.workload == "c2" and nrhs == 1 and not args.no_c5_leg and cells >= 64:
        try:
            k5, K5 = 16, max(2, min(K, 5))
            b16 = torch.from_numpy(np.ascontiguousarray(mg.seeded_rhs(A, k5))).to(dev)      # row-major [n][16]: the library's block layout
            x16 = torch.zeros_like(b16)
            h.set_nrhs(k5)
            h.solve_dev(b16, x16, 0.0, 2)
            d5 = []
            for _ in range(3):
                x16.zero_()
                barrier()
                t0 = time.perf_counter()
                it5, res5 = h.solve_dev(b16, x16, 0.0, K5)
                barrier()
                d5.append(time.perf_counter() - t0)
            dt5 = sorted(d5)[1]
            c5_leg = {"workload": f"the same hierarchy, 16 right-hand sides (BASELINE configs[4]): solveMG on the n x 16 block, Frobenius criterion",
                      "ms_per_step": round(dt5 / K5 * 1e3, 4), "dof_updates_per_s": round(n * k5 * K5 / dt5, 1), "steps": K5,
                      "timed_regions_ms_per_step": [round(v / K5 * 1e3, 4) for v in d5], "relres_after_steps": float(res5[-1] / res5[0]),
                      "path": "column by column on the single-vector kernels, columns alternating between two streams (solve_dev_columns)"
                              if not os.environ.get("MG_NO_COLUMNS") else "block SpMM kernels (MG_NO_COLUMNS=1)"}
            del b16, x16
            h.set_nrhs(1)
            torch.cuda.empty_cache()
        except Exception as e:
            c5_leg = {"error": f"{type(e).__name__}: {e}"}
            try:
                h.set_nrhs(1)
            except Exception:
                pass
    if c5_leg is not None:
        roofline["c5_leg"] = c5_leg
    # ---- the same K steps with the STREAMING formats forced for this handle (mg_set_option no_rowclass = 1): what
    # an operator without repeated rows gets (variable coefficients, SA-AMG); its SURVEY 8d fraction ------------------
    if nrhs == 1 and roofline.get("row_classes_L1") and not args.no_generic_pass:
        def generic_pass(opts):
            hg = mg.device.DeviceHierarchy(p, device_id=local_rank, nrhs=nrhs, options=opts)
            xg = torch.zeros_like(b)
            hg.solve_dev(b, xg, 0.0, max(1, W))
            dts = []
            for _ in range(3):          # (median of 3 regions of K steps: the first region of a fresh handle runs slow)
                xg.zero_()
                barrier()
                t0 = time.perf_counter()
                itg, resg = hg.solve_dev(b, xg, 0.0, K)
                barrier()
                dts.append(time.perf_counter() - t0)
            dtg = sorted(dts)[1]
            profg, movedg, totg = profiled_pass(hg, b, xg, K, torch)
            formg, _ = hg.sweep_residual_form(1)
            kg, fg = kernel_symbol(hg, mg, p, 1, nrhs)
            hg.close()
            del xg
            return dtg, resg, profg, movedg, formg, kg, fg

        def launch_line(prof_, moved_, key):
            if key not in prof_ or prof_[key][1] == 0:
                return None
            ms_, cnt_, bts_ = prof_[key]
            a_ = ms_ / cnt_
            return {"avg_launch_ms": round(a_, 5), "launches": cnt_, "algorithmic_bytes_per_launch": bts_,
                    "achieved": round(bts_ / a_ / 1e6, 1), "frac": round(bts_ / a_ / 1e6 / HBM_PEAK_GBS, 4),
                    "moved_bytes_per_launch": moved_[key], "moved_frac": round(moved_[key] / a_ / 1e6 / HBM_PEAK_GBS, 4)}

        dtg, resg, profg, movedg, formg, kg, fg = generic_pass({"no_rowclass": 1})
        gen = {"ms_per_step": round(dtg / K * 1e3, 4), "dof_updates_per_s": round(n * nrhs * K / dtg, 1),
               "resvec_rel_diff_vs_default": float(np.abs(resg - resvec).max() / resvec[0]),
               "note": "the same workload with row classes disabled through the API for a second handle: what an operator "
                       "without repeated rows gets (variable coefficients; jInv's div sigma grad).  `frac` is SURVEY 8d's "
                       "ALGORITHMIC CSR bytes / time / peak (the north_star's 60 % target), `moved_frac` the bytes the kernel streams"}
        if formg == 4:      # band form: sweep + residual in one pass, the values streamed once from planar arrays
            pair = launch_line(profg, movedg, (1, "smooth+residual"))
            pairn = launch_line(profg, movedg, (1, "smooth+residual+norm"))
            gen.update({"kernel": "mgk::csr_rowclass_march3_spmv<false, 5, 512, K1, 2, true>", "device_format": "band form (structure "
                        "classes as a product map + 7 planar value arrays, 4 of them read where the values are symmetric), fine level; band-27 / pattern-coded CSR below", "sweep_residual_pair": pair,
                        "sweep_residual_norm_pair": pairn})
            if pair:
                gen.update({k: pair[k] for k in ("avg_launch_ms", "launches", "algorithmic_bytes_per_launch", "achieved", "frac",
                                                 "moved_bytes_per_launch", "moved_frac")})
            # the CSR kernels alone (round 2's path for such operators), for comparison
            dt0, res0, prof0, moved0, _, k0, f0 = generic_pass({"no_rowclass": 1, "no_band": 1})
            line0 = launch_line(prof0, moved0, (1, "smooth")) or {}
            line0.update({"kernel": k0, "device_format": f0, "ms_per_step": round(dt0 / K * 1e3, 4),
                          "resvec_rel_diff_vs_default": float(np.abs(res0 - resvec).max() / resvec[0])})
            gen["without_band"] = line0
        else:
            line = launch_line(profg, movedg, (1, "smooth")) or {}
            gen.update(line)
            gen.update({"kernel": kg, "device_format": fg})
        roofline["generic_csr"] = gen

    # ---- the workload jInv actually sends: nodal div sigma grad with a log-normal cell coefficient
    # (testGMG.jl:57-75 / testSAforDivSigGrad.jl:96-100 idiom) on the same grid, same solver parameters, DEFAULT format
    # selection - every row of A has its own values (band form on the fine level, CSR kernels on the Galerkin levels below),
    # P and R keep their row classes.  Host setup as in the reference (MGsetup, Galerkin products). -------------------------
    if args.workload == "c2" and nrhs == 1 and not args.no_divsiggrad and not args.no_generic_pass:
        import scipy.sparse as sp
        t0 = time.perf_counter()
        mesh2 = mg.getRegularMesh([0.0, 1.0] * 3, [cells] * 3)
        sigma = np.exp(np.random.default_rng(5).standard_normal(cells ** 3))
        A2 = mg.getNodalDivSigGradMatrix(mesh2, sigma)
        A2 = (A2 + 1e-3 * abs(A2).sum(axis=0).max() * sp.identity(A2.shape[0], format="csr")).tocsr()
        A2.sort_indices()
        p2 = mg.getMGparam(np.float64, np.int64, levels, os.cpu_count() or 8, K, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0,
                           "FullWeighting")
        mg.MGsetup(A2, mesh2, p2, 1)
        t_set2 = time.perf_counter() - t0
        h2 = mg.device.DeviceHierarchy(p2, device_id=local_rank, nrhs=1)
        b2 = torch.from_numpy(np.ascontiguousarray(mg.seeded_rhs(A2, 1))).to(dev)
        x2 = torch.zeros_like(b2)
        h2.solve_dev(b2, x2, 0.0, max(1, W))
        dts = []
        for _ in range(3):
            x2.zero_()
            barrier()
            t0 = time.perf_counter()
            it2, res2 = h2.solve_dev(b2, x2, 0.0, K)
            barrier()
            dts.append(time.perf_counter() - t0)
        dt2 = sorted(dts)[1]
        prof2, moved2, tot2 = profiled_pass(h2, b2, x2, K, torch)
        form2, _ = h2.sweep_residual_form(1)
        kt2 = {}
        for (l, k), v in sorted(prof2.items()):
            if v[0] / tot2 > 0.02:
                a_ = v[0] / v[1]
                kt2[f"L{l}:{k}"] = {"avg_ms": round(a_, 5), "launches_per_step": round(v[1] / K, 2), "moved_MB": round(moved2[(l, k)] / 1e6, 2),
                                    "frac": round(moved2[(l, k)] / a_ / 1e6 / HBM_PEAK_GBS, 4)}
        roofline["divsiggrad"] = {
            "workload": f"nodal div sigma grad, log-normal sigma (seed 5), {cells}^3 cells + 1e-3 * max column sum * I, GMG V(2,1) Jacobi w=0.8, {p2.levels} levels "
                        "(Galerkin coarse operators), default format selection",
            "ms_per_step": round(dt2 / K * 1e3, 4), "dof_updates_per_s": round(n * K / dt2, 1),
            "relres_after_steps": float(res2[-1] / res2[0]), "fine_level_form": {4: "band form (two-stage pass, values streamed once)",
                                                                                 0: "two launches (CSR kernels)"}.get(form2, str(form2)),
            "host_setup_s": round(t_set2, 1), "kernels": kt2}
        # opt-in (mg_set_option band_sym_tol): symmetric reads on the 27-point Galerkin levels too, whose entries are symmetric only up to the
        # rounding of R*(A*P) - the operator applied then differs from the stored one at rounding level (the default above is bit-faithful)
        try:
            h2t = mg.device.DeviceHierarchy(p2, device_id=local_rank, nrhs=1, options={"band_sym_tol": 1})
            x2t = torch.zeros_like(b2)
            h2t.solve_dev(b2, x2t, 0.0, max(1, W))
            dtt = []
            for _ in range(3):
                x2t.zero_()
                barrier()
                t0 = time.perf_counter()
                h2t.solve_dev(b2, x2t, 0.0, K)
                barrier()
                dtt.append(time.perf_counter() - t0)
            roofline["divsiggrad"]["ms_per_step_with_band_sym_tol"] = round(sorted(dtt)[1] / K * 1e3, 4)
            roofline["divsiggrad"]["band_form_L2"] = {"default": h2.band_form(2), "band_sym_tol": h2t.band_form(2)}
            h2t.close()
            del x2t
        except Exception as e:
            roofline["divsiggrad"]["ms_per_step_with_band_sym_tol"] = f"{type(e).__name__}: {e}"
        # 16 right-hand sides on this hierarchy: the matrix streams are real here (no row classes), so the block SpMM kernels - one pass
        # over A for all columns - are what runs (the column-wise path needs the four-stage pass of a constant-coefficient fine level)
        if not args.no_c5_leg:
            try:
                k5, K5 = 16, max(2, min(K, 4))
                b16 = torch.from_numpy(np.ascontiguousarray(mg.seeded_rhs(A2, k5))).to(dev)
                x16 = torch.zeros_like(b16)
                h2.set_nrhs(k5)
                h2.solve_dev(b16, x16, 0.0, 1)
                d5 = []
                for _ in range(3):
                    x16.zero_()
                    barrier()
                    t0 = time.perf_counter()
                    it5, res5 = h2.solve_dev(b16, x16, 0.0, K5)
                    barrier()
                    d5.append(time.perf_counter() - t0)
                dt5 = sorted(d5)[1]
                roofline["divsiggrad"]["block16"] = {
                    "workload": "the same hierarchy, 16 right-hand sides: block SpMM kernels (csr_stream_spmm: A streamed once for all columns)",
                    "ms_per_step": round(dt5 / K5 * 1e3, 4), "dof_updates_per_s": round(n * k5 * K5 / dt5, 1), "steps": K5,
                    "sixteen_single_vector_steps_ms": round(16 * dt2 / K * 1e3, 4),
                    "speedup_over_column_by_column": round(16 * (dt2 / K) / (dt5 / K5), 3),
                    "relres_after_steps": float(res5[-1] / res5[0])}
                del b16, x16
            except Exception as e:
                roofline["divsiggrad"]["block16"] = {"error": f"{type(e).__name__}: {e}"}
        h2.close()
        del x2, b2, A2, p2

    # ---- BASELINE configs[2] as a leg of the driver's line: SA-AMG on anisotropic diffusion, general CSR, at
    # 128^3 cells (256^3 needs a minute of host setup: `--workload c3`).  Host setup as SA_AMGsetup does it, with its largest Galerkin
    # products on the GPU (MG_SETUP_GPU=1, opt-in: same pattern, values to rounding - the leg times the CYCLE, not the setup). ---------
    if args.workload == "c2" and nrhs == 1 and not args.no_c3_leg and not args.no_generic_pass and cells >= 128:
        keep_env = os.environ.get("MG_SETUP_GPU")
        try:
            import contextlib
            os.environ["MG_SETUP_GPU"] = "1"
            c3c, K3 = 128, max(2, min(K, 5))
            t0 = time.perf_counter()
            A3, _ = mg.anisotropic_divsiggrad([c3c] * 3, weights=(1.0, 0.25, 0.0625))
            p3 = mg.getMGparam(np.float64, np.int64, 14, os.cpu_count() or 8, K3, 0.0, "SPAI", 1.0, 1, 1, "V", "Julia", 0.4, 0.0)
            with contextlib.redirect_stdout(sys.stderr):
                mg.SA_AMGsetup(A3, p3, True, 1)
            t_set3 = time.perf_counter() - t0
            h3 = mg.device.DeviceHierarchy(p3, device_id=local_rank, nrhs=1)
            n3 = A3.shape[0]
            b3 = torch.from_numpy(np.ascontiguousarray(mg.seeded_rhs(A3, 1))).to(dev)
            x3 = torch.zeros_like(b3)
            h3.solve_dev(b3, x3, 0.0, 2)
            d3 = []
            for _ in range(3):
                x3.zero_()
                barrier()
                t0 = time.perf_counter()
                it3, res3 = h3.solve_dev(b3, x3, 0.0, K3)
                barrier()
                d3.append(time.perf_counter() - t0)
            dt3 = sorted(d3)[1]
            prof3, moved3, tot3 = profiled_pass(h3, b3, x3, K3, torch)
            lev3 = {}
            for (l, k), v in sorted(prof3.items()):
                e = lev3.setdefault(l, [0.0, 0.0])
                e[0] += v[0]
                e[1] += moved3[(l, k)] * v[1]          # (bytes per launch x launches)
            roofline["c3_leg"] = {
                "workload": f"SA-AMG (theta 0.4, V(1,1) SPAI) on anisotropic diffusion {c3c}^3 cells, edge weights 16:4:1 x log-normal sigma, general CSR",
                "ms_per_step": round(dt3 / K3 * 1e3, 4), "dof_updates_per_s": round(n3 * K3 / dt3, 1), "steps": K3, "N": int(n3),
                "levels": int(p3.levels), "level_rows": [int(a.shape[0]) for a in p3.As], "level_nnz": [int(a.nnz) for a in p3.As],
                "operator_complexity": round(sum(a.nnz for a in p3.As) / A3.nnz, 2), "relres_after_steps": float(res3[-1] / res3[0]),
                "host_setup_s": round(t_set3, 1), "setup": "SA_AMGsetup on the host, its largest Galerkin products on the GPU (MG_SETUP_GPU=1)",
                "per_level": {f"L{l}": {"ms_per_step": round(e[0] / K3, 4), "moved_MB_per_step": round(e[1] / K3 / 1e6, 1),
                                        "frac": round(e[1] / e[0] / 1e6 / HBM_PEAK_GBS, 4) if e[0] > 0 else None} for l, e in sorted(lev3.items())}}
            h3.close()
            del b3, x3, A3, p3
        except Exception as e:
            roofline["c3_leg"] = {"error": f"{type(e).__name__}: {e}"}
        finally:
            if keep_env is None:
                os.environ.pop("MG_SETUP_GPU", None)
            else:
                os.environ["MG_SETUP_GPU"] = keep_env

    # ---- CPU baseline: the C/OpenMP oracle ("port") on a bounded sample of the same workload ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, p, b_host, n, nrhs, cells, resvec)

    if rank == 0:
        out = {
            "metric": "V-cycle DoF-updates/s", "value": round(dof_per_s, 1), "unit": "DoF-updates/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4),
            "higher_is_better": True, "scaling": None, "vs_baseline": None, "dtype": "f64",   # (one GPU: neither weak nor strong)
            "data": "synthetic",
            "config": {"workload": f"{desc}, {p.levels} levels, nrhs={nrhs}, fp64"[:118],      # (kept under 120 characters: the driver's record cuts longer ones)
                       "step": "one step = one cycle + the residual and norm of solveMG's stopping test (SolveFuncs.jl:24-37)",
                       "cells": cells, "levels": p.levels, "nrhs": nrhs, "N": n, "nnz": int(A.nnz),
                       "level_rows": [int(a.shape[0]) for a in p.As], "level_nnz": [int(a.nnz) for a in p.As],
                       "parallelism": "1 process per GPU"},
            "timed_regions_ms_per_step": [round(r / K * 1e3, 4) for r in region],
            "timing": f"median of {len(region)} regions of exactly {K} steps each (barrier + synchronize on both sides)",
            "relres_after_steps": relres,
            "setup_s": {"operator": round(t_op, 2), "MGsetup": round(t_setup, 2), "upload": round(t_upload, 2)},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        emit(out)
    mg.clear_(p)
    if world > 1:
        dist.destroy_process_group()


def profiled_pass(h, b, x, K, torch):
    """K instrumented steps: every launch bracketed by HIP events on the library's stream."""
    h.profile_reset()
    h.profile_enable(True)
    x.zero_()
    torch.cuda.synchronize()
    h.solve_dev(b, x, 0.0, K)
    h.profile_enable(False)
    prof = h.profile()
    return prof, h.profile_moved(), sum(v[0] for v in prof.values())


def kernel_symbol(h, mg, p, l, nrhs):
    """(kernel symbol, device format) of the fused sweep on level l - the name rocprofv3 --stats lists."""
    D = mg.device
    ntl = "true" if 12.0 * p.As[l - 1].nnz > 128.0e6 else "false"
    if nrhs > 1:
        var, _ = h.operator_kernel_info(l, D.MG_OP_A)
        if var == 6:
            return "mgk::csr_rowclass_lane_spmm2<2>", "row classes (block right-hand sides, two columns per lane)"
        if var == 5:
            return "mgk::csr_rowclass_lane_spmm<2>", "row classes (block right-hand sides)"
        return f"mgk::csr_stream_spmm<2, {ntl}>", "plain CSR (block right-hand sides)"
    rc = h.operator_rowclasses(l, D.MG_OP_A)
    if rc[0] > 0:
        var, nexc = h.operator_kernel_info(l, D.MG_OP_A)
        inl = "true" if 0 < nexc <= 256 else "false"   # a short list of exception rows is handled in-kernel
        return ({0: "mgk::csr_rowclass_spmv<2, %s, false>", 1: "mgk::csr_rowclass_window_spmv<2, %s>",
                 2: "mgk::csr_rowclass_tile_spmv<2, %s>", 3: "mgk::csr_rowclass_march_spmv<2, %s, false>",
                 4: "mgk::csr_rowclass_lane_spmv<2, %s, true>"}[var] % inl,
                "row classes")
    fm = h.operator_format(l, D.MG_OP_A)
    if fm[0] > 0:
        dl = "true" if (fm[1] <= 1024 and fm[0] < 1024) else "false"
        return f"mgk::csr_pattern_spmv<2, {ntl}, {dl}>", "pattern-coded"
    return f"mgk::csr_stream_spmv<2, {ntl}>", "plain CSR"


def host_info():
    """Logical / physical cores and NUMA nodes of this host (Linux)."""
    logical = os.cpu_count() or 1
    phys, nodes = set(), 0
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
        nodes = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
    except Exception:
        pass
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = logical
    return {"logical": logical, "physical": len(phys) or logical, "numa_nodes": nodes or None, "usable": avail}


def reference_width_bytes(p, nrhs):
    """Bytes ONE solveMG step of the unfused reference sequence moves at the reference's widths (Int64 indices and
    pointers: 16 B/nnz + 8 B/row, every vector operand of every separate op; MGcycle.jl:26-102, SolveFuncs.jl:24-37),
    V-cycle, x != 0 on the fine level, x = 0 on the coarse ones."""
    tot = 0.0
    nl = len(p.As)

    def spmv(M, beta_nonzero):
        r, c = M.shape
        return nrhs * (16.0 * M.nnz + 8.0 * (r + 1) + 8.0 * c + (16.0 if beta_nonzero else 8.0) * r)

    for l in range(nl - 1):
        A = p.As[l]
        nloc = A.shape[0] * nrhs
        npre, npost = max(1, int(p.relaxPre(l + 1))), max(1, int(p.relaxPost(l + 1)))
        sweep = 32.0 * nloc + spmv(A, False) + 24.0 * nloc        # x += d.*r ; r = -A x ; r += b
        tot += 16.0 * nloc + 8.0 * nloc                           # r = b ; norm(x)
        if l == 0:
            tot += spmv(A, True)                                  # r -= A x
        tot += (npre - 1) * sweep + 32.0 * nloc                   # relax pre
        tot += spmv(A, False) + 24.0 * nloc                       # r = b - A x
        tot += 8.0 * p.As[l + 1].shape[0] * nrhs + spmv(p.Rs[l], False) + spmv(p.Ps[l], True)
        tot += 16.0 * nloc + spmv(A, True)                        # r = b ; r -= A x
        tot += (npost - 1) * sweep + 32.0 * nloc                  # relax post
    nc = p.As[-1].shape[0]
    tot += 8.0 * nc * nc + 24.0 * nc * nrhs
    A = p.As[0]
    tot += spmv(A, False) + 24.0 * A.shape[0] * nrhs + 8.0 * A.shape[0] * nrhs     # solveMG: r = b - A x ; norm
    return tot


def cpu_baseline(args, p, b_host, n, nrhs, cells, resvec_gpu):
    """The C/OpenMP oracle ("port") timed on this host: (a) all PHYSICAL cores, (b) numCores = 8, the reference's default
    (MGdef.jl:156).  The hierarchy, the scratch and b/x are first-touched by the threads that stream them."""
    from oracle import c_oracle
    info = host_info()
    cores = max(1, min(info["physical"], info["usable"], c_oracle.max_threads()))
    ref_bytes = reference_width_bytes(p, nrhs)
    out = {}
    for tag, nthr, ncyc in (("all", cores, args.cpu_cycles or (16 if cells >= 200 else 50)),
                            ("numCores8", min(8, cores), args.cpu_cycles or (3 if cells >= 200 else 20))):
        if nrhs > 1:
            ncyc = args.cpu_cycles or 1
        co = c_oracle.COracle(p, nrhs, first_touch_threads=nthr)
        co.solveMG_placed(b_host, 0.0, 1, nthr)          # warm-up cycle
        it, rv, _, tc = co.solveMG_placed(b_host, 0.0, ncyc, nthr)
        co.close()
        out[tag] = {"value": round(n * nrhs * it / tc, 1), "cores": nthr, "steps": it, "seconds": round(tc, 2),
                    "gbs_reference_widths": round(ref_bytes * it / tc / 1e9, 1)}
        if tag == "all":
            k = min(len(rv), len(resvec_gpu))
            out["resvec_rel_diff_vs_gpu"] = float(np.abs(rv[:k] - resvec_gpu[:k]).max() / resvec_gpu[0])
            out["relres_after_sample"] = float(rv[-1] / rv[0])
    a = out["all"]
    return {"value": a["value"], "unit": "DoF-updates/s", "cores": a["cores"], "kind": "port",
            "sample": f"{a['steps']} solveMG steps of the same workload ({cells}^3 cells, nrhs={nrhs}) on the C/OpenMP oracle: "
                      f"Int64 indices, unfused op sequence, hierarchy and vectors first-touched by the streaming threads, "
                      f"OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')}, {a['seconds']}s",
            "gbs_reference_widths": a["gbs_reference_widths"],
            "reference_width_bytes_per_step": ref_bytes,
            "numCores8": out["numCores8"], "host": info,
            "relres_after_sample": out["relres_after_sample"], "resvec_rel_diff_vs_gpu": out["resvec_rel_diff_vs_gpu"]}


def bench_weak(args, mg, torch, dist, cells, K, W, rank, world, local_rank):
    """N > 1, WEAK scaling: `cells`^3 per GPU; the global grid is cells x (boxes per dimension) - 512^3 on 8
    GPUs (BASELINE.json configs[3]).  No global matrix exists anywhere: every rank builds its part of the
    hierarchy on an overlapping box (multigrid.jl_amd/structured_setup.py)."""
    from multigrid_jl_amd import distributed as dd, structured_setup as ss
    share = os.environ.get("MG_BENCH_SHARE_GPU") == "1"
    dev = torch.device("cuda", local_rank)
    domains = dd.default_domains(world, 3)
    strong = args.scaling == "strong"
    if strong:      # the SAME cells^3 grid cut into `world` boxes (e.g. --cells 512: BASELINE configs[3] at any N)
        gcells = [cells] * 3
        domain = np.ravel([[0.0, 1.0]] * 3)
    else:           # cells^3 PER GPU: 512^3 on 8 GPUs
        gcells = [cells * d for d in domains]
        domain = np.ravel([[0.0, float(d)] for d in domains])      # h = 1/cells in every direction, as on one GPU
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
    Hpy = H
    native_note = None
    if not args.python_sequencer:      # the hot loop in C++ behind the C ABI (mg_dist_*), RCCL send/recv on a side stream
        ok = 1.0
        try:
            H = dd.NativeDistributedHierarchy(Hpy, transport="plugin" if share else "rccl")
        except Exception as e:          # (every rank must take the same path: agree on the outcome below)
            ok, native_note = 0.0, f"{type(e).__name__}: {e}"
            log(f"[rank {rank}] native sequencer unavailable: {native_note}")
        flag = torch.tensor([ok], device=red_dev, dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) < 1.0:
            if H is not Hpy:
                H.close()      # (gives the tail hierarchy back to the Python sequencer's stream)
            H = Hpy
            native_note = native_note or "another rank could not create the native sequencer"

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
    lb = torch.tensor([Hpy.local_algorithmic_bytes()], device=red_dev, dtype=torch.float64)
    dist.all_reduce(lb, op=dist.ReduceOp.MAX)
    rccl_ranks = H.comm_count() if hasattr(H, "comm_count") else None
    if rank == 0:
        ach = float(lb.item()) / (dt / K) / 1e9
        out = {
            "metric": "V-cycle DoF-updates/s", "value": round(n * K / dt, 1), "unit": "DoF-updates/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"3D 7-pt Poisson {gcells} cells" + ("" if strong else f" ({cells}^3 per GPU)") + f", GMG V(2,1) Jacobi w=0.8, {levels} levels, nrhs=1, fp64")[:118],
                       "step": "one step = one cycle + the residual and norm of solveMG's stopping test",
                       "cells_per_gpu": cells, "global_cells": gcells, "levels": levels, "nrhs": 1, "N": n,
                       "parallelism": f"DomainDecomposition boxes {domains}, {len(Hpy.levels)} sharded levels, replicated "
                                      f"tail from {Hpy.n_tail} rows, " +
                                      ("Python sequencer, one all_to_all_single halo exchange per SpMV (torch.distributed)"
                                       + (f" [native sequencer unavailable: {native_note}]" if native_note else "")
                                       if (args.python_sequencer or native_note) else
                                       "native C++ sequencer (mg_dist_*): ncclSend/ncclRecv halo exchange per SpMV on a side "
                                       "stream overlapped with the interior rows, one scalar all-reduce per step") +
                                      ", sharded host setup"},
            "relres_after_steps": float(resvec[-1] / resvec[0]),
            "rccl_comm_ranks": rccl_ranks,      # ncclCommCount of the sequencer's communicator (None/0: host-staged transport)
            "transport": ("plug-in (host-staged, ranks share one GPU)" if share else "RCCL") if not (args.python_sequencer or native_note) else "torch.distributed",
            "setup_s": {"sharded_setup_incl_upload": round(t_setup, 2)},
            "roofline": {"bound": "hbm", "kernel": "sharded levels of one V-cycle, per GPU (max over ranks)",
                         "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                         "csr_equivalent": {"achieved": round(ach, 1), "ratio_to_peak": round(ach / HBM_PEAK_GBS, 4)},
                         "note": "per-kernel bandwidth fractions are reported by the N=1 line; here only the CSR-priced "
                                 "(SURVEY 8d) bytes of the rank's sharded levels / step time: the local operators of this "
                                 "constant-coefficient workload are stored as row classes (no matrix stream), so that is "
                                 "traffic avoided, not a bandwidth"},
            "cpu_baseline": None,
        }
        if os.environ.get("MG_BENCH_CHILD") != "1":
            attach_strong_reference(out, STRONG_REF)
        emit(out)
    dist.barrier()
    dist.destroy_process_group()


def bench_ghost(args, mg, torch, dist, cells, K, W, rank, world, local_rank):
    """N > 1 in the GHOST-LAYER form (multigrid.jl_amd/ghost_dist.py, csrc/mg_ghost.inc): every rank runs the single-GPU kernels - the
    four-stage pass, the 27-point marching form, the marching restriction, the pipelined stopping test - on its box extended by ghost
    layers, one exchange per fused pass.  Weak (cells^3 per GPU: 512^3 on 8 GPUs = BASELINE configs[3]) or strong (--scaling strong:
    the same cells^3 grid in N boxes).  --force-sharded-path: a world of one through the same code (and RCCL); --ghost-dry R/N: rank R
    of N alone on this GPU, nothing travels (one GPU's compute share, redundant ghost rows included)."""
    from multigrid_jl_amd import distributed as dd, ghost_dist as gd, structured_setup as ss
    share = os.environ.get("MG_BENCH_SHARE_GPU") == "1"
    dev = torch.device("cuda", local_rank)
    dry = None
    if args.ghost_dry:
        dry = tuple(int(v) for v in args.ghost_dry.split("/"))
        if world != 1 or len(dry) != 2 or not (0 <= dry[0] < dry[1]):
            raise SystemExit("--ghost-dry R/N needs ONE process and 0 <= R < N")
    grank, gworld = (dry if dry else (rank, world))
    domains = dd.default_domains(gworld, 3)
    strong = args.scaling == "strong"
    if strong:
        gcells = [cells] * 3
        domain = np.ravel([[0.0, 1.0]] * 3)
    else:
        gcells = [cells * d for d in domains]
        domain = np.ravel([[0.0, float(d)] for d in domains])      # h = 1/cells in every direction, as on one GPU
    levels = args.levels or levels_for(min(gcells))
    p = mg.getMGparam(np.float64, np.int64, levels, os.cpu_count() or 8, K, 0.0, "Jac", 0.8, 2, 1, "V",
                      "NoMUMPS", 0.5, 0.0, "FullWeighting")
    t0 = time.perf_counter()
    G = gd.ghost_gmg(gcells, domains, grank, gworld, p, ss.poisson_operator(gcells, domain), domain=domain, nrhs=1, dry_tail=bool(dry))
    t_host = time.perf_counter() - t0
    t0 = time.perf_counter()
    transport = "dry" if dry else ("plugin" if share else "rccl")
    H = gd.NativeGhostHierarchy(G, local_rank, transport=transport)
    t_up = time.perf_counter() - t0
    n = int(np.prod(np.asarray(gcells) + 1))
    b_ext, ssq = gd.local_rhs(G)
    red_dev = torch.device("cpu") if share else dev
    tot = torch.tensor([ssq], device=red_dev, dtype=torch.float64)
    if dist is not None and not dry:
        dist.all_reduce(tot)
    b = torch.from_numpy(np.ascontiguousarray(b_ext / float(tot.item()) ** 0.5)).to(dev)
    x = torch.zeros_like(b)
    L0 = G.levels[0]
    n_own = int(np.prod([L0.own_hi[k] - L0.own_lo[k] for k in range(3)]))
    ok4, geo4 = H.dev.four_stage_form(1)
    log(f"[rank {grank}/{gworld}] global {gcells} cells over boxes {domains}: extended box {L0.ext_n} = {L0.n} rows for {n_own} owned "
        f"({L0.n / n_own:.3f} x), {G.a} sharded levels (ghost layers {[L.gmin for L in G.levels]}) + replicated levels from {G.n_tail} rows; "
        f"four-stage pass {'on' if ok4 else 'OFF'}; host setup {t_host:.1f}s, upload {t_up:.1f}s, HBM {H.dev.device_bytes() / 1e9:.2f} GB")

    def barrier():
        torch.cuda.synchronize()
        if dist is not None and not dry:
            dist.barrier()
            torch.cuda.synchronize()

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.prewarm and (dist is None or dry or world == 1):
        H.dev.time_op(1, mg.device.MG_K_RESIDUAL, 50)
    for _ in range(10 if (args.prewarm > 0 and world > 1) else 0):      # fixed count: every rank enters the same exchanges
        H.cycle(b, x, False)
    if W > 0:
        x.zero_()
        H.solve(b, x, 0.0, W)
    region = []
    e0 = H.exchanges()
    for _ in range(max(1, args.repeats)):
        x.zero_()
        barrier()
        t0 = time.perf_counter()
        iters, resvec = H.solve(b, x, 0.0, K)
        barrier()
        region.append(time.perf_counter() - t0)
        assert iters == K
    e1 = H.exchanges()
    reg = torch.tensor(region, device=red_dev, dtype=torch.float64)
    if dist is not None and not dry:
        dist.all_reduce(reg, op=dist.ReduceOp.MAX)      # every region: the slowest rank
    region = [float(v) for v in reg.cpu().numpy()]
    dt = float(np.median(region))
    nreg = max(1, args.repeats)
    per_step_exch = (e1[0] - e0[0]) / (nreg * K)
    per_step_sent = (e1[1] - e0[1]) * 8.0 / (nreg * K)
    rccl_ranks = H.comm_count()
    kern_table = None
    if world == 1:      # (one process: a world of one, or a dry rank) where the step goes, launch by launch (instrumented pass: graphs off, events on)
        prof, moved, tot_ms = profiled_pass(H.dev, b, x, K, torch)
        kern_table = {f"L{l}:{k}": {"avg_us": round(v[0] / v[1] * 1e3, 2), "launches_per_step": round(v[1] / K, 2), "us_per_step": round(v[0] / K * 1e3, 1)}
                      for (l, k), v in sorted(prof.items())}
        kern_table["kernel_ms_per_step"] = round(tot_ms / K, 4)
    rows_ext = torch.tensor([float(sum(L.n for L in G.levels)), float(L0.n) / n_own], device=red_dev, dtype=torch.float64)
    if dist is not None and not dry:
        dist.all_reduce(rows_ext, op=dist.ReduceOp.MAX)
    # the same box through the plain single-GPU path, same process, same GPU (a world of one only: the extended box IS the grid)
    same_job = None
    if world == 1 and not dry and not os.environ.get("MG_BENCH_NO_SAME_JOB"):
        try:
            H.close()
            A1, mesh1 = mg.poisson_shifted(gcells, [v for g in gcells for v in (0.0, g / float(cells))])
            p1 = mg.getMGparam(np.float64, np.int64, levels, os.cpu_count() or 8, K, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0, "FullWeighting")
            mg.MGsetup(A1, mesh1, p1, 1)
            h1 = mg.to_device(p1, device_id=local_rank)
            b1 = torch.from_numpy(np.ascontiguousarray(mg.seeded_rhs(A1, 1))).to(dev)
            x1 = torch.zeros_like(b1)
            if W > 0:
                h1.solve_dev(b1, x1, 0.0, W)
            r1 = []
            for _ in range(nreg):
                x1.zero_()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                it1, rv1 = h1.solve_dev(b1, x1, 0.0, K)
                torch.cuda.synchronize()
                r1.append(time.perf_counter() - t0)
            d1 = float(np.median(r1))
            same_job = {"single_gpu_ms_per_step": round(d1 / K * 1e3, 4), "sharded_over_single": round(dt / d1, 4),
                        "resvec_rel_diff": float(np.abs(np.asarray(rv1) - np.asarray(resvec)).max() / rv1[0]),
                        "note": "the same grid through the plain single-GPU path (mg_create ... mg_solve_dev_FP64), same process, same GPU, right after the sharded run"}
        except Exception as e:
            same_job = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        out = {
            "metric": "V-cycle DoF-updates/s", "value": round((n_own if dry else n) * K / dt, 1), "unit": "DoF-updates/s",
            "n_gpus": 1 if dry else world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4),
            "timed_regions_ms_per_step": [round(v / K * 1e3, 4) for v in region],
            "higher_is_better": True, "scaling": (None if (world == 1) else args.scaling), "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"3D 7-pt Poisson {gcells} cells" + ("" if strong else f" ({cells}^3 per GPU)") + f", GMG V(2,1) Jacobi w=0.8, {levels} levels, nrhs=1, fp64")[:118],
                       "step": "one step = one cycle + the residual and norm of solveMG's stopping test",
                       "cells_per_gpu": cells, "global_cells": gcells, "levels": levels, "nrhs": 1, "N": n,
                       "parallelism": f"DomainDecomposition boxes {domains} with ghost layers (the reference's overlap, DDIndices.jl:61-92): {G.a} sharded levels on "
                                      f"extended boxes, ghost widths {[L.gmin for L in G.levels]}, replicated levels from {G.n_tail} rows; every rank runs the "
                                      f"single-GPU kernels (four-stage pass {'on' if ok4 else 'off'}), one exchange per fused pass on a side stream, one "
                                      f"all-reduce into the first replicated level, norms over owned rows; sharded host setup"},
            "relres_after_steps": float(resvec[-1] / resvec[0]),
            "sharded_form": "ghost layers (mg_ghost_*)",
            "transport": {"dry": "none (dry run: ONE rank of the world alone on its GPU - timing of its compute share only, numbers meaningless)",
                          "plugin": "plug-in (host-staged, ranks share one GPU)", "rccl": "RCCL"}[transport],
            "rccl_comm_ranks": rccl_ranks,
            "ghost": {"extended_over_owned_rows_fine": round(float(rows_ext[1].item()), 4), "exchanges_per_step": round(per_step_exch, 3),
                      "bytes_sent_per_step_per_rank": round(per_step_sent, 1), "fine_extended_box": [int(v) for v in L0.ext_n],
                      "four_stage_geometry": geo4},
            "setup_s": {"host": round(t_host, 2), "upload": round(t_up, 2)},
            "roofline": {"bound": "hbm", "kernel": "per-kernel fractions are reported by the N = 1 line (the same kernels run here)", "achieved": None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None},
            "cpu_baseline": None,
        }
        if dry:
            out["dry_run"] = {"rank": grank, "world": gworld, "owned_rows": n_own,
                              "note": "value = owned rows x steps / time of THIS rank alone; x world = the N-GPU rate with free communication"}
        if kern_table is not None:
            out["kernels"] = kern_table
        if same_job is not None:
            out["same_job_single_gpu"] = same_job
        if os.environ.get("MG_BENCH_CHILD") != "1" and not dry:
            attach_strong_reference(out, STRONG_REF)
        emit(out)
    if dist is not None and not dry:
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
            "config": {"workload": f"3D 7-pt Poisson {cells}^3 cells, GMG V(2,1) Jacobi w=0.8, {p.levels} levels, nrhs={nrhs}, fp64"[:118],
                       "step": "one step = one cycle + the residual and norm of solveMG's stopping test",
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
        emit(out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
