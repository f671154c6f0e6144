"""Multi-GPU path (multigrid.jl_amd/distributed.py).

CPU (-m "not gpu"): world_size-2 and -4 ``gloo`` runs with the test-only checker backend
(tests/dist_cpu_backend.py): partition, halo plans, exchange, replicated tail and the cycle schedule must
reproduce the oracle's single-process solveMG on the same seeded problem.
GPU (-m gpu): the same schedule with the HIP kernels - world_size 1 in-process and world_size 2 as two
processes sharing the one GPU of the box (gloo with host staging stands in for RCCL).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(kind, nrhs, cyc):
    import multigrid_jl_amd as mg
    if kind == "gmg3d":
        A, mesh = mg.poisson_shifted([16, 16, 16])
        p = mg.getMGparam(np.float64, np.int64, 4, 8, 6, 1e-10, "Jac", 0.8, 2, 1, cyc, "NoMUMPS", 0.5, 0.0)
        mg.MGsetup(A, mesh, p, nrhs)
        nodes = mesh.n + 1
    elif kind == "gmg2d":
        A, mesh = mg.poisson_shifted([40, 24])
        p = mg.getMGparam(np.float64, np.int64, 3, 8, 6, 1e-10, "SPAI", 1.0, 1, 1, cyc, "NoMUMPS", 0.5, 0.0)
        mg.MGsetup(A, mesh, p, nrhs)
        nodes = mesh.n + 1
    elif kind == "gmg3d-jacgmres":   # Jac-GMRES smoother (FGMRES_relaxation, MGcycle.jl:48-50,96-98): native sequencer only
        A, mesh = mg.poisson_shifted([16, 16, 16])
        p = mg.getMGparam(np.float64, np.int64, 4, 8, 6, 1e-10, "Jac-GMRES", 0.75, 2, 2, cyc, "NoMUMPS", 0.5, 0.0)
        mg.MGsetup(A, mesh, p, nrhs)
        nodes = mesh.n + 1
    else:  # "sa": general CSR, contiguous row blocks
        A, _ = mg.anisotropic_divsiggrad([12, 12, 12], weights=(1, 0.5, 0.25))
        p = mg.getMGparam(np.float64, np.int64, 4, 8, 6, 1e-10, "SPAI", 1.0, 1, 1, cyc, "Julia", 0.4)
        mg.SA_AMGsetup(A, p, True, nrhs)
        nodes = None
    b = mg.seeded_rhs(A, nrhs)
    return mg, A, p, b, nodes


def _worker(rank, world, port, kind, nrhs, cyc, use_hip, q, backend="gloo", native=None, box=False):
    if box and box != "plain" and use_hip:       # let the small local operators of the test take the row-class / staged kernels
        os.environ.update(MG_NO_SMALL="1", MG_ROWCLASS_MIN_ROWS="0", MG_ROWCLASS_MAX_PASSES="64", MG_ROWCLASS_MIN_COVER="0.3",
                          MG_MARCH_MIN_WG="0", MG_TILE_MIN_WG="0", MG_WINDOW_MIN_WG="0", MG_MARCH_MAX_LEN="64",
                          MG_WINP_MIN_ROWS="0")
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        if backend == "nccl":
            torch.cuda.set_device(0)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        from multigrid_jl_amd import distributed as dd
        mg, A, p, b, nodes = _problem(kind, nrhs, cyc)
        if nodes is not None:
            owner = dd.box_owner(nodes, dd.default_domains(world, len(nodes)))
        else:
            owner = dd.block_owner(A.shape[0], world)
        if use_hip:
            be = dd.HipBackend(0)
            comm = dd.TorchComm(stage_through_host=(backend != "nccl"))
        else:
            from dist_cpu_backend import CpuCheckerBackend
            be = CpuCheckerBackend()
            comm = dd.TorchComm()
        level_nodes = None
        if box and nodes is not None:      # BOX form of the sharded levels: rows in natural box order, A one square operator
            level_nodes = [((np.asarray(nodes) - 1) >> l) + 1 for l in range(len(p.As))]
        H = dd.DistributedHierarchy.from_global(p, comm, be, owner, nrhs, replicate_below=200, level_nodes=level_nodes,
                                                native_only=bool(native) and (cyc == "K" or p.relaxType == "Jac-GMRES"))
        assert len(H.levels) >= 2, "the test must exercise at least two sharded levels"
        assert H.box_form == bool(box and nodes is not None and nrhs == 1)
        if box and box != "plain" and use_hip and nodes is not None and nrhs == 1:
            var = [L.A_int.kernel_variant() for L in H.levels]
            assert var[0][0] in (1, 2, 3) and (world == 1 or var[0][1] > 0), var   # a staged kernel + exception rows
            if len(nodes) == 3:
                assert var[0][0] == 3, var                                     # 3-D fine level: z-marching
            pv = [L.P.kernel_variant() for L in H.levels]
            assert pv[0][0] == 5, pv                  # grid form of P: coarse windows staged in LDS ...
            assert world == 1 or rank != 0 or pv[0][1] > 0, pv   # ... + the rows that read halo columns (the interface nodes
                                                                 # belong to the upper box: rank 0 reads them), behind the exchange
        S = H
        if native:          # the same local operators and plans, the loop in C++ (mg_dist_*): "plugin" or "rccl" transport
            S = dd.NativeDistributedHierarchy(H, transport=native)
        b_loc = H.scatter_fine(b)
        x_loc = torch.zeros_like(b_loc)
        it, resvec = S.solve(b_loc, x_loc, 1e-10, 6)
        # one more single cycle from the non-zero x through the public cycle() entry
        x2 = x_loc.clone()
        S.cycle(b_loc, x2, False)
        be.synchronize()
        out = [None] * world
        dist.all_gather_object(out, (H.rows_fine, x_loc.cpu().numpy(), x2.cpu().numpy()))   # (NCCL: via cuda:0)
        if rank == 0:
            x = np.zeros_like(b)
            xc = np.zeros_like(b)
            for rows, xl, x2l in out:
                x[rows] = xl
                xc[rows] = x2l
            q.put(("ok", it, resvec, x, xc))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc()))
        raise


def _run(world, kind, nrhs, cyc, use_hip=False, backend="gloo", native=None, box=False):
    from oracle import mg_oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, nrhs, cyc, use_hip, q, backend, native, box)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = q.get(timeout=300)
    for pr in procs:
        pr.join(timeout=120)
    assert res[0] == "ok", res[1]
    _, it, resvec, x, xc = res
    mg, A, p, b, _ = _problem(kind, nrhs, cyc)
    xo = np.zeros_like(b)
    hist = {}
    _, _, ito = orc.solveMG(p, b, xo, False, hist)
    assert it == ito
    assert np.abs(resvec - hist["resvec"]).max() / hist["resvec"][0] < 1e-10
    assert np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()
    xo2 = orc.recursiveCycle(p, b, xo.copy(), 1)
    assert np.abs(xc - xo2).max() <= 1e-10 * np.abs(xo2).max()


# ---- CPU: host logic --------------------------------------------------------------------------------------
def test_box_owner_follows_dd_rule():
    from multigrid_jl_amd import distributed as dd
    own = dd.box_owner([9, 9], [2, 2]).reshape(9, 9)            # 8x8 cells -> cellSize 4
    assert own[0, 0] == 0 and own[0, 3] == 0 and own[0, 4] == 1 and own[0, 8] == 1
    assert own[4, 0] == 2 and own[8, 8] == 3                        # loc2cs: x fastest (DDService.jl:27-36)
    own = dd.box_owner([8, 6], [3, 1])                              # 7 cells / 3 -> cellSize 2, last box takes the rest
    assert np.array_equal(own[:8], [0, 0, 1, 1, 2, 2, 2, 2])
    own3 = dd.box_owner([5, 5, 5], [2, 2, 2])
    assert sorted(set(own3)) == list(range(8)) and own3.size == 125
    assert dd.default_domains(8, 3) == [2, 2, 2] and dd.default_domains(2, 3) == [1, 1, 2]


def test_halo_plan_reproduces_global_spmv():
    """Without any process group: emulate the exchange of all ranks in one process."""
    import multigrid_jl_amd as mg
    from multigrid_jl_amd import distributed as dd
    A, mesh = mg.poisson_shifted([10, 8, 6])
    nr = 4
    part = dd.Partition(dd.box_owner(mesh.n + 1, [1, 2, 2]), nr)
    x = np.random.default_rng(0).standard_normal(A.shape[0])
    y = A @ x
    locs = [dd.localize(A, part, part, r) for r in range(nr)]
    for r in range(nr):
        Ml, plan = locs[r]
        assert sum(plan.recv_splits) == plan.n_halo
        halo = []
        for q in range(nr):                                         # what q sends to r, in q's send order
            Mq, pq = locs[q]
            off = int(np.sum(pq.send_splits[:r]))
            idx = pq.send_idx[off: off + pq.send_splits[r]]
            assert len(idx) == plan.recv_splits[q]
            halo.append(x[part.rows[q]][idx])
        xl = np.concatenate([x[part.rows[r]]] + halo)
        assert np.allclose(Ml @ xl, y[part.rows[r]], rtol=1e-14, atol=1e-14)


def test_coarse_ownership_follows_coincident_node():
    import multigrid_jl_amd as mg
    from multigrid_jl_amd import distributed as dd
    P, nc = mg.getFWInterp([9, 9, 9])
    part = dd.Partition(dd.box_owner([9, 9, 9], [2, 2, 2]), 8)
    cpart = part.coarsen(P)
    fine = part.owner.reshape(9, 9, 9)[::2, ::2, ::2].reshape(-1)
    assert np.array_equal(cpart.owner, fine)


@pytest.mark.parametrize("world,kind,nrhs,cyc", [(2, "gmg3d", 1, "V"), (2, "gmg3d", 3, "W"), (4, "gmg3d", 1, "F"),
                                                 (2, "gmg2d", 1, "V"), (2, "sa", 2, "V")])
def test_gloo_distributed_solve_matches_oracle(built, world, kind, nrhs, cyc):
    _run(world, kind, nrhs, cyc, use_hip=False)


@pytest.mark.parametrize("world,kind,cyc", [(2, "gmg3d", "V"), (4, "gmg3d", "W"), (2, "gmg2d", "V"), (8, "gmg3d", "V")])
def test_gloo_distributed_box_form(built, world, kind, cyc):
    """BOX form of the sharded levels (rows in natural box order, A one square operator [owned box | halo], rows that
    read the halo computed after the exchange): host logic under the checker backend."""
    _run(world, kind, 1, cyc, use_hip=False, box=True)


# ---- GPU: the HIP kernels under the same schedule ---------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("world,kind,nrhs,cyc", [(2, "gmg3d", 4, "W"), (2, "sa", 1, "V")])
def test_hip_distributed_solve_matches_oracle(built, world, kind, nrhs, cyc):
    _run(world, kind, nrhs, cyc, use_hip=True)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_hip_distributed_rowclass_with_exception_rows(built, world, monkeypatch):
    """Local operators of a sharded level are regular except next to the sub-domain faces (rows renumbered into the
    [interior | boundary] order, halo columns appended): with the size thresholds lifted they are stored as row classes
    plus EXCEPTION rows (csr_rows_spmv) and the sharded solve must still match the oracle."""
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_ROWCLASS_MIN_COVER", "0.2")
    monkeypatch.setenv("MG_ROWCLASS_KEEP_SINGLETONS", "0")
    _run(world, "gmg3d", 1, "V", use_hip=True)


# ---- the native sequencer behind the C ABI (mg_dist_*) ----------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("world,kind,cyc", [(2, "gmg3d", "W"), (2, "gmg3d", "F"), (2, "sa", "V")])
def test_native_sequencer_plugin_transport(built, world, kind, cyc):
    """mg_dist_* with the host-staged exchange plug-in (gloo underneath), `world` fresh processes sharing the one GPU:
    the C++ level schedule, pack kernels, interior/boundary split and replicated tail must reproduce the oracle."""
    _run(world, kind, 1, cyc, use_hip=True, native="plugin")


@pytest.mark.gpu
@pytest.mark.parametrize("world,kind,nrhs,cyc", [(2, "sa", 2, "V"), (2, "gmg3d", 16, "V")])
def test_native_sequencer_blocks_of_right_hand_sides(built, world, kind, nrhs, cyc):
    """mg_dist_* with nrhs > 1 (MGdef.jl:163-176: the reference is block-capable everywhere; one Frobenius criterion for the
    block, SolveFuncs.jl:30): row-major [n][nrhs] blocks through the pack kernels, the halo exchange, the SpMM kernels, the
    padded all-gather into the replicated tail - GMG and SA-AMG (general CSR, contiguous row blocks) hierarchies, against the
    oracle's block solve."""
    _run(world, kind, nrhs, cyc, use_hip=True, native="plugin")


@pytest.mark.gpu
@pytest.mark.parametrize("world,kind,cyc,native", [(4, "gmg3d", "F", "plugin"), (2, "gmg3d", "W", None), (2, "gmg2d", "V", "plugin")])
def test_box_form_local_operators_hip(built, world, kind, cyc, native):
    """Sharded levels in BOX form on the device: the local A is one square grid operator (z-marching / plane-tile kernels
    for the rows of the owned box that do not read the halo, csr_rows_spmv for those that do, after the exchange), driven
    by the native sequencer (phase-split launches around the side-stream exchange) and by the Python one."""
    _run(world, kind, 1, cyc, use_hip=True, native=native, box=True)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_native_box_form_default_thresholds(built, world):
    """Box-form levels WITHOUT the row-class overrides of the tests above: local operators of fewer than
    `rowclass_min_rows` rows (and every variable-coefficient operator) are not stored as row classes, so phase 1 of the
    phase-split launches computes nothing and phase 2 the whole product.  The fused residual + norm of the native solve
    loop must then sum phase 2's partials only (round-2 ADVICE: it added stale partials of the previous reduction)."""
    _run(world, "gmg3d", 1, "V", use_hip=True, native="plugin", box="plain")


@pytest.mark.gpu
@pytest.mark.parametrize("world,kind,cyc,box", [(2, "gmg3d", "K", False), (2, "gmg3d-jacgmres", "W", False), (2, "gmg3d-jacgmres", "K", True)])
def test_native_sequencer_kcycle_and_jac_gmres(built, world, kind, cyc, box):
    """The K-cycle (2 FGMRES steps per level preconditioned by the next level's K-cycle, MGcycle.jl:72-76) and the Jac-GMRES
    smoother (FGMRES.jl:48-126) in the native sharded sequencer: products with the halo exchanged, dots all-reduced, the
    data-dependent exits identical on every rank; the level above the replicated tail lands in mg_kcycle_step_async_dev_FP64.
    Against the oracle (1e-10).  The Python sequencer refuses both (loudly)."""
    _run(world, kind, 1, cyc, use_hip=True, native="plugin", box=box)


@pytest.mark.gpu
def test_native_sequencer_rccl_world1(built):
    """The RCCL transport with the one rank a single-GPU box allows: ncclCommInitRank from the library's own unique id,
    ncclAllReduce / ncclAllGather on the compute stream, side stream and events created (no peer to send to)."""
    _run(1, "gmg3d", 1, "V", use_hip=True, backend="nccl", native="rccl")


# ---- sharded SETUP (structured_setup.py): every rank builds only its part -------------------------------
def _worker_structured(rank, world, port, cells, levels, cyc, nrhs, use_hip, q):
    try:
        if not use_hip:      # the operator-by-operator comparison below is written for the [interior | boundary] form
            os.environ["MG_DIST_NO_BOX"] = "1"
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import multigrid_jl_amd as mg
        from multigrid_jl_amd import distributed as dd, structured_setup as ss
        if use_hip:
            be, comm = dd.HipBackend(0), dd.TorchComm(stage_through_host=True)
        else:
            from dist_cpu_backend import CpuCheckerBackend
            be, comm = CpuCheckerBackend(), dd.TorchComm()
        doms = dd.default_domains(world, len(cells))
        p = mg.getMGparam(np.float64, np.int64, levels, 8, 5, 1e-10, "Jac", 0.8, 2, 1, cyc, "NoMUMPS", 0.5, 0.0)
        H, info = ss.structured_gmg(cells, doms, comm, be, p, ss.poisson_operator(cells), nrhs=nrhs, replicate_below=1000)
        # reference: the global hierarchy cut by from_global on the same partition
        A, mesh = mg.poisson_shifted(cells)
        pg = mg.getMGparam(np.float64, np.int64, levels, 8, 5, 1e-10, "Jac", 0.8, 2, 1, cyc, "NoMUMPS", 0.5, 0.0)
        mg.MGsetup(A, mesh, pg, nrhs)
        owner = dd.box_owner(mesh.n + 1, doms)
        if not use_hip:
            G = dd.DistributedHierarchy.from_global(pg, comm, be, owner, nrhs, replicate_below=1000)
            assert len(G.levels) == len(H.levels) == info["sharded_levels"] >= 2
            assert np.array_equal(G.rows_fine, H.rows_fine)
            for Lg, Lh in zip(G.levels, H.levels):
                assert Lg.n_int == Lh.n_int
                for og, oh in ((Lg.A_int, Lh.A_int), (Lg.A_bnd, Lh.A_bnd), (Lg.R, Lh.R), (Lg.P, Lh.P)):
                    assert og.shape == oh.shape and og.nnz == oh.nnz
                    assert np.array_equal(og.indptr, oh.indptr) and np.array_equal(og.indices, oh.indices)
                    assert np.allclose(og.data, oh.data, rtol=1e-13, atol=0)
                assert np.allclose(Lg.d.numpy(), Lh.d.numpy(), rtol=1e-13)
                for pg_, ph_ in ((Lg.planA, Lh.planA), (Lg.planR, Lh.planR), (Lg.planP, Lh.planP)):
                    if pg_ is None:
                        assert ph_ is None
                        continue
                    assert pg_.send_splits == ph_.send_splits and pg_.recv_splits == ph_.recv_splits
                    assert np.array_equal(pg_.send_idx, ph_.send_idx)
        # the right-hand side built locally equals the global seeded one
        b_glob = mg.seeded_rhs(A, nrhs)
        b_own, ss2 = ss.local_rhs(info, nrhs)
        tot = torch.tensor([ss2], dtype=torch.float64)
        dist.all_reduce(tot)
        b_own = H.order_fine(b_own) / float(tot.item()) ** 0.5
        assert np.allclose(b_own, b_glob[H.rows_fine], rtol=1e-12, atol=1e-15)
        b_loc = be.from_numpy(b_own)
        x_loc = torch.zeros_like(b_loc)
        it, resvec = H.solve(b_loc, x_loc, 1e-10, 5)
        be.synchronize()
        out = [None] * world
        dist.all_gather_object(out, (H.rows_fine, x_loc.cpu().numpy()))
        if rank == 0:
            x = np.zeros_like(b_glob)
            for rows, xl in out:
                x[rows] = xl
            q.put(("ok", it, resvec, x))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc()))
        raise


def _run_structured(world, cells, levels, cyc, nrhs, use_hip=False):
    from oracle import mg_oracle as orc
    import multigrid_jl_amd as mg
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_structured, args=(r, world, port, cells, levels, cyc, nrhs, use_hip, q))
             for r in range(world)]
    for pr in procs:
        pr.start()
    res = q.get(timeout=600)
    for pr in procs:
        pr.join(timeout=120)
    assert res[0] == "ok", res[1]
    _, it, resvec, x = res
    A, mesh = mg.poisson_shifted(cells)
    p = mg.getMGparam(np.float64, np.int64, levels, 8, 5, 1e-10, "Jac", 0.8, 2, 1, cyc, "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p, nrhs)
    b = mg.seeded_rhs(A, nrhs)
    xo = np.zeros_like(b)
    hist = {}
    orc.solveMG(p, b, xo, False, hist)
    assert np.abs(resvec - hist["resvec"]).max() / hist["resvec"][0] < 1e-10
    assert np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()


@pytest.mark.parametrize("world,cells,levels,cyc,nrhs", [(2, [32, 32, 32], 4, "V", 1), (4, [32, 32, 32], 4, "W", 2),
                                                        (2, [128, 64], 4, "V", 1), (8, [32, 32, 32], 4, "V", 1)])
def test_structured_setup_equals_global(built, world, cells, levels, cyc, nrhs):
    _run_structured(world, cells, levels, cyc, nrhs)


@pytest.mark.gpu
def test_hip_distributed_over_rccl_world1(built):
    """The un-staged RCCL code path (device tensors straight into torch.distributed 'nccl': all_gather_into_tensor,
    all_reduce, all_gather_object) with the one rank a single-GPU box allows."""
    _run(1, "gmg3d", 1, "V", use_hip=True, backend="nccl")


@pytest.mark.gpu
def test_structured_setup_hip(built):
    _run_structured(2, [32, 32, 32], 4, "V", 1, use_hip=True)


def _worker_c4box(rank, world, port, cells, levels, q):
    """One rank of the C4-sized run: sharded host setup of its 257^3 box, native sequencer, plug-in transport."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["MG_HOST_THREADS"] = str(max(1, (os.cpu_count() or 2) // world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import multigrid_jl_amd as mg
        from multigrid_jl_amd import distributed as dd, structured_setup as ss
        be, comm = dd.HipBackend(0), dd.TorchComm(stage_through_host=True)
        doms = dd.default_domains(world, 3)
        p = mg.getMGparam(np.float64, np.int64, levels, 8, 2, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
        H, info = ss.structured_gmg(cells, doms, comm, be, p, ss.poisson_operator(cells), nrhs=1)
        assert H.box_form and len(H.levels) >= 2
        b_own, ss2 = ss.local_rhs(info, 1)
        tot = torch.tensor([ss2], dtype=torch.float64)
        dist.all_reduce(tot)
        b_loc = be.from_numpy(H.order_fine(b_own) / float(tot.item()) ** 0.5)
        x_loc = torch.zeros_like(b_loc)
        S = dd.NativeDistributedHierarchy(H, transport="plugin")
        yes, l1, l2 = H.levels[0].A_int.can_sweep_residual(H.levels[0].x0, H.levels[0].d)
        it, resvec = S.solve(b_loc, x_loc, 0.0, 2)
        be.synchronize()
        sums = torch.tensor([float(x_loc.sum().item()), float((x_loc * x_loc).sum().item())], dtype=torch.float64)
        dist.all_reduce(sums)
        if rank == 0:
            q.put(("ok", it, resvec, sums.numpy(), (bool(yes), l1, l2, H.levels[0].n_own)))
        dist.barrier()
        S.close()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc()))
        raise


@pytest.mark.gpu
def test_halo_form_large_boxes_two_ranks_vs_c_oracle(built):
    """The HALO form (mg_dist_*: general CSR, block cycles, K-cycle, Jac-GMRES) on two 129^3-node boxes - 128 x 128 x 256 cells,
    4.3 M rows - on two ranks sharing this box's one GPU (host-staged plug-in transport): sharded host setup, box-form levels, the
    fused sweep + residual pairs with their face-layer list kernels, replicated tail; two solveMG steps against the C/OpenMP
    oracle on the global hierarchy (residual history to 1e-10, sum and norm of the iterate).  BASELINE.json configs[3]'s per-GPU
    box size (two 257^3 boxes) runs in the ghost-layer form, the default sharded path:
    tests/test_ghost_dist.py::test_c4_per_gpu_box_size_two_ranks_ghost_form_vs_c_oracle (until round 5 it ran in both forms)."""
    from oracle import c_oracle
    import multigrid_jl_amd as mg
    cells, levels, world = [128, 128, 256], 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_c4box, args=(r, world, port, cells, levels, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    A, mesh = mg.poisson_shifted(cells)                      # (the checker's hierarchy, while the ranks set up theirs)
    p = mg.getMGparam(np.float64, np.int64, levels, 8, 2, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p, 1)
    b = mg.seeded_rhs(A)
    co = c_oracle.COracle(p, 1)
    xo = np.zeros_like(b)
    ito, rv = co.solveMG(b, xo, 0.0, 2, c_oracle.max_threads())
    res = q.get(timeout=900)
    for pr in procs:
        pr.join(timeout=300)
    assert res[0] == "ok", res[1]
    _, it, resvec, sums, info = res
    assert info[0] and info[1] > 15000 and info[2] > 15000, info      # the two-stage pass with a real face layer on each rank
    assert it == ito == 2
    assert np.abs(np.asarray(resvec) - rv).max() / rv[0] < 1e-10
    assert abs(sums[0] - xo.sum()) <= 1e-9 * np.abs(xo).sum() and abs(sums[1] - xo @ xo) <= 1e-10 * (xo @ xo)


def test_unsupported_settings_fail_loudly():
    import multigrid_jl_amd as mg
    from multigrid_jl_amd import distributed as dd
    for kw in (dict(relaxType="Jac-GMRES"), dict(cycleType="K", relaxType="Jac")):
        p = mg.getMGparam(**kw)
        with pytest.raises(NotImplementedError):
            dd.DistributedHierarchy.check_supported(p)
