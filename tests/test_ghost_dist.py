"""Ghost-layer form of the sharded cycle (multigrid.jl_amd/ghost_dist.py, csrc/mg_ghost.inc).

CPU (-m "not gpu"): gloo worlds of 2, 4 and 8 ranks - geometry, local hierarchies on extended boxes, exchange plans and the
validity rules (results poisoned with NaN beyond the depth the rules claim) through the test-only numpy sequencer
(tests/ghost_cpu_checker.py) against the oracle's single-process solveMG on the global hierarchy.
GPU (-m gpu): the library's own schedule (mg_ghost_*) with 1, 2 and 4 processes sharing the one GPU of the box through the
host-staged transport, and RCCL at a world of one, against the oracle.
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


class _ThreadWorld:
    """`size` ranks that are THREADS of this process (one GPU context: the pool allows at most 6 GPU processes, C4 has 8 ranks).
    The collectives of the plug-in transport over shared memory and a barrier; every rank enters every collective (the library's
    schedule takes the same decisions on all ranks)."""

    def __init__(self, size):
        import threading
        self.size = size
        self.bar = threading.Barrier(size)
        self.slot = [None] * size

    def view(self, rank):
        world = self

        class _C:
            def all_to_all(self, send, ss, rs):
                world.slot[rank] = (send, ss)
                world.bar.wait()
                parts = []
                for q in range(world.size):
                    sq, ssq = world.slot[q]
                    o = int(sum(ssq[:rank]))
                    assert ssq[rank] == rs[q], "exchange plans of two ranks disagree"
                    parts.append(sq[o:o + ssq[rank]])
                out = np.concatenate(parts) if parts else np.zeros(0)
                world.bar.wait()
                return out

            def all_reduce(self, values):
                world.slot[rank] = values
                world.bar.wait()
                tot = np.sum([world.slot[q] for q in range(world.size)], axis=0)
                world.bar.wait()
                return tot

        return _C()



# format switches that make the tests' small grids take the kernels of the large ones (per handle: mg_set_option)
TEST_OPTS = dict(no_small=1, rowclass_min_rows=0, rowclass_max_passes=64, rowclass_min_cover=0.3, march_min_wg=0, tile_min_wg=0, window_min_wg=0,
                 winp_min_rows=0, march27_min_rows=0, marchr_min_rows=0)


def _setup_all_ranks(mg, world, case, cyc, nrhs=1, nd=None, g_fine=None):
    """Host setup of EVERY rank in this process: what all_gather_object would deliver is captured in a first pass."""
    from multigrid_jl_amd import ghost_dist as gd
    from multigrid_jl_amd.structured_setup import poisson_operator
    _, cells, rb = _param(mg, case, cyc)
    nd = nd or _domains(world, len(cells), case)
    kw = {} if g_fine is None else dict(g_fine=g_fine)
    pieces = {}
    if world == 1:       # (nothing to gather)
        return [gd.ghost_gmg(cells, nd, 0, 1, _param(mg, case, cyc)[0], poisson_operator(cells), replicate_below=rb, nrhs=nrhs, **kw)]

    class _Captured(Exception):
        pass

    for r in range(world):
        def capture(obj, _r=r):
            pieces[_r] = obj
            raise _Captured()
        try:
            gd.ghost_gmg(cells, nd, r, world, _param(mg, case, cyc)[0], poisson_operator(cells), replicate_below=rb, gather_objects=capture, nrhs=nrhs, **kw)
        except _Captured:
            pass
    full = [pieces[r] for r in range(world)]
    return [gd.ghost_gmg(cells, nd, r, world, _param(mg, case, cyc)[0], poisson_operator(cells), replicate_below=rb, gather_objects=lambda o: full, nrhs=nrhs, **kw)
            for r in range(world)]


def _thread_world_run(Gs, body, options=None):
    """One thread per rank, one ghost-attached handle each (host-staged transport over shared memory): [body(rank, G, H)]."""
    import threading
    from multigrid_jl_amd import ghost_dist as gd
    world = len(Gs)
    W = _ThreadWorld(world)
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            H = gd.NativeGhostHierarchy(Gs[r], 0, transport="plugin", collectives=W.view(r), options=dict(TEST_OPTS if options is None else options))
            try:
                out[r] = body(r, Gs[r], H)
            finally:
                H.close()
        except Exception:   # pragma: no cover
            import traceback
            errs.append(f"rank {r}: {traceback.format_exc()}")
            W.bar.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errs, errs[0]
    return out


def _run_threads(mg, world, case, cyc, tol, maxit, g_fine=None):
    """What _run(..., "plugin") returns, with the ranks as threads of this process."""
    Gs = _setup_all_ranks(mg, world, case, cyc, g_fine=g_fine)
    _, cells, _ = _param(mg, case, cyc)
    A, mesh = mg.poisson_shifted(cells)
    b = mg.seeded_rhs(A)

    def body(r, G, H):
        b_ext = b[G.gid_fine]
        own = G.levels[0].own_mask()
        bt = torch.from_numpy(b_ext).cuda()
        xt = torch.zeros_like(bt)
        it, resvec = H.solve(bt, xt, tol, maxit)
        x_ext = xt.cpu().numpy()
        x2t = xt.clone()
        H.cycle(bt, x2t, False)
        extra = dict(zip(("exchanges", "sent"), H.exchanges()))
        try:                # entry points whose sums would run over ghost rows (here: the host-pointer PCG) are refused, not computed
            H.dev.pcg(b_ext, np.zeros_like(b_ext), 1e-8, 2)
            extra["pcg_refused"] = False
        except mg.device.MGDeviceError as e:
            extra["pcg_refused"] = "sharded" in str(e)
        extra["four_stage"] = H.dev.four_stage_form(1)[0]
        extra["comm_count"] = H.comm_count()
        return G.gid_fine[own], x_ext[own], x2t.cpu().numpy()[own], int(it), np.asarray(resvec), extra

    out = _thread_world_run(Gs, body)
    x, x2 = np.zeros_like(b), np.zeros_like(b)
    for gid, xl, x2l, *_ in out:
        x[gid] = xl
        x2[gid] = x2l
    return [o[3] for o in out], [o[4] for o in out], x, x2, [o[5] for o in out]


CASES = {
    # name: (cells, levels, replicate_below, smoother, omega, npre, npost)
    "3d-a2": ([32, 32, 32], 4, 1000, "Jac", 0.8, 2, 1),
    "3d-a3": ([32, 32, 64], 5, 1000, "Jac", 0.8, 2, 1),
    "3d-v11": ([32, 32, 32], 4, 1000, "SPAI", 1.0, 1, 1),
    "3d-v32": ([32, 32, 32], 4, 1000, "Jac", 0.8, 3, 2),
    "2d": ([64, 96], 4, 300, "Jac", 0.8, 2, 1),
    "3d-jg": ([32, 32, 32], 4, 1000, "Jac-GMRES", 0.75, 2, 2),  # Jac-GMRES smoother (FGMRES_relaxation, MGcycle.jl:48-50,96-98; FGMRES.jl:48-126)
    "3d-a3-jg": ([32, 32, 64], 5, 1000, "Jac-GMRES", 0.75, 2, 1),
    "3d-a1": ([32, 32, 32], 4, 10000, "Jac", 0.8, 2, 1),        # ONE sharded level (the fine level's own ghost width must carry the four-stage pass)
}


def _param(mg, case, cyc):
    cells, levels, rb, rel, om, npre, npost = CASES[case]
    return mg.getMGparam(np.float64, np.int64, levels, 8, 8, 1e-10, rel, om, npre, npost, cyc, "NoMUMPS", 0.5, 0.0), cells, rb


def _domains(world, dim, case):
    from multigrid_jl_amd import distributed as dd
    if dim == 2:
        return {1: [1, 1], 2: [1, 2], 4: [2, 2], 8: [2, 4]}[world]
    return dd.default_domains(world, 3)


def _global_reference(mg, case, cyc, tol, maxit):
    """The oracle's solveMG on the GLOBAL hierarchy (single process)."""
    from oracle import mg_oracle as orc
    p, cells, _ = _param(mg, case, cyc)
    A, mesh = mg.poisson_shifted(cells)
    mg.MGsetup(A, mesh, p)
    b = mg.seeded_rhs(A)
    x = np.zeros_like(b)
    p.maxOuterIter, p.relativeTol = maxit, tol
    hist = {}
    orc.solveMG(p, b, x, False, hist)
    return A, b, x, np.asarray(hist["resvec"])


def _worker(rank, world, port, case, cyc, mode, q, tol, maxit):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import multigrid_jl_amd as mg
        from multigrid_jl_amd import ghost_dist as gd
        from multigrid_jl_amd.structured_setup import poisson_operator
        p, cells, rb = _param(mg, case, cyc)
        kw = dict(g_fine=int(os.environ["MG_TEST_G_FINE"])) if os.environ.get("MG_TEST_G_FINE") else {}
        G = gd.ghost_gmg(cells, _domains(world, len(cells), case), rank, world, p, poisson_operator(cells), replicate_below=rb, **kw)
        A, mesh = mg.poisson_shifted(cells)
        b = mg.seeded_rhs(A)
        b_ext = b[G.gid_fine]
        own = G.levels[0].own_mask()
        extra = {}
        if mode == "cpu":
            from ghost_cpu_checker import GhostCpuSequencer
            S = GhostCpuSequencer(G)
            it, resvec, x_ext = S.solve(b_ext, np.zeros_like(b_ext), tol, maxit)
            S2 = GhostCpuSequencer(G)
            it2, resvec2, x2 = S2.solve(b_ext, np.zeros_like(b_ext), tol, maxit, fused4=False)
            assert it2 == it and np.allclose(resvec2, resvec, rtol=1e-12, atol=0) and np.allclose(x2[own], x_ext[own], rtol=0, atol=1e-12 * np.abs(x_ext[own]).max())
            xc = S.cycle(b_ext, x_ext, False)
            extra["exchanges"] = S.exchanges
        else:
            torch.cuda.set_device(0)
            transport = "plugin" if world > 1 else mode
            os.environ.update(MG_NO_SMALL="1", MG_ROWCLASS_MIN_ROWS="0", MG_ROWCLASS_MAX_PASSES="64", MG_ROWCLASS_MIN_COVER="0.3", MG_MARCH_MIN_WG="0",
                              MG_TILE_MIN_WG="0", MG_WINDOW_MIN_WG="0", MG_WINP_MIN_ROWS="0", MG_MARCH27_MIN_ROWS="0", MG_MARCHR_MIN_ROWS="0")
            H = gd.NativeGhostHierarchy(G, 0, transport=("rccl" if transport == "rccl" else "plugin"))
            bt = torch.from_numpy(b_ext).cuda()
            xt = torch.zeros_like(bt)
            it, resvec = H.solve(bt, xt, tol, maxit)
            x_ext = xt.cpu().numpy()
            x2t = xt.clone()
            H.cycle(bt, x2t, False)
            xc = x2t.cpu().numpy()
            extra["exchanges"], extra["sent"] = H.exchanges()
            try:                # entry points whose sums would run over ghost rows (here: the host-pointer PCG) are refused, not computed
                H.dev.pcg(b_ext, np.zeros_like(b_ext), 1e-8, 2)
                extra["pcg_refused"] = False
            except mg.device.MGDeviceError as e:
                extra["pcg_refused"] = "sharded" in str(e)
            extra["four_stage"] = H.dev.four_stage_form(1)[0]
            extra["comm_count"] = H.comm_count()
            H.close()
        out = [None] * world
        dist.all_gather_object(out, (G.gid_fine[own], x_ext[own], xc[own], int(it), np.asarray(resvec), extra))
        if rank == 0:
            x = np.zeros_like(b)
            x2 = np.zeros_like(b)
            for gid, xl, x2l, *_ in out:
                x[gid] = xl
                x2[gid] = x2l
            its = [o[3] for o in out]
            q.put(("ok", its, [o[4] for o in out], x, x2, [o[5] for o in out]))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(("err", f"rank {rank}: {e!r}\n{traceback.format_exc()}"))


def _run(world, case, cyc, mode, tol=1e-8, maxit=6):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, cyc, mode, q, tol, maxit)) for r in range(world)]
    for pr in procs:
        pr.start()
    try:
        res = q.get(timeout=600)
    finally:
        for pr in procs:
            pr.join(timeout=60)
            if pr.is_alive():
                pr.kill()
    assert res[0] == "ok", res[1]
    return res[1:]


def _check(mg, world, case, cyc, mode, tol=1e-8, maxit=6):
    from oracle import mg_oracle as orc
    A, b, x_ref, res_ref = _global_reference(mg, case, cyc, tol, maxit)
    if mode == "threads":      # the ranks are threads of this process (no spawn, no second GPU context: seconds instead of tens of seconds)
        its, resvecs, x, x2, extra = _run_threads(mg, world, case, cyc, tol, maxit)
    else:
        its, resvecs, x, x2, extra = _run(world, case, cyc, mode, tol, maxit)
    assert all(i == len(res_ref) - 1 for i in its), (its, len(res_ref) - 1)
    for rv in resvecs:                      # every rank holds the same, global, residual history
        assert len(rv) == len(res_ref)
        assert np.abs(rv - res_ref).max() <= 1e-10 * res_ref[0], np.abs(rv - res_ref).max() / res_ref[0]
    assert np.abs(x - x_ref).max() <= 1e-10 * np.abs(x_ref).max()
    # one more cycle from the non-zero iterate through the public cycle entry
    p, cells, _ = _param(mg, case, cyc)
    Ag, mesh = mg.poisson_shifted(cells)
    mg.MGsetup(Ag, mesh, p)
    xc = orc.recursiveCycle(p, b, x_ref.copy(), 1, None, cyc)
    assert np.abs(x2 - xc).max() <= 1e-10 * np.abs(xc).max()
    return extra


@pytest.fixture(scope="module")
def mg():
    import multigrid_jl_amd as m
    return m


def test_ghost_boxes_cover_and_carry_their_own_widths(mg):
    """Round 6: every sharded level has the ghost width its own passes consume (fine: 5 = four-stage pass + restriction, coarser: 3);
    a level's box ends on nodes of the next level and the next level's box holds the parents of all its nodes."""
    from multigrid_jl_amd import ghost_dist as gd
    cells, nd, a = [64, 64, 128], [2, 2, 2], 3
    for rank in range(8):
        lv = gd.ghost_boxes(cells, nd, gd.default_box_of(rank, nd), a)
        old = gd.ghost_boxes(cells, nd, gd.default_box_of(rank, nd), a, nested=True)
        for l in range(a):
            n_l = [(c >> l) + 1 for c in cells]
            for k in range(3):
                (olo, ohi), (elo, ehi) = lv[l]["own"][k], lv[l]["ext"][k]
                assert 0 <= elo <= olo <= ohi <= ehi <= n_l[k] - 1
                assert old[l]["ext"][k][0] <= elo and ehi <= old[l]["ext"][k][1]                          # never larger than the nested boxes
                want = gd.G_FINE if l == 0 else gd.G_LAST
                assert olo == 0 or want <= olo - elo <= want + 1
                assert ohi == n_l[k] - 1 or want <= ehi - ohi <= want + 1
                if l + 1 < a:
                    assert elo % 2 == 0 and ehi % 2 == 0                                                  # ends on nodes of the next level
                    clo, chi = lv[l + 1]["ext"][k]
                    assert clo <= elo // 2 and ehi // 2 <= chi                                            # the parents of every node
                    assert lv[l + 1]["own"][k] == (-(-olo // 2), ohi // 2)                               # owner of the coincident node
                    assert (old[l]["ext"][k][0], old[l]["ext"][k][1]) == (2 * old[l + 1]["ext"][k][0], 2 * old[l + 1]["ext"][k][1])
        assert lv[0]["gmin"] >= gd.G_FINE and all(lv[l]["gmin"] >= gd.G_LAST for l in range(1, a))
        assert old[a - 1]["gmin"] == gd.G_LAST and old[0]["gmin"] >= 4 * gd.G_LAST - 3
    # one sharded level: the fine level's own width (3 layers cannot carry the four-stage pass)
    assert gd.ghost_boxes([32, 32, 32], [1, 1, 2], [0, 0, 1], 1)[0]["gmin"] >= 5
    # the owned boxes of a level tile the grid
    for l in range(a):
        seen = np.zeros([(c >> l) + 1 for c in cells][::-1], dtype=int)
        for rank in range(8):
            o = gd.ghost_boxes(cells, nd, gd.default_box_of(rank, nd), a)[l]["own"]
            seen[o[2][0]:o[2][1] + 1, o[1][0]:o[1][1] + 1, o[0][0]:o[0][1] + 1] += 1
        assert (seen == 1).all()


def test_local_hierarchy_rows_equal_global(mg):
    """World of 4 computed in ONE process (no communication needed for the check): every extended-box operator row that is not
    in the outermost cut layer equals the global hierarchy's row; R into the replicated level holds the owned nodes' rows."""
    from multigrid_jl_amd import ghost_dist as gd
    from multigrid_jl_amd.structured_setup import poisson_operator
    case = "3d-a2"
    p, cells, rb = _param(mg, case, "V")
    pg, _, _ = _param(mg, case, "V")
    A, mesh = mg.poisson_shifted(cells)
    mg.MGsetup(A, mesh, pg)
    world, nd = 4, [1, 2, 2]
    pieces = {}

    class _Captured(Exception):
        pass

    for rank in range(world):          # two passes: the tail pieces first (what all_gather_object would deliver)
        def capture(obj, _r=rank):
            pieces[_r] = obj
            raise _Captured()
        try:
            gd.ghost_gmg(cells, nd, rank, world, p, poisson_operator(cells), replicate_below=rb, gather_objects=capture)
        except _Captured:
            pass
    full = [pieces[r] for r in range(world)]
    cover = None
    for rank in range(world):
        G = gd.ghost_gmg(cells, nd, rank, world, p, poisson_operator(cells), replicate_below=rb, gather_objects=lambda obj: full)
        a = G.a
        assert a == 2 and len(G.param.As) == len(pg.As)
        for l in range(a):
            L = G.levels[l]
            nodes = [(c >> l) + 1 for c in cells]
            ax = [np.arange(L.ext_n[k]) + L.ext_lo[k] for k in range(3)]
            gid = (ax[0][None, None, :] + nodes[0] * (ax[1][None, :, None] + nodes[1] * ax[2][:, None, None])).reshape(-1)
            inner = L.depth_mask(L.gmin - 1)
            Aloc = G.param.As[l]
            Ag = pg.As[l][gid, :][:, gid]
            diff = abs(Aloc - Ag).tocsr()
            rows_bad = np.unique(diff.nonzero()[0][np.abs(diff.data if diff.nnz else np.zeros(0)) > 1e-12 * abs(Ag).max()]) if diff.nnz else np.zeros(0, dtype=int)
            assert not inner[rows_bad].any(), f"level {l}: a row inside the valid region differs from the global operator"
            assert np.allclose(G.param.relaxPrecs[l][inner], pg.relaxPrecs[l][gid][inner], rtol=1e-13)
        # into the replicated level: rows of the owned nodes only, together one copy of the global restriction
        Rt = G.param.Rs[a - 1]
        owned_rows = np.diff(Rt.indptr) > 0
        cover = owned_rows.astype(int) if cover is None else cover + owned_rows.astype(int)
        for j in range(a, len(pg.As)):
            assert abs(G.param.As[j] - pg.As[j]).max() <= 1e-12 * abs(pg.As[j]).max()
    assert (cover == 1).all()


@pytest.mark.parametrize("world,case,cyc", [(2, "3d-a2", "V"), (4, "3d-a2", "W"), (8, "3d-a2", "V"), (2, "3d-a3", "V"), (4, "3d-a3", "F"),
                                            (2, "3d-v11", "V"), (4, "3d-v32", "V"), (2, "2d", "V"), (4, "2d", "W"), (2, "3d-a1", "V"), (8, "3d-a1", "W")])
def test_ghost_form_cpu_vs_oracle(mg, world, case, cyc):
    extra = _check(mg, world, case, cyc, "cpu")
    assert all(e["exchanges"] > 0 for e in extra)


def test_ghost_form_three_fine_layers_cpu(mg, monkeypatch):
    """The validity rules with a fine level of three ghost layers (what round 5 gave a single sharded level): the four-stage pass is not
    taken (it consumes four), the NaN-poisoned sequencer finds no rule claiming too much, the history matches the oracle."""
    monkeypatch.setenv("MG_TEST_G_FINE", "3")
    _check(mg, 2, "3d-a1", "V", "cpu")


def test_ghost_form_early_stop_cpu(mg):
    """The stopping test ends the loop in the middle: same step count and iterate as the oracle."""
    _check(mg, 2, "3d-a2", "V", "cpu", tol=1e-3, maxit=8)


@pytest.mark.gpu
@pytest.mark.parametrize("world,case,cyc,mode", [(1, "3d-a2", "V", "threads"), (2, "3d-a2", "V", "plugin"), (4, "3d-a2", "W", "threads"), (2, "3d-a3", "F", "threads"),
                                                 (4, "3d-v32", "V", "threads"), (2, "3d-v11", "V", "threads"), (2, "2d", "V", "threads"), (2, "3d-a1", "V", "threads"),
                                                 (8, "3d-a1", "W", "threads")])
def test_ghost_form_hip_plugin_vs_oracle(mg, world, case, cyc, mode):
    """The library's schedule (mg_ghost_*) with `world` ranks sharing the GPU through the host-staged transport: processes over
    torch.distributed ("plugin": one case) or threads of this process over shared memory ("threads": same library path, no spawn)."""
    extra = _check(mg, world, case, cyc, mode)
    assert all(e["pcg_refused"] for e in extra)
    if world > 1:
        assert all(e["exchanges"] > 0 for e in extra)


@pytest.mark.gpu
@pytest.mark.parametrize("world,case,cyc", [(2, "3d-a2", "K"), (4, "3d-a3", "K"), (2, "3d-jg", "V"), (4, "3d-jg", "W"), (2, "3d-a3-jg", "K")])
def test_ghost_form_kcycle_and_jac_gmres(mg, world, case, cyc):
    """Round 6: the K-cycle (2 FGMRES steps per level preconditioned by the next level's K-cycle, MGcycle.jl:72-76) and
    the Jac-GMRES smoother (FGMRES.jl:48-126) on ghost-attached handles: the FGMRES dots are sums over the owned rows of all ranks, every
    product follows one exchange of its input's ghost layers, the data-dependent exits are identical on every rank.  Until round 5 these
    ran in the halo form only (tests/test_distributed.py::test_native_sequencer_kcycle_and_jac_gmres)."""
    extra = _check(mg, world, case, cyc, "threads", maxit=4)
    assert all(e["exchanges"] > 0 for e in extra)


@pytest.mark.gpu
def test_ghost_form_three_fine_layers_take_the_two_stage_steps(mg):
    """the four-stage pass consumes FOUR ghost layers of x in one kernel; a fine level that has only three (round 5: one
    sharded level) silently produced a wrong restriction input.  With g_fine = 3 the pass must not be taken (two-stage steps instead), the
    residual history still matches the oracle, and nothing is clamped: gh_set reports a negative depth as MG_ERR_STATE."""
    from oracle import mg_oracle as orc
    case, cyc, world, tol, maxit = "3d-a1", "V", 2, 1e-8, 6
    A, b, x_ref, res_ref = _global_reference(mg, case, cyc, tol, maxit)
    its, resvecs, x, x2, extra = _run_threads(mg, world, case, cyc, tol, maxit, g_fine=3)
    assert all(i == len(res_ref) - 1 for i in its)
    for rv in resvecs:
        assert np.abs(rv - res_ref).max() <= 1e-10 * res_ref[0]
    assert np.abs(x - x_ref).max() <= 1e-10 * np.abs(x_ref).max()


@pytest.mark.gpu
def test_ghost_form_hip_rccl_world1(mg):
    """RCCL transport at a world of one: ncclCommInitRank from the library's own id, all-reduces on the compute stream."""
    extra = _check(mg, 1, "3d-a2", "V", "rccl")
    assert extra[0]["comm_count"] == 1


@pytest.mark.gpu
def test_ghost_form_hip_early_stop(mg):
    _check(mg, 2, "3d-a2", "V", "threads", tol=1e-3, maxit=8)


@pytest.mark.gpu
@pytest.mark.parametrize("world,cyc,nrhs,tol,x_nonzero", [(2, "V", 3, 1e-30, False), (4, "W", 2, 3e-3, False), (2, "V", 2, 1e-30, True), (1, "V", 2, 1e-30, False)])
def test_ghost_form_block_of_right_hand_sides(mg, world, cyc, nrhs, tol, x_nonzero):
    """solveMG on an n x k block (SolveFuncs.jl:3-39, one Frobenius stopping test) with the grid cut over `world` ranks in the
    ghost-layer form: every column plays the single-vector kernels on the rank's extended boxes, the norms are sums over the owned
    rows of all ranks.  Against the oracle's block solve on the global hierarchy (ranks = threads of this process)."""
    from oracle import mg_oracle as orc
    case, maxit = "3d-a2", 5
    p, cells, _ = _param(mg, case, cyc)
    A, mesh = mg.poisson_shifted(cells)
    mg.MGsetup(A, mesh, p, nrhs)
    B = mg.seeded_rhs(A, nrhs)
    X0 = np.random.default_rng(5).standard_normal(B.shape)
    Xo = np.asfortranarray(X0.copy()) if x_nonzero else np.zeros_like(B)
    p.maxOuterIter, p.relativeTol = maxit, tol
    hist = {}
    _, _, ito = orc.solveMG(p, B, Xo, False, hist)
    res_ref = np.asarray(hist["resvec"])
    Gs = _setup_all_ranks(mg, world, case, cyc, nrhs=nrhs)
    opts = dict(TEST_OPTS, rowclass_min_cover=0.05, march_max_len=64, march4_ty_max=12)     # (33 lines: an odd count needs more than one tile row)

    def body(r, G, H):
        own = G.levels[0].own_mask()
        assert H.nrhs == nrhs and H.dev.four_stage_form(1)[0]
        bt = torch.from_numpy(np.ascontiguousarray(B[G.gid_fine])).cuda()          # n_ext x nrhs, row-major: the device layout of a block
        if x_nonzero:
            Xl = X0[G.gid_fine].copy()
            Xl[~own] = np.nan                                                      # (only the owned rows of x are the caller's to give)
            xt = torch.from_numpy(np.ascontiguousarray(Xl)).cuda()
        else:
            xt = torch.zeros_like(bt)
        e0 = H.exchanges()
        it, resvec = H.solve(bt, xt, tol, maxit)
        e1 = H.exchanges()
        x2 = xt.clone()            # one more cycle on the block from the iterate (round 6: a block cycle = its columns' cycles)
        H.dev.cycle_dev(bt, x2, 0)
        return G.gid_fine[own], xt.cpu().numpy()[own], int(it), np.asarray(resvec), e1[0] - e0[0], x2.cpu().numpy()[own]

    out = _thread_world_run(Gs, body, options=opts)
    X, X2 = np.zeros((A.shape[0], nrhs)), np.zeros((A.shape[0], nrhs))
    for gid, xl, it, rv, exch, x2l in out:
        X[gid] = xl
        X2[gid] = x2l
        assert it == ito, (it, ito)
        assert np.abs(rv - res_ref).max() <= 1e-10 * res_ref[0]
        assert world == 1 or exch > 0
    Xc = orc.recursiveCycle(p, B, np.asfortranarray(Xo.copy()), 1, None, cyc)
    assert np.abs(X2 - Xc).max() <= 1e-10 * np.abs(Xc).max()
    if tol == 3e-3:
        assert 1 < ito < maxit                     # (meant to stop early)
    assert np.abs(X - Xo).max() <= 1e-10 * np.abs(Xo).max()


def _worker_c4box(rank, world, port, cells, levels, q, steps):
    """One rank of the C4-sized run in the ghost-layer form: its 257^3 box + ghost layers, plug-in transport."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["MG_HOST_THREADS"] = str(max(1, (os.cpu_count() or 2) // world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        import multigrid_jl_amd as mg
        from multigrid_jl_amd import distributed as dd, ghost_dist as gd, structured_setup as ss
        p = mg.getMGparam(np.float64, np.int64, levels, 8, steps, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
        G = gd.ghost_gmg(cells, dd.default_domains(world, 3), rank, world, p, ss.poisson_operator(cells))
        b_ext, ss2 = gd.local_rhs(G)
        tot = torch.tensor([ss2], dtype=torch.float64)
        dist.all_reduce(tot)
        H = gd.NativeGhostHierarchy(G, 0, transport="plugin")
        bt = torch.from_numpy(b_ext / float(tot.item()) ** 0.5).cuda()
        xt = torch.zeros_like(bt)
        e0 = H.exchanges()
        it, resvec = H.solve(bt, xt, 0.0, steps)
        e1 = H.exchanges()
        own = torch.from_numpy(G.levels[0].own_mask()).cuda()
        xo = xt[own]
        sums = torch.tensor([float(xo.sum().item()), float((xo * xo).sum().item())], dtype=torch.float64)
        dist.all_reduce(sums)
        info = dict(four_stage=H.dev.four_stage_form(1)[0], a=G.a, gmin=[L.gmin for L in G.levels], ext=[L.ext_n for L in G.levels],
                    exchanges=e1[0] - e0[0], sent=e1[1] - e0[1], sweep_form=[H.dev.sweep_residual_form(l + 1)[0] for l in range(G.a)])
        if rank == 0:
            q.put(("ok", it, resvec, sums.numpy(), info))
        dist.barrier()
        H.close()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put(("err", traceback.format_exc()))
        raise


@pytest.mark.gpu
def test_c4_per_gpu_box_size_two_ranks_ghost_form_vs_c_oracle(mg):
    """BASELINE.json configs[3] (512^3 cells over 8 GPUs) puts a 257^3-node box on every GPU.  Two such boxes - 256 x 256 x 512
    cells, 33.9 M rows - on two ranks sharing this box's one GPU in the GHOST-LAYER form (host-staged transport): the fine level of
    each rank runs the four-stage pass on 257 x 257 x 261-263 nodes, level 2 the 27-point marching form, the restriction the marching
    form; four solveMG steps (from-zero step, two four-stage steps, last step) against the C/OpenMP oracle on the global hierarchy."""
    from oracle import c_oracle
    cells, levels, world, steps = [256, 256, 512], 6, 2, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_c4box, args=(r, world, port, cells, levels, q, steps)) for r in range(world)]
    for pr in procs:
        pr.start()
    A, mesh = mg.poisson_shifted(cells)                      # (the checker's hierarchy, while the ranks set up theirs)
    p = mg.getMGparam(np.float64, np.int64, levels, 8, steps, 0.0, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p, 1)
    b = mg.seeded_rhs(A)
    co = c_oracle.COracle(p, 1)
    xo = np.zeros_like(b)
    ito, rv = co.solveMG(b, xo, 0.0, steps, c_oracle.max_threads())
    res = q.get(timeout=900)
    for pr in procs:
        pr.join(timeout=300)
    assert res[0] == "ok", res[1]
    _, it, resvec, sums, info = res
    print("ghost form, C4 box size:", info)
    # round 6: every level carries its own ghost width - 5 fine layers (four-stage pass + restriction), 3 on the coarser levels
    assert info["four_stage"] == 1 and info["a"] == 3 and info["gmin"] == [5, 3, 3], info
    assert info["ext"][0][2] <= 257 + 6 and info["ext"][1][2] <= 129 + 4 and info["ext"][2][2] <= 65 + 3, info
    assert info["sweep_form"][0] == 3 and info["sweep_form"][1] == 5, info      # 2-D tile form on the fine level, 27-point marching form on level 2
    assert it == ito == steps
    assert np.abs(np.asarray(resvec) - rv).max() / rv[0] < 1e-10
    assert abs(sums[0] - xo.sum()) <= 1e-9 * np.abs(xo).sum() and abs(sums[1] - xo @ xo) <= 1e-10 * (xo @ xo)
    # communication the schedule issued: at most one fine-level exchange per step + two per coarser sharded level and step (+ b once)
    assert info["exchanges"] <= steps * (1 + 2 * (info["a"] - 1)) + 1, info


# ---- sharded MG-preconditioned Krylov (round 6): solveCG_MG / solveBiCGSTAB_MG / solveGMRES_MG on ghost-attached handles ---------------
def _worker_krylov(rank, world, port, case, cyc, mode, method, q, tol, maxit):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import multigrid_jl_amd as mg
        from multigrid_jl_amd import ghost_dist as gd
        from multigrid_jl_amd.structured_setup import poisson_operator
        p, cells, rb = _param(mg, case, cyc)
        G = gd.ghost_gmg(cells, _domains(world, len(cells), case), rank, world, p, poisson_operator(cells), replicate_below=rb)
        A, mesh = mg.poisson_shifted(cells)
        b = mg.seeded_rhs(A)
        b_ext = b[G.gid_fine]
        own = G.levels[0].own_mask()
        x0 = np.zeros_like(b_ext)
        if method == "bicgstab":            # (a non-zero start: the initial residual needs x's ghost layers)
            x0 = (0.01 * np.random.default_rng(3).standard_normal(A.shape[0]))[G.gid_fine]
        info = {}
        if mode == "cpu":
            from ghost_cpu_checker import GhostCpuSequencer, GhostKrylovCpu
            S = GhostCpuSequencer(G)
            K = GhostKrylovCpu(S)
            if method == "pcg":
                x, flag, it, resvec = K.cg(b_ext, x0, tol, maxit)
            elif method == "bicgstab":
                x, flag, it, resvec = K.bicgstb(b_ext, x0, tol, maxit)
            else:
                x, flag, it, resvec = K.fgmres(b_ext, x0, 3, tol, maxit)
            info = dict(exchanges=S.exchanges, allreduces=K.allreduces)
        else:
            torch.cuda.set_device(0)
            os.environ.update(MG_NO_SMALL="1", MG_ROWCLASS_MIN_ROWS="0", MG_ROWCLASS_MAX_PASSES="64", MG_ROWCLASS_MIN_COVER="0.3", MG_MARCH_MIN_WG="0",
                              MG_TILE_MIN_WG="0", MG_WINDOW_MIN_WG="0", MG_WINP_MIN_ROWS="0", MG_MARCH27_MIN_ROWS="0", MG_MARCHR_MIN_ROWS="0")
            H = gd.NativeGhostHierarchy(G, 0, transport="plugin" if world > 1 else "rccl")
            bt = torch.from_numpy(b_ext).cuda()
            # what ONE application of the preconditioner communicates (a cycle from x = 0)
            e0, a0 = H.exchanges()[0], H.allreduces()
            H.cycle(bt, torch.zeros_like(bt), True)
            e_cyc, a_cyc = H.exchanges()[0] - e0, H.allreduces() - a0
            x0n = x0.copy()
            x0n[~own] = np.nan if method == "bicgstab" else 0.0      # (only the owned rows of x are the caller's to give)
            xt = torch.from_numpy(x0n).cuda()
            e0, a0 = H.exchanges()[0], H.allreduces()
            if method == "pcg":
                flag, it, resvec = H.pcg(bt, xt, tol, maxit)
            elif method == "bicgstab":
                flag, it, resvec = H.bicgstab(bt, xt, tol, maxit)
            else:
                flag, it, resvec = H.fgmres(bt, xt, 3, tol, maxit)
            info = dict(exchanges=H.exchanges()[0] - e0, allreduces=H.allreduces() - a0, e_cyc=e_cyc, a_cyc=a_cyc)
            x = xt.cpu().numpy()
            H.close()
        out = [None] * world
        dist.all_gather_object(out, (G.gid_fine[own], x[own], int(flag), int(it), np.asarray(resvec), info))
        if rank == 0:
            xg = np.zeros_like(b)
            for gid, xl, *_ in out:
                xg[gid] = xl
            q.put(("ok", [o[2] for o in out], [o[3] for o in out], [o[4] for o in out], xg, [o[5] for o in out]))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(("err", f"rank {rank}: {e!r}\n{traceback.format_exc()}"))


def _check_krylov(mg, world, case, cyc, mode, method, tol=1e-9, maxit=12):
    from oracle import mg_oracle as orc
    p, cells, _ = _param(mg, case, cyc)
    A, mesh = mg.poisson_shifted(cells)
    mg.MGsetup(A, mesh, p)
    b = mg.seeded_rhs(A)
    p.relativeTol, p.maxOuterIter = tol, maxit
    if method == "pcg":
        x_ref, flag_ref, it_ref, res_ref = orc.solveCG_MG(p, b, np.zeros_like(b))
    elif method == "bicgstab":
        x_ref, flag_ref, it_ref, res_ref = orc.solveBiCGSTAB_MG(p, b, 0.01 * np.random.default_rng(3).standard_normal(A.shape[0]))
    else:
        x_ref, flag_ref, it_ref, res_ref = orc.solveGMRES_MG(p, b, np.zeros_like(b), 3)
    if mode == "threads":
        res = _krylov_threads(mg, world, case, cyc, method, tol, maxit)
    else:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker_krylov, args=(r, world, port, case, cyc, mode, method, q, tol, maxit)) for r in range(world)]
        for pr in procs:
            pr.start()
        try:
            res = q.get(timeout=600)
        finally:
            for pr in procs:
                pr.join(timeout=60)
                if pr.is_alive():
                    pr.kill()
    assert res[0] == "ok", res[1]
    _, flags, its, resvecs, x, infos = res
    assert all(f == flag_ref for f in flags) and all(i == it_ref for i in its), (flags, flag_ref, its, it_ref)
    for rv in resvecs:                      # every rank holds the same, global, residual history
        assert len(rv) == len(res_ref) and np.abs(rv - res_ref).max() <= 1e-10 * max(1.0, np.abs(res_ref).max()), (rv, res_ref)
    assert np.abs(x - x_ref).max() <= 1e-10 * np.abs(x_ref).max()
    return it_ref, infos


@pytest.mark.parametrize("world,case,cyc,method", [(2, "3d-a2", "V", "pcg"), (4, "3d-a2", "V", "fgmres"), (8, "3d-a2", "V", "pcg"), (2, "3d-a3", "V", "bicgstab"),
                                                   (4, "2d", "W", "pcg"), (2, "3d-a1", "V", "fgmres")])
def test_ghost_form_krylov_cpu_vs_oracle(mg, world, case, cyc, method):
    """The sharded drivers' three pieces - owned-row dots + all-reduce, one exchange in front of every product with A, the sharded cycle
    as M - reproduce the oracle's iterates and iteration counts (numpy sequencer, NaN-poisoned ghost rows, gloo)."""
    it, infos = _check_krylov(mg, world, case, cyc, "cpu", method)
    assert it > 1 and all(i["exchanges"] > 0 for i in infos)


def _krylov_threads(mg, world, case, cyc, method, tol, maxit):
    """The `plugin` mode of _worker_krylov with the ranks as threads of this process."""
    Gs = _setup_all_ranks(mg, world, case, cyc)
    _, cells, _ = _param(mg, case, cyc)
    A, mesh = mg.poisson_shifted(cells)
    b = mg.seeded_rhs(A)
    x0g = 0.01 * np.random.default_rng(3).standard_normal(A.shape[0]) if method == "bicgstab" else np.zeros_like(b)

    def body(r, G, H):
        own = G.levels[0].own_mask()
        bt = torch.from_numpy(b[G.gid_fine]).cuda()
        e0, a0 = H.exchanges()[0], H.allreduces()
        H.cycle(bt, torch.zeros_like(bt), True)          # what ONE application of the preconditioner communicates
        e_cyc, a_cyc = H.exchanges()[0] - e0, H.allreduces() - a0
        x0 = x0g[G.gid_fine].copy()
        x0[~own] = np.nan if method == "bicgstab" else 0.0      # (only the owned rows of x are the caller's to give)
        xt = torch.from_numpy(x0).cuda()
        e0, a0 = H.exchanges()[0], H.allreduces()
        if method == "pcg":
            flag, it, resvec = H.pcg(bt, xt, tol, maxit)
        elif method == "bicgstab":
            flag, it, resvec = H.bicgstab(bt, xt, tol, maxit)
        else:
            flag, it, resvec = H.fgmres(bt, xt, 3, tol, maxit)
        info = dict(exchanges=H.exchanges()[0] - e0, allreduces=H.allreduces() - a0, e_cyc=e_cyc, a_cyc=a_cyc)
        return G.gid_fine[own], xt.cpu().numpy()[own], int(flag), int(it), np.asarray(resvec), info

    out = _thread_world_run(Gs, body)
    xg = np.zeros_like(b)
    for gid, xl, *_ in out:
        xg[gid] = xl
    return "ok", [o[2] for o in out], [o[3] for o in out], [o[4] for o in out], xg, [o[5] for o in out]


@pytest.mark.gpu
@pytest.mark.parametrize("world,case,cyc,method,mode", [(1, "3d-a2", "V", "bicgstab", "plugin"),    # (a world of one as a process of its own: the RCCL transport - all-reduces of 1 and 2 scalars on the stream)
                                                        (2, "3d-a2", "V", "pcg", "plugin"), (4, "3d-a2", "V", "fgmres", "threads"),
                                                        (2, "3d-a3", "V", "bicgstab", "threads"), (4, "3d-a2", "W", "pcg", "threads"), (2, "3d-a1", "V", "fgmres", "threads")])
def test_ghost_form_krylov_hip_vs_oracle(mg, world, case, cyc, method, mode):
    """mg_pcg_dev / mg_bicgstab_dev / mg_fgmres_dev on ghost-attached handles (`world` ranks sharing the GPU through the host-staged
    transport: threads of this process, one case as processes over torch.distributed) against the oracle's solveCG_MG / solveBiCGSTAB_MG /
    solveGMRES_MG on the global hierarchy: same iterates, same iteration counts; the communication per iteration is what the design says."""
    k, infos = _check_krylov(mg, world, case, cyc, mode, method)
    if world == 1:
        return
    for i in infos:
        ec, ac = i["e_cyc"], i["a_cyc"]
        assert ec > 0 and ac >= 1
        if method == "pcg":        # k iterations, the k-th converged: products 1 + k (one exchange each), cycles k, scalars ||b||, r'z, k x (p'Ap, ||r||), (k-1) x z'r
            assert i["exchanges"] == 1 + k + k * ec, (i, k)
            assert i["allreduces"] == 1 + 3 * k + k * ac, (i, k)
        elif method == "bicgstab":  # per iteration: 2 cycles, 2 products, scalars rho, r~'v, ||s||, (t's, t't) as ONE, ||r||
            assert i["exchanges"] == 1 + k * (2 + 2 * ec), (i, k)
            assert i["allreduces"] == 2 + k * (5 + 2 * ac), (i, k)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["solve", "pcg"])
def test_c4_topology_eight_ranks_ghost_form_hip(mg, method):
    """BASELINE.json configs[3] cuts the grid into numDomains = [2,2,2] (DDIndices.jl:41-47): every rank has 3 face, 3 edge and 1 corner
    neighbour.  Eight ghost-attached handles on the one GPU of the box (ranks = threads of this process - the pool allows 6 GPU
    processes - host-staged transport), three sharded levels with their own ghost widths: solveMG and the sharded PCG against the
    oracle on the global hierarchy."""
    from oracle import mg_oracle as orc
    case, cyc, world, nd, tol, maxit = "3d-a3", "V", 8, [2, 2, 2], 1e-8, 6
    p, cells, rb = _param(mg, case, cyc)
    A, mesh = mg.poisson_shifted(cells)
    mg.MGsetup(A, mesh, p)
    b = mg.seeded_rhs(A)
    p.maxOuterIter, p.relativeTol = maxit, tol
    flag_ref = it_ref = None
    if method == "solve":
        hist = {}
        x_ref = np.zeros_like(b)
        orc.solveMG(p, b, x_ref, False, hist)
        res_ref = np.asarray(hist["resvec"])
    else:
        x_ref, flag_ref, it_ref, res_ref = orc.solveCG_MG(p, b, np.zeros_like(b))
    Gs = _setup_all_ranks(mg, world, case, cyc, nd=nd)
    assert all(G.a == 3 for G in Gs)
    for G in Gs:      # every rank exchanges with its 7 neighbours on the fine level: 3 faces, 3 edges, 1 corner
        assert sum(1 for s in G.levels[0].send_splits if s > 0) == 7 and sum(1 for s in G.levels[0].recv_splits if s > 0) == 7

    def body(r, G, H):
        bt = torch.from_numpy(b[G.gid_fine]).cuda()
        xt = torch.zeros_like(bt)
        if method == "solve":
            it, resvec = H.solve(bt, xt, tol, maxit)
            flag = 0
        else:
            flag, it, resvec = H.pcg(bt, xt, tol, maxit)
        own = G.levels[0].own_mask()
        return G.gid_fine[own], xt.cpu().numpy()[own], int(it), np.asarray(resvec), int(flag), H.exchanges()

    out = _thread_world_run(Gs, body)
    x = np.zeros_like(b)
    for gid, xl, it, rv, flag, exch in out:
        x[gid] = xl
        assert len(rv) == len(res_ref) and np.abs(rv - res_ref).max() <= 1e-10 * max(1.0, res_ref[0]), (rv, res_ref)
        assert exch[0] > 0
        if method == "pcg":
            assert flag == flag_ref and it == it_ref
    assert np.abs(x - x_ref).max() <= 1e-10 * np.abs(x_ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("world,nrhs", [(2, 3), (4, 2)])
def test_ghost_form_block_krylov_hip_vs_oracle(mg, world, nrhs):
    """The BLOCK Krylov drivers on ghost-attached handles (round 6): blockCG / blockBiCGSTB / blockFGMRES as the reference's wrappers call
    them for several right-hand sides (SolveFuncs.jl:95,113,130; MGWrapper.jl:67-78).  Blocks [n_ext][k] with their owned rows valid: a
    Gram matrix is the box Gram over the owned rows + ONE all-reduce of k x k doubles, a product with A follows one exchange of the block's
    ghost layers, the preconditioner is the block's columns through the sharded cycle.  Against the oracle on the global hierarchy."""
    from oracle import mg_oracle as orc
    case, cyc, tol, maxit = "3d-a2", "V", 1e-8, 12
    p, cells, _ = _param(mg, case, cyc)
    A, mesh = mg.poisson_shifted(cells)
    mg.MGsetup(A, mesh, p, nrhs)
    B = mg.seeded_rhs(A, nrhs)
    p.relativeTol, p.maxOuterIter = tol, maxit
    Af = lambda V: A @ V
    M = orc.getMultigridPreconditioner(p, B)
    Xcg, fcg, rcg, icg = orc.blockCG(Af, B, tol, maxit, M)
    Xbi, fbi, ibi, rbi = orc.blockBiCGSTB(Af, B, tol, maxit, M)
    Xgm, fgm, igm, rgm = orc.blockFGMRES(Af, B, 3, tol, maxit, M)
    Gs = _setup_all_ranks(mg, world, case, cyc, nrhs=nrhs)
    opts = dict(TEST_OPTS, rowclass_min_cover=0.05, march_max_len=64, march4_ty_max=12)

    def body(r, G, H):
        own = G.levels[0].own_mask()
        bt = torch.from_numpy(np.ascontiguousarray(B[G.gid_fine])).cuda()
        outs = []
        for name in ("pcg", "bicgstab", "fgmres"):
            xt = torch.zeros_like(bt)
            if name == "pcg":
                flag, it, res = H.pcg(bt, xt, tol, maxit)
            elif name == "bicgstab":
                flag, it, res = H.bicgstab(bt, xt, tol, maxit)
            else:
                flag, it, res = H.fgmres(bt, xt, 3, tol, maxit)
            outs.append((xt.cpu().numpy()[own], int(flag), int(it), np.asarray(res)))
        return G.gid_fine[own], outs

    out = _thread_world_run(Gs, body, options=opts)
    for j, (Xo, fo, io, ro) in enumerate(((Xcg, fcg, icg, rcg.max(axis=1)), (Xbi, fbi, ibi, rbi), (Xgm, fgm, igm, rgm))):
        X = np.zeros_like(B)
        for gid, outs in out:
            xl, flag, it, res = outs[j]
            X[gid] = xl
            assert flag == fo and it == io, (j, flag, fo, it, io)
            res = res.max(axis=1) if res.ndim == 2 else res
            assert len(res) == len(ro) and np.abs(res - ro).max() <= 1e-8, (j, res, ro)
        assert np.abs(X - Xo).max() <= 1e-8 * np.abs(Xo).max()
