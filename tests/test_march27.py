"""27-point levels (the Galerkin coarse operators of the 7-point grid operator) on the z-marching form csr_rowclass_march27_spmv.

Reference operations: relax's sweep x + d.*(b - A x) (MGcycle.jl:129-131), the residual r = b - A x (MGcycle.jl:58-60), and the two in
one walk.  Same products in the same order as the launches it replaces: compared bit for bit with the handle that has the form switched
off, against numpy to kernel tolerance, and through solveMG against the oracle (1e-10 on the residual history: BASELINE north_star)."""
import numpy as np
import pytest

from oracle import mg_oracle as orc

RES_TOL = 1e-10


def _env(monkeypatch, nt=0, tiles_x=0, segs=0):
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_ROWCLASS_MIN_COVER", "0.05")
    monkeypatch.setenv("MG_MARCH_MIN_WG", "0")
    monkeypatch.setenv("MG_MARCH27_MIN_ROWS", "0")
    monkeypatch.setenv("MG_MARCH27_NT", str(nt))
    monkeypatch.setenv("MG_MARCH27_TILES_X", str(tiles_x))
    monkeypatch.setenv("MG_MARCH27_SEGS", str(segs))


def _setup(mg, ncells, levels, cyc="V", pre=2, post=1, maxIter=6, tol=1e-10):
    A, mesh = mg.poisson_shifted(ncells)
    p = mg.getMGparam(np.float64, np.int64, levels, 8, maxIter, tol, "Jac", 0.8, pre, post, cyc, "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p, 1)
    return A, p, mg.seeded_rhs(A, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("cells,nt,tiles_x,segs", [([64, 48, 40], 0, 0, 0), ([48, 34, 36], 512, 2, 3), ([38, 70, 44], 256, 0, 0)])
def test_27_point_level_single_products_and_pair_bit_identical(mg, built, monkeypatch, cells, nt, tiles_x, segs):
    """Level 2 of a 3-level hierarchy (27-point Galerkin operator): sweep, residual and the pair on the marching form against
    the same handle with MG_NO_MARCH27=1 (plane tiles, two launches) - equal bits - and against numpy."""
    import torch
    from multigrid_jl_amd import device as dev
    _env(monkeypatch, nt, tiles_x, segs)
    A, p, b = _setup(mg, cells, 3)
    out = {}
    Al, dl = p.As[1], p.relaxPrecs[1]
    rng = np.random.default_rng(sum(cells))
    xh, bh = rng.standard_normal(Al.shape[0]), rng.standard_normal(Al.shape[0])
    for name, off in (("m27", "0"), ("tile", "1")):
        monkeypatch.setenv("MG_NO_MARCH27", off)
        h = mg.to_device(p)
        form, geo = h.sweep_residual_form(2)
        assert (form == 5) == (off == "0"), (form, geo)
        if off == "0":
            assert (nt == 0 or geo[8] == nt) and (tiles_x == 0 or geo[0] == tiles_x) and (segs == 0 or geo[9] == segs), geo
        x, bb = torch.from_numpy(xh).cuda(), torch.from_numpy(bh).cuda()
        s, r = torch.full_like(x, np.nan), torch.full_like(x, np.nan)
        h.fused_dev(2, dev.MG_K_SMOOTH, bb, x, s)
        h.fused_dev(2, dev.MG_K_RESIDUAL, bb, x, r)
        t2, r2 = torch.full_like(x, np.nan), torch.full_like(x, np.nan)
        if off == "0":
            h.sweep_residual_dev(2, bb, x, t2, r2)
        else:
            h.fused_dev(2, dev.MG_K_SMOOTH, bb, x, t2)
            h.fused_dev(2, dev.MG_K_RESIDUAL, bb, t2, r2)
        out[name] = [v.cpu().numpy() for v in (s, r, t2, r2)]
        h.close()
        p.device = None
    for a, c in zip(out["m27"], out["tile"]):
        assert np.array_equal(a, c)                        # (0.0 == -0.0)
    s_w = xh + dl * (bh - Al @ xh)
    r_w = bh - Al @ xh
    r2_w = bh - Al @ s_w
    for got, want in zip(out["m27"], (s_w, r_w, s_w, r2_w)):
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    mg.clear_(p)


@pytest.mark.gpu
@pytest.mark.parametrize("cells,levels,cyc,pre,post", [([32, 48, 32], 4, "W", 1, 1), ([40, 36, 48], 3, "F", 3, 2)])
def test_solve_with_27_point_marching_levels(mg, built, monkeypatch, cells, levels, cyc, pre, post):
    """solveMG with levels 2.. on the marching form: history and iterate against the oracle, and the same iterates as with the form off."""
    _env(monkeypatch)
    runs = {}
    for name, off in (("m27", "0"), ("tile", "1")):
        monkeypatch.setenv("MG_NO_MARCH27", off)
        A, p, b = _setup(mg, cells, levels, cyc=cyc, pre=pre, post=post)
        h = mg.to_device(p)
        assert (h.sweep_residual_form(2)[0] == 5) == (off == "0")
        x = np.zeros_like(b)
        _, _, it = mg.solveMG(p, b, x)
        hist = {}
        xo = np.zeros_like(b)
        _, _, ito = orc.solveMG(p, b, xo, False, hist)
        assert it == ito
        assert np.abs(p.resvec - hist["resvec"]).max() / hist["resvec"][0] < RES_TOL
        assert np.abs(x - xo).max() <= RES_TOL * np.abs(xo).max()
        runs[name] = (x.copy(), np.asarray(p.resvec).copy())
        mg.clear_(p)
    assert np.array_equal(runs["m27"][0], runs["tile"][0])
    assert np.array_equal(runs["m27"][1], runs["tile"][1])


@pytest.mark.gpu
@pytest.mark.parametrize("cells,levels,cyc", [([32, 48, 32], 4, "V"), ([38, 70, 44], 3, "W")])
def test_27_point_pair_from_zero(mg, built, monkeypatch, cells, levels, cyc):
    """V(2,1): a 27-point level is entered with x = 0 (MGcycle.jl:29), so its first update is x1 = d.*b - formed INSIDE the pair's walk
    (x1 = d.*b ; sweep ; residual: three stages, x not read; the restriction into the level writes no x1).  Same bits as the pair
    reading the x1 the restriction wrote (MG_NO_MARCH27_ZERO=1) and as the plane tiles; history and iterate against the oracle; the
    pair of level 2 moved one vector less per launch."""
    _env(monkeypatch)
    runs, moved = {}, {}
    for name, env in (("zero", {}), ("x1", {"MG_NO_MARCH27_ZERO": "1"}), ("tile", {"MG_NO_MARCH27": "1"})):
        for k in ("MG_NO_MARCH27_ZERO", "MG_NO_MARCH27"):
            monkeypatch.setenv(k, env.get(k, "0"))
        A, p, b = _setup(mg, cells, levels, cyc=cyc, pre=2, post=1, maxIter=5)
        h = mg.to_device(p)
        h.profile_enable(True)
        x = np.zeros_like(b)
        _, _, it = mg.solveMG(p, b, x)
        prof, mv = h.profile(), h.profile_moved()
        h.profile_enable(False)
        if name != "tile":
            assert h.sweep_residual_form(2)[0] == 5
            moved[name] = mv[(2, "smooth+residual")]
            assert (2, "dscale") not in prof
        hist = {}
        xo = np.zeros_like(b)
        _, _, ito = orc.solveMG(p, b, xo, False, hist)
        assert it == ito
        assert np.abs(p.resvec - hist["resvec"]).max() / hist["resvec"][0] < RES_TOL
        assert np.abs(x - xo).max() <= RES_TOL * np.abs(xo).max()
        runs[name] = (x.copy(), np.asarray(p.resvec).copy())
        n2 = p.As[1].shape[0]
        mg.clear_(p)
    for other in ("x1", "tile"):
        assert np.array_equal(runs["zero"][0], runs[other][0])
        assert np.array_equal(runs["zero"][1], runs[other][1])
    if cyc == "V":      # every visit of level 2 starts from x = 0: each pair launch read b only
        assert moved["x1"] - moved["zero"] == 8.0 * n2, (moved, n2)
    else:               # (W: the second visit carries x)
        assert 0.0 < moved["x1"] - moved["zero"] < 8.0 * n2, (moved, n2)
