"""The solve loop's two fine-level passes across the stopping test as ONE four-stage pass (csr_rowclass_march4_spmv).

Reference order being reproduced: SolveFuncs.jl:24-37 (cycle, r = b - A x, norm, stopping test) around MGcycle.jl:26-31
(r = b - A x on entry), 54 / 122-136 (relax: nu1 sweeps), 58-60 (residual for the restriction).  The pass must give the
same bits as the two two-stage passes it replaces, and the solve the same iterates whether or not the loop stops early."""
import numpy as np
import pytest

from oracle import mg_oracle as orc

RES_TOL = 1e-10      # BASELINE.json north_star: residual history within 1e-10 relative


def _small_grid_env(monkeypatch):
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_ROWCLASS_MIN_COVER", "0.05")
    monkeypatch.setenv("MG_MARCH_MIN_WG", "0")
    monkeypatch.setenv("MG_MARCH_MAX_LEN", "64")


def _setup(mg, ncells, levels, tol=1e-10, maxIter=6, pre=2, post=1, cyc="V", relax="Jac", omega=0.8):
    A, mesh = mg.poisson_shifted(ncells)
    p = mg.getMGparam(np.float64, np.int64, levels, 8, maxIter, tol, relax, omega, pre, post, cyc, "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p, 1)
    return A, p, mg.seeded_rhs(A, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("cells,nt,k1,tiles_x,ty_max", [([33, 25, 15], 768, 3, 0, 0), ([40, 31, 17], 1024, 2, 2, 0), ([23, 23, 23], 512, 4, 1, 9),
                                                        ([130, 9, 9], 1024, 2, 0, 0), ([30, 50, 11], 768, 4, 2, 10)])
def test_four_stage_pass_bit_identical_to_the_two_passes(mg, built, monkeypatch, cells, nt, k1, tiles_x, ty_max):
    """t', r' of the four-stage pass = the outputs of the two two-stage passes chained through xn = t + d.*r, bit for bit
    (same products, same order, same epilogue expressions); ||r|| to rounding (another partition of the partial sums);
    against numpy to kernel tolerance.  Geometries forced: 2 / 3 / 4 rows per lane, partial tiles, several tiles per line,
    several tile rows (strips shifted so that the first / last line of the grid ends / starts a strip)."""
    import torch
    _small_grid_env(monkeypatch)
    monkeypatch.setenv("MG_MARCH4_NT", str(nt))
    monkeypatch.setenv("MG_MARCH4_K1", str(k1))
    monkeypatch.setenv("MG_MARCH4_TILES_X", str(tiles_x))
    monkeypatch.setenv("MG_MARCH4_TY_MAX", str(ty_max))
    A, p, b = _setup(mg, cells, 2)
    h = mg.to_device(p)
    ok, geo = h.four_stage_form(1)
    assert ok and geo[8] == nt and geo[4] == k1 and (tiles_x == 0 or geo[0] == tiles_x) and (ty_max == 0 or geo[3] <= ty_max), geo
    assert h.sweep_residual_form(1)[0] == 3
    Al, dl = p.As[0], p.relaxPrecs[0]
    rng = np.random.default_rng(sum(cells) + nt)
    xh, bh = rng.standard_normal(Al.shape[0]), rng.standard_normal(Al.shape[0])
    x, bb = torch.from_numpy(xh).cuda(), torch.from_numpy(bh).cuda()
    tp, rp = torch.full_like(x, np.nan), torch.full_like(x, np.nan)
    nrm = h.four_stage_dev(1, bb, x, tp, rp)
    # the two passes it replaces
    t, xn = torch.zeros_like(x), torch.zeros_like(x)
    nrm2 = h.sweep_residual_dev(1, bb, x, t, None, xn, True)
    t2, r2 = torch.zeros_like(x), torch.zeros_like(x)
    h.sweep_residual_dev(1, bb, xn, t2, r2)
    assert torch.equal(tp, t2) and torch.equal(rp, r2)     # (0.0 == -0.0: a zero may change its sign)
    assert abs(nrm - nrm2) <= 1e-14 * nrm2
    # numpy
    t_w = xh + dl * (bh - Al @ xh)
    r_w = bh - Al @ t_w
    xn_w = t_w + dl * r_w
    tp_w = xn_w + dl * (bh - Al @ xn_w)
    rp_w = bh - Al @ tp_w
    assert np.abs(tp.cpu().numpy() - tp_w).max() <= 1e-12 * np.abs(tp_w).max()
    assert np.abs(rp.cpu().numpy() - rp_w).max() <= 1e-11 * np.abs(rp_w).max()
    assert abs(nrm - np.linalg.norm(r_w)) <= 1e-12 * np.linalg.norm(r_w)
    mg.clear_(p)


@pytest.mark.gpu
def test_four_stage_pass_with_more_workgroups_than_cus(mg, built, monkeypatch):
    """A plane with more tiles than the chip has CUs (513-node lines) takes several rounds of shorter segments: 183 tiles x 4
    segments = 732 workgroups here, same bits as the two passes."""
    import torch
    _small_grid_env(monkeypatch)
    monkeypatch.setenv("MG_MARCH4_TILES_X", "3")
    monkeypatch.setenv("MG_MARCH4_TY_MAX", "2")
    monkeypatch.setenv("MG_MARCH4_SEGS", "4")
    A, p, b = _setup(mg, [48, 120, 33], 2)
    h = mg.to_device(p)
    ok, geo = h.four_stage_form(1)
    assert ok and geo[5] > 512 and geo[9] == 4, geo
    rng = np.random.default_rng(11)
    x, bb = torch.from_numpy(rng.standard_normal(A.shape[0])).cuda(), torch.from_numpy(rng.standard_normal(A.shape[0])).cuda()
    tp, rp = torch.full_like(x, np.nan), torch.full_like(x, np.nan)
    nrm = h.four_stage_dev(1, bb, x, tp, rp)
    t, xn = torch.zeros_like(x), torch.zeros_like(x)
    nrm2 = h.sweep_residual_dev(1, bb, x, t, None, xn, True)
    t2, r2 = torch.zeros_like(x), torch.zeros_like(x)
    h.sweep_residual_dev(1, bb, xn, t2, r2)
    assert torch.equal(tp, t2) and torch.equal(rp, r2)
    assert abs(nrm - nrm2) <= 1e-14 * nrm2
    mg.clear_(p)


@pytest.mark.gpu
@pytest.mark.parametrize("cells,levels,cyc,tol", [([40, 31, 17], 3, "W", 1e-10), ([23, 23, 23], 2, "F", 1e-30),
                                                  ([33, 25, 15], 2, "V", 3e-3), ([36, 33, 12], 2, "V", 1e-1)])
def test_solve_with_the_four_stage_pass(mg, built, monkeypatch, cells, levels, cyc, tol):
    """solveMG through the four-stage pass: residual history and iterate against the oracle (1e-10), and against the same
    solve with MG_NO_MARCH4=1 - iterates bit-identical, also when the stopping test ends the loop before the step count
    (the speculative stages are dropped and the iterate re-created), on the first step, or never."""
    _small_grid_env(monkeypatch)
    runs = {}
    for name, off in (("four", "0"), ("two", "1")):
        monkeypatch.setenv("MG_NO_MARCH4", off)
        A, p, b = _setup(mg, cells, levels, tol=tol, maxIter=7, cyc=cyc)
        h = mg.to_device(p)
        assert h.four_stage_form(1)[0] == (off == "0")
        x = np.zeros_like(b)
        _, _, it = mg.solveMG(p, b, x)
        hist = {}
        xo = np.zeros_like(b)
        _, _, ito = orc.solveMG(p, b, xo, False, hist)
        assert it == ito, (it, ito)
        assert np.abs(p.resvec - hist["resvec"]).max() / hist["resvec"][0] < RES_TOL
        assert np.abs(x - xo).max() <= RES_TOL * np.abs(xo).max()
        # from a given non-zero iterate
        x2 = np.random.default_rng(7).standard_normal(b.shape)
        xo2 = x2.copy()
        mg.solveMG(p, b, x2)
        orc.solveMG(p, b, xo2, False, {})
        assert np.abs(x2 - xo2).max() <= RES_TOL * np.abs(xo2).max()
        runs[name] = (x.copy(), np.asarray(p.resvec).copy(), x2.copy(), it)
        mg.clear_(p)
    if tol == 3e-3:
        assert 1 < runs["four"][3] < 7          # (the case is meant to stop early, after more than one step)
    assert runs["four"][3] == runs["two"][3]
    assert np.array_equal(runs["four"][0], runs["two"][0])
    assert np.array_equal(runs["four"][2], runs["two"][2])
    assert np.abs(runs["four"][1] - runs["two"][1]).max() <= 1e-14 * runs["two"][1][0]



@pytest.mark.gpu
@pytest.mark.parametrize("tol", [0.0, 2e-3])
def test_final_sum_inside_the_fine_restriction(mg, built, monkeypatch, tol):
    """Default formats: the ||r||^2 partials of a four-stage pass are added up by one more workgroup of the launch behind it - the fine
    restriction of the next cycle (grid_wave_restrict's FinalSum) - instead of a launch of their own: the same history bit for bit as with
    MG_NO_DEFER_SUM=1, by count and when the stopping test ends the loop early, and the oracle's to 1e-10."""
    runs = {}
    monkeypatch.setenv("MG_MARCH_MIN_WG", "0")      # (the marching forms on a grid of 181 000 rows; everything else as shipped)
    for name, off in (("inside", "0"), ("own", "1")):
        monkeypatch.setenv("MG_NO_DEFER_SUM", off)
        A, p, b = _setup(mg, [64, 48, 56], 4, tol=tol, maxIter=6, cyc="V")
        h = mg.to_device(p)
        from multigrid_jl_amd import device as D
        assert h.four_stage_form(1)[0] and h.operator_kernel_variant(1, D.MG_OP_R) == 11      # (the wavefront restriction)
        x = np.zeros_like(b)
        _, _, it = mg.solveMG(p, b, x)
        runs[name] = (x.copy(), np.asarray(p.resvec).copy(), it)
        if name == "inside":
            hist = {}
            xo = np.zeros_like(b)
            _, _, ito = orc.solveMG(p, b, xo, False, hist)
            assert it == ito
            assert np.abs(p.resvec - hist["resvec"]).max() / hist["resvec"][0] < RES_TOL
            assert np.abs(x - xo).max() <= RES_TOL * np.abs(xo).max()
        mg.clear_(p)
    if tol > 0.0:
        assert 1 < runs["inside"][2] < 6
    assert runs["inside"][2] == runs["own"][2]
    assert np.array_equal(runs["inside"][0], runs["own"][0])
    assert np.array_equal(runs["inside"][1], runs["own"][1])
