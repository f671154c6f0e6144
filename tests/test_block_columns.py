"""A block of right-hand sides solved column by column on the single-vector kernels (solve_dev_columns).

solveMG on a block (SolveFuncs.jl:3-39 with B of several columns): every column gets the cycle, ONE stopping test on the block norm.
Checked against the oracle's block solve (1e-10 on the history: BASELINE north_star), against the block kernels (MG_NO_COLUMNS=1),
and - iterate by iterate - against solveMG on each column alone with the same step count (bit-identical: the same kernels)."""
import numpy as np
import pytest

from oracle import mg_oracle as orc

RES_TOL = 1e-10


def _env(monkeypatch):
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_ROWCLASS_MIN_COVER", "0.05")
    monkeypatch.setenv("MG_MARCH_MIN_WG", "0")
    monkeypatch.setenv("MG_MARCH_MAX_LEN", "64")


def _setup(mg, cells, levels, nrhs, cyc="V", tol=1e-10, maxIter=6):
    A, mesh = mg.poisson_shifted(cells)
    p = mg.getMGparam(np.float64, np.int64, levels, 8, maxIter, tol, "Jac", 0.8, 2, 1, cyc, "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p, nrhs)
    return A, p, mg.seeded_rhs(A, nrhs)


@pytest.mark.gpu
@pytest.mark.parametrize("cells,levels,nrhs,cyc,tol", [([40, 31, 17], 3, 3, "W", 1e-10), ([33, 25, 15], 2, 2, "V", 3e-3), ([23, 23, 23], 2, 5, "F", 1e-30)])
def test_block_solved_column_by_column(mg, built, monkeypatch, cells, levels, nrhs, cyc, tol):
    _env(monkeypatch)
    runs = {}
    for name, off in (("columns", "0"), ("block", "1")):
        monkeypatch.setenv("MG_NO_COLUMNS", off)
        A, p, B = _setup(mg, cells, levels, nrhs, cyc=cyc, tol=tol, maxIter=7)
        h = mg.to_device(p)
        assert h.four_stage_form(1)[0]
        X = np.zeros_like(B)
        _, _, it = mg.solveMG(p, B, X)
        hist = {}
        Xo = np.zeros_like(B)
        _, _, ito = orc.solveMG(p, B, Xo, False, hist)
        assert it == ito, (it, ito)
        assert np.abs(p.resvec - hist["resvec"]).max() / hist["resvec"][0] < RES_TOL
        assert np.abs(X - Xo).max() <= RES_TOL * np.abs(Xo).max()
        # from a given non-zero block
        X2 = np.asfortranarray(np.random.default_rng(5).standard_normal(B.shape))
        Xo2 = X2.copy(order="F")
        mg.solveMG(p, B, X2)
        orc.solveMG(p, B, Xo2, False, {})
        assert np.abs(X2 - Xo2).max() <= RES_TOL * np.abs(Xo2).max()
        runs[name] = (X.copy(), np.asarray(p.resvec).copy(), it)
        mg.clear_(p)
    assert runs["columns"][2] == runs["block"][2]
    if tol == 3e-3:
        assert 1 < runs["columns"][2] < 7          # (meant to stop early)
    assert np.abs(runs["columns"][0] - runs["block"][0]).max() <= 1e-12 * np.abs(runs["block"][0]).max()
    assert np.abs(runs["columns"][1] - runs["block"][1]).max() <= 1e-13 * runs["block"][1][0]
    # every column = solveMG on that column alone with the block's step count (tol 0: the count decides)
    monkeypatch.setenv("MG_NO_COLUMNS", "0")
    steps = runs["columns"][2]
    A, p1, b1 = _setup(mg, cells, levels, 1, cyc=cyc, tol=0.0, maxIter=steps)
    _, _, B = _setup(mg, cells, levels, nrhs, cyc=cyc)
    for c in range(nrhs):
        xc = np.zeros(A.shape[0])
        mg.solveMG(p1, np.ascontiguousarray(B[:, c]), xc)
        assert np.array_equal(xc, runs["columns"][0][:, c])
    mg.clear_(p1)


@pytest.mark.gpu
def test_block_cycle_and_column_solve_share_one_handle(mg, built, monkeypatch):
    """the column-wise solve plays one vector at a time on the handle's OWN coarse buffers - the pointers a block cycle
    (mg_cycle_dev with nrhs = k: the preconditioner call of a block Krylov method) keys its HIP graphs with.  A W-cycle on a grid
    small enough that the graphs start at level 1: block cycle, column-wise solve, block cycle again on ONE handle - every result
    must equal the oracle's (a graph captured for k columns replayed for one, or the reverse, would not)."""
    _env(monkeypatch)
    monkeypatch.setenv("MG_NO_COLUMNS", "0")
    A, p, B = _setup(mg, [24, 19, 16], 3, 4, cyc="W", tol=0.0, maxIter=3)      # (20 lines: an even count fits one tile row of the four-stage pass)
    h = mg.to_device(p)
    assert h.four_stage_form(1)[0]
    for rep in range(2):
        Z = np.zeros_like(B)
        mg.recursiveCycle(p, B, Z, 1)                          # block cycle (SpMM kernels, graphs keyed with nrhs = 4)
        Zo = orc.recursiveCycle(p, B, np.zeros_like(B), 1, None, "W")
        assert np.abs(Z - Zo).max() <= RES_TOL * np.abs(Zo).max(), rep
        X = np.zeros_like(B)
        mg.solveMG(p, B, X)                                    # column-wise solve (single-vector kernels, nrhs = 1 inside)
        Xo = np.zeros_like(B)
        hist = {}
        orc.solveMG(p, B, Xo, False, hist)
        assert np.abs(p.resvec - hist["resvec"]).max() / hist["resvec"][0] < RES_TOL, rep
        assert np.abs(X - Xo).max() <= RES_TOL * np.abs(Xo).max(), rep
    assert h.graph_launches()[1] >= 2          # (graphs of both kinds are cached side by side)
    mg.clear_(p)
