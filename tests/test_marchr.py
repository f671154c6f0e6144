"""The restriction bc = R r (MGcycle.jl:66) as a walk along z over the fine planes (csr_rowclass_marchr_spmv).

Same products in the same order as the gather form (csr_rowclass_lane_spmv): compared bit for bit with the handle that has the form
switched off, against numpy to kernel tolerance, and through solveMG (where the restriction also writes the coarse level's first
update d.*bc) against the oracle - 1e-10 on the residual history (BASELINE north_star)."""
import numpy as np
import pytest

from oracle import mg_oracle as orc

RES_TOL = 1e-10


def _env(monkeypatch, segs=0, tx=0, ty=0):
    monkeypatch.setenv("MG_ROWCLASS_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_SMALL", "1")           # (this test is about the row-class forms: keep the small-level kernels out)
    monkeypatch.setenv("MG_ROWCLASS_MAX_PASSES", "64")
    monkeypatch.setenv("MG_ROWCLASS_MIN_COVER", "0.05")
    monkeypatch.setenv("MG_MARCH_MIN_WG", "0")
    monkeypatch.setenv("MG_MARCHR_MIN_ROWS", "0")
    monkeypatch.setenv("MG_NO_WAVE_RESTRICT", "1")   # (... and the wavefront form, which takes such grid pairs first)
    monkeypatch.setenv("MG_MARCHR_SEGS", str(segs))
    monkeypatch.setenv("MG_MARCHR_TX", str(tx))
    monkeypatch.setenv("MG_MARCHR_TY", str(ty))


def _setup(mg, ncells, levels, cyc="V", pre=2, post=1, maxIter=6, tol=1e-10):
    A, mesh = mg.poisson_shifted(ncells)
    p = mg.getMGparam(np.float64, np.int64, levels, 8, maxIter, tol, "Jac", 0.8, pre, post, cyc, "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p, 1)
    return A, p, mg.seeded_rhs(A, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("cells,segs,tx,ty", [([64, 48, 40], 0, 0, 0), ([40, 72, 44], 5, 8, 16), ([16, 16, 16], 2, 4, 4)])
def test_marching_restriction_bit_identical_to_the_gather_form(mg, built, monkeypatch, cells, segs, tx, ty):
    import torch
    from multigrid_jl_amd import device as dev
    _env(monkeypatch, segs, tx, ty)
    A, p, b = _setup(mg, cells, 3)
    rng = np.random.default_rng(sum(cells))
    rhs = {lvl: rng.standard_normal(p.Rs[lvl - 1].shape[1]) for lvl in (1, 2)}
    out = {}
    for name, off in (("march", "0"), ("gather", "1")):
        monkeypatch.setenv("MG_NO_MARCHR", off)
        h = mg.to_device(p)
        for lvl in (1, 2):
            R = p.Rs[lvl - 1]
            if lvl == 1:      # (level 2 of the smallest case has classes of one row each: exception rows, the gather form stays)
                assert (h.operator_kernel_variant(lvl, dev.MG_OP_R) == 7) == (off == "0")
            else:
                assert off == "0" or h.operator_kernel_variant(lvl, dev.MG_OP_R) != 7
            r = torch.from_numpy(rhs[lvl]).cuda()
            bc = torch.full((R.shape[0],), np.nan, dtype=torch.float64, device="cuda")
            h.spmv_dev(lvl, dev.MG_OP_R, 1.0, r, 0.0, bc)
            out[(name, lvl)] = bc.cpu().numpy()
        h.close()
        p.device = None
    for lvl in (1, 2):
        assert np.array_equal(out[("march", lvl)], out[("gather", lvl)])
        want = p.Rs[lvl - 1] @ rhs[lvl]
        assert np.abs(out[("march", lvl)] - want).max() <= 1e-13 * np.abs(want).max()
    mg.clear_(p)


@pytest.mark.gpu
@pytest.mark.parametrize("cells,levels,cyc", [([32, 48, 32], 4, "W")])
def test_solve_with_the_marching_restriction(mg, built, monkeypatch, cells, levels, cyc):
    """solveMG with the restrictions on the marching form (bc and the coarse level's first update d.*bc in one launch): history and
    iterate against the oracle, and the same iterates as with the form off."""
    _env(monkeypatch)
    runs = {}
    for name, off in (("march", "0"), ("gather", "1")):
        monkeypatch.setenv("MG_NO_MARCHR", off)
        A, p, b = _setup(mg, cells, levels, cyc=cyc)
        mg.to_device(p)
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
    assert np.array_equal(runs["march"][0], runs["gather"][0])
    assert np.array_equal(runs["march"][1], runs["gather"][1])
