"""-m gpu: the one-trip kernels of the small grid levels (csrc/mg_small.hpp: grid27_small_spmv / _restrict, grid_small_prolong).
They replace the streaming kernels on grid levels below `rowclass_min_rows`; every product is compared with numpy on the host's
operators and with the kernels they replace (a second handle with no_small = 1), whole solves with the oracle."""
import numpy as np
import pytest

from oracle import mg_oracle as orc

pytestmark = pytest.mark.gpu


def _hier(mg, cells, levels, relax="Jac", om=0.8, cyc="V"):
    A, mesh = mg.poisson_shifted(cells)
    p = mg.getMGparam(np.float64, np.int64, levels, 8, 6, 1e-10, relax, om, 2, 1, cyc, "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p)
    return A, p


@pytest.mark.parametrize("cells,levels,relax", [([24, 16, 32], 3, "Jac"), ([16, 16, 16], 3, "SPAI"), ([40, 24], 3, "Jac"), ([8, 64, 12], 2, "Jac")])
def test_small_level_kernels_vs_numpy_and_streaming_kernels(mg, built, cells, levels, relax):
    from multigrid_jl_amd import device as D
    A, p = _hier(mg, cells, levels, relax, 1.0 if relax == "SPAI" else 0.8)
    h = D.DeviceHierarchy(p, 0, 1)
    h0 = D.DeviceHierarchy(p, 0, 1, options={"no_small": 1, "no_cell_prolong": 1, "no_wave_restrict": 1})
    h1 = D.DeviceHierarchy(p, 0, 1, options={"no_cell_prolong": 1, "no_wave_restrict": 1})
    import torch
    rng = np.random.default_rng(5)
    try:
        nl = len(p.As)
        for l in range(1, nl):
            assert h.operator_kernel_variant(l, D.MG_OP_A) == 8, (l, h.operator_kernel_variant(l, D.MG_OP_A))
            # P: a lane per coarse cell (10; any size) in front of the lane-per-fine-row kernel (8)
            # R: 62 coarse nodes per wavefront (11; any size) in front of the lane-per-coarse-row kernel with its records (8)
            assert h.operator_kernel_variant(l, D.MG_OP_R) == 11 and h.operator_kernel_variant(l, D.MG_OP_P) == 10
            assert h1.operator_kernel_variant(l, D.MG_OP_P) == 8 and h0.operator_kernel_variant(l, D.MG_OP_P) not in (8, 10)
            assert h1.operator_kernel_variant(l, D.MG_OP_R) == 8 and h0.operator_kernel_variant(l, D.MG_OP_R) not in (8, 11)
            assert h0.operator_kernel_variant(l, D.MG_OP_A) != 8
            Al, Pl, Rl, d = p.As[l - 1], p.Ps[l - 1], p.Rs[l - 1], np.asarray(p.relaxPrecs[l - 1])
            n, nc = Al.shape[0], Pl.shape[1]
            x, b, xc = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(nc)
            xt, bt, xct = (torch.from_numpy(v).cuda() for v in (x, b, xc))
            for kern, ref in ((D.MG_K_RESIDUAL, b - Al @ x), (D.MG_K_SMOOTH, x + d * (b - Al @ x))):
                out, out0 = torch.zeros(n, dtype=torch.float64, device="cuda"), torch.zeros(n, dtype=torch.float64, device="cuda")
                h.fused_dev(l, kern, bt, xt, out)
                h0.fused_dev(l, kern, bt, xt, out0)
                scale = np.abs(ref).max()
                assert np.abs(out.cpu().numpy() - ref).max() <= 1e-13 * scale
                assert np.abs(out.cpu().numpy() - out0.cpu().numpy()).max() <= 1e-13 * scale
            # A x, R x, x += P xc
            y = torch.zeros(n, dtype=torch.float64, device="cuda")
            h.spmv_dev(l, D.MG_OP_A, 1.0, xt, 0.0, y)
            assert np.abs(y.cpu().numpy() - Al @ x).max() <= 1e-13 * np.abs(Al @ x).max()
            yc = torch.zeros(nc, dtype=torch.float64, device="cuda")
            h.spmv_dev(l, D.MG_OP_R, 1.0, xt, 0.0, yc)
            assert np.abs(yc.cpu().numpy() - Rl @ x).max() <= 1e-13 * np.abs(Rl @ x).max()
            yc1 = torch.full((nc,), 7.0, dtype=torch.float64, device="cuda")
            h1.spmv_dev(l, D.MG_OP_R, 1.0, xt, 0.0, yc1)
            assert torch.equal(yc, yc1)          # (the wavefront form adds the same products in the same order as the record kernel)
            yf = bt.clone()
            h.spmv_dev(l, D.MG_OP_P, 1.0, xct, 1.0, yf)
            yf0 = bt.clone()
            h0.spmv_dev(l, D.MG_OP_P, 1.0, xct, 1.0, yf0)
            yf1 = bt.clone()
            h1.spmv_dev(l, D.MG_OP_P, 1.0, xct, 1.0, yf1)
            assert np.abs(yf.cpu().numpy() - (b + Pl @ xc)).max() <= 1e-13 * np.abs(b + Pl @ xc).max()
            assert np.abs(yf.cpu().numpy() - yf0.cpu().numpy()).max() <= 1e-13 * np.abs(b + Pl @ xc).max()
            assert torch.equal(yf, yf1)          # (the two arithmetic kernels add the same products in the same order)
    finally:
        h.close()
        h0.close()
        h1.close()


@pytest.mark.parametrize("cells,levels,cyc", [([32, 32, 32], 4, "V"), ([32, 16, 24], 3, "W"), ([64, 48], 4, "F")])
def test_solves_with_small_level_kernels_match_the_oracle(mg, built, cells, levels, cyc):
    A, p = _hier(mg, cells, levels, cyc=cyc)
    b = mg.seeded_rhs(A)
    x = np.zeros_like(b)
    mg.solveMG(p, b, x)
    from multigrid_jl_amd import device as D
    assert p.device.operator_kernel_variant(2, D.MG_OP_A) == 8
    hist = {}
    xo = np.zeros_like(b)
    orc.solveMG(p, b, xo, False, hist)
    assert np.abs(p.resvec - hist["resvec"]).max() <= 1e-10 * hist["resvec"][0]
    assert np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()
    mg.clear_(p)


@pytest.mark.parametrize("cells,levels,cyc,pre,post", [([32, 32, 32], 4, "V", 2, 1), ([36, 20, 28], 3, "W", 1, 2), ([64, 48], 4, "F", 2, 2), ([24, 40, 16], 3, "K", 2, 1)])
def test_two_launches_of_a_small_level_as_one(mg, built, monkeypatch, cells, levels, cyc, pre, post):
    """grid27_small_resid_restrict (r = b - A x formed in LDS inside the restriction's launch, MGcycle.jl:58-66) and
    grid27_small_prolong_smooth (x + P xc formed in LDS inside the first post-sweep's launch, MGcycle.jl:90-102): the same iterates
    and residual history, bit for bit, as the launches they replace (MG_NO_SMALL_FUSE=1), and the oracle's to 1e-10; the profile
    shows the fused launches and no residual launch on the levels they serve."""
    runs = {}
    for name, off in (("fused", "0"), ("plain", "1")):
        monkeypatch.setenv("MG_NO_SMALL_FUSE", off)
        A, mesh = mg.poisson_shifted(cells)
        p = mg.getMGparam(np.float64, np.int64, levels, 8, 5, 1e-10, "Jac", 0.8, pre, post, cyc, "NoMUMPS", 0.5, 0.0)
        mg.MGsetup(A, mesh, p)
        b = mg.seeded_rhs(A)
        h = mg.to_device(p)
        h.profile_enable(True)
        x = np.zeros_like(b)
        _, _, it = mg.solveMG(p, b, x)
        prof = h.profile()
        h.profile_enable(False)
        fused_levels = [l for (l, k) in prof if k == "smooth+prolong"]
        if off == "0":
            assert fused_levels, prof.keys()
            if cyc != "K":
                for l in fused_levels:      # (level 1's residual launches are the solve loop's own: the stopping test)
                    assert ((l, "residual") not in prof or l == 1) and (l, "prolong") not in prof, (l, prof.keys())
        else:
            assert not fused_levels
        hist = {}
        xo = np.zeros_like(b)
        _, _, ito = orc.solveMG(p, b, xo, False, hist)
        assert it == ito
        assert np.abs(p.resvec - hist["resvec"]).max() <= 1e-10 * hist["resvec"][0]
        assert np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()
        runs[name] = (x.copy(), np.asarray(p.resvec).copy())
        mg.clear_(p)
    assert np.array_equal(runs["fused"][0], runs["plain"][0])
    assert np.array_equal(runs["fused"][1], runs["plain"][1])


def test_transfer_kernels_take_unaligned_vectors_through_another_form(mg, built):
    """grid_wave_restrict / grid_cell_prolong read the fine vector in 16-byte pairs (the pair at the last entry of an odd-length vector stays
    inside one aligned granule ONLY for a 16-byte aligned base): a vector at an 8-byte aligned address takes another kernel, same result."""
    import torch
    from multigrid_jl_amd import device as D
    A, p = _hier(mg, [24, 16, 32], 3)
    h = D.DeviceHierarchy(p, 0, 1)
    try:
        rng = np.random.default_rng(11)
        Pl, Rl = p.Ps[0], p.Rs[0]
        n, nc = Pl.shape
        x, xc = rng.standard_normal(n), rng.standard_normal(nc)
        big = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
        xa = torch.from_numpy(x).cuda()
        xu = big[1:]
        xu.copy_(xa)
        assert xa.data_ptr() % 16 == 0 and xu.data_ptr() % 16 == 8
        ya, yu = torch.zeros(nc, dtype=torch.float64, device="cuda"), torch.zeros(nc, dtype=torch.float64, device="cuda")
        h.spmv_dev(1, D.MG_OP_R, 1.0, xa, 0.0, ya)
        h.spmv_dev(1, D.MG_OP_R, 1.0, xu, 0.0, yu)
        ref = Rl @ x
        assert np.abs(ya.cpu().numpy() - ref).max() <= 1e-13 * np.abs(ref).max()
        assert np.abs(yu.cpu().numpy() - ref).max() <= 1e-13 * np.abs(ref).max()
        xct = torch.from_numpy(xc).cuda()
        fa = xa.clone()
        big2 = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
        fu = big2[1:]
        fu.copy_(xa)
        h.spmv_dev(1, D.MG_OP_P, 1.0, xct, 1.0, fa)
        h.spmv_dev(1, D.MG_OP_P, 1.0, xct, 1.0, fu)
        ref = x + Pl @ xc
        assert np.abs(fa.cpu().numpy() - ref).max() <= 1e-13 * np.abs(ref).max()
        assert np.abs(fu.cpu().numpy() - ref).max() <= 1e-13 * np.abs(ref).max()
        assert big2[0].item() == 0.0
    finally:
        h.close()
    mg.clear_(p)


def test_small_levels_follow_new_values(mg, built):
    """replaceMatrixInHierarchy on the device (mg_rap_FP64): the position-code records are rebuilt from the new values."""
    A, p = _hier(mg, [16, 16, 16], 3)
    b = mg.seeded_rhs(A)
    x = np.zeros_like(b)
    mg.solveMG(p, b, x)
    A2 = (A * 1.75).tocsr()
    mg.replaceMatrixInHierarchy(p, A2)
    x = np.zeros_like(b)
    mg.solveMG(p, b, x)
    hist = {}
    xo = np.zeros_like(b)
    orc.solveMG(p, b, xo, False, hist)
    assert np.abs(p.resvec - hist["resvec"]).max() <= 1e-10 * hist["resvec"][0]
    assert np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()
    mg.clear_(p)


@pytest.mark.parametrize("n,m,avg", [(3000, 3000, 400), (517, 2100, 1500), (64, 64, 64), (2000, 900, 120), (300, 200000, 1200), (1500, 1500, 1100)])
def test_long_row_kernel_vs_numpy(mg, built, n, m, avg):
    """csr_longrow_spmv (one wavefront per row; operators without row classes whose rows average >= 1000 entries - the Galerkin
    levels of an SA-AMG hierarchy; 16-bit column offsets where every row spans < 65536 columns, 32-bit ones in the 200 000-column
    case; the shorter-row cases stay on csr_stream_spmv): y = alpha A x + beta y [+ d.*y], b - A x, x + d.*(b - A x) against numpy, ragged rows included
    (empty rows, one row far longer than the rest)."""
    import scipy.sparse as sp
    import torch
    from multigrid_jl_amd import device as D
    rng = np.random.default_rng(n + m)
    lens = np.clip(rng.poisson(avg, n), 0, m)
    lens[0] = 0
    lens[n // 2] = min(m, 4 * avg)
    rows = np.repeat(np.arange(n), lens)
    cols = np.concatenate([np.sort(rng.choice(m, size=int(k), replace=False)) for k in lens]) if lens.sum() else np.zeros(0, dtype=int)
    M = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(n, m))
    M.sort_indices()
    op = D.DeviceOperator(M, 0)
    op0 = None
    try:
        x = rng.standard_normal(m)
        y0 = rng.standard_normal(n)
        xt = torch.from_numpy(x).cuda()
        y = torch.from_numpy(y0.copy()).cuda()
        op.apply(D.MG_K_SPMV, xt, y, alpha=-1.5, beta=0.25)
        ref = -1.5 * (M @ x) + 0.25 * y0
        assert np.abs(y.cpu().numpy() - ref).max() <= 1e-13 * max(1.0, np.abs(ref).max())
        if n == m:
            b, d = rng.standard_normal(n), rng.standard_normal(n)
            bt, dt = torch.from_numpy(b).cuda(), torch.from_numpy(d).cuda()
            out = torch.zeros(n, dtype=torch.float64, device="cuda")
            op.apply(D.MG_K_RESIDUAL, xt, out, b=bt)
            ref = b - M @ x
            assert np.abs(out.cpu().numpy() - ref).max() <= 1e-13 * np.abs(ref).max()
            op.apply(D.MG_K_SMOOTH, xt, out, b=bt, d=dt)
            ref = x + d * (b - M @ x)
            assert np.abs(out.cpu().numpy() - ref).max() <= 1e-13 * np.abs(ref).max()
    finally:
        op.close()


def _divsiggrad(mg, cells, levels, seed=3):
    import scipy.sparse as sp
    mesh = mg.getRegularMesh([0.0, 1.0] * len(cells), cells)
    sigma = np.exp(np.random.default_rng(seed).standard_normal(int(np.prod(cells))))
    A = mg.getNodalDivSigGradMatrix(mesh, sigma)
    A = (A + 1e-3 * abs(A).sum(axis=0).max() * sp.identity(A.shape[0], format="csr")).tocsr()
    A.sort_indices()
    p = mg.getMGparam(np.float64, np.int64, levels, 8, 6, 1e-10, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
    mg.MGsetup(A, mesh, p)
    return A, mesh, p


@pytest.mark.parametrize("cells,levels", [([24, 16, 20], 3), ([16, 16, 16], 3), ([40, 24], 3)])
def test_band27_variable_coefficient_levels(mg, built, cells, levels):
    """Variable coefficients (div sigma grad: testGMG.jl:57-75): the 27-point (2-D: 9-point) Galerkin levels have no two equal rows;
    they are held as 27 planar value arrays (grid27_band_spmv, kernel variant 9).  Products against numpy and against the
    pattern-coded CSR kernels (no_band27), solve against the oracle, and the planar values follow replaceMatrixInHierarchy."""
    import torch
    from multigrid_jl_amd import device as D
    A, mesh, p = _divsiggrad(mg, cells, levels)
    h = D.DeviceHierarchy(p, 0, 1, options={"band_sym_tol": 1})
    h0 = D.DeviceHierarchy(p, 0, 1, options={"no_band27": 1})
    rng = np.random.default_rng(9)
    try:
        assert h.operator_kernel_variant(2, D.MG_OP_A) == 9 and h0.operator_kernel_variant(2, D.MG_OP_A) != 9
        # a Galerkin operator of a symmetric fine one is symmetric up to the rounding of R*(A*P): with the option band_sym_tol 14 of the
        # 27 planes are read; by DEFAULT (round 6) only bit-for-bit symmetry takes the symmetric reads - the device operator is the stored one
        assert h.band_form(2) == [2, 1, 1, 14] and h0.band_form(2)[0] == 0
        hd = D.DeviceHierarchy(p, 0, 1)
        A2 = p.As[1].tocsr()
        exact = (A2 != A2.T).nnz == 0
        assert hd.band_form(2) == ([2, 1, 1, 14] if exact else [2, 1, 0, 27])
        hd.close()
        h7 = D.DeviceHierarchy(p, 0, 1, options={"no_band_sym": 1})
        assert h7.band_form(2) == [2, 1, 0, 27]
        h7.close()
        for l in range(2, len(p.As)):
            Al, d = p.As[l - 1], np.asarray(p.relaxPrecs[l - 1])
            n = Al.shape[0]
            x, b = rng.standard_normal(n), rng.standard_normal(n)
            xt, bt = torch.from_numpy(x).cuda(), torch.from_numpy(b).cuda()
            for kern, ref in ((D.MG_K_RESIDUAL, b - Al @ x), (D.MG_K_SMOOTH, x + d * (b - Al @ x))):
                out, out0 = torch.zeros(n, dtype=torch.float64, device="cuda"), torch.zeros(n, dtype=torch.float64, device="cuda")
                h.fused_dev(l, kern, bt, xt, out)
                h0.fused_dev(l, kern, bt, xt, out0)
                assert np.abs(out.cpu().numpy() - ref).max() <= 1e-13 * np.abs(ref).max()
                assert np.abs(out.cpu().numpy() - out0.cpu().numpy()).max() <= 1e-13 * np.abs(ref).max()
            y = torch.from_numpy(b.copy()).cuda()
            h.spmv_dev(l, D.MG_OP_A, -0.5, xt, 2.0, y)
            ref = -0.5 * (Al @ x) + 2.0 * b
            assert np.abs(y.cpu().numpy() - ref).max() <= 1e-13 * np.abs(ref).max()
    finally:
        h.close()
        h0.close()
    b = mg.seeded_rhs(A)
    x = np.zeros_like(b)
    mg.solveMG(p, b, x)
    hist = {}
    xo = np.zeros_like(b)
    orc.solveMG(p, b, xo, False, hist)
    assert np.abs(p.resvec - hist["resvec"]).max() <= 1e-10 * hist["resvec"][0]
    assert np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()
    # new sigma, same pattern: the resident planar arrays are refilled (mg_rap_FP64)
    A2, _, _ = _divsiggrad(mg, cells, levels, seed=4)
    dev_before = p.device
    mg.replaceMatrixInHierarchy(p, A2)
    assert p.device is dev_before and p.device.operator_kernel_variant(2, D.MG_OP_A) == 9
    b2 = mg.seeded_rhs(A2)
    x = np.zeros_like(b2)
    mg.solveMG(p, b2, x)
    hist = {}
    xo = np.zeros_like(b2)
    orc.solveMG(p, b2, xo, False, hist)
    assert np.abs(p.resvec - hist["resvec"]).max() <= 1e-10 * hist["resvec"][0]
    assert np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max()
    mg.clear_(p)


def test_galerkin_product_of_a_nearly_dense_level_on_the_gpu(mg, built, monkeypatch):
    """Setup only: R*(A*P) of a nearly dense SA-AMG level as dense GEMMs on the GPU (hostlib.galerkin_dense_gpu) - the structural
    pattern of the host's sparse product (explicit zeros and cancelling sums kept), values to rounding; and the switch that
    hands such levels to it (size, density, MG_SETUP_GPU)."""
    import scipy.sparse as sp
    from multigrid_jl_amd import hostlib as H
    rng = np.random.default_rng(12)
    n, nc = 700, 90
    A = sp.random(n, n, density=0.2, random_state=3, format="csr", dtype=np.float64)
    A = (A + A.T + sp.identity(n) * 10.0).tocsr()
    A.sort_indices()
    A.data[5] = 0.0                                          # an explicit zero stays a structural entry
    P = sp.random(n, nc, density=0.15, random_state=4, format="lil", dtype=np.float64)
    P[0, 0] = 1.0
    P[1, 0] = -1.0                                           # (sums that may cancel stay entries too)
    P = sp.csr_matrix(P)
    P.sort_indices()
    R = H.transpose_csr(P)
    want = H.spgemm(R, H.spgemm(A, P))
    got = H.galerkin_dense_gpu(R, A, P)
    assert got.shape == want.shape and np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
    assert np.abs(got.data - want.data).max() <= 1e-13 * np.abs(want.data).max()
    got2 = H.galerkin_sparse_gpu(R, A, P)                    # rocSPARSE's SpGEMM (levels too large for dense blocks)
    assert got2 is not None and np.array_equal(got2.indptr, want.indptr) and np.array_equal(got2.indices, want.indices)
    assert np.abs(got2.data - want.data).max() <= 1e-13 * np.abs(want.data).max()
    monkeypatch.setenv("MG_SETUP_GPU_MIN_ROWS", "500")
    monkeypatch.delenv("MG_SETUP_GPU", raising=False)
    assert not H.galerkin_dense_gpu_ok(A, P)                 # OPT-IN: by default the setup stays on the host, as the reference's does
    monkeypatch.setenv("MG_SETUP_GPU", "1")
    assert H.galerkin_dense_gpu_ok(A, P)
    assert not H.galerkin_sparse_gpu_ok(A, P)                # (10^7 products: the host's)
    monkeypatch.setenv("MG_SETUP_GPU_MIN_ROWS", "3000")
    assert not H.galerkin_dense_gpu_ok(A, P)                 # (700 rows: below the size it pays from)
    monkeypatch.setenv("MG_SETUP_GPU_MIN_ROWS", "500")
    thin = sp.random(4000, 4000, density=0.001, random_state=5, format="csr")
    assert not H.galerkin_dense_gpu_ok(thin, sp.random(4000, 300, density=0.01, random_state=6, format="csr"))


@pytest.mark.parametrize("cells,nd,rank", [([32, 32, 64], [1, 1, 2], 1), ([32, 32, 64], [2, 2, 2], 0), ([32, 32, 64], [2, 2, 2], 7), ([64, 96], [2, 2], 3)])
def test_embedded_grid_pair_transfers(mg, built, cells, nd, rank):
    """Round 6: the extended boxes of the sharded cycle carry their own ghost widths, so a level's box pairs with a SUB-BOX of the
    next level's box (ghost_dist.ghost_boxes).  grid_wave_restrict / grid_cell_prolong take that sub-box at its offset: R's rows
    outside it are empty (written as 0), P reads no column outside it.  One rank's local operators against scipy."""
    import torch
    from multigrid_jl_amd import device as D, ghost_dist as gd
    from multigrid_jl_amd.structured_setup import poisson_operator
    world = int(np.prod(nd))
    p = mg.getMGparam(np.float64, np.int64, 5 if len(cells) == 3 else 4, 8, 6, 1e-10, "Jac", 0.8, 2, 1, "V", "NoMUMPS", 0.5, 0.0)
    G = gd.ghost_gmg(cells, nd, rank, world, p, poisson_operator(cells), replicate_below=1000 if len(cells) == 3 else 300, dry_tail=True)
    assert G.a >= 2
    h = D.DeviceHierarchy(G.param, 0, 1)
    h0 = D.DeviceHierarchy(G.param, 0, 1, options={"no_small": 1, "no_cell_prolong": 1, "no_wave_restrict": 1})
    rng = np.random.default_rng(11)
    try:
        embedded = 0
        for l in range(1, G.a):
            Pl, Rl = G.param.Ps[l - 1], G.param.Rs[l - 1]
            nf, nc = Pl.shape
            fine_n, coarse_n = G.levels[l - 1].ext_n, G.levels[l].ext_n
            embedded += any(2 * c - 1 != f for f, c in zip(fine_n, coarse_n))
            assert h.operator_kernel_variant(l, D.MG_OP_R) == 11 and h.operator_kernel_variant(l, D.MG_OP_P) == 10, (l, fine_n, coarse_n)
            assert h0.operator_kernel_variant(l, D.MG_OP_R) not in (8, 11) and h0.operator_kernel_variant(l, D.MG_OP_P) not in (8, 10)
            r, xc, x = rng.standard_normal(nf), rng.standard_normal(nc), rng.standard_normal(nf)
            rt, xct = torch.from_numpy(r).cuda(), torch.from_numpy(xc).cuda()
            ref = Rl @ r
            for hh in (h, h0):
                yc = torch.full((nc,), 7.0, dtype=torch.float64, device="cuda")
                hh.spmv_dev(l, D.MG_OP_R, 1.0, rt, 0.0, yc)
                assert np.abs(yc.cpu().numpy() - ref).max() <= 1e-13 * np.abs(ref).max()
            assert (yc.cpu().numpy()[np.diff(Rl.indptr) == 0] == 0.0).all()
            ref = x + Pl @ xc
            outs = []
            for hh in (h, h0):
                yf = torch.from_numpy(x).cuda()
                hh.spmv_dev(l, D.MG_OP_P, 1.0, xct, 1.0, yf)
                assert np.abs(yf.cpu().numpy() - ref).max() <= 1e-13 * np.abs(ref).max()
                outs.append(yf)
            assert np.abs(outs[0].cpu().numpy() - outs[1].cpu().numpy()).max() <= 1e-13 * np.abs(ref).max()
        assert embedded >= 1
    finally:
        h.close()
        h0.close()
