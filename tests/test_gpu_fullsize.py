"""Size-independent properties of the hot path at benchmark-scale shapes (2^20 / 2^22 features,
millions of rows, generated on the device).  The oracle cannot run at these sizes in seconds, so
the checks are properties that hold at any size: bookkeeping identities, linearity of X^T,
reproducibility, the directional derivative of the objective, monotone line search, consensus
fixed points, and agreement of the exact and the mini-batch FTRL at batch = 1 row panels."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROWS, DIM, DRAWS = 3_000_000, 1 << 22, 131


@pytest.fixture(scope="module")
def big():
    from admm_amd import capi
    ctx = capi.Context(0)
    s = ctx.csr_synth(ROWS, DIM, DRAWS, seed=11)
    s.build_transposed(col_block=16384).relabel_by_frequency()
    yield ctx, s
    s.destroy()
    ctx.close()


def test_bookkeeping_identities(big):
    ctx, s = big
    assert s.nrows == ROWS and s.dim == DIM
    # nnz = stored features + one bias entry per row; shards generated separately add up
    a = ctx.csr_synth(1_000_000, DIM, DRAWS, seed=11, row_offset=0)
    b = ctx.csr_synth(2_000_000, DIM, DRAWS, seed=11, row_offset=1_000_000)
    assert a.nnz + b.nnz == s.nnz
    a.destroy()
    b.destroy()
    # generated values are all 1, so the set is a pattern matrix: no value arrays
    row_major, copy = s.bytes()
    assert row_major == 8 * (ROWS + 1) + 4 * s.nnz + 4 * ROWS
    assert 0 < copy < 6 * (s.nnz - ROWS) + (1 << 24)       # the hot-mask entries are not copied


def test_zero_weights_give_log2_and_half(big):
    ctx, s = big
    ll, auc, n = s.metrics([ctx.vec(DIM)])
    assert n == ROWS and "%.7f" % ll == "0.6931472" and auc == pytest.approx(0.5, abs=1e-12)


def test_gradient_linear_reproducible_and_consistent(big):
    ctx, s = big
    rng = np.random.default_rng(0)
    x = rng.normal(0, 0.05, DIM).astype(np.float32)
    d = rng.normal(0, 1.0, DIM).astype(np.float32)
    d /= np.linalg.norm(d)
    dx, g = ctx.vec(DIM, x), ctx.vec(DIM)
    f0 = s.logistic_fg(dx, g, rho=0.01)
    g0 = g.download()
    f0b = s.logistic_fg(dx, g, rho=0.01)
    assert f0 == f0b
    np.testing.assert_array_equal(g0, g.download())          # integer accumulation: bit-reproducible
    # directional derivative by central differences (f is a sum over 3M rows: step large enough
    # for float residuals, small enough for the curvature)
    h = 2e-2
    fp = s.logistic_fg(ctx.vec(DIM, x + np.float32(h) * d), g, rho=0.01)
    fm = s.logistic_fg(ctx.vec(DIM, x - np.float32(h) * d), g, rho=0.01)
    dd = float(np.dot(g0.astype(np.float64), d.astype(np.float64)))
    assert (fp - fm) / (2 * h) == pytest.approx(dd, rel=2e-3, abs=1e-3 * abs(f0) / ROWS)
    # the regulariser's gradient is exactly rho (x - z) + y: difference of two calls isolates it
    z = rng.normal(0, 0.05, DIM).astype(np.float32)
    y = rng.normal(0, 0.05, DIM).astype(np.float32)
    s.logistic_fg(dx, g, z=ctx.vec(DIM, z), ydual=ctx.vec(DIM, y), rho=0.5)
    g1 = g.download()
    s.logistic_fg(dx, g, rho=0.0)
    g2 = g.download()
    np.testing.assert_allclose(g1 - g2, y + np.float32(0.5) * (x - z), rtol=0, atol=2e-6 * np.abs(g1).max())


@pytest.mark.parametrize("linesearch", [0, 1, 2])
def test_lbfgs_decreases_objective(big, linesearch):
    """every line search mode lowers the objective, reports the objective of the point it returns,
    and (two solves in a row, as two ADMM iterations) restarts from its own last evaluation"""
    ctx, s = big
    x, g = ctx.vec(DIM), ctx.vec(DIM)
    f_start = s.logistic_fg(x, g, rho=0.01)
    assert f_start == pytest.approx(ROWS * np.log(2.0), rel=1e-6)
    res = s.lbfgs_solve(x, rho=0.01, max_iter=3, linesearch=linesearch)
    assert res["evals"] >= 2 and res["f"] < f_start * 0.99
    assert s.logistic_fg(x, g, rho=0.01) == pytest.approx(res["f"], rel=1e-7)
    # second solve from the same x with a consensus point and a dual that have moved
    z = ctx.vec(DIM, np.full(DIM, 1e-3, np.float32))
    f2_start = s.logistic_fg(x, g, z=z, rho=0.01)
    res2 = s.lbfgs_solve(x, z=z, rho=0.01, max_iter=3, linesearch=linesearch)
    assert res2["f"] < f2_start
    assert s.logistic_fg(x, g, z=z, rho=0.01) == pytest.approx(res2["f"], rel=1e-7)
    # its first evaluation cost no sweep: forward sweeps = one X d per iteration (+ full evaluations)
    if linesearch == 1:
        assert res2["forward_sweeps"] == res2["evals"] - 1
    else:
        assert res2["transposed_sweeps"] == res2["iters"]


def test_lbfgs_modes_agree(big):
    """inside one solve the three modes walk the same points (to float rounding)"""
    ctx, s = big
    out = []
    for mode in (1, 0, 2):
        x = ctx.vec(DIM)
        res = s.lbfgs_solve(x, rho=0.01, max_iter=4, linesearch=mode)
        out.append((res, x.download()))
    (r1, x1), (r0, x0), (r2, x2) = out
    assert r0["evals"] == r1["evals"] == r2["evals"] and r0["iters"] == r1["iters"] == r2["iters"]
    assert r0["f"] == pytest.approx(r1["f"], rel=1e-7) and r2["f"] == pytest.approx(r1["f"], rel=1e-6)
    scale = np.abs(x1).max()
    assert np.abs(x0 - x1).max() <= 1e-5 * scale and np.abs(x2 - x1).max() <= 1e-4 * scale


def test_consensus_fixed_point(big):
    """with identical workers, z = payload-sum / (N rho + lambda) and x = z leaves the dual unchanged"""
    ctx, _ = big
    rng = np.random.default_rng(3)
    x = rng.normal(size=DIM).astype(np.float32)
    dx, dy, dp, dz = ctx.vec(DIM, x), ctx.vec(DIM), ctx.vec(DIM), ctx.vec(DIM, x)
    ctx.payload(dx, dy, 1.0, dp)                             # rho = 1, y = 0 -> payload = x
    r2, s2 = ctx.consensus(dp, 1, 1.0, 0.0, 0.0, dx, dy, dz)  # N = 1, lambda = 0 -> z = x
    np.testing.assert_array_equal(dz.download(), x)
    assert r2 == 0.0 and s2 == 0.0 and not dy.download().any()


def test_ftrl_minibatch_of_one_panel_matches_sums(big):
    """One mini-batch = the whole set: n = sum of g^2 per column, z = sum g (w is 0 below l_1),
    checked through totals that do not need the matrix on the host."""
    from admm_amd import capi
    ctx, _ = big
    s = ctx.csr_synth(500_000, 1 << 20, 47, seed=5).build_transposed(panel_rows=500_000, col_block=8192)
    dim = 1 << 20
    w, z, n, off, reg = (ctx.vec(dim) for _ in range(5))
    s.ftrl_pass(w, z, n, off, reg, 0.1, 1.0, 1.0, 200.0, batch=500_000)
    hz, hn = z.download().astype(np.float64), n.download().astype(np.float64)
    _, _, _, label = s.download()
    # from zero state every prediction is 0.5: g = 0.5 - y, g^2 = 0.25
    assert hn[0] == pytest.approx(0.25 * s.nrows, rel=1e-6)              # bias column: every row
    assert hz[0] == pytest.approx(float((0.5 - label).sum()), rel=1e-6, abs=1e-3)
    assert hn.sum() == pytest.approx(0.25 * s.nnz, rel=1e-6)             # every stored entry once
    s.destroy()


def test_single_pass_ftrl_at_ten_million_rows():
    """The single-pass FTRL shape (10M rows x 2^20 features, ~40 entries per row): the row-by-row
    pass of 1024 workers in one launch, and the mini-batch pass of one worker.  Properties: the
    first step of every worker from zero state is known in closed form, squared-gradient sums are
    bounded by the entry counts, both modes learn (logloss below ln 2 on fresh rows)."""
    from admm_amd import capi
    ctx = capi.Context(0)
    dim, workers, per = 1 << 20, 1024, 9766
    shards = [ctx.csr_synth(per, dim, 47, seed=21, row_offset=k * per) for k in range(workers)]
    assert sum(s.nrows for s in shards) >= 10_000_000
    w, z, n = ([ctx.vec(dim) for _ in shards] for _ in range(3))
    off, reg = ctx.vec(dim), ctx.vec(dim)
    ctx.ftrl_pass_multi(shards, w, z, n, [off] * workers, [reg] * workers, 0.1, 1.0, 1.0, 1.0)
    test = ctx.csr_synth(200_000, dim, 47, seed=22)
    for k in (0, 511, 1023):
        hn = n[k].download().astype(np.float64)
        # every row adds g^2 <= 1 to the bias column and to each of its stored columns
        assert 0 < hn[0] <= shards[k].nrows and hn.sum() <= shards[k].nnz
        assert hn[0] == hn.max()
        ll, auc, _ = test.metrics([w[k]])
        assert ll < 0.6931 and auc > 0.6
    # a second pass over the same shard moves the state on and stays finite
    n_before = n[5].download()
    ctx.ftrl_pass_multi(shards[5:6], w[5:6], z[5:6], n[5:6], [off], [reg], 0.1, 1.0, 1.0, 1.0)
    n_after = n[5].download()
    assert np.isfinite(n_after).all() and (n_after >= n_before).all() and n_after[0] > n_before[0]
    for v in w + z + n:
        v.free()
    for s in shards:
        s.destroy()
    one = ctx.csr_synth(10_000_000, dim, 47, seed=21).build_transposed(panel_rows=65536, col_block=8192)
    wv, zv, nv = (ctx.vec(dim) for _ in range(3))
    one.ftrl_pass(wv, zv, nv, off, reg, 0.1, 1.0, 1.0, 1.0, batch=65536)
    ll, auc, _ = test.metrics([wv])
    assert ll < 0.6931 and auc > 0.7
    hn = nv.download().astype(np.float64)
    assert hn.sum() <= one.nnz and hn[0] <= one.nrows
    one.destroy()
    test.destroy()
    ctx.close()


def test_heavy_tailed_rows_over_sixteen_million_features():
    """Log-normal row lengths (a few rows of many hundreds of entries among short ones) hashed over
    2^24 features: the sweeps' answers do not depend on how the rows fall into tiles and panels."""
    from admm_amd import capi
    ctx = capi.Context(0)
    dim = 1 << 24
    kw = dict(seed=31, len_dist=1, tail_sigma=1.2, max_len=1024)
    a = ctx.csr_synth(400_000, dim, 40, **kw).build_transposed(col_block=16384).relabel_by_frequency()
    b = ctx.csr_synth(400_000, dim, 40, **kw).build_transposed(panel_rows=50_000, col_block=4096)
    rp = a.download()[0].astype(np.int64)
    lens = np.diff(rp)
    assert lens.max() > 8 * lens.mean()                                  # the tail is there
    rng = np.random.default_rng(2)
    x = ctx.vec(dim, (rng.normal(0, 0.05, dim) * (rng.random(dim) < 0.05)).astype(np.float32))
    ga, gb = ctx.vec(dim), ctx.vec(dim)
    fa = a.logistic_fg(x, ga, rho=0.0)
    fb = b.logistic_fg(x, gb, rho=0.0)
    assert fa == pytest.approx(fb, rel=1e-6)
    ha, hb = ga.download(), gb.download()
    np.testing.assert_allclose(ha, hb, rtol=2e-5, atol=2e-5 * np.abs(hb).max())
    # the bias coordinate of X^T r is the sum of the residuals, whose sum over the loss is checkable
    assert ha[0] == pytest.approx(hb[0], rel=1e-6)
    a.destroy()
    b.destroy()
    ctx.close()


def test_heavy_tailed_rows_over_sixteen_million_features_against_the_oracle():
    """The Criteo-like shape (2^24 hashed features, log-normal row lengths with rows of many hundred
    entries among short ones) at a size the CPU restatement still handles in seconds: objective and
    gradient, predictions, logloss and AUC, the row-by-row FTRL pass and the mini-batch one, each
    against the oracle on the downloaded sample set."""
    from admm_amd import capi
    from oracle import oracle as orc
    orc.build()
    ctx = capi.Context(0)
    dim = 1 << 24
    dev = ctx.csr_synth(400_000, dim, 40, seed=31, len_dist=1, tail_sigma=1.2, max_len=1024)
    dev.build_transposed(panel_rows=100_000, col_block=8192).relabel_by_frequency()
    tr = orc.Csr(*dev.download(), dim)
    lens = np.diff(tr.rowptr.astype(np.int64))
    assert tr.nnz == dev.nnz and lens.max() > 8 * lens.mean() and lens.min() >= 1
    rng = np.random.default_rng(2)
    x = (rng.normal(0, 0.3, dim) * (rng.random(dim) < 0.05)).astype(np.float32)
    x[:4096] = rng.normal(0, 0.1, 4096)
    b = (rng.normal(0, 0.05, dim) * (rng.random(dim) < 0.01)).astype(np.float32)
    z = (rng.normal(0, 0.1, dim) * (rng.random(dim) < 0.05)).astype(np.float32)
    dx, db, dz = ctx.vec(dim, x), ctx.vec(dim, b), ctx.vec(dim, z)
    # objective and gradient
    f_ref, g_ref = orc.logistic_fg(tr, dim, x, b, z, None, 0.05)
    g = ctx.vec(dim)
    f = dev.logistic_fg(dx, g, offset=db, z=dz, rho=0.05)
    assert f == pytest.approx(f_ref, rel=1e-5)
    np.testing.assert_allclose(g.download(), g_ref, rtol=2e-5, atol=2e-5 * np.abs(g_ref).max())
    # predictions, logloss, AUC
    s64, p_ref = orc.logloss_sum_f64(tr, [x, b], want_p=True)
    np.testing.assert_allclose(dev.predict([dx, db]), p_ref, rtol=1e-5, atol=1e-6)
    ll, auc, n = dev.metrics([dx, db])
    assert n == tr.nrows and ll == pytest.approx(s64 / tr.nrows, rel=1e-5)
    assert auc == pytest.approx(orc.auc_from_scores(p_ref, tr.label, f64=True), rel=1e-5)
    # FTRL: the reference's row-by-row pass (hyper-parameters of script/start.sh), then one mini-batch pass
    for batch, rtol in ((1, 1e-5), (100_000, 5e-5)):
        hw, hz, hn = (np.zeros(dim, np.float32) for _ in range(3))
        orc.ftrl_run(tr, hw, hz, hn, b, z, 0.01, 1.0, 0.4, 2.5, batch=batch)
        w, zz, nn = ctx.vec(dim), ctx.vec(dim), ctx.vec(dim)
        dev.ftrl_pass(w, zz, nn, db, dz, 0.01, 1.0, 0.4, 2.5, batch=batch)
        for got, ref in ((w, hw), (zz, hz), (nn, hn)):
            np.testing.assert_allclose(got.download(), ref, rtol=rtol, atol=1e-6 * max(1.0, np.abs(ref).max()))
        for v in (w, zz, nn):
            v.free()
    dev.destroy()
    ctx.close()


@pytest.mark.parametrize("name,rows,dim,draws,kw,nshards", [
    ("c3", 100_000_000, 1 << 22, 131.0, {}, 8),
    ("c5", 125_000_000, 1 << 24, 40.0, dict(len_dist=1, tail_sigma=1.2, max_len=1024), 5),
])
def test_one_rank_shard_equals_the_sum_of_its_pieces(name, rows, dim, draws, kw, nshards):
    """The two shapes the benchmark runs, at their full one-GPU size.  Objective and gradient of the
    whole shard (default layout: row masks, hot stream, column stream, cells) must equal the sums over
    disjoint row ranges built as separate sets (other panel boundaries, other frequency ranks, larger
    fixed-point scale), and the gradient must be the derivative of the objective along a direction."""
    from admm_amd import capi
    ctx = capi.Context(0)
    rng = np.random.default_rng(5)
    x = rng.normal(0, 0.05, dim).astype(np.float32)
    d = rng.normal(0, 1.0, dim).astype(np.float32)
    d /= np.linalg.norm(d)
    s = ctx.csr_synth(rows, dim, draws, seed=31, **kw)
    s.build_transposed(col_block=16384).relabel_by_frequency()
    nnz = s.nnz
    dx, g = ctx.vec(dim, x), ctx.vec(dim)
    f = s.logistic_fg(dx, g, rho=0.0)
    gw = g.download().astype(np.float64)
    h = 2e-2
    fp = s.logistic_fg(ctx.vec(dim, x + np.float32(h) * d), g, rho=0.0)
    fm = s.logistic_fg(ctx.vec(dim, x - np.float32(h) * d), g, rho=0.0)
    assert (fp - fm) / (2 * h) == pytest.approx(float(np.dot(gw, d.astype(np.float64))), rel=2e-3, abs=1e-3 * f / rows)
    # the all-zero point: every residual is +-1/2, the sums of the frequent columns are as large as they get
    f0 = s.logistic_fg(ctx.vec(dim), g, rho=0.0)
    g0 = g.download().astype(np.float64)
    assert f0 == pytest.approx(rows * np.log(2.0), rel=1e-6)
    s.destroy()
    fsum, gsum, g0sum, nsum = 0.0, np.zeros(dim), np.zeros(dim), 0
    per = rows // nshards
    for k in range(nshards):
        part = ctx.csr_synth(per, dim, draws, seed=31, row_offset=k * per, **kw)
        part.build_transposed(col_block=16384).relabel_by_frequency()
        nsum += part.nnz
        fsum += part.logistic_fg(dx, g, rho=0.0)
        gsum += g.download()
        part.logistic_fg(ctx.vec(dim), g, rho=0.0)
        g0sum += g.download()
        part.destroy()
    assert nsum == nnz
    assert f == pytest.approx(fsum, rel=1e-9)
    scale = np.abs(gsum).max()
    np.testing.assert_allclose(gw, gsum, rtol=1e-6, atol=1e-7 * scale)
    np.testing.assert_allclose(g0, g0sum, rtol=1e-6, atol=1e-7 * np.abs(g0sum).max())
    ctx.close()
