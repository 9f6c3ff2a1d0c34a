"""Parity of the HIP path (through the C ABI) with the CPU restatement on the same inputs.

Tolerances: index / nnz bookkeeping bit-exact; floating point 1e-5 relative (the figure the
reference hot path is specified to), with an absolute floor for values that are sums of
cancelling terms."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, make_synth_csr

pytestmark = pytest.mark.gpu

RTOL = 1e-5


@pytest.fixture(scope="module")
def ctx():
    from admm_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def upload(ctx, raw, dim):
    return ctx.csr_upload(raw.rowptr, raw.col, raw.val, raw.label, dim, add_bias=True)


def close(a, b, rtol=RTOL, atol=1e-7):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


# ------------------------------------------------------------------------------------------------
# sample set bookkeeping
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("nrows,mean", [(0, 5), (1, 0), (257, 7), (1500, 33)])
def test_bias_shift_bit_exact(ctx, oracle, nrows, mean):
    raw = make_synth_csr(nrows, 97, mean, seed=11, binary=False)
    tr = oracle.translate(raw)
    dev = upload(ctx, raw, 97)
    assert (dev.nrows, dev.nnz, dev.dim) == (tr.nrows, tr.nnz, 97)
    assert dev.nnz == raw.nnz + raw.nrows
    rowptr, col, val, label = dev.download()
    np.testing.assert_array_equal(rowptr, tr.rowptr)
    np.testing.assert_array_equal(col, tr.col)
    np.testing.assert_array_equal(val, tr.val)
    np.testing.assert_array_equal(label, tr.label)
    dev.destroy()


def test_upload_rejects_out_of_range_index(ctx):
    from admm_amd import capi
    with pytest.raises(capi.AdmmError):
        ctx.csr_upload(np.array([0, 1], np.uint64), np.array([5], np.uint32), None,
                       np.array([1], np.float32), dim=6, add_bias=True)  # 5+1 needs dim >= 7


def test_wtest_known_answers_on_gpu(ctx, oracle):
    with open(os.path.join(GOLDEN, "wtest_known_answers.json")) as f:
        known = json.load(f)
    raw = oracle.parse_libsvm(os.path.join(GOLDEN, "wtest.libsvm"))
    prm = known["ftrl_params"]
    dev = upload(ctx, raw, prm["dim"])
    w, z, n, off, reg = (ctx.vec(prm["dim"]) for _ in range(5))
    for expect in known["ftrl_passes"]:
        dev.ftrl_pass(w, z, n, off, reg, prm["alpha"], prm["beta"], prm["l_1"], prm["l_2"])
        hw, hz, hn = w.download(), z.download(), n.download()
        for j in known["touched"]:
            close(hw[j], expect["w"], atol=0)
            close(hz[j], expect["z"], atol=0)
            close(hn[j], expect["n"], atol=0)
    ll, auc, n_rows = dev.metrics([off, reg], want_auc=False)
    assert "%.7f" % ll == known["zero_weight_logloss"] and n_rows == 1
    dev.destroy()


# ------------------------------------------------------------------------------------------------
# metrics
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("binary,label01,mean", [(True, True, 9), (False, False, 40), (True, True, 130)])
def test_predict_logloss_auc(ctx, oracle, binary, label01, mean):
    dim = 513
    raw = make_synth_csr(3000, dim, mean, seed=21, binary=binary, label01=label01)
    tr = oracle.translate(raw)
    rng = np.random.default_rng(5)
    x = rng.normal(0, 0.4, dim).astype(np.float32)
    b = rng.normal(0, 0.05, dim).astype(np.float32)
    dev = upload(ctx, raw, dim)
    dx, db = ctx.vec(dim, x), ctx.vec(dim, b)
    p = dev.predict([dx, db])
    s64, p_ref = oracle.logloss_sum_f64(tr, [x, b], want_p=True)
    close(p, p_ref, atol=1e-7)
    ll, auc, n = dev.metrics([dx, db])
    assert n == 3000
    close(ll, s64 / 3000)
    close(ll, oracle.logloss(tr, [x, b]), rtol=2e-5)      # the float running sum of the reference
    close(auc, oracle.auc_from_scores(p_ref, tr.label, f64=True))
    close(auc, oracle.auc(tr, [x, b]), rtol=5e-5)         # float ranks / float accumulators
    dev.destroy()


def test_auc_with_ties_and_zero_weights(ctx, oracle):
    dim = 64
    raw = make_synth_csr(800, dim, 3, seed=8)       # short rows: many identical scores
    tr = oracle.translate(raw)
    dev = upload(ctx, raw, dim)
    zero = ctx.vec(dim)
    ll, auc, _ = dev.metrics([zero])
    assert "%.7f" % ll == "0.6931472"
    assert auc == pytest.approx(0.5, abs=1e-12)
    x = np.zeros(dim, np.float32)
    x[1:6] = [0.5, -0.5, 0.25, 0.0, 1.0]
    _, auc, _ = dev.metrics([ctx.vec(dim, x)])
    close(auc, oracle.auc(tr, [x]))
    dev.destroy()


def test_auc_on_arrays_and_all_ranks_metrics(ctx, oracle):
    """admm_auc_scores on bare (score, label) arrays with ties, and the all-ranks test metrics of
    the driver, which for one rank must be the sample set's own metrics"""
    from admm_amd import driver
    rng = np.random.default_rng(21)
    n = 5000
    scores = np.round(rng.random(n), 2).astype(np.float32)          # many ties
    labels = rng.integers(0, 2, n).astype(np.float32)
    got = ctx.auc_scores(ctx.vec(n, scores), ctx.vec(n, labels), n)
    close(got, oracle.auc_from_scores(scores, labels, f64=True))
    dim = 2000
    raw = make_synth_csr(3000, dim, 12, seed=17)
    dev = upload(ctx, raw, dim)
    w = ctx.vec(dim, rng.normal(0, 0.3, dim).astype(np.float32))
    b = ctx.vec(dim, rng.normal(0, 0.05, dim).astype(np.float32))
    ll, auc, rows = dev.metrics([w, b])
    g_ll, g_auc, g_rows = driver.global_test_metrics(ctx, dev, [w, b])
    assert g_rows == rows and g_auc == auc
    close(g_ll, ll, rtol=1e-12)
    dev.destroy()


# ------------------------------------------------------------------------------------------------
# FTRL
# ------------------------------------------------------------------------------------------------

def _ftrl_both(ctx, oracle, raw, dim, params, passes=1, batch=1, panel=0):
    tr = oracle.translate(raw)
    rng = np.random.default_rng(17)
    reg = rng.normal(0, 0.02, dim).astype(np.float32)
    off = rng.normal(0, 0.02, dim).astype(np.float32)
    hw, hz, hn = (np.zeros(dim, np.float32) for _ in range(3))
    dev = upload(ctx, raw, dim)
    if batch > 1:
        dev.build_transposed(panel_rows=panel or batch, col_block=128)
    w, z, n = ctx.vec(dim), ctx.vec(dim), ctx.vec(dim)
    doff, dreg = ctx.vec(dim, off), ctx.vec(dim, reg)
    for _ in range(passes):
        oracle.ftrl_run(tr, hw, hz, hn, off, reg, *params, batch=batch)
        dev.ftrl_pass(w, z, n, doff, dreg, *params, batch=batch)
    out = (w.download(), z.download(), n.download()), (hw, hz, hn)
    dev.destroy()
    return out


@pytest.mark.parametrize("binary,label01,mean", [(True, True, 12), (False, False, 30), (True, True, 90)])
def test_ftrl_sequential_pass(ctx, oracle, binary, label01, mean):
    dim = 301
    raw = make_synth_csr(2000, dim, mean, seed=31, binary=binary, label01=label01)
    # hyper-parameters of script/start.sh:15-31
    got, ref = _ftrl_both(ctx, oracle, raw, dim, (0.01, 1.0, 0.4, 2.5), passes=2)
    for g, r in zip(got, ref):
        close(g, r, atol=1e-6 * max(1.0, np.abs(r).max()))


def test_ftrl_sequential_admm_constants(ctx, oracle):
    dim = 200
    raw = make_synth_csr(1500, dim, 20, seed=33)
    got, ref = _ftrl_both(ctx, oracle, raw, dim, (0.1, 1.0, 1.0, 200.0))   # src/workers.cc:53-54
    for g, r in zip(got, ref):
        close(g, r, atol=1e-6)


def test_ftrl_previous_occurrence_table_in_chunks(ctx, oracle, monkeypatch):
    """the table behind the pipelined row-by-row pass is built in chunks of rows with a halo: tiny
    chunks give the same pass, bit for bit, as one chunk; and a column whose last use lies more than
    65534 rows back is read from memory again (no entry in the table)"""
    from oracle import oracle as orc
    dim = 5000
    raw = make_synth_csr(3000, dim, 25, seed=35)
    outs = []
    for chunk in (None, "97"):
        if chunk:
            monkeypatch.setenv("ADMM_DEP_CHUNK_ROWS", chunk)
        got, ref = _ftrl_both(ctx, oracle, raw, dim, (0.05, 1.0, 0.1, 1.0))
        if chunk:
            monkeypatch.delenv("ADMM_DEP_CHUNK_ROWS")
        for g, r in zip(got, ref):
            close(g, r, atol=1e-6 * max(1.0, np.abs(r).max()))
        outs.append(got)
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    # 70000 rows of one entry each; column 7 is stored by the first and the last row only
    n = 70000
    col = (8 + np.arange(n) % 50).astype(np.uint32)
    col[0] = col[-1] = 7
    far = orc.Csr(np.arange(n + 1, dtype=np.uint64), col, np.ones(n, np.float32),
                  (np.arange(n) % 3 == 0).astype(np.float32), 100)
    got, ref = _ftrl_both(ctx, oracle, far, 100, (0.05, 1.0, 0.1, 1.0))
    for g, r in zip(got, ref):
        close(g, r, atol=1e-6 * max(1.0, np.abs(r).max()))


def test_ftrl_rows_with_repeated_columns(ctx, oracle):
    from oracle import oracle as orc
    rowptr = np.array([0, 3, 5, 8], np.uint64)
    col = np.array([2, 2, 5, 1, 1, 0, 4, 0], np.uint32)
    val = np.array([1, 2, 1, 1, 1, .5, 1, 1], np.float32)
    raw = orc.Csr(rowptr, col, val, np.array([1, 0, 1], np.float32), 8)
    got, ref = _ftrl_both(ctx, oracle, raw, 8, (0.05, 1.0, 0.01, 0.5), passes=3)
    for g, r in zip(got, ref):
        close(g, r, atol=1e-6)


def test_ftrl_multi_pass_equals_single_passes(ctx):
    """many workers' row-by-row passes in one launch: each bit-identical to its own launch"""
    from oracle import oracle as orc
    dim = 40000
    raws = [make_synth_csr(900, dim, 30, seed=61),                       # rows up to 64 entries, unit values
            make_synth_csr(700, dim, 90, seed=62),                       # rows up to 128 entries
            make_synth_csr(800, dim, 25, seed=63, binary=False),         # stored values
            make_synth_csr(300, dim, 200, seed=64),                      # rows too long for the pipelined kernel
            orc.Csr(np.array([0, 3, 5], np.uint64), np.array([2, 2, 5, 1, 1], np.uint32),
                    np.ones(5, np.float32), np.array([1, 0], np.float32), dim),  # repeated columns
            make_synth_csr(0, dim, 5, seed=65)]                          # no rows
    sets = [ctx.csr_upload(r.rowptr, r.col, r.val, r.label, dim) for r in raws]
    sets[1].relabel_by_frequency()
    rng = np.random.default_rng(6)
    off = ctx.vec(dim, rng.normal(0, 0.01, dim).astype(np.float32))
    reg = ctx.vec(dim, rng.normal(0, 0.01, dim).astype(np.float32))
    prm = (0.05, 1.0, 0.01, 0.5)

    def fresh():
        return [[ctx.vec(dim) for _ in sets] for _ in range(3)]
    w1, z1, n1 = fresh()
    w2, z2, n2 = fresh()
    for _ in range(2):                                                   # state carried over, as between parts
        for k, s in enumerate(sets):
            s.ftrl_pass(w1[k], z1[k], n1[k], off, reg, *prm)
        ctx.ftrl_pass_multi(sets, w2, z2, n2, [off] * len(sets), [reg] * len(sets), *prm)
    for k in range(len(sets)):
        for a, b in ((w1, w2), (z1, z2), (n1, n2)):
            np.testing.assert_array_equal(a[k].download(), b[k].download())
    assert np.abs(z2[0].download()).max() > 0
    from admm_amd import capi
    with pytest.raises(capi.AdmmError):
        ctx.ftrl_pass_multi(sets[:2], [w2[0], w2[0]], z2[:2], n2[:2], [off] * 2, [reg] * 2, *prm)
    for s in sets:
        s.destroy()


def test_ftrl_pass_relabel_pass(ctx, oracle):
    """a row-by-row pass, then the frequency relabelling, then a second pass: the tables the first
    pass built describe the old numbering and must not be reused"""
    dim = 5000
    raw = make_synth_csr(1200, dim, 25, seed=71)
    tr = oracle.translate(raw)
    rng = np.random.default_rng(2)
    off = rng.normal(0, 0.02, dim).astype(np.float32)
    reg = rng.normal(0, 0.02, dim).astype(np.float32)
    prm = (0.05, 1.0, 0.1, 1.0)
    hw, hz, hn = (np.zeros(dim, np.float32) for _ in range(3))
    dev = upload(ctx, raw, dim)
    w, z, n = ctx.vec(dim), ctx.vec(dim), ctx.vec(dim)
    doff, dreg = ctx.vec(dim, off), ctx.vec(dim, reg)
    dev.ftrl_pass(w, z, n, doff, dreg, *prm)
    oracle.ftrl_run(tr, hw, hz, hn, off, reg, *prm)
    dev.relabel_by_frequency()
    dev.ftrl_pass(w, z, n, doff, dreg, *prm)
    oracle.ftrl_run(tr, hw, hz, hn, off, reg, *prm)
    for got, ref in ((w, hw), (z, hz), (n, hn)):
        close(got.download(), ref, atol=1e-6 * max(1.0, np.abs(ref).max()))
    dev.destroy()


def test_ftrl_minibatch_wide_column_block(ctx, oracle):
    """col_block 4096 in the mini-batch form: 66 KB of LDS in the 512-thread transposed sweep"""
    dim = 9000
    raw = make_synth_csr(1500, dim, 25, seed=36)
    tr = oracle.translate(raw)
    prm = (0.05, 1.0, 0.2, 1.5)
    zero = np.zeros(dim, np.float32)
    hw, hz, hn = (np.zeros(dim, np.float32) for _ in range(3))
    oracle.ftrl_run(tr, hw, hz, hn, zero, zero, *prm, batch=500)
    dev = upload(ctx, raw, dim).build_transposed(panel_rows=500, col_block=4096)
    w, z, n = ctx.vec(dim), ctx.vec(dim), ctx.vec(dim)
    dev.ftrl_pass(w, z, n, ctx.vec(dim), ctx.vec(dim), *prm, batch=500)
    for got, ref in ((w, hw), (z, hz), (n, hn)):
        close(got.download(), ref, rtol=5e-5, atol=2e-6)
    dev.destroy()


@pytest.mark.parametrize("batch", [64, 500])
def test_ftrl_minibatch(ctx, oracle, batch):
    dim = 301
    raw = make_synth_csr(1700, dim, 25, seed=35)
    got, ref = _ftrl_both(ctx, oracle, raw, dim, (0.05, 1.0, 0.2, 1.5), passes=2, batch=batch)
    for g, r in zip(got, ref):
        close(g, r, rtol=5e-5, atol=2e-6)    # float atomics order inside a batch


@pytest.mark.parametrize("binary", [True, False])
@pytest.mark.parametrize("relabel", [False, True])
def test_ftrl_minibatch_on_panel_boundaries(ctx, oracle, relabel, binary):
    """batches of 512 and 1024 rows: a multiple of the 256-row panels a mini-batch copy is cut into, so
    the forward sweep of a batch runs over the row-panel copy (hot stream, lists, mask columns with
    both sums); 40000 columns, so that entries fall outside the hot stream too"""
    dim = 40000
    raw = make_synth_csr(2900, dim, 40, seed=37, binary=binary)
    tr = oracle.translate(raw)
    rng = np.random.default_rng(17)
    reg = rng.normal(0, 0.02, dim).astype(np.float32)
    off = rng.normal(0, 0.02, dim).astype(np.float32)
    for batch in (512, 1024):
        hw, hz, hn = (np.zeros(dim, np.float32) for _ in range(3))
        dev = upload(ctx, raw, dim)
        if relabel:
            dev.relabel_by_frequency()
        dev.build_transposed(panel_rows=batch, col_block=2048)
        w, z, n = ctx.vec(dim), ctx.vec(dim), ctx.vec(dim)
        doff, dreg = ctx.vec(dim, off), ctx.vec(dim, reg)
        for _ in range(2):
            oracle.ftrl_run(tr, hw, hz, hn, off, reg, 0.05, 1.0, 0.2, 1.5, batch=batch)
            dev.ftrl_pass(w, z, n, doff, dreg, 0.05, 1.0, 0.2, 1.5, batch=batch)
        for got, ref in ((w, hw), (z, hz), (n, hn)):
            close(got.download(), ref, rtol=5e-5, atol=2e-6 * max(1.0, np.abs(ref).max()))
        dev.destroy()


# ------------------------------------------------------------------------------------------------
# batch objective, L-BFGS
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("binary,mean,panel", [(True, 15, 0), (False, 60, 300)])
def test_logistic_fg(ctx, oracle, binary, mean, panel):
    dim = 700
    raw = make_synth_csr(2500, dim, mean, seed=41, binary=binary)
    tr = oracle.translate(raw)
    rng = np.random.default_rng(3)
    x = rng.normal(0, 0.3, dim).astype(np.float32)
    z = rng.normal(0, 0.3, dim).astype(np.float32)
    yd = rng.normal(0, 0.05, dim).astype(np.float32)
    f_ref, g_ref = oracle.logistic_fg(tr, dim, x, None, z, yd, 0.7)
    dev = upload(ctx, raw, dim).build_transposed(panel_rows=panel, col_block=256)
    g = ctx.vec(dim)
    f = dev.logistic_fg(ctx.vec(dim, x), g, z=ctx.vec(dim, z), ydual=ctx.vec(dim, yd), rho=0.7)
    close(f, f_ref)
    close(g.download(), g_ref, rtol=2e-5, atol=2e-5 * np.abs(g_ref).max())
    # a second call must start from clean accumulators
    f2 = dev.logistic_fg(ctx.vec(dim, x), g, z=ctx.vec(dim, z), ydual=ctx.vec(dim, yd), rho=0.7)
    close(f2, f_ref)
    close(g.download(), g_ref, rtol=2e-5, atol=2e-5 * np.abs(g_ref).max())
    dev.destroy()


def test_lbfgs_short_memory_follows_the_oracle(ctx, oracle):
    """two correction pairs only: the ring of pairs wraps every other iteration, and the newest pair is
    written before it is known to be usable (it must not land on a pair still in use)"""
    dim = 90
    raw = make_synth_csr(1200, dim, 9, seed=47)
    tr = oracle.translate(raw)
    z = np.zeros(dim, np.float32)

    def fg(xx):
        return oracle.logistic_fg(tr, dim, xx, None, z, None, 0.5)

    dev = upload(ctx, raw, dim).build_transposed(col_block=64)
    for m, iters in ((1, 6), (2, 7), (3, 9)):
        x_ref, f_ref, ev_ref, it_ref = oracle.lbfgs_solve(np.zeros(dim, np.float32), fg, m=m, max_iter=iters)
        x = ctx.vec(dim)
        res = dev.lbfgs_solve(x, z=ctx.vec(dim), rho=0.5, m=m, max_iter=iters, linesearch=1)
        assert (res["evals"], res["iters"]) == (ev_ref, it_ref)
        close(res["f"], f_ref)
        close(x.download(), x_ref, rtol=1e-4, atol=1e-6)
    dev.destroy()


def test_timing_can_be_restricted_to_kernel_classes(ctx):
    from admm_amd import capi
    dim = 300
    raw = make_synth_csr(3000, dim, 12, seed=5)
    dev = upload(ctx, raw, dim).build_transposed(col_block=64)
    x, g = ctx.vec(dim), ctx.vec(dim)
    ctx.timing_enable(True, classes=[capi.K_FORWARD, capi.K_TRANSPOSED])
    ctx.timing_reset()
    f1 = dev.logistic_fg(x, g, rho=0.1)
    assert ctx.timing_get(capi.K_FORWARD)[1] == 1 and ctx.timing_get(capi.K_TRANSPOSED)[1] == 1
    assert ctx.timing_get(capi.K_ELEMENTWISE)[1] == 0
    ctx.timing_enable(True)
    ctx.timing_reset()
    f2 = dev.logistic_fg(x, g, rho=0.1)
    assert ctx.timing_get(capi.K_ELEMENTWISE)[1] >= 1
    ctx.timing_enable(False)
    assert f1 == f2
    dev.destroy()


@pytest.mark.parametrize("linesearch", [0, 1, 2])
def test_lbfgs_solve(ctx, oracle, linesearch):
    """linesearch 0 evaluates trial points from the margins, 1 by full sweeps: same trajectory"""
    dim = 120
    raw = make_synth_csr(1500, dim, 10, seed=43)
    tr = oracle.translate(raw)
    z = np.zeros(dim, np.float32)

    def fg(xx):
        return oracle.logistic_fg(tr, dim, xx, None, z, None, 1.0)

    x_ref, f_ref, _, _ = oracle.lbfgs_solve(np.zeros(dim, np.float32), fg, max_iter=80, epsilon=1e-6)
    dev = upload(ctx, raw, dim).build_transposed(col_block=64)
    x = ctx.vec(dim)
    res = dev.lbfgs_solve(x, z=ctx.vec(dim), rho=1.0, max_iter=80, epsilon=1e-6, linesearch=linesearch)
    close(res["f"], f_ref, rtol=1e-6)
    close(x.download(), x_ref, rtol=1e-3, atol=2e-4)   # both converged to the same minimiser
    # and one short run follows the same trajectory step for step
    x_ref, f_ref, ev_ref, it_ref = oracle.lbfgs_solve(np.zeros(dim, np.float32), fg, max_iter=4)
    x.zero()
    res = dev.lbfgs_solve(x, z=ctx.vec(dim), rho=1.0, max_iter=4, linesearch=linesearch)
    assert (res["evals"], res["iters"]) == (ev_ref, it_ref)
    if linesearch == 1:
        assert res["forward_sweeps"] == res["transposed_sweeps"] == ev_ref
    elif linesearch == 0:   # one X d sweep per iteration, one full evaluation per accepted point
        assert res["transposed_sweeps"] == 1 + it_ref
        assert res["forward_sweeps"] == 1 + 2 * it_ref
    else:   # an accepted point takes a forward sweep only now and then
        assert res["transposed_sweeps"] == 1 + it_ref
        assert 1 + it_ref < res["forward_sweeps"] < 1 + 2 * it_ref
    close(res["f"], f_ref)
    close(x.download(), x_ref, rtol=1e-4, atol=1e-6)
    dev.destroy()


@pytest.mark.parametrize("linesearch", [0, 1])
def test_lbfgs_warm_start_reuses_the_last_evaluation(ctx, linesearch, monkeypatch):
    """a solve that starts where the previous one on the same set ended (same x and offset; z and the
    dual moved, as between ADMM iterations) takes its first loss and loss gradient from that solve:
    one sweep pair fewer, bit-identical results"""
    dim = 5000
    raw = make_synth_csr(4000, dim, 20, seed=7)
    rng = np.random.default_rng(3)
    off = rng.normal(0, 0.05, dim).astype(np.float32)
    z1 = rng.normal(0, 0.1, dim).astype(np.float32)
    z2 = rng.normal(0, 0.1, dim).astype(np.float32)
    y2 = rng.normal(0, 0.01, dim).astype(np.float32)
    out = {}
    for cache in ("1", "0"):
        monkeypatch.setenv("ADMM_FG_CACHE", cache)
        dev = upload(ctx, raw, dim).build_transposed(col_block=1024).relabel_by_frequency()
        x = ctx.vec(dim)
        first = dev.lbfgs_solve(x, offset=ctx.vec(dim, off), z=ctx.vec(dim, z1), rho=0.5, max_iter=3, linesearch=linesearch)
        second = dev.lbfgs_solve(x, offset=ctx.vec(dim, off), z=ctx.vec(dim, z2), ydual=ctx.vec(dim, y2), rho=0.5,
                                 max_iter=3, linesearch=linesearch)
        # a different offset must not be served from the cache
        third = dev.lbfgs_solve(x, offset=ctx.vec(dim, 2 * off), z=ctx.vec(dim, z2), rho=0.5, max_iter=2,
                                linesearch=linesearch)
        out[cache] = (first, second, third, x.download())
        dev.destroy()
    (f1, s1, t1, x1), (f0, s0, t0, x0) = out["1"], out["0"]
    np.testing.assert_array_equal(x1, x0)
    for a_, b_ in ((f1, f0), (s1, s0), (t1, t0)):
        assert (a_["f"], a_["evals"], a_["iters"]) == (b_["f"], b_["evals"], b_["iters"])
    assert f1["forward_sweeps"] == f0["forward_sweeps"] and t1["forward_sweeps"] == t0["forward_sweeps"]
    assert s1["forward_sweeps"] == s0["forward_sweeps"] - 1
    assert s1["transposed_sweeps"] == s0["transposed_sweeps"] - 1


@pytest.mark.parametrize("linesearch", [0, 1, 2])
@pytest.mark.parametrize("nrows", [0, 1, 3])
def test_lbfgs_on_tiny_sets(ctx, oracle, nrows, linesearch):
    """no rows (only the proximal term is left), one row, and a handful: the solve runs and agrees
    with the oracle's solver"""
    dim = 300
    raw = make_synth_csr(nrows, dim, 6, seed=5)
    tr = oracle.translate(raw)
    z = np.full(dim, 0.25, np.float32)

    def fg(xx):
        return oracle.logistic_fg(tr, dim, xx, None, z, None, 2.0)

    x_ref, f_ref, ev_ref, it_ref = oracle.lbfgs_solve(np.zeros(dim, np.float32), fg, max_iter=5)
    dev = upload(ctx, raw, dim).build_transposed(col_block=128)
    if nrows:
        dev.relabel_by_frequency()
    x = ctx.vec(dim)
    res = dev.lbfgs_solve(x, z=ctx.vec(dim, z), rho=2.0, max_iter=5, linesearch=linesearch)
    assert (res["evals"], res["iters"]) == (ev_ref, it_ref)
    close(res["f"], f_ref, atol=1e-6)
    close(x.download(), x_ref, rtol=1e-4, atol=1e-6)
    dev.destroy()


# ------------------------------------------------------------------------------------------------
# consensus algebra
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("l1", [0.0, 0.3])
def test_consensus_kernels(ctx, oracle, l1):
    dim = 100003
    rng = np.random.default_rng(9)
    x = rng.normal(size=dim).astype(np.float32)
    y = rng.normal(size=dim).astype(np.float32)
    zold = rng.normal(size=dim).astype(np.float32)
    rho, gv = np.float32(0.01), np.float32(4.0)
    dx, dy, dz, dp = ctx.vec(dim, x), ctx.vec(dim, y), ctx.vec(dim, zold), ctx.vec(dim)
    ctx.payload(dx, dy, rho, dp)
    p_ref = oracle.payload(x, y, rho)
    np.testing.assert_array_equal(dp.download(), p_ref)
    total = (p_ref * 3).astype(np.float32)
    dt = ctx.vec(dim, total)
    z_ref = oracle.global_update(total, 3, rho, gv, np.float32(l1))
    # separate kernels
    dz2, dy2 = ctx.vec(dim), ctx.vec(dim, y)
    ctx.global_update(dt, 3, rho, gv, l1, dz2)
    ctx.dual_update(dy2, dx, dz2, rho)
    # fused kernel
    r2, s2 = ctx.consensus(dt, 3, rho, gv, l1, dx, dy, dz)
    y_ref = y.copy()
    oracle.dual_update(y_ref, x, z_ref, rho)
    r2_ref, s2_ref = oracle.residuals(x, z_ref, zold, rho)
    for zz, yy in ((dz2, dy2), (dz, dy)):
        close(zz.download(), z_ref, rtol=2e-7, atol=0)
        close(yy.download(), y_ref, rtol=1e-6, atol=1e-7)
    close(r2, r2_ref, rtol=1e-9)
    close(s2, s2_ref, rtol=1e-9)


# ------------------------------------------------------------------------------------------------
# synthetic generator
# ------------------------------------------------------------------------------------------------

def test_synth_generator_properties(ctx):
    dim = 1 << 14
    a = ctx.csr_synth(5000, dim, 40, seed=7)
    rowptr, col, val, label = a.download()
    assert rowptr[0] == 0 and rowptr[-1] == a.nnz == col.shape[0]
    lens = np.diff(rowptr.astype(np.int64))
    assert lens.min() >= 1 and 30 < lens.mean() < 45
    assert (val == 1.0).all() and set(np.unique(label)) <= {0.0, 1.0}
    assert 0.05 < label.mean() < 0.95
    for r in range(0, 5000, 97):
        row = col[rowptr[r]:rowptr[r + 1]].astype(np.int64)
        assert row[0] == 0 and (np.diff(row) > 0).all() and row[-1] < dim
    # power law: the most frequent feature is far more frequent than the median one
    counts = np.bincount(col[col > 0], minlength=dim)
    assert counts.max() > 50 * max(1, np.median(counts[counts > 0]))
    # determinism and shard consistency: rows [2000, 3000) generated alone are identical
    b = ctx.csr_synth(1000, dim, 40, seed=7, row_offset=2000)
    rp2, col2, _, lab2 = b.download()
    np.testing.assert_array_equal(col2, col[rowptr[2000]:rowptr[3000]])
    np.testing.assert_array_equal(lab2, label[2000:3000])
    # the values (all 1) are stored only on request; both forms are the same sample set and give
    # the same objective and gradient
    c = ctx.csr_synth(5000, dim, 40, seed=7, explicit_values=True)
    rp3, col3, val3, lab3 = c.download()
    np.testing.assert_array_equal(col3, col)
    np.testing.assert_array_equal(lab3, label)
    assert (val3 == 1.0).all()
    assert c.bytes()[0] == a.bytes()[0] + 4 * a.nnz
    x = ctx.vec(dim, np.random.default_rng(1).normal(0, 0.1, dim).astype(np.float32))
    ga, gc = ctx.vec(dim), ctx.vec(dim)
    a.build_transposed().relabel_by_frequency()
    c.build_transposed().relabel_by_frequency()
    fa = a.logistic_fg(x, ga, rho=0.0)
    fc = c.logistic_fg(x, gc, rho=0.0)
    # (the two forms reduce a row with a different number of lanes: equal up to float rounding)
    assert fa == pytest.approx(fc, rel=1e-6)
    np.testing.assert_allclose(ga.download(), gc.download(), rtol=1e-5, atol=1e-5 * np.abs(gc.download()).max())
    a.destroy()
    b.destroy()
    c.destroy()


def test_sweeps_on_generated_set(ctx, oracle):
    """the generated set (long rows, lanes-per-row 32) through both sweeps against the oracle"""
    from oracle import oracle as orc
    dim = 1 << 15
    dev = ctx.csr_synth(20000, dim, 100, seed=3).build_transposed(panel_rows=6000)
    tr = orc.Csr(*dev.download(), dim)
    rng = np.random.default_rng(4)
    x = rng.normal(0, 0.1, dim).astype(np.float32)
    z = rng.normal(0, 0.1, dim).astype(np.float32)
    f_ref, g_ref = oracle.logistic_fg(tr, dim, x, None, z, None, 0.01)
    g = ctx.vec(dim)
    f = dev.logistic_fg(ctx.vec(dim, x), g, z=ctx.vec(dim, z), rho=0.01)
    close(f, f_ref)
    close(g.download(), g_ref, rtol=2e-5, atol=2e-5 * np.abs(g_ref).max())
    dev.destroy()


@pytest.mark.parametrize("explicit_values", [False, True])
@pytest.mark.parametrize("relabel", [False, True])
def test_sweeps_many_tiles_per_workgroup(ctx, oracle, relabel, explicit_values):
    """enough rows that every workgroup of the forward sweep walks a dozen tiles (its software
    pipeline runs in steady state), in all three forms of the sweep, against the oracle"""
    from oracle import oracle as orc
    dim = 1 << 18
    nrows = 400_000
    dev = ctx.csr_synth(nrows, dim, 60, seed=12, explicit_values=explicit_values)
    dev.build_transposed(panel_rows=100_000, col_block=8192)
    if relabel:
        dev.relabel_by_frequency()
    tr = orc.Csr(*dev.download(), dim)
    rng = np.random.default_rng(8)
    x = rng.normal(0, 0.1, dim).astype(np.float32)
    b = rng.normal(0, 0.02, dim).astype(np.float32)
    z = rng.normal(0, 0.1, dim).astype(np.float32)
    # objective / gradient
    f_ref, g_ref = oracle.logistic_fg(tr, dim, x, b, z, None, 0.05)
    g = ctx.vec(dim)
    f = dev.logistic_fg(ctx.vec(dim, x), g, offset=ctx.vec(dim, b), z=ctx.vec(dim, z), rho=0.05)
    close(f, f_ref)
    close(g.download(), g_ref, rtol=2e-5, atol=2e-5 * np.abs(g_ref).max())
    # predictions and metrics
    s64, p_ref = oracle.logloss_sum_f64(tr, [x, b], want_p=True)
    close(dev.predict([ctx.vec(dim, x), ctx.vec(dim, b)]), p_ref, atol=1e-6)
    ll, auc, _ = dev.metrics([ctx.vec(dim, x), ctx.vec(dim, b)])
    close(ll, s64 / tr.nrows)
    # mini-batch FTRL
    hw, hz, hn = (np.zeros(dim, np.float32) for _ in range(3))
    oracle.ftrl_run(tr, hw, hz, hn, b, z, 0.05, 1.0, 0.1, 1.0, batch=100_000)
    w, zz, n = ctx.vec(dim), ctx.vec(dim), ctx.vec(dim)
    dev.ftrl_pass(w, zz, n, ctx.vec(dim, b), ctx.vec(dim, z), 0.05, 1.0, 0.1, 1.0, batch=100_000)
    for got, ref in ((w, hw), (zz, hz), (n, hn)):
        close(got.download(), ref, rtol=5e-5, atol=1e-6 * max(1.0, np.abs(ref).max()))
    # L-BFGS: the line search from the margins walks the same points as the one by full sweeps
    runs = []
    for mode in (1, 0, 2):
        xs = ctx.vec(dim, x)
        res = dev.lbfgs_solve(xs, offset=ctx.vec(dim, b), z=ctx.vec(dim, z), ydual=ctx.vec(dim, b), rho=0.05,
                              max_iter=3, linesearch=mode)
        runs.append((res, xs.download()))
    (r1, x1), (r0, x0), (r2, x2) = runs
    # mode 2: two accepted points from the margins, the last one by a full evaluation
    assert (r2["evals"], r2["iters"]) == (r1["evals"], r1["iters"])
    assert r2["transposed_sweeps"] == 1 + r2["iters"] and r2["forward_sweeps"] == 1 + r2["iters"] + 1
    close(r2["f"], r1["f"], rtol=1e-6)
    close(x2, x1, rtol=1e-4, atol=1e-6)
    assert (r0["evals"], r0["iters"]) == (r1["evals"], r1["iters"])
    assert r1["forward_sweeps"] == r1["transposed_sweeps"] == r1["evals"]
    assert r0["transposed_sweeps"] == 1 + r0["iters"] and r0["forward_sweeps"] == 1 + 2 * r0["iters"]
    close(r0["f"], r1["f"], rtol=1e-6)
    close(x0, x1, rtol=1e-4, atol=1e-6)
    f_ref, _ = oracle.logistic_fg(tr, dim, x0, b, z, b, 0.05)
    close(r0["f"], f_ref)
    dev.destroy()


# ------------------------------------------------------------------------------------------------
# frequency relabelling is invisible at the boundary
# ------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("relabel_first", [False, True])
def test_relabel_by_frequency_is_invisible(ctx, oracle, relabel_first):
    dim = 40000                      # larger than the LDS-cached head of the gathered vector
    raw = make_synth_csr(3000, dim, 50, seed=91, binary=False)
    tr = oracle.translate(raw)
    dev = upload(ctx, raw, dim)
    if relabel_first:
        dev.relabel_by_frequency().build_transposed(panel_rows=1000, col_block=1024)
    else:
        dev.build_transposed(panel_rows=1000, col_block=1024).relabel_by_frequency()
    rowptr, col, val, label = dev.download()
    np.testing.assert_array_equal(rowptr, tr.rowptr)
    np.testing.assert_array_equal(col, tr.col)
    np.testing.assert_array_equal(val, tr.val)
    rng = np.random.default_rng(6)
    x = rng.normal(0, 0.3, dim).astype(np.float32)
    b = rng.normal(0, 0.05, dim).astype(np.float32)
    z = rng.normal(0, 0.3, dim).astype(np.float32)
    # metrics
    s64, p_ref = oracle.logloss_sum_f64(tr, [x, b], want_p=True)
    ll, auc, _ = dev.metrics([ctx.vec(dim, x), ctx.vec(dim, b)])
    close(ll, s64 / tr.nrows)
    close(auc, oracle.auc_from_scores(p_ref, tr.label, f64=True))
    # objective / gradient
    f_ref, g_ref = oracle.logistic_fg(tr, dim, x, b, z, None, 0.3)
    g = ctx.vec(dim)
    f = dev.logistic_fg(ctx.vec(dim, x), g, offset=ctx.vec(dim, b), z=ctx.vec(dim, z), rho=0.3)
    close(f, f_ref)
    close(g.download(), g_ref, rtol=2e-5, atol=2e-5 * np.abs(g_ref).max())
    # sequential and mini-batch FTRL
    for batch in (1, 1000):
        hw, hz, hn = (np.zeros(dim, np.float32) for _ in range(3))
        oracle.ftrl_run(tr, hw, hz, hn, b, z, 0.05, 1.0, 0.1, 1.0, batch=batch)
        w, zz, n = ctx.vec(dim), ctx.vec(dim), ctx.vec(dim)
        dev.ftrl_pass(w, zz, n, ctx.vec(dim, b), ctx.vec(dim, z), 0.05, 1.0, 0.1, 1.0, batch=batch)
        for got, ref in ((w, hw), (zz, hz), (n, hn)):
            close(got.download(), ref, rtol=5e-5, atol=1e-6 * max(1.0, np.abs(ref).max()))
    dev.destroy()


def test_gradient_is_reproducible(ctx):
    """the transposed sweep adds in 64-bit fixed point: two evaluations agree bit for bit"""
    dim = 1 << 15
    dev = ctx.csr_synth(30000, dim, 60, seed=5).build_transposed(panel_rows=8192, col_block=2048)
    x = ctx.vec(dim, np.random.default_rng(1).normal(0, 0.1, dim).astype(np.float32))
    g = ctx.vec(dim)
    f1 = dev.logistic_fg(x, g, rho=0.1)
    g1 = g.download()
    f2 = dev.logistic_fg(x, g, rho=0.1)
    assert f1 == f2
    np.testing.assert_array_equal(g1, g.download())
    dev.destroy()


@pytest.mark.parametrize("relabel", [False, True])
def test_rows_longer_than_a_staging_pass(ctx, oracle, relabel):
    """tiles whose rows hold more entries than one LDS staging pass (and one row longer than a pass)"""
    from oracle import oracle as orc
    dim = 60000
    rng = np.random.default_rng(77)
    lens = rng.integers(300, 1500, 400)
    lens[17] = 20000
    lens[255] = 0
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    col = np.concatenate([np.sort(rng.choice(dim - 1, l, replace=False)) for l in lens]).astype(np.uint32)
    val = np.where(rng.random(col.shape[0]) < 0.7, 1.0, rng.uniform(0.5, 1.5, col.shape[0])).astype(np.float32)
    label = (rng.random(400) < 0.5).astype(np.float32)
    raw = orc.Csr(rowptr, col, val, label, dim)
    tr = oracle.translate(raw)
    dev = upload(ctx, raw, dim).build_transposed(panel_rows=128, col_block=4096)
    if relabel:
        dev.relabel_by_frequency()
    x = rng.normal(0, 0.02, dim).astype(np.float32)
    f_ref, g_ref = oracle.logistic_fg(tr, dim, x, None, None, None, 0.2)
    g = ctx.vec(dim)
    f = dev.logistic_fg(ctx.vec(dim, x), g, rho=0.2)
    close(f, f_ref)
    close(g.download(), g_ref, rtol=2e-5, atol=2e-5 * np.abs(g_ref).max())
    s64, p_ref = oracle.logloss_sum_f64(tr, [x], want_p=True)
    close(dev.predict([ctx.vec(dim, x)]), p_ref, atol=1e-6)
    dev.destroy()


@pytest.mark.parametrize("binary", [True, False])
def test_predict_and_update_row_by_row_equal_the_pass(ctx, oracle, binary):
    """FtrlSolver::Run is Predict then Update for every row and one more Predict on the last row
    (src/ftrl.cc:101-145): driving the two entry points row by row from the host gives the state of
    admm_ftrl_pass(batch = 1) (up to the order in which a row's products are added: element by element
    here, as the reference does, across the wavefront in the pass); at zero state a prediction is 0.5
    (test/ftrl_test.cc:37)"""
    dim = 300
    raw = make_synth_csr(400, dim, 9, seed=91, binary=binary)
    tr = oracle.translate(raw)
    rng = np.random.default_rng(2)
    off = rng.normal(0, 0.05, dim).astype(np.float32)
    reg = rng.normal(0, 0.05, dim).astype(np.float32)
    prm = (0.1, 1.0, 0.05, 0.5)
    w, z, n = ctx.vec(dim), ctx.vec(dim), ctx.vec(dim)
    zero = ctx.vec(dim)
    doff, dreg = ctx.vec(dim, off), ctx.vec(dim, reg)
    rows = [(tr.col[int(tr.rowptr[r]):int(tr.rowptr[r + 1])], tr.val[int(tr.rowptr[r]):int(tr.rowptr[r + 1])])
            for r in range(tr.nrows)]
    assert ctx.ftrl_predict(rows[0][0], rows[0][1], w, z, n, zero, zero, *prm) == 0.5
    for r, (idx, val) in enumerate(rows):
        p = ctx.ftrl_predict(idx, None if binary else val, w, z, n, doff, dreg, *prm)
        ctx.ftrl_update(idx, tr.label[r], p, w, z, n, prm[0])
    ctx.ftrl_predict(rows[-1][0], None if binary else rows[-1][1], w, z, n, doff, dreg, *prm)
    dev = upload(ctx, raw, dim)
    w2, z2, n2 = ctx.vec(dim), ctx.vec(dim), ctx.vec(dim)
    dev.ftrl_pass(w2, z2, n2, doff, dreg, *prm)
    for a, b in ((w, w2), (z, z2), (n, n2)):
        close(a.download(), b.download(), rtol=1e-5, atol=1e-7)
    hw, hz, hn = (np.zeros(dim, np.float32) for _ in range(3))
    oracle.ftrl_run(tr, hw, hz, hn, off, reg, *prm)
    for got, ref in ((w, hw), (z, hz), (n, hn)):
        close(got.download(), ref, rtol=1e-5, atol=1e-7)
    dev.destroy()


def test_tied_ranks_are_the_reference_sort(ctx, oracle):
    """Metrics::Sort (src/metrics.cc:14-46): ties share the mean of their 1-based ranks; negative, zero,
    repeated and single scores"""
    rng = np.random.default_rng(8)
    for n in (1, 2, 7, 1000, 70001):
        x = np.round(rng.normal(0, 1, n), 1).astype(np.float32)      # many ties
        if n > 2:
            x[:3] = [0.0, -0.0, 0.0]
        ranks = ctx.vec(n)
        ctx.tied_ranks(ctx.vec(n, x), ranks)
        np.testing.assert_array_equal(ranks.download(), oracle.tied_rank(x))
    np.testing.assert_array_equal(oracle.tied_rank(np.array([3, 1, 3, 2], np.float32)), [3.5, 1.0, 3.5, 2.0])


def test_tied_ranks_beyond_two_to_the_24(ctx, oracle):
    """more scores than a float counts exactly: the reference rounds the integer (first + last position) of a tie
    group to float before it adds 1 (or, for the last group, 2 to the sum with the last INDEX: src/metrics.cc:33,39),
    and those two roundings differ up there -- a big tied group at the end shows it"""
    n = (1 << 24) + 70_003
    rng = np.random.default_rng(12)
    x = np.round(rng.normal(0, 1, n), 2).astype(np.float32)
    x[rng.choice(n, 40_001, replace=False)] = 9.0            # the last tie group, an odd number of members
    ranks = ctx.vec(n)
    ctx.tied_ranks(ctx.vec(n, x), ranks)
    np.testing.assert_array_equal(ranks.download(), oracle.tied_rank(x))


def test_upload_checks_a_long_unsorted_row_quickly(ctx):
    """a row of several hundred thousand entries out of order: the device leaves its search for a repeated column
    to the host (a sort) instead of comparing pair by pair; both answers come back in moments"""
    import time
    dim, n = 1 << 20, 400_000
    rng = np.random.default_rng(4)
    ids = rng.choice(dim - 1, n, replace=False).astype(np.uint32)          # unsorted, no repeats
    short = np.arange(40, dtype=np.uint32)
    for repeat in (False, True):
        row = ids.copy()
        if repeat:
            row[-1] = row[1234]
        col = np.concatenate([short, row, short])
        rowptr = np.array([0, 40, 40 + n, 80 + n], np.uint64)
        t0 = time.perf_counter()
        dev = ctx.csr_upload(rowptr, col, None, np.array([1, 0, 1], np.float32), dim, add_bias=True)
        dev.relabel_by_frequency()
        assert time.perf_counter() - t0 < 20.0
        # a set whose rows repeat a column gets no row masks
        assert (dev.layout_counts()["mask_entries"] == 0) == repeat
        dev.destroy()


def test_upload_checks_on_the_device(ctx, oracle):
    """what admm_csr_upload validates: a row pointer that decreases, an index beyond dim, and (accepted, but
    noted: such sets run the element-by-element FTRL kernel) a row that stores a column twice -- the
    reference applies its update twice then, in sequence (src/ftrl.cc:80-87)"""
    from admm_amd import capi
    lab = np.array([1, 0, 1], np.float32)
    with pytest.raises(capi.AdmmError, match="rowptr decreases at row 1"):
        ctx.csr_upload(np.array([0, 2, 1, 3], np.uint64), np.array([1, 2, 3], np.uint32), None, lab, dim=10, add_bias=True)
    with pytest.raises(capi.AdmmError, match="feature index 9 of row 2"):
        ctx.csr_upload(np.array([0, 1, 2, 3], np.uint64), np.array([1, 2, 9], np.uint32), None, lab, dim=10, add_bias=True)
    # unsorted rows without a repeated column are ordinary sets; with one, the exact pass still matches
    dim = 12
    rowptr = np.array([0, 3, 6, 8], np.uint64)
    col = np.array([4, 1, 4, 7, 2, 5, 3, 3], np.uint32)          # rows 0 and 2 repeat a column
    val = np.array([1, 1, 1, 1, 1, 1, 0.5, 2.0], np.float32)
    raw = oracle.Csr(rowptr, col, val, lab, dim)
    tr = oracle.translate(raw)
    dev = ctx.csr_upload(rowptr, col, val, lab, dim, add_bias=True)
    w, z, n = ctx.vec(dim), ctx.vec(dim), ctx.vec(dim)
    zero = ctx.vec(dim)
    hw, hz, hn = (np.zeros(dim, np.float32) for _ in range(3))
    for _ in range(3):
        dev.ftrl_pass(w, z, n, zero, zero, 0.1, 1.0, 0.01, 0.1)
        oracle.ftrl_run(tr, hw, hz, hn, np.zeros(dim, np.float32), np.zeros(dim, np.float32), 0.1, 1.0, 0.01, 0.1)
    for got, ref in ((w, hw), (z, hz), (n, hn)):
        close(got.download(), ref, rtol=1e-5, atol=1e-7)
    dev.destroy()
