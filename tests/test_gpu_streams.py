"""The two 16-bit streams of a whole-set copy -- the forward sweep's hot stream (entries in the most
frequent columns, row after row) and the transposed sweep's column stream (the same entries, column
after column) -- against the CPU restatement, and against the forms that do without them: every
form adds the same integers, so they agree bit for bit."""
import numpy as np
import pytest

from conftest import make_synth_csr

pytestmark = pytest.mark.gpu

RTOL = 1e-5


@pytest.fixture(scope="module")
def ctx():
    from admm_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def close(a, b, rtol=RTOL, atol=1e-7):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def ragged_set(nrows, dim, seed, binary):
    """rows of very different kinds: empty rows, rows whose ids all lie in the rare tail, rows of one
    entry, one row of 9000 entries (longer than a lane's run), the rest power-law rows"""
    from oracle import oracle as orc
    base = make_synth_csr(nrows, dim, 40, seed=seed, binary=binary)
    rng = np.random.default_rng(seed + 1)
    rows = [base.col[int(base.rowptr[r]):int(base.rowptr[r + 1])] for r in range(nrows)]
    vals = [base.val[int(base.rowptr[r]):int(base.rowptr[r + 1])] for r in range(nrows)]
    # the rarest ids: the least frequent columns of the base set
    counts = np.bincount(base.col, minlength=dim - 1)
    rare = np.argsort(counts, kind="stable")[:dim // 2]
    for r in range(0, nrows, 7):
        ids = np.sort(rng.choice(rare, 5, replace=False)).astype(np.uint32)
        rows[r], vals[r] = ids, np.ones(5, np.float32)
    for r in range(3, nrows, 11):
        rows[r], vals[r] = rows[r][:0], vals[r][:0]
    for r in range(5, nrows, 13):
        rows[r], vals[r] = rows[r][:1], vals[r][:1]
    long_ids = np.sort(rng.choice(dim - 1, 9000, replace=False)).astype(np.uint32)
    rows[nrows // 2] = long_ids
    vals[nrows // 2] = np.ones(9000, np.float32) if binary else rng.uniform(0.25, 2.0, 9000).astype(np.float32)
    rowptr = np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.uint64)
    return orc.Csr(rowptr, np.concatenate(rows).astype(np.uint32), np.concatenate(vals).astype(np.float32),
                   base.label, dim)


@pytest.mark.parametrize("binary", [True, False])
@pytest.mark.parametrize("relabel", [False, True])
def test_streams_against_the_oracle(ctx, oracle, relabel, binary):
    dim = 40000                       # more columns than the streams take: both forms of every sweep run
    nrows = 20_011                    # three panels of the column stream, five of the hot stream, the last ones partial
    raw = ragged_set(nrows, dim, 301, binary)
    tr = oracle.translate(raw)
    dev = ctx.csr_upload(raw.rowptr, raw.col, raw.val, raw.label, dim, add_bias=True)
    dev.build_transposed(col_block=4096)          # automatic panels: the copy has a column stream
    if relabel:
        dev.relabel_by_frequency()
    rng = np.random.default_rng(9)
    x = rng.normal(0, 0.1, dim).astype(np.float32)
    b = rng.normal(0, 0.02, dim).astype(np.float32)
    z = rng.normal(0, 0.1, dim).astype(np.float32)
    f_ref, g_ref = oracle.logistic_fg(tr, dim, x, b, z, None, 0.05)
    g = ctx.vec(dim)
    f = dev.logistic_fg(ctx.vec(dim, x), g, offset=ctx.vec(dim, b), z=ctx.vec(dim, z), rho=0.05)
    close(f, f_ref)
    close(g.download(), g_ref, rtol=2e-5, atol=2e-5 * np.abs(g_ref).max())
    s64, p_ref = oracle.logloss_sum_f64(tr, [x, b], want_p=True)
    close(dev.predict([ctx.vec(dim, x), ctx.vec(dim, b)]), p_ref, atol=1e-6)
    # large weights: the fixed-point shift of the margins follows max |w| (margins of a few hundred:
    # the float sums of the restatement carry a relative error of 1e-6 of that into the exponent)
    xl = (x * 300).astype(np.float32)
    s64, p_ref = oracle.logloss_sum_f64(tr, [xl], want_p=True)
    close(dev.predict([ctx.vec(dim, xl)]), p_ref, rtol=3e-4, atol=1e-6)
    # a mini-batch FTRL pass sweeps panel ranges: the copy is rebuilt without the column stream
    hw, hz, hn = (np.zeros(dim, np.float32) for _ in range(3))
    oracle.ftrl_run(tr, hw, hz, hn, b, z, 0.05, 1.0, 0.1, 1.0, batch=nrows)
    w, zz, n = ctx.vec(dim), ctx.vec(dim), ctx.vec(dim)
    dev.ftrl_pass(w, zz, n, ctx.vec(dim, b), ctx.vec(dim, z), 0.05, 1.0, 0.1, 1.0, batch=nrows)
    for got, ref in ((w, hw), (zz, hz), (n, hn)):
        close(got.download(), ref, rtol=5e-5, atol=1e-6 * max(1.0, np.abs(ref).max()))
    # and the objective still comes out the same afterwards
    f2 = dev.logistic_fg(ctx.vec(dim, x), g, offset=ctx.vec(dim, b), z=ctx.vec(dim, z), rho=0.05)
    assert f2 == f
    dev.destroy()


@pytest.mark.parametrize("binary", [True, False])
def test_forms_agree_bit_for_bit(ctx, monkeypatch, binary):
    """objective, gradient and predictions of one set built four ways: with both streams; without
    the column stream; with a hot stream of 300 columns only; with the row-major forward sweep"""
    dim = 30000
    raw = ragged_set(9000, dim, 77, binary)
    rng = np.random.default_rng(3)
    x = rng.normal(0, 0.2, dim).astype(np.float32)
    z = rng.normal(0, 0.2, dim).astype(np.float32)

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dev = ctx.csr_upload(raw.rowptr, raw.col, raw.val, raw.label, dim, add_bias=True)
        dev.build_transposed(col_block=2048).relabel_by_frequency()
        for k in env:
            monkeypatch.delenv(k)
        g = ctx.vec(dim)
        f = dev.logistic_fg(ctx.vec(dim, x), g, z=ctx.vec(dim, z), rho=0.1)
        p = dev.predict([ctx.vec(dim, x)])
        dev.destroy()
        return f, g.download(), p

    f0, g0, p0 = run({})
    f1, g1, p1 = run({"ADMM_COLUMN_STREAM": "0"})
    assert f1 == f0
    np.testing.assert_array_equal(g1, g0)
    f2, g2, p2 = run({"ADMM_FORWARD_HOT": "300"})
    assert f2 == f0
    np.testing.assert_array_equal(g2, g0)
    np.testing.assert_array_equal(p2, p0)
    # the row-major forward sweep adds floats: close, not equal
    f3, g3, p3 = run({"ADMM_FORWARD_PANELS": "0"})
    close(f3, f0)
    close(p3, p0, atol=1e-6)
    close(g3, g0, rtol=2e-5, atol=2e-5 * np.abs(g0).max())
