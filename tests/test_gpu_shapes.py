"""The two training paths over a spread of shapes drawn from fixed seeds -- rows that do not fill a
panel, dims below and above the hot stream's 8192 ids, very short and long rows, mini-batches of
256 .. 2048 rows, column blocks of 64 .. 4096 -- each checked against the CPU restatement."""
import os

import numpy as np
import pytest

from conftest import make_synth_csr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from admm_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def close(a, b, rtol, atol):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def draw(seed):
    rng = np.random.default_rng(1000 + seed)
    dim = int(rng.choice([300, 2500, 9000, 20000, 70000]))
    nrows = int(rng.integers(300, 5200))
    mean = float(rng.choice([3, 12, 35, 80]))
    binary = bool(rng.integers(0, 2))
    relabel = bool(rng.integers(0, 2))
    return rng, dim, nrows, mean, binary, relabel


@pytest.mark.parametrize("seed", range(int(os.environ.get("ADMM_SHAPE_SEEDS", "10"))))
def test_objective_and_gradient_over_shapes(ctx, oracle, seed):
    rng, dim, nrows, mean, binary, relabel = draw(seed)
    raw = make_synth_csr(nrows, dim, mean, seed=seed, binary=binary, label01=bool(seed % 2))
    tr = oracle.translate(raw)
    dev = ctx.csr_upload(raw.rowptr, raw.col, raw.val, raw.label, dim, add_bias=True)
    cb = int(rng.choice([64, 256, 1024, 4096]))
    dev.build_transposed(panel_rows=int(rng.choice([0, 0, 700])), col_block=min(cb, max(64, (dim // 64) * 64)))
    if relabel:
        dev.relabel_by_frequency()
    x = rng.normal(0, 0.2, dim).astype(np.float32)
    off = rng.normal(0, 0.05, dim).astype(np.float32)
    z = rng.normal(0, 0.2, dim).astype(np.float32)
    yd = rng.normal(0, 0.02, dim).astype(np.float32)
    f_ref, g_ref = oracle.logistic_fg(tr, dim, x, off, z, yd, 0.3)
    g = ctx.vec(dim)
    for _ in range(2):   # the second call starts from whatever the first left in the accumulators
        f = dev.logistic_fg(ctx.vec(dim, x), g, offset=ctx.vec(dim, off), z=ctx.vec(dim, z), ydual=ctx.vec(dim, yd),
                            rho=0.3)
        close(f, f_ref, 1e-5, 1e-7)
        close(g.download(), g_ref, 2e-5, 2e-5 * np.abs(g_ref).max())
    _, p_ref = oracle.logloss_sum_f64(tr, [x, off], want_p=True)
    close(dev.predict([ctx.vec(dim, x), ctx.vec(dim, off)]), p_ref, 1e-5, 1e-6)
    dev.destroy()


@pytest.mark.parametrize("seed", range(int(os.environ.get("ADMM_SHAPE_SEEDS", "8"))))
def test_minibatch_pass_over_shapes(ctx, oracle, seed):
    rng, dim, nrows, mean, binary, relabel = draw(100 + seed)
    batch = int(rng.choice([256, 512, 768, 2048]))
    raw = make_synth_csr(nrows, dim, mean, seed=50 + seed, binary=binary)
    tr = oracle.translate(raw)
    dev = ctx.csr_upload(raw.rowptr, raw.col, raw.val, raw.label, dim, add_bias=True)
    if relabel:
        dev.relabel_by_frequency()
    cb = int(rng.choice([64, 512, 2048, 4096]))
    dev.build_transposed(panel_rows=batch, col_block=min(cb, max(64, (dim // 64) * 64)))
    reg = rng.normal(0, 0.02, dim).astype(np.float32)
    off = rng.normal(0, 0.02, dim).astype(np.float32)
    hw, hz, hn = (np.zeros(dim, np.float32) for _ in range(3))
    w, z, n = ctx.vec(dim), ctx.vec(dim), ctx.vec(dim)
    doff, dreg = ctx.vec(dim, off), ctx.vec(dim, reg)
    for _ in range(2):
        oracle.ftrl_run(tr, hw, hz, hn, off, reg, 0.05, 1.0, 0.2, 1.5, batch=batch)
        dev.ftrl_pass(w, z, n, doff, dreg, 0.05, 1.0, 0.2, 1.5, batch=batch)
    for got, ref in ((w, hw), (z, hz), (n, hn)):
        close(got.download(), ref, 5e-5, 2e-6 * max(1.0, np.abs(ref).max()))
    dev.destroy()
