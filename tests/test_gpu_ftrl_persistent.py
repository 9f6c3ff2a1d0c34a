"""The mini-batch FTRL pass as one launch (resident workgroups, device-wide barriers between a batch's two
phases) against the same pass as two launches per batch: same arithmetic in the same order, so w, z and n agree
bit for bit -- which also shows that no workgroup read a stale line of what another one wrote inside the launch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from admm_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def one_pass(ctx, monkeypatch, persistent, rows, dim, batch, col_block, passes=1, seed=11, heavy=False):
    monkeypatch.setenv("ADMM_FTRL_PERSISTENT", "1" if persistent else "0")
    kw = dict(len_dist=1, tail_sigma=1.2, max_len=1024) if heavy else {}
    train = ctx.csr_synth(rows, dim, 40.0, seed=seed, **kw)
    train.relabel_by_frequency()
    train.build_transposed(panel_rows=batch, col_block=col_block)
    rng = np.random.default_rng(seed)
    off = ctx.vec(dim, rng.normal(0, 0.01, dim).astype(np.float32))
    reg = ctx.vec(dim, rng.normal(0, 0.05, dim).astype(np.float32))
    w, z, n = (ctx.vec(dim) for _ in range(3))
    for _ in range(passes):
        train.ftrl_pass(w, z, n, off, reg, 0.05, 1.0, 0.1, 1.0, batch=batch)
    out = tuple(v.download() for v in (w, z, n))
    train.destroy()
    for v in (w, z, n, off, reg):
        v.free()
    monkeypatch.delenv("ADMM_FTRL_PERSISTENT")
    return out


@pytest.mark.parametrize("rows,dim,batch,col_block,heavy", [
    (300_000, 1 << 17, 4096, 2048, False),     # 16 panels per batch, 64 column blocks
    (200_001, 1 << 20, 65536, 1024, False),    # 256 panels per batch, four cells per workgroup, a partial last batch
    (70_000, 1 << 15, 256, 8192, False),       # one panel per batch, 274 batches, wide column blocks
    (150_000, 1 << 18, 8192, 2048, True),      # heavy-tailed rows: lists and cells longer than what is asked for ahead
])
def test_one_launch_equals_two_launches_per_batch(ctx, monkeypatch, rows, dim, batch, col_block, heavy):
    ref = one_pass(ctx, monkeypatch, False, rows, dim, batch, col_block, passes=2, heavy=heavy)
    got = one_pass(ctx, monkeypatch, True, rows, dim, batch, col_block, passes=2, heavy=heavy)
    assert np.count_nonzero(ref[1]) > 100
    for a, b, name in zip(got, ref, "wzn"):
        np.testing.assert_array_equal(a, b, err_msg=name)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("ADMM_FUZZ_SEEDS", "6"))))
def test_one_launch_equals_two_launches_over_shapes(ctx, monkeypatch, seed):
    """seeded shapes: short sets (fewer rows than a batch), few and many column blocks, short and heavy-tailed rows"""
    rng = np.random.default_rng(9100 + seed)
    rows = int(rng.choice([300, 5000, 40_000, 120_000]))
    dim = int(rng.choice([300, 4096, 1 << 15, 1 << 17, 1 << 20]))
    batch = int(rng.choice([256, 512, 1024, 4096, 65536]))
    col_block = min(int(rng.choice([64, 256, 2048, 8192])), (dim // 64) * 64)
    col_block = max(col_block, dim // 8192)      # the library takes up to 8192 column blocks
    heavy = bool(rng.integers(0, 2))
    ref = one_pass(ctx, monkeypatch, False, rows, dim, batch, col_block, passes=2, seed=seed, heavy=heavy)
    got = one_pass(ctx, monkeypatch, True, rows, dim, batch, col_block, passes=2, seed=seed, heavy=heavy)
    for a, b, name in zip(got, ref, "wzn"):
        np.testing.assert_array_equal(a, b, err_msg="%s rows %d dim %d batch %d col_block %d heavy %s" %
                                      (name, rows, dim, batch, col_block, heavy))
