"""admm_consensus_static (the consensus step for callers that cannot allocate) against
admm_consensus on the same inputs, through the C ABI: z and the dual bit for bit, the two residuals
to 1e-12 relative (double sums over a different partition of the coordinates).  Two calls in a row,
the second with a smaller input, so that what the first call left in the shared static array would
show if it were read."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(ctx, dim, seed, l1, workers, stream=None):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=dim).astype(np.float32)
    y = rng.normal(size=dim).astype(np.float32)
    zold = rng.normal(size=dim).astype(np.float32)
    total = rng.normal(size=dim).astype(np.float32) * np.float32(workers)
    rho, gv = 0.01, 4.0
    # the existing entry point
    dx, dy, dz, dt = ctx.vec(dim, x), ctx.vec(dim, y), ctx.vec(dim, zold), ctx.vec(dim, total)
    r2, s2 = ctx.consensus(dt, workers, rho, gv, l1, dx, dy, dz)
    # the one that allocates nothing: its own copies of the in/out vectors, device residuals
    sy, sz = ctx.vec(dim, y), ctx.vec(dim, zold)
    res = ctx.vec(4)                      # 16 bytes: two doubles
    ctx.sync()
    import torch
    if stream is None:
        ctx.consensus_static(dt, workers, rho, gv, l1, dx, sy, sz, residuals=res)
    else:                                 # a stream of the caller's: the inputs were written on the context's
        ctx.consensus_static(dt, workers, rho, gv, l1, dx, sy, sz, residuals=res, stream=stream.cuda_stream)
    torch.cuda.synchronize()              # the call did not wait for its stream
    got = res.download().view(np.float64)
    np.testing.assert_array_equal(sz.download(), dz.download())
    np.testing.assert_array_equal(sy.download(), dy.download())
    np.testing.assert_allclose(got, [r2, s2], rtol=1e-12, atol=0)
    assert r2 > 0 and s2 > 0
    for v in (dx, dy, dz, dt, sy, sz, res):
        v.free()


def test_consensus_static_matches_consensus_twice():
    from admm_amd import capi
    ctx = capi.Context(0)
    _case(ctx, 300_007, seed=1, l1=0.3, workers=3)     # 74 workgroups' partial slots written
    _case(ctx, 5_001, seed=2, l1=0.0, workers=1)       # 2 workgroups: the other 72 slots are stale
    import torch
    _case(ctx, 70_001, seed=3, l1=0.1, workers=2, stream=torch.cuda.Stream())   # not the default stream
    # without residuals, and past the static array's limit
    dim = 4097
    v = [ctx.vec(dim, np.full(dim, k + 1, np.float32)) for k in range(4)]
    ctx.consensus_static(v[0], 2, 0.5, 1.0, 0.0, v[1], v[2], v[3])
    import torch
    torch.cuda.synchronize()
    np.testing.assert_array_equal(v[3].download(), np.full(dim, 0.5, np.float32))          # 1 / (2*0.5 + 1)
    np.testing.assert_array_equal(v[2].download(), np.full(dim, 3 + 0.5 * (2 - 0.5), np.float32))
    before = v[3].download()
    with pytest.raises(capi.AdmmError, match="exceeds the static scratch"):
        ctx.consensus_static(v[0], 2, 0.5, 1.0, 0.0, v[1], v[2], v[3], dim=(1 << 28) + 1)
    np.testing.assert_array_equal(v[3].download(), before)
    ctx.close()
