import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from admm_amd import capi
ctx = capi.Context(0)
dim = 1 << 22
res = {}
for mode in ("1", "0"):
    os.environ["ADMM_FORWARD_PANELS"] = mode
    s = ctx.csr_synth(10_000_000, dim, 131, seed=1)
    s.build_transposed(col_block=16384).relabel_by_frequency()
    rng = np.random.default_rng(0)
    x = ctx.vec(dim, rng.normal(0, 0.05, dim).astype(np.float32))
    g = ctx.vec(dim)
    f = s.logistic_fg(x, g, rho=0.01)
    ll, auc, _ = s.metrics([x])
    res[mode] = (f, g.download(), ll, auc)
    s.destroy()
(f1, g1, l1, a1), (f0, g0, l0, a0) = res["1"], res["0"]
print("f", f1, f0, abs(f1 - f0) / abs(f0))
print("g maxdiff", np.abs(g1 - g0).max(), "gmax", np.abs(g0).max())
print("ll", l1, l0, "auc", a1, a0)
