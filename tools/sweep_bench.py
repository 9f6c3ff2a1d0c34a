"""Development aid: time the two sweeps of one objective/gradient evaluation on a synthetic set,
optionally with the transposed sweep's gathers switched off (--no-gather: a run that moves a
known number of bytes, used to calibrate the hardware traffic counters)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from admm_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=1 << 22)
ap.add_argument("--draws", type=float, default=131)
ap.add_argument("--panel-rows", type=int, default=0)
ap.add_argument("--col-block", type=int, default=16384)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--no-gather", action="store_true")
ap.add_argument("--explicit-values", action="store_true")
ap.add_argument("--no-relabel", action="store_true")
ap.add_argument("--heavy", action="store_true", help="log-normal row lengths (sigma 1.2, up to 1024 draws)")
args = ap.parse_args()

ctx = capi.Context(0)
kw = dict(len_dist=1, tail_sigma=1.2, max_len=1024) if args.heavy else {}
s = ctx.csr_synth(args.rows, args.dim, args.draws, seed=1, explicit_values=args.explicit_values, **kw)
s.build_transposed(panel_rows=args.panel_rows, col_block=args.col_block)
if not args.no_relabel:
    s.relabel_by_frequency()
x, g = ctx.vec(args.dim), ctx.vec(args.dim)
import numpy as np
x.upload(np.random.default_rng(0).normal(0, 0.05, args.dim).astype(np.float32))
rows, nnz = s.nrows, s.nnz
print("hbm bytes: row-major %d, column-block copy %d" % s.bytes(), flush=True)
fb = 8.0 * nnz + 16.0 * rows + 4.0 * args.dim
tb = 8.0 * (nnz - rows) + 4.0 * rows + 4.0 * args.dim
if args.no_gather:
    os.environ["ADMM_CALIBRATE_NO_GATHER"] = "1"
for ab in ["no-gather" if args.no_gather else "normal"]:
    s.logistic_fg(x, g, rho=0.01)
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(args.reps):
        s.logistic_fg(x, g, rho=0.01)
    f_ms, f_n = ctx.timing_get(capi.K_FORWARD)
    t_ms, t_n = ctx.timing_get(capi.K_TRANSPOSED)
    ctx.timing_enable(False)
    print("%-9s rows=%d nnz=%d  forward %.3f ms (%.0f GB/s alg)  transposed %.3f ms (%.0f GB/s alg)"
          % (ab, rows, nnz, f_ms / f_n, fb / (f_ms / f_n) / 1e6, t_ms / t_n, tb / (t_ms / t_n) / 1e6), flush=True)
