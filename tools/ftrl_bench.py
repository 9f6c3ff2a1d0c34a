"""Development aid: time the FTRL pass (exact sequential and mini-batch) on a synthetic set."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from admm_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--dim", type=int, default=1 << 20)
ap.add_argument("--draws", type=float, default=47)
ap.add_argument("--batch", type=str, default="1")
ap.add_argument("--workers", type=int, default=0,
                help="also time this many independent workers (rows / workers each) in one launch")
args = ap.parse_args()

ctx = capi.Context(0)
s = ctx.csr_synth(args.rows, args.dim, args.draws, seed=1)
print("rows %d nnz %d (%.1f per row)" % (s.nrows, s.nnz, s.nnz / s.nrows), flush=True)
for b in [int(x) for x in args.batch.split(",")]:
    if b > 1:
        s.build_transposed(panel_rows=b)
    w, z, n, off, reg = (ctx.vec(args.dim) for _ in range(5))
    for rep in range(2):  # the first pass also prepares per-set side data
        ctx.sync()
        t0 = time.perf_counter()
        s.ftrl_pass(w, z, n, off, reg, 0.01, 1.0, 0.4, 2.5, batch=b)
        ctx.sync()
        dt = time.perf_counter() - t0
        if rep == 0:
            print("first pass %.3f s" % dt, flush=True)
    ll, auc, _ = s.metrics([w])
    print("batch %-8d %.3f s  %.3e nnz/s  %.3e rows/s   train logloss %.5f auc %.5f"
          % (b, dt, s.nnz / dt, s.nrows / dt, ll, auc), flush=True)

if args.workers > 0:
    per = args.rows // args.workers
    shards = [ctx.csr_synth(per, args.dim, args.draws, seed=1, row_offset=k * per) for k in range(args.workers)]
    state = [[ctx.vec(args.dim) for _ in shards] for _ in range(3)]
    off, reg = ctx.vec(args.dim), ctx.vec(args.dim)
    total = sum(x.nnz for x in shards)
    for rep in range(2):
        ctx.sync()
        t0 = time.perf_counter()
        ctx.ftrl_pass_multi(shards, state[0], state[1], state[2], [off] * len(shards), [reg] * len(shards),
                            0.01, 1.0, 0.4, 2.5)
        ctx.sync()
        dt = time.perf_counter() - t0
        print("%d workers x %d rows%s: %.3f s  %.3e nnz/s  %.3e rows/s"
              % (args.workers, per, " (first pass)" if rep == 0 else "", dt, total / dt, per * args.workers / dt), flush=True)
