#!/usr/bin/env python3
"""Throughput of one ADMM iteration of the sparse logistic-regression hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload: ADMM-LR with the L1+L2 consensus
update on a synthetic power-law set of 100M rows x 2^22 features, ~100 stored entries per row,
row-sharded over the N GPUs (the same 100M rows for every N: strong scaling).  One step = one
ADMM iteration on every rank: local x-update (L-BFGS on logistic loss + y.(x-z) + rho/2|x-z|^2,
every objective/gradient evaluation = one forward CSR sweep + one transposed sweep over the
rank's shard), payload rho*x+y, sum all-reduce over ranks (RCCL through torch.distributed),
logloss/AUC on the rank's test shard, fused consensus + dual update.

value = stored entries swept per second, whole job: sum over ranks of
        (shard nnz incl. the bias column) x (objective/gradient evaluations in the K steps) / time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=100_000_000, help="total rows over all ranks")
    ap.add_argument("--dim", type=int, default=1 << 22)
    ap.add_argument("--mean-nnz", type=float, default=131.0,
                    help="feature draws per row; ~99 stay after de-duplication at 2^22 features, +1 bias")
    ap.add_argument("--lbfgs-iters", type=int, default=3)
    ap.add_argument("--linesearch", type=int, default=1, choices=[0, 1, 2],
                    help="1 (default here): every trial point of the L-BFGS line search is a full forward + "
                         "transposed evaluation, the mix the metric is defined on (entries per objective/gradient "
                         "evaluation); 0: trial points from the per-row margins, the library's default and the "
                         "faster ADMM iteration, timed beside the timed region as config.other_line_search")
    ap.add_argument("--panel-rows", type=int, default=0)
    ap.add_argument("--col-block", type=int, default=16384)
    ap.add_argument("--test-frac", type=float, default=0.01)
    ap.add_argument("--cpu-sample-rows", type=int, default=400_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-linesearch", action="store_true",
                    help="skip the two extra iterations that time the other line search mode")
    ap.add_argument("--seed", type=int, default=20251111)
    ap.add_argument("--explicit-values", action="store_true",
                    help="store the generated values (all 1.0) as an f32 array instead of keeping the set as "
                         "a pattern matrix, which is how the library keeps any all-ones sample set")
    return ap.parse_args()


def cpu_baseline(ctx, args, rho):
    """The CPU restatement (oracle, one thread) on a bounded sample of the same workload: the same
    generator, the same objective/gradient evaluation, timed on this machine's host cores."""
    from oracle import oracle as orc
    orc.build()
    n = args.cpu_sample_rows
    dev = ctx.csr_synth(n, args.dim, args.mean_nnz, seed=args.seed, row_offset=0,
                        explicit_values=args.explicit_values)
    csr = orc.Csr(*dev.download(), args.dim)
    dev.destroy()
    x = np.zeros(args.dim, np.float32)
    z = np.zeros(args.dim, np.float32)
    y = np.zeros(args.dim, np.float32)
    orc.logistic_fg(csr, args.dim, x, None, z, y, rho)          # warm caches / page in
    evals, t0 = 0, time.perf_counter()
    while True:
        f, g = orc.logistic_fg(csr, args.dim, x, None, z, y, rho)
        x = (x - np.float32(1e-4) * g.astype(np.float32)).astype(np.float32)
        evals += 1
        dt = time.perf_counter() - t0
        if dt > 12.0 or evals >= 48:   # about 12 s of CPU work
            break
    fg_rate = csr.nnz * evals / dt
    # the reference's own local solver on the same sample: one sequential FTRL pass
    w, zz, nn, off = (np.zeros(args.dim, np.float32) for _ in range(4))
    t0 = time.perf_counter()
    orc.ftrl_run(csr, w, zz, nn, off, z, 0.1, 1.0, 1.0, 200.0)
    ftrl_rate = csr.nnz / (time.perf_counter() - t0)
    return {"value": fg_rate, "unit": "nnz/s", "cores": 1, "kind": "port",
            "sample": "%d rows / %d nnz of the same generator; %d objective+gradient evaluations of "
                      "oracle/admm_oracle.c (gcc -O2, 1 thread)" % (n, csr.nnz, evals),
            "ftrl_pass_nnz_per_s": ftrl_rate}


def main():
    args = parse_args()
    import torch
    from admm_amd import capi, driver
    from admm_amd import dist as adist

    rank, world, local_rank = adist.env_rank_world()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE %d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU implementation")
    torch.cuda.set_device(local_rank)
    adist.init("nccl", device=torch.device("cuda", local_rank))

    # one HIP stream for the library's kernels and for torch's collectives
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)
    ctx = capi.Context(local_rank, stream=tstream.cuda_stream)

    # ---- the rank's shard -------------------------------------------------------------------
    rows_total = args.rows
    r0, r1 = adist.shard_range(rows_total, rank, world)
    t_gen = time.perf_counter()
    train = ctx.csr_synth(r1 - r0, args.dim, args.mean_nnz, seed=args.seed, row_offset=r0,
                          explicit_values=args.explicit_values)
    train.build_transposed(panel_rows=args.panel_rows, col_block=args.col_block)
    train.relabel_by_frequency()
    n_test = max(1, int((r1 - r0) * args.test_frac))
    test = ctx.csr_synth(n_test, args.dim, args.mean_nnz, seed=args.seed + 1, row_offset=r0,
                         explicit_values=args.explicit_values)
    test.relabel_by_frequency()
    ctx.sync()
    t_gen = time.perf_counter() - t_gen

    prm = driver.AdmmConfig()
    prm.step_size, prm.global_var, prm.global_l1 = 0.01, 4.0, 0.01     # script/start.sh:25-27
    prm.dim, prm.local_solver = args.dim, "lbfgs"
    prm.lbfgs.update(max_iter=args.lbfgs_iters, linesearch=args.linesearch)
    worker = driver.Worker(ctx)
    worker.InitWorker(args.dim)
    master = driver.Master(ctx)
    zg = ctx.vec(args.dim)
    payload_t = torch.zeros(args.dim, dtype=torch.float32, device="cuda")
    payload = capi.DeviceVec.from_ptr(ctx, payload_t.data_ptr(), args.dim)

    def allreduce(_buf):
        adist.allreduce_sum_(payload_t)

    def step(last=False):
        return driver.admm_iteration(worker, master, [train], test, prm, zg, payload, allreduce,
                                     world, last)

    def barrier():
        adist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.timing_enable(True)
    ctx.timing_reset()
    evals = fwd_sweeps = tr_sweeps = 0
    t0 = time.perf_counter()
    last_rec = None
    for _ in range(args.steps):
        last_rec = step()
        st = last_rec["stats"]
        evals += st["evals"]
        fwd_sweeps += st.get("forward_sweeps", st["evals"])
        tr_sweeps += st.get("transposed_sweeps", st["evals"])
    barrier()
    dt = time.perf_counter() - t0
    fwd_ms, fwd_n = ctx.timing_get(capi.K_FORWARD)
    tr_ms, tr_n = ctx.timing_get(capi.K_TRANSPOSED)
    ctx.timing_enable(False)

    # Beside the timed region: two more ADMM iterations in each of the other line search modes (after one
    # untimed iteration each), so that the line shows what an ADMM iteration costs in the library's default
    # mode and in the fastest one as well.  Not part of `value`.
    other = None
    if not args.no_other_linesearch:
        names = ("from margins", "by full sweeps", "from margins, accepted point too")
        other = []
        for mode in (0, 1, 2):
            if mode == args.linesearch:
                continue
            prm.lbfgs["linesearch"] = mode
            step()
            barrier()
            o_sweeps, o_t0 = 0.0, time.perf_counter()
            for _ in range(2):
                st = step()["stats"]
                o_sweeps += 0.5 * (st.get("forward_sweeps", st["evals"]) + st.get("transposed_sweeps", st["evals"]))
            barrier()
            o_dt = time.perf_counter() - o_t0
            o = adist.allreduce_sum_(torch.tensor([float(train.nnz) * o_sweeps], dtype=torch.float64, device="cuda"))
            o_dt = float(adist.allreduce_max_(torch.tensor([o_dt], dtype=torch.float64, device="cuda")).item())
            other.append({"line_search": names[mode], "lbfgs_linesearch": mode, "ms_per_step": o_dt / 2 * 1e3,
                          "nnz_per_s": float(o.item()) / o_dt, "steps": 2})
        prm.lbfgs["linesearch"] = args.linesearch

    # test logloss / AUC over the test rows of all ranks together, once, after the timed work
    g_loss, g_auc, g_rows = driver.global_test_metrics(ctx, test, [worker.base_vec_, worker.bias_vec_])

    # max time and total swept entries over ranks
    # entries swept: every sweep of the sample set counts its stored entries once, so a forward +
    # transposed pair (one objective/gradient evaluation) counts them twice over two: nnz per pair
    swept = float(train.nnz) * 0.5 * (fwd_sweeps + tr_sweeps)
    dt = float(adist.allreduce_max_(torch.tensor([dt], dtype=torch.float64, device="cuda")).item())
    s = adist.allreduce_sum_(torch.tensor([swept, float(train.nnz), float(evals), float(fwd_sweeps), float(tr_sweeps)],
                                          dtype=torch.float64, device="cuda"))
    swept, nnz_total, evals_total, fwd_total, tr_total = (float(v) for v in s.tolist())

    if rank == 0:
        rows = r1 - r0
        # algorithmic HBM bytes of one launch over this rank's shard (idx u32 + val f32 per stored
        # entry; rowptr 8 + label 4 + residual 4 per row forward, residual 4 per row transposed;
        # one dim-sized vector each).  The figure is a property of the problem, not of the storage:
        # a pattern-matrix set moves fewer bytes than this, as do the entries held in row masks.
        tnnz = train.nnz - rows                       # the bias column is reduced by the forward sweep
        fwd_bytes = 8.0 * train.nnz + 16.0 * rows + 4.0 * args.dim
        tr_bytes = 8.0 * tnnz + 4.0 * rows + 4.0 * args.dim
        fwd_avg = fwd_ms / max(fwd_n, 1) * 1e-3
        tr_avg = tr_ms / max(tr_n, 1) * 1e-3
        kernels = {
            "forward_csr_sweep": {"avg_ms": fwd_avg * 1e3, "launches": fwd_n, "algorithmic_bytes": fwd_bytes,
                                  "achieved_GBs": fwd_bytes / fwd_avg / 1e9 if fwd_avg > 0 else None},
            "transposed_block_sweep": {"avg_ms": tr_avg * 1e3, "launches": tr_n, "algorithmic_bytes": tr_bytes,
                                       "achieved_GBs": tr_bytes / tr_avg / 1e9 if tr_avg > 0 else None},
        }
        dom = "transposed_block_sweep" if tr_ms >= fwd_ms else "forward_csr_sweep"
        ach = kernels[dom]["achieved_GBs"]
        # HBM bytes per launch from the PMC passes kept under profiles/ (FETCH_SIZE / WRITE_SIZE of this
        # same command, corrected as that file explains); only for the workload they were taken on
        traffic, traffic_src = None, None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic_c3.json")
        if os.path.exists(tfile):
            pmc = json.load(open(tfile))
            wl = pmc.get("workload", {})
            if (wl.get("rows_total"), wl.get("dim"), wl.get("n_gpus"), bool(wl.get("explicit_values"))) == \
                    (rows_total, args.dim, world, args.explicit_values):
                sym = "k_transposed<false" if dom == "transposed_block_sweep" else "k_forward<1"
                for name, rec in pmc["kernels"].items():
                    if sym in name:
                        traffic, traffic_src = rec["traffic_bytes"], "profiles/pmc_traffic_c3.json"
        out = {
            "metric": "nnz/sec per ADMM iteration + % HBM roofline at 1/2/4/8 MI355X",
            "value": swept / dt,
            "unit": "nnz/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "ADMM-LR L1+L2, synthetic %d rows x %d features, ~%d nnz/row, row-sharded over %d GPU(s)"
                            % (rows_total, args.dim, round(nnz_total / rows_total), world),
                "local_solver": "lbfgs (max_iter %d, line search %s)" % (
                    args.lbfgs_iters, ("from margins", "by full sweeps", "from margins, accepted point too")[args.linesearch]),
                "values": "all 1.0, stored as f32" if args.explicit_values else
                          "all 1.0, implicit (pattern matrix: no value arrays in HBM)",
                "rows_total": rows_total, "dim": args.dim, "nnz_total": nnz_total,
                "objective_evals_total": evals_total,
                "forward_sweeps_total": fwd_total, "transposed_sweeps_total": tr_total,
                "admm_iterations_per_s": args.steps / dt,
                "other_line_search": other,
                "global_test_metrics": {"logloss": g_loss, "auc": g_auc, "rows": g_rows},
                "generate_and_build_s": t_gen,
                "rank0_hbm_bytes": dict(zip(("row_major", "column_block_copy"), train.bytes())),
                "last_line": driver.format_line("rank0", *last_rec["line"]) if last_rec and last_rec["line"] else None,
                "last_residuals": [last_rec["r2"], last_rec["s2"]] if last_rec else None,
            },
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_source": traffic_src, "algorithmic_bytes": kernels[dom]["algorithmic_bytes"],
                         "kernels": kernels},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ctx, args, float(prm.step_size))
        print(json.dumps(out), flush=True)

    train.destroy()
    test.destroy()
    ctx.close()
    adist.finalize()


if __name__ == "__main__":
    main()
