#!/usr/bin/env python3
"""Throughput of the ADMM / FTRL sparse logistic-regression hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config c3|c5|ftrl]

--config c3 (default): ADMM-LR with the L1+L2 consensus update on a synthetic power-law set of
    100M rows x 2^22 features, ~100 stored entries per row, row-sharded over the N GPUs (the same
    100M rows for every N: strong scaling).  One step = one ADMM iteration on every rank: local
    x-update (L-BFGS on logistic loss + y.(x-z) + rho/2|x-z|^2, every objective/gradient evaluation
    = one forward CSR sweep + one transposed sweep over the rank's shard), payload rho*x+y, sum
    all-reduce over ranks (RCCL behind admm_allreduce_sum_f32), logloss/AUC on the rank's test
    shard, fused consensus + dual update.
--config c5: the same iteration on the Criteo-like shape: 2^24 hashed features, log-normal row
    lengths (mean ~33 stored entries, rows of many hundred among them), 125M rows PER GPU (N = 8 is
    the 1B-row job; N = 1 is one of its shards: weak scaling).
--config ftrl: FTRL-proximal single pass, 10M rows x 2^20 features, ~40 entries per row, one GPU.
    One step = one pass over the set in the mini-batch form; the row-by-row form (the reference's
    semantics) is timed beside it on a bounded sample.

value = stored entries swept per second, whole job: sum over ranks of
        (shard nnz incl. the bias column) x (objective/gradient evaluations in the K steps) / time
        (ftrl: nnz x passes / time).
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
METRIC = "nnz/sec per ADMM iteration + % HBM roofline at 1/2/4/8 MI355X"

CONFIGS = {
    # rows: total over all ranks (c3) or per rank (c5); draws: feature draws per row before de-duplication
    "c3": dict(rows=100_000_000, per_rank=False, dim=1 << 22, draws=131.0, heavy=False, col_block=16384),
    "c5": dict(rows=125_000_000, per_rank=True, dim=1 << 24, draws=40.0, heavy=True, col_block=16384),
    "ftrl": dict(rows=10_000_000, per_rank=False, dim=1 << 20, draws=47.0, heavy=False, col_block=2048),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3")
    ap.add_argument("--rows", type=int, default=0, help="override the configuration's row count")
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--mean-nnz", type=float, default=0.0,
                    help="feature draws per row; c3: 131 draws leave ~99 after de-duplication at 2^22 features, +1 bias")
    ap.add_argument("--lbfgs-iters", type=int, default=3)
    ap.add_argument("--linesearch", type=int, default=1, choices=[0, 1, 2],
                    help="1 (default here): every trial point of the L-BFGS line search is a full forward + "
                         "transposed evaluation, the mix the metric is defined on (entries per objective/gradient "
                         "evaluation); 0: trial points from the per-row margins, the library's default and the "
                         "faster ADMM iteration, timed beside the timed region as config.other_line_search")
    ap.add_argument("--panel-rows", type=int, default=0)
    ap.add_argument("--col-block", type=int, default=0)
    ap.add_argument("--test-frac", type=float, default=0.01)
    ap.add_argument("--ftrl-batch", type=int, default=65536, help="rows per mini-batch of the timed FTRL pass")
    ap.add_argument("--cpu-sample-rows", type=int, default=400_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-linesearch", action="store_true",
                    help="skip the two extra iterations that time the other line search mode")
    ap.add_argument("--ftrl-pass-only", action="store_true",
                    help="ftrl configuration: skip the row-by-row passes and the quality comparison beside the timed "
                         "region (counter passes want the mini-batch pass's launches alone)")
    ap.add_argument("--keep-rows", action="store_true",
                    help="keep the training set's row-major forms in HBM beside its copies (admm_csr_set_lean off)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="c3 on one GPU also runs the ftrl configuration and a c5 shard at full size after its "
                         "timed region and attaches them as config.other_configs; this switch skips that")
    ap.add_argument("--time-every", type=int, default=4,
                    help="HIP events bracket every n-th launch of each sweep (the kernel averages are over that "
                         "sample); 1 brackets them all")
    ap.add_argument("--seed", type=int, default=20251111)
    ap.add_argument("--explicit-values", action="store_true",
                    help="store the generated values (all 1.0) as an f32 array instead of keeping the set as "
                         "a pattern matrix, which is how the library keeps any all-ones sample set")
    ap.add_argument("--allreduce", choices=["lib", "torch"], default=None,
                    help="lib: admm_allreduce_sum_f32 (RCCL through the library, the product's path); "
                         "torch: torch.distributed's all-reduce on the same buffer.  Not given: lib, and when the "
                         "library's communicator fails or gives no answer the run goes on with torch's and says so "
                         "in config.allreduce_fallback; given as lib, such a failure ends the run with exit code 3")
    args = ap.parse_args()
    args.overridden = bool(args.rows or args.dim or args.mean_nnz or args.panel_rows or args.col_block)
    return apply_config(args, args.config)


def apply_config(args, name, **override):
    """a copy of `args` set up for configuration `name` (explicit --rows etc. survive for the named one only)"""
    import copy
    a = copy.copy(args)
    cfg = CONFIGS[name]
    primary = name == args.config
    a.config = name
    a.rows = (args.rows if primary else 0) or cfg["rows"]
    a.dim = (args.dim if primary else 0) or cfg["dim"]
    a.mean_nnz = (args.mean_nnz if primary else 0.0) or cfg["draws"]
    a.col_block = (args.col_block if primary else 0) or cfg["col_block"]
    if not primary:
        a.panel_rows = 0
    a.heavy, a.per_rank = cfg["heavy"], cfg["per_rank"]
    for k, v in override.items():
        setattr(a, k, v)
    return a


def synth_kw(args):
    kw = dict(explicit_values=args.explicit_values)
    if args.heavy:
        kw.update(len_dist=1, tail_sigma=1.2, max_len=1024)
    return kw


# ---------------------------------------------------------------------------------------------------
# CPU side by side
# ---------------------------------------------------------------------------------------------------

def cpu_baseline(ctx, args, rho):
    """The CPU restatement (oracle/, the reference's arithmetic in plain C) timed on this machine's host
    cores on a bounded sample of the same workload.  Two legs:
      - one thread, in RAM: objective/gradient evaluations (the L-BFGS path) and one row-by-row FTRL pass;
      - the reference's own shape of a run: N worker processes (N = the host cores given to this job, at
        most 64) under tools/dmlc_local.py, each an ADMM worker on its own libsvm shard with the
        host shared-memory sum standing in for rabit's all-reduce, timed both with the shard parsed
        once and with the text parsed again in every iteration as the reference does."""
    from oracle import oracle as orc
    orc.build()
    n = args.cpu_sample_rows
    dev = ctx.csr_synth(n, args.dim, args.mean_nnz, seed=args.seed, row_offset=0, **synth_kw(args))
    csr = orc.Csr(*dev.download(), args.dim)
    dev.destroy()
    out = {"unit": "nnz/s", "kind": "port"}
    if args.config != "ftrl":
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
            if dt > 8.0 or evals >= 32:   # about 8 s of CPU work
                break
        out.update(value=csr.nnz * evals / dt, cores=1,
                   sample="%d rows / %d nnz of the same generator; %d objective+gradient evaluations of "
                          "oracle/admm_oracle.c (gcc -O2, 1 thread)" % (n, csr.nnz, evals))
    # the reference's own local solver on the same sample: one sequential FTRL pass
    w, zz, nn, off, reg = (np.zeros(args.dim, np.float32) for _ in range(5))
    t0 = time.perf_counter()
    orc.ftrl_run(csr, w, zz, nn, off, reg, 0.1, 1.0, 1.0, 200.0)
    ftrl_rate = csr.nnz / (time.perf_counter() - t0)
    out["ftrl_pass_nnz_per_s"] = ftrl_rate
    if args.config == "ftrl":
        out.update(value=ftrl_rate, cores=1,
                   sample="%d rows / %d nnz of the same generator; one row-by-row FTRL pass of "
                          "oracle/admm_oracle.c (gcc -O2, 1 thread)" % (n, csr.nnz))
    try:
        out["multi_process"] = cpu_baseline_multiprocess(csr, args)
    except Exception as e:     # the side-by-side leg must not take the GPU line down with it
        out["multi_process"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def cpu_baseline_multiprocess(csr, args):
    """N oracle worker processes under tools/dmlc_local.py: see tools/cpu_admm_worker.py."""
    import shutil
    import subprocess
    import tempfile
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", cores) or cores), 64))
    rows_per = max(2000, min(50000, csr.nrows // cores))
    tmp = tempfile.mkdtemp(prefix="admm_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        from tools import cpu_admm_worker as caw
        nnz_total = caw.write_shards(csr, tmp, cores, rows_per)
        res = {}
        for mode in ("resident", "reparse"):
            cmd = [sys.executable, os.path.join(ROOT, "tools", "dmlc_local.py"), "-n", str(cores),
                   sys.executable, os.path.join(ROOT, "tools", "cpu_admm_worker.py"), tmp, str(args.dim), mode, "8"]
            t0 = time.perf_counter()
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            wall = time.perf_counter() - t0
            if p.returncode != 0:
                raise RuntimeError("cpu workers failed: " + (p.stdout + p.stderr)[-400:])
            line = [l for l in p.stdout.splitlines() if l.startswith("CPU_ADMM ")][-1]
            rec = json.loads(line[len("CPU_ADMM "):])
            res[mode] = {"nnz_per_s": nnz_total * rec["iterations"] / rec["loop_s"], "iterations": rec["iterations"],
                         "loop_s": rec["loop_s"], "wall_s": wall}
        cpu_model = ""
        try:
            cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            pass
        return {"cores": cores, "cpu_model": cpu_model, "rows_per_worker": rows_per, "nnz_total": nnz_total,
                "what": "the reference's ADMM iteration (one row-by-row FTRL pass per worker, payload, sum all-reduce, "
                        "z-update, dual update) restated in oracle/, one process per core under tools/dmlc_local.py; "
                        "the all-reduce goes through a shared-memory file",
                "in_ram_sweep_nnz_per_s": res["resident"]["nnz_per_s"],
                "reparse_each_iteration_nnz_per_s": res["reparse"]["nnz_per_s"], "detail": res}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------
# helpers shared by the configurations
# ---------------------------------------------------------------------------------------------------

class Job:
    """rank bookkeeping, the library context with its communicator, and the control-plane helpers"""

    def __init__(self, args):
        import torch
        from admm_amd import capi
        from admm_amd import dist as adist
        self.torch, self.capi, self.adist = torch, capi, adist
        self.rank, self.world, self.local_rank = adist.env_rank_world()
        if self.world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE %d: launch with torch.distributed.run" % (args.gpus, self.world))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the hot path has no CPU implementation")
        # Rehearsal of the N > 1 code path on a box with ONE GPU (ADMM_BENCH_SHARE_GPU=1): every rank drives
        # GPU 0 and the payload is summed on the host through gloo, because RCCL cannot place two ranks on one
        # GPU.  It checks the rank logic, not the collective; its numbers mean nothing.
        self.share_gpu = os.environ.get("ADMM_BENCH_SHARE_GPU") == "1" and self.world > 1
        if self.share_gpu:
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        if self.world > 1:
            # control plane only (barriers, max of the timings, the communicator id): host tensors on gloo;
            # the nccl half is created lazily and only used with --allreduce torch
            adist.init("gloo" if self.share_gpu else "cpu:gloo,cuda:nccl")
        self.ctx = capi.Context(self.local_rank)
        self.allreduce_path = "single rank"
        self.comm_stuck = False
        self.allreduce_fallback = None
        lib_demanded = args.allreduce == "lib"
        if args.allreduce is None:
            args.allreduce = "lib"
        if self.share_gpu:
            self.ctx.rank, self.ctx.world = self.rank, self.world
            self.ctx.exchange = self._host_exchange
            self.allreduce_path = "REHEARSAL: host sum through gloo, all ranks on one GPU"
            args.allreduce = "lib"
        elif self.world > 1:
            self.allreduce_path = "admm_allreduce_sum_f32 (RCCL through libadmm_hip.so)"
            if args.allreduce == "lib":
                # communicator + self test in a thread of their own: an error OR no answer within two
                # minutes sends every rank to torch.distributed's all-reduce instead
                import threading
                state = {}

                def bring_up():
                    try:
                        torch.cuda.set_device(self.local_rank)
                        adist.lib_comm_init(self.ctx)
                        self.ctx.comm_selftest()
                        state["ok"] = True
                    except Exception as e:   # noqa: BLE001 - reported below, the run goes on without the library path
                        state["error"] = e

                th = threading.Thread(target=bring_up, daemon=True)
                th.start()
                th.join(timeout=float(os.environ.get("ADMM_BENCH_COMM_TIMEOUT", "120")))
                if not state.get("ok"):
                    why = state.get("error", "no answer within %s s" % os.environ.get("ADMM_BENCH_COMM_TIMEOUT", "120"))
                    self.allreduce_fallback = "rank %d: %s" % (self.rank, why)
                    sys.stderr.write("rank %d: library communicator failed (%s); using torch.distributed\n" % (self.rank, why))
                    args.allreduce = "torch"
                    self.comm_stuck = th.is_alive()
            ok = self.host_sum([1.0 if args.allreduce == "lib" else 0.0])[0]
            if ok != self.world:        # every rank takes the same path
                if self.allreduce_fallback is None and args.allreduce == "lib":
                    self.allreduce_fallback = "the library communicator failed on %d other rank(s)" % int(self.world - ok)
                args.allreduce = "torch"
                if lib_demanded:
                    sys.stderr.write("rank %d: --allreduce lib was asked for and the library's communicator is not usable "
                                     "(%s)\n" % (self.rank, self.allreduce_fallback))
                    sys.stderr.flush()
                    os._exit(3)
            if args.allreduce == "torch":
                self.allreduce_path = "torch.distributed all_reduce (nccl = RCCL)"
        self.use_lib = args.allreduce == "lib" or self.world == 1

    def allreduce(self, vec, n=None):
        """in-place float sum over ranks of a library device vector, on the library's communicator"""
        if self.world > 1:
            self.ctx.allreduce_sum(vec, n)

    def _host_exchange(self, vec, cnt):
        part = self.capi.DeviceVec.from_ptr(self.ctx, vec.ptr.value, cnt)
        t = self.torch.from_numpy(part.download())
        self.adist.allreduce_sum_(t)
        part.upload(t.numpy())

    def host_sum(self, values):
        if self.world == 1:
            return list(values)
        t = self.torch.tensor(values, dtype=self.torch.float64)
        self.adist.allreduce_sum_(t)
        return t.tolist()

    def host_max(self, value):
        if self.world == 1:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64)
        self.adist.allreduce_max_(t)
        return float(t.item())

    def barrier(self):
        self.ctx.sync()
        self.adist.barrier()
        self.torch.cuda.synchronize()

    def close(self):
        if self.comm_stuck:      # a thread still sits in the communicator's bring-up: leave without joining it
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)
        self.ctx.close()
        self.adist.finalize()


def pmc_traffic(args, world, rows_total):
    """HBM bytes per launch of the two sweeps from the PMC passes kept under profiles/ (FETCH_SIZE / WRITE_SIZE
    of this same command, corrected as the file explains); only for the workload they were taken on.
    Returns ({"forward_csr_sweep": bytes, "transposed_block_sweep": bytes}, source) or (None, None)."""
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % args.config)
    if not os.path.exists(tfile):
        return None, None
    pmc = json.load(open(tfile))
    wl = pmc.get("workload", {})
    if (wl.get("rows_total"), wl.get("dim"), wl.get("n_gpus"), bool(wl.get("explicit_values"))) != \
            (rows_total, args.dim, world, args.explicit_values):
        return None, None
    got = {}
    for name, rec in pmc["kernels"].items():
        if "k_forward_rp<1" in name:
            got["forward_csr_sweep"] = rec["traffic_bytes"]
        elif "k_transposed<false" in name and "transposed_sweep_total" not in pmc:
            got["transposed_block_sweep"] = rec["traffic_bytes"]
    if "transposed_sweep_total" in pmc:   # cells + column stream: one sweep
        got["transposed_block_sweep"] = pmc["transposed_sweep_total"]["traffic_bytes"]
    return (got or None), "profiles/pmc_traffic_%s.json" % args.config


def sweep_kernels(train, rows, dim, fwd_ms, fwd_n, tr_ms, tr_n, margins_written=True):
    """per-kernel roofline records of the two sweeps over one rank's shard.
    algorithmic bytes: idx u32 + val f32 per stored entry; rowptr 8 + label 4 + residual 4 per row forward,
    residual 4 per row transposed; one dim-sized vector each.  A property of the problem, not of the storage.
    stored bytes: what the layout the kernel streams holds (a pattern-matrix set has no value arrays, the
    entries held in row masks cost a bit each) plus the per-row vectors it reads and writes."""
    lay = train.layout_bytes()
    tnnz = train.nnz - rows                       # the bias column is reduced by the forward sweep
    fwd_bytes = 8.0 * train.nnz + 16.0 * rows + 4.0 * dim
    tr_bytes = 8.0 * tnnz + 4.0 * rows + 4.0 * dim
    fwd_stored = lay["forward_form"] + 4.0 * rows * (2 if margins_written else 1) + 4.0 * dim
    tr_stored = lay["column_block_copy"] + 4.0 * rows + 16.0 * dim
    out = {}
    for name, ms, n, alg, stored in (("forward_csr_sweep", fwd_ms, fwd_n, fwd_bytes, fwd_stored),
                                     ("transposed_block_sweep", tr_ms, tr_n, tr_bytes, tr_stored)):
        avg = ms / max(n, 1) * 1e-3
        out[name] = {"avg_ms": avg * 1e3, "launches": n, "algorithmic_bytes": alg, "stored_bytes": stored,
                     "achieved_GBs": alg / avg / 1e9 if avg > 0 else None,
                     "stored_GBs": stored / avg / 1e9 if avg > 0 else None}
    return out


def roofline_of(kernels, dom, traffic, traffic_src):
    """roofline record of the dominant sweep `dom`.  traffic: {kernel: PMC bytes per launch} or None; with it every
    kernel also gets frac_counters = PMC bytes / its HIP-event time / peak, the share of the HBM peak that the
    bytes the kernel really moved amount to (frac divides the problem's 8 bytes per stored entry instead)."""
    for name, k in kernels.items():
        t = (traffic or {}).get(name)
        k["traffic_bytes"] = t
        k["frac"] = k["achieved_GBs"] / HBM_PEAK_GBS if k["achieved_GBs"] else None
        k["frac_stored"] = k["stored_GBs"] / HBM_PEAK_GBS if k["stored_GBs"] else None
        k["frac_counters"] = (t / (k["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if (t and k["avg_ms"] > 0) else None
    k = kernels[dom]
    return {"bound": "hbm", "kernel": dom, "achieved": k["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": k["frac"],
            "algorithmic_bytes": k["algorithmic_bytes"],
            "stored_bytes": k["stored_bytes"], "frac_stored": k["frac_stored"],
            "traffic": k["traffic_bytes"], "traffic_unit": "bytes per launch", "traffic_source": traffic_src,
            "frac_counters": k["frac_counters"],
            "traffic_measured_in_run": False,
            "note": "achieved/frac divide the ALGORITHMIC bytes (8 B per stored entry) by the kernel's HIP-event time; "
                    "stored_bytes/frac_stored divide what the kernel's layout actually holds; traffic is a PMC "
                    "measurement of an earlier run of this command kept under profiles/, not of this run, and "
                    "frac_counters = traffic / this run's kernel time / peak: the kernels move fewer bytes than the "
                    "8 per entry the metric counts (pattern matrix, row masks, 16-bit stream words), so frac says how "
                    "close the sweep is to the metric's roofline and frac_counters how busy the HBM really is",
            "kernels": kernels}


# ---------------------------------------------------------------------------------------------------
# ADMM iteration (c3, c5)
# ---------------------------------------------------------------------------------------------------

def run_admm(args, job, side=False):
    """the ADMM iteration benchmark; returns rank 0's record (None on the other ranks).  side: the configuration
    is run beside the primary one (no CPU baseline, no other line search modes)."""
    ctx, capi, rank, world = job.ctx, job.capi, job.rank, job.world
    from admm_amd import driver
    from admm_amd import dist as adist

    sharding = "contiguous row ranges of equal size; synthetic rows are identically distributed, so the shards' entry " \
               "counts differ by the factor below"
    if args.per_rank:
        rows_total = args.rows * world
        scaling = "weak"
        if world > 1 and args.heavy:
            # heavy-tailed rows: cut the job's rows so that every rank gets the same number of ENTRIES, from the
            # generator's draws per block of rows (known without generating anything); blocks of at most 65536
            # rows, about a thousand per rank
            block = 256
            while block < 65536 and block * 1024 * world < rows_total:
                block *= 2
            draws = ctx.synth_row_draws(rows_total, args.dim, args.mean_nnz, block, seed=args.seed, **synth_kw(args))
            prefix = np.concatenate([[0], np.cumsum(draws.astype(np.uint64))])
            b0, b1 = adist.shard_range_by_nnz(prefix, rank, world)
            r0, r1 = b0 * block, min(rows_total, b1 * block)
            sharding = "contiguous row ranges cut at multiples of %d rows so that the ranks' generator draws (entries " \
                       "before a row's repeats are dropped) are equal: adist.shard_range_by_nnz" % block
        else:
            r0, r1 = args.rows * rank, args.rows * (rank + 1)
    else:
        rows_total = args.rows
        r0, r1 = adist.shard_range(rows_total, rank, world)
        scaling = "strong"
    t_gen = time.perf_counter()
    train = ctx.csr_synth(r1 - r0, args.dim, args.mean_nnz, seed=args.seed, row_offset=r0, **synth_kw(args))
    if not args.keep_rows:
        train.set_lean()                  # whole-set sweeps only: each row-major form is freed once its copies exist
    train.relabel_by_frequency()          # first: the copies are then built once, in the final numbering
    train.build_transposed(panel_rows=args.panel_rows, col_block=args.col_block)
    n_test = max(1, int((r1 - r0) * args.test_frac))
    test = ctx.csr_synth(n_test, args.dim, args.mean_nnz, seed=args.seed + 1, row_offset=r0, **synth_kw(args))
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
    torch_payload = None
    if job.use_lib:
        payload = ctx.vec(args.dim + driver.RESIDUAL_TAIL)

        def allreduce(buf):
            job.allreduce(buf)
    else:
        torch = job.torch
        torch_payload = torch.zeros(args.dim + driver.RESIDUAL_TAIL, dtype=torch.float32, device="cuda")
        payload = capi.DeviceVec.from_ptr(ctx, torch_payload.data_ptr(), args.dim + driver.RESIDUAL_TAIL)

        def allreduce(_buf):
            ctx.sync()
            adist.allreduce_sum_(torch_payload)
            torch.cuda.current_stream().synchronize()

    loop = driver.AdmmLoop(worker, master, [train], test, prm, zg, payload, allreduce, world)

    def step(last=False):
        return loop.iteration(last)

    for _ in range(args.warmup):
        step()
    job.barrier()
    # only the two sweeps are timed: an event pair around each of the step's small kernels as well would
    # stretch the step it measures
    # ... and of those only every args.time_every-th launch: the two event records of a timed launch hold the
    # next kernel up by a few microseconds, which a small shard's iteration feels
    ctx.timing_enable(True, classes=[capi.K_FORWARD, capi.K_TRANSPOSED], every=args.time_every)
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
    job.barrier()
    dt = time.perf_counter() - t0
    fwd_ms, fwd_n = ctx.timing_get(capi.K_FORWARD)
    tr_ms, tr_n = ctx.timing_get(capi.K_TRANSPOSED)
    ctx.timing_enable(False)

    # Beside the timed region: two more ADMM iterations in each of the other line search modes (after one
    # untimed iteration each), so that the line shows what an ADMM iteration costs in the library's default
    # mode and in the fastest one as well.  Not part of `value`.
    other = None
    out = None
    if not args.no_other_linesearch and not side:
        names = ("from margins", "by full sweeps", "from margins, accepted point too")
        other = []
        for mode in (0, 1, 2):
            if mode == args.linesearch:
                continue
            prm.lbfgs["linesearch"] = mode
            step()
            job.barrier()
            o_sweeps, o_t0 = 0.0, time.perf_counter()
            for _ in range(2):
                st = step()["stats"]
                o_sweeps += 0.5 * (st.get("forward_sweeps", st["evals"]) + st.get("transposed_sweeps", st["evals"]))
            job.barrier()
            o_dt = job.host_max(time.perf_counter() - o_t0)
            o_swept = job.host_sum([float(train.nnz) * o_sweeps])[0]
            other.append({"line_search": names[mode], "lbfgs_linesearch": mode, "ms_per_step": o_dt / 2 * 1e3,
                          "nnz_per_s": o_swept / o_dt, "steps": 2})
        prm.lbfgs["linesearch"] = args.linesearch

    # residuals of the last iteration over all ranks (the loop carries them one iteration behind, in the
    # tail of the payload; this flushes the last pair), and test logloss / AUC over the test rows of all ranks
    global_res = loop.flush_residuals()
    g_loss, g_auc, g_rows = driver.global_test_metrics(ctx, test, [worker.base_vec_, worker.bias_vec_])

    # max time and total swept entries over ranks; every sweep of the sample set counts its stored entries
    # once, so a forward + transposed pair (one objective/gradient evaluation) counts them twice over two
    swept = float(train.nnz) * 0.5 * (fwd_sweeps + tr_sweeps)
    dt = job.host_max(dt)
    swept, nnz_total, evals_total, fwd_total, tr_total = job.host_sum(
        [swept, float(train.nnz), float(evals), float(fwd_sweeps), float(tr_sweeps)])
    nnz_max = job.host_max(float(train.nnz))

    if rank == 0:
        rows = r1 - r0
        kernels = sweep_kernels(train, rows, args.dim, fwd_ms, fwd_n, tr_ms, tr_n)
        dom = "transposed_block_sweep" if tr_ms >= fwd_ms else "forward_csr_sweep"
        traffic, traffic_src = pmc_traffic(args, world, rows_total)
        sweep_ms = (fwd_ms / max(fwd_n, 1) * fwd_sweeps + tr_ms / max(tr_n, 1) * tr_sweeps) / max(args.steps, 1)
        out = {
            "metric": METRIC,
            "value": swept / dt,
            "unit": "nnz/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "ADMM-LR L1+L2, synthetic %d rows x %d features, ~%d nnz/row%s, row-sharded over %d GPU(s)"
                            % (rows_total, args.dim, round(nnz_total / rows_total),
                               " (log-normal row lengths, hashed power-law ids)" if args.heavy else "", world),
                "name": args.config,
                "local_solver": "lbfgs (max_iter %d, line search %s)" % (
                    args.lbfgs_iters, ("from margins", "by full sweeps", "from margins, accepted point too")[args.linesearch]),
                "values": "all 1.0, stored as f32" if args.explicit_values else
                          "all 1.0, implicit (pattern matrix: no value arrays in HBM)",
                "allreduce": job.allreduce_path,
                "allreduce_fallback": job.allreduce_fallback,
                "sharding": sharding,
                "shard_nnz_max_over_mean": nnz_max / (nnz_total / world),
                "rows_total": rows_total, "dim": args.dim, "nnz_total": nnz_total,
                "objective_evals_total": evals_total,
                "forward_sweeps_total": fwd_total, "transposed_sweeps_total": tr_total,
                "admm_iterations_per_s": args.steps / dt,
                "ms_per_step_outside_the_sweeps": dt / args.steps * 1e3 - sweep_ms,
                "sweep_launches_timed": "every %d-th launch of each sweep is bracketed by HIP events; roofline.kernels.*."
                                        "launches counts those" % args.time_every,
                "other_line_search": other,
                "global_test_metrics": {"logloss": g_loss, "auc": g_auc, "rows": g_rows},
                "generate_and_build_s": t_gen,
                "rank0_hbm_bytes": train.layout_bytes(),
                "rank0_layout_counts": train.layout_counts(),
                "last_line": driver.format_line("rank0", *last_rec["line"]) if last_rec and last_rec["line"] else None,
                "last_residuals_rank0": [last_rec["r2"], last_rec["s2"]] if last_rec else None,
                "last_residuals_all_ranks": global_res,
            },
            "roofline": roofline_of(kernels, dom, traffic, traffic_src),
        }
        if world == 1 and not args.no_cpu_baseline and not side:
            out["cpu_baseline"] = cpu_baseline(ctx, args, float(prm.step_size))

    train.destroy()
    test.destroy()
    for v in (zg, worker.base_vec_, worker.bias_vec_, worker.langr_vec_, worker.nt_vec_, worker.zt_vec_,
              worker._n, worker._z):
        v.free()
    if job.use_lib:
        payload.free()
    return out if rank == 0 else None


# ---------------------------------------------------------------------------------------------------
# FTRL single pass
# ---------------------------------------------------------------------------------------------------

def run_ftrl(args, job, side=False):
    ctx, capi, rank, world = job.ctx, job.capi, job.rank, job.world
    out = None
    from admm_amd import dist as adist
    rows_total = args.rows
    r0, r1 = adist.shard_range(rows_total, rank, world)       # N > 1: independent passes over row shards
    dim, batch = args.dim, args.ftrl_batch
    prm = (0.01, 1.0, 0.4, 2.5)                                # script/start.sh:15-31
    t_gen = time.perf_counter()
    train = ctx.csr_synth(r1 - r0, dim, args.mean_nnz, seed=args.seed, row_offset=r0, **synth_kw(args))
    train.relabel_by_frequency()
    train.build_transposed(panel_rows=batch, col_block=args.col_block)
    test = ctx.csr_synth(max(1, int((r1 - r0) * args.test_frac)), dim, args.mean_nnz, seed=args.seed + 1,
                         row_offset=r0, **synth_kw(args))
    ctx.sync()
    t_gen = time.perf_counter() - t_gen
    w, z, n, off, reg = (ctx.vec(dim) for _ in range(5))

    def one_pass():
        for v in (w, z, n):
            v.zero()
        train.ftrl_pass(w, z, n, off, reg, *prm, batch=batch)

    for _ in range(args.warmup):
        one_pass()
    job.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    job.barrier()
    dt = job.host_max(time.perf_counter() - t0)
    # per-kernel breakdown from one more pass, outside the timed region: a pass is 600 small launches, and a
    # pair of HIP events around each of them costs more than a tenth of the pass
    ctx.timing_enable(True)
    ctx.timing_reset()
    one_pass()
    ctx.sync()
    fwd_ms, fwd_n = ctx.timing_get(capi.K_FORWARD)
    tr_ms, tr_n = ctx.timing_get(capi.K_TRANSPOSED)
    ew_ms, ew_n = ctx.timing_get(capi.K_ELEMENTWISE)
    ctx.timing_enable(False)
    ll, auc, _ = test.metrics([w])
    nnz_total = job.host_sum([float(train.nnz)])[0]

    # the reference's semantics beside it: the row-by-row pass of one worker on a bounded sample, and of
    # 1024 workers (one wavefront each, the ADMM x-update of many workers resident on one GPU) on the set
    exact = {}
    if rank == 0 and not args.ftrl_pass_only:
        sample = ctx.csr_synth(200_000, dim, args.mean_nnz, seed=args.seed, row_offset=0, **synth_kw(args))
        sw, sz, sn = (ctx.vec(dim) for _ in range(3))
        sample.ftrl_pass(sw, sz, sn, off, reg, *prm, batch=1)      # builds the dependence table
        ctx.sync()
        for v in (sw, sz, sn):
            v.zero()
        ctx.sync()
        t1 = time.perf_counter()
        sample.ftrl_pass(sw, sz, sn, off, reg, *prm, batch=1)
        ctx.sync()
        t1 = time.perf_counter() - t1
        exact["one_worker"] = {"rows": sample.nrows, "nnz": sample.nnz, "seconds": t1, "nnz_per_s": sample.nnz / t1,
                               "full_pass_estimate_s": t1 * train.nnz / sample.nnz}
        sample.destroy()
        workers, per = 1024, max(1, (r1 - r0) // 1024)
        shards = [ctx.csr_synth(per, dim, args.mean_nnz, seed=args.seed, row_offset=k * per, **synth_kw(args))
                  for k in range(workers)]
        st = [[ctx.vec(dim) for _ in shards] for _ in range(3)]
        ctx.ftrl_pass_multi(shards, st[0], st[1], st[2], [off] * workers, [reg] * workers, *prm)
        ctx.sync()
        t2 = time.perf_counter()
        ctx.ftrl_pass_multi(shards, st[0], st[1], st[2], [off] * workers, [reg] * workers, *prm)
        ctx.sync()
        t2 = time.perf_counter() - t2
        tot = sum(s.nnz for s in shards)
        exact["many_workers"] = {"workers": workers, "rows_each": per, "nnz": tot, "seconds": t2, "nnz_per_s": tot / t2}
        for s in shards:
            s.destroy()
        for grp in st:
            for v in grp:
                v.free()
        # what the batched semantics cost in model quality: the same 1M rows, one pass each way
        from admm_amd import driver
        exact["quality"] = driver.ftrl_exact_vs_minibatch(ctx, 1_000_000, dim, args.mean_nnz, args.seed, prm,
                                                          batches=(4096, batch), col_block=args.col_block, **synth_kw(args))

    if rank == 0:
        rows = r1 - r0
        pass_s = dt / args.steps
        # algorithmic bytes of one pass: idx + val per stored entry, rowptr 8 + label 4 + prediction 4 per row,
        # and the dim-sized state read and written once per pass (z, n, reg read; z, n, w written)
        alg = 8.0 * train.nnz + 16.0 * rows + 4.0 * dim * 6
        lay = train.layout_bytes()
        stored = lay["forward_form"] + lay["column_block_copy"] + 12.0 * rows
        npanels = (rows + batch - 1) // batch
        traffic = traffic_src = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic_ftrl.json")
        if os.path.exists(tfile):
            pmc = json.load(open(tfile))
            wl = pmc.get("workload", {})
            if (wl.get("rows_total"), wl.get("dim"), wl.get("n_gpus")) == (rows_total, dim, world) and "pass_total" in pmc:
                traffic, traffic_src = pmc["pass_total"]["traffic_bytes"], "profiles/pmc_traffic_ftrl.json"
        out = {
            "metric": METRIC, "value": nnz_total * args.steps / dt, "unit": "nnz/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": pass_s * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "FTRL-proximal single pass, synthetic %d rows x %d features, ~%d nnz/row, %d GPU(s)"
                            % (rows_total, dim, round(nnz_total / rows_total), world),
                "name": args.config,
                "mode": "mini-batch of %d rows: every row of a batch is predicted from the state at the batch's start, "
                        "then the batch's summed update is applied per coordinate (a generalisation of the "
                        "reference's row-by-row pass, which is batch 1)" % batch,
                "hyper_parameters": dict(zip(("alpha", "beta", "l_1", "l_2"), prm)),
                "batches_per_pass": npanels,
                "test_logloss_after_one_pass": ll, "test_auc_after_one_pass": auc,
                "row_by_row_pass": exact,
                "kernel_ms_per_pass": {"forward": fwd_ms, "transposed": tr_ms, "elementwise": ew_ms,
                                       "launches": fwd_n + tr_n + ew_n,
                                       "from": "one extra pass after the timed region, HIP events around every kernel"},
                "rows_total": rows_total, "dim": dim, "nnz_total": nnz_total, "generate_and_build_s": t_gen,
                "rank0_hbm_bytes": lay, "rank0_layout_counts": train.layout_counts(),
            },
            "roofline": {"bound": "hbm", "kernel": "ftrl_minibatch_pass (forward sweep + transposed sweep + per-coordinate apply, per batch)",
                         "achieved": alg / pass_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / pass_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg,
                         "stored_bytes": stored, "frac_stored": stored / pass_s / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_unit": "bytes per pass", "traffic_source": traffic_src,
                         "frac_counters": (traffic / pass_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "traffic_measured_in_run": False,
                         "note": "the pass is a chain of per-batch launches; achieved = algorithmic bytes of one pass / "
                                 "wall time of one pass between two stream synchronisations (kernel_ms_per_pass: HIP events "
                                 "around every kernel of one extra pass)"},
        }
        if world == 1 and not args.no_cpu_baseline and not side:
            out["cpu_baseline"] = cpu_baseline(ctx, args, 0.01)
    train.destroy()
    test.destroy()
    for v in (w, z, n, off, reg):
        v.free()
    return out


def summary_of(rec):
    """what config.other_configs keeps of a configuration run beside the primary one"""
    rf = rec["roofline"]
    keep = {"name": rec["config"]["name"], "workload": rec["config"]["workload"], "steps": rec["steps"],
            "warmup": rec["warmup"], "ms_per_step": rec["ms_per_step"], "value": rec["value"], "unit": rec["unit"],
            "scaling": rec["scaling"],
            "roofline": {k: rf.get(k) for k in ("kernel", "achieved", "frac", "frac_stored", "traffic", "frac_counters",
                                                "traffic_source", "algorithmic_bytes", "stored_bytes")}}
    if "kernels" in rf:
        keep["roofline"]["kernels"] = {
            n: {k: v.get(k) for k in ("avg_ms", "launches", "frac", "frac_stored", "frac_counters", "traffic_bytes")}
            for n, v in rf["kernels"].items()}
    for k in ("admm_iterations_per_s", "ms_per_step_outside_the_sweeps", "global_test_metrics", "mode", "batches_per_pass",
              "test_logloss_after_one_pass", "test_auc_after_one_pass", "row_by_row_pass", "exact_vs_minibatch",
              "kernel_ms_per_pass", "generate_and_build_s", "rank0_hbm_bytes", "shard_nnz_max_over_mean"):
        if k in rec["config"]:
            keep[k] = rec["config"][k]
    return keep


def main():
    args = parse_args()
    job = Job(args)
    run = run_ftrl if args.config == "ftrl" else run_admm
    out = run(args, job)
    # One GPU, default workload: the other two single-GPU configurations run in this same process after the
    # timed region, at full size, and ride along in the line (the c3 record stays the headline).
    if job.world == 1 and args.config == "c3" and not args.overridden and not args.explicit_values \
            and not args.no_other_configs:
        others = []
        for name, kw in (("ftrl", dict(steps=20, warmup=3)), ("c5", dict(steps=3, warmup=1))):
            t0 = time.perf_counter()
            try:
                a = apply_config(args, name, no_cpu_baseline=True, no_other_linesearch=True, **kw)
                rec = (run_ftrl if name == "ftrl" else run_admm)(a, job, side=True)
                rec = summary_of(rec)
            except Exception as e:   # noqa: BLE001 - a side configuration must not take the headline down with it
                rec = {"name": name, "error": "%s: %s" % (type(e).__name__, e)}
            rec["wall_s"] = time.perf_counter() - t0
            others.append(rec)
        out["config"]["other_configs"] = others
    if job.rank == 0 and out is not None:
        print(json.dumps(out), flush=True)
    job.close()


if __name__ == "__main__":
    main()
