"""bench.py's multi-rank control flow on ONE GPU: two ranks under torch.distributed.run share GPU 0
(ADMM_BENCH_SHARE_GPU=1; the payload is then summed on the host through gloo, because RCCL cannot
place two ranks on one GPU).  Checks the rank logic -- row shards, barriers, max-over-ranks timing,
the residual tail, one JSON line from rank 0 -- not the collective."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_bench(extra, ranks=2):
    env = dict(os.environ, ADMM_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(ranks), "--no-cpu-baseline"] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 alone reports
    return json.loads(lines[0])


def test_two_ranks_share_the_rows_of_the_admm_workload():
    rows = 600_000
    one = run_bench(["--rows", str(rows), "--steps", "2", "--warmup", "1", "--no-other-linesearch"], ranks=1)
    two = run_bench(["--rows", str(rows), "--steps", "2", "--warmup", "1", "--no-other-linesearch"])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["rows_total"] == rows == one["config"]["rows_total"]
    # the same rows whatever the number of ranks: the generator is keyed by the global row number
    assert two["config"]["nnz_total"] == one["config"]["nnz_total"]
    assert two["config"]["allreduce"].startswith("REHEARSAL")
    assert two["value"] > 0 and two["ms_per_step"] > 0
    res = two["config"]["last_residuals_all_ranks"]
    assert res["primal_r"] > 0 and res["iterations_recorded"] >= 2
    # both ranks' test rows enter the metrics of all ranks
    assert two["config"]["global_test_metrics"]["rows"] == one["config"]["global_test_metrics"]["rows"]


def test_two_ranks_run_independent_minibatch_passes():
    two = run_bench(["--config", "ftrl", "--rows", "400000", "--steps", "2", "--warmup", "1"])
    assert two["n_gpus"] == 2 and two["value"] > 0


def test_two_ranks_cut_the_heavy_tailed_rows_by_entries():
    """the c5 shape (heavy-tailed row lengths, rows per rank fixed): the ranks' row ranges are cut where the
    generator's draws balance"""
    two = run_bench(["--config", "c5", "--rows", "400000", "--steps", "1", "--warmup", "1", "--no-other-linesearch"])
    cfg = two["config"]
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and cfg["rows_total"] == 800000
    assert "shard_range_by_nnz" in cfg["sharding"]
    assert cfg["shard_nnz_max_over_mean"] <= 1.01


def test_generator_draws_per_block_bound_the_stored_entries():
    from admm_amd import capi
    ctx = capi.Context(0)
    kw = dict(len_dist=1, tail_sigma=1.2, max_len=1024)
    rows, dim, block = 200_000, 1 << 20, 8192
    draws = ctx.synth_row_draws(rows, dim, 40.0, block, seed=9, **kw)
    s = ctx.csr_synth(rows, dim, 40.0, seed=9, **kw)
    rowptr = s.download()[0].astype(np.int64)
    stored = np.array([rowptr[min(rows, (b + 1) * block)] - rowptr[b * block] for b in range(len(draws))])
    assert len(draws) == (rows + block - 1) // block
    assert np.all(draws.astype(np.int64) >= stored)                     # a row stores its draws minus its repeats
    assert np.all(stored >= 0.75 * draws.astype(np.int64))
    # a shard of the same rows generated on its own (row_offset) draws the same
    part = ctx.synth_row_draws(block * 3, dim, 40.0, block, seed=9, row_offset=block * 5, **kw)
    np.testing.assert_array_equal(part, draws[5:8])
    s.destroy()
    ctx.close()
