"""Host logic of the multi-process path that needs no GPU: the file rendezvous of the library
communicator's id (used when no torch.distributed group is up) and the entry-balanced row shards."""
import os
import subprocess
import sys
import time

import numpy as np

from conftest import ROOT

CHILD = r'''
import os, sys
sys.path.insert(0, os.environ["ADMM_ROOT"])
from admm_amd import dist as adist
rank = int(os.environ["RANK"])
uid = bytes([int(os.environ["UID_BYTE"])]) * 128 if rank == 0 else None
got, path = adist.exchange_uid_by_file(rank, uid, timeout_s=float(os.environ.get("T", "30")))
# (the communicator's collective bring-up would sit here; rank 0 removes the file after it)
sys.stdout.write("%d %d %s\n" % (rank, got[0], path))
'''


def launch(tmp, port, byte, ranks=(0, 1), delay_rank0=0.0, timeout="30"):
    """the ranks of one launch: children of one parent process (this one), same port"""
    procs = {}
    for r in ranks:
        env = dict(os.environ, ADMM_ROOT=ROOT, RANK=str(r), WORLD_SIZE="2", MASTER_PORT=str(port),
                   ADMM_RENDEZVOUS_DIR=str(tmp), UID_BYTE=str(byte), T=timeout)
        env.pop("TORCHELASTIC_RUN_ID", None)
        if r == 0 and delay_rank0:
            time.sleep(delay_rank0)
        procs[r] = subprocess.Popen([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE,
                                    stderr=subprocess.PIPE, text=True)
    return {r: p.communicate(timeout=60) + (p.returncode,) for r, p in procs.items()}


def test_uid_file_twice_on_one_port_and_a_stale_file(tmp_path):
    from admm_amd import dist as adist
    port = 29500                                    # torchrun's constant default
    # a crashed earlier job on the same port left its id file behind: another launch (another parent) ...
    stale = tmp_path / ("admm_uid_py_%d_ppid%d" % (port, 1))
    stale.write_bytes(b"\x07" * 128)
    # ... and even a file with THIS launch's name, if it is old enough, is not read
    os.environ.pop("TORCHELASTIC_RUN_ID", None)
    for byte in (11, 12):                           # two jobs in a row on the same port
        out = launch(tmp_path, port, byte, ranks=(1, 0), delay_rank0=0.3)
        for r in (0, 1):
            so, se, rc = out[r]
            assert rc == 0, se
            assert int(so.split()[1]) == byte, (so, se)
        path = out[0][0].split()[2]
        assert "ppid%d" % os.getpid() in path
        # rank 0 removes the file when the communicator is up; here the test stands in for that
        os.unlink(path)
    # an old file under the launch's own name: ignored by age, the rank gives up with a clear error
    mine = tmp_path / ("admm_uid_py_%d_ppid%d" % (port, os.getpid()))
    mine.write_bytes(b"\x09" * 128)
    old = time.time() - 3600
    os.utime(mine, (old, old))
    out = launch(tmp_path, port, 13, ranks=(1,), timeout="1.0")
    so, se, rc = out[1]
    assert rc != 0 and "TimeoutError" in se and "no communicator id" in se
    assert adist.uid_file_path  # (exported for launchers that want to clean up)


def heavy_tailed_lengths(n, seed, sigma=1.2, mean=40.0, cap=1024):
    """row lengths as the benchmark's heavy-tailed generator draws them: log-normal, capped, plus the bias entry"""
    rng = np.random.default_rng(seed)
    mu = np.log(mean) - 0.5 * sigma * sigma
    ln = np.floor(np.exp(mu + sigma * rng.standard_normal(n)) + 0.5).astype(np.int64)
    return np.clip(ln, 0, cap) + 1


def test_shards_balanced_by_entries():
    from admm_amd import dist as adist
    n = 400_000
    ln = heavy_tailed_lengths(n, 5)
    # make it worse than iid: the long rows come first, as a file sorted by session length would have them
    ln = np.sort(ln)[::-1]
    rp = np.concatenate([[0], np.cumsum(ln)]).astype(np.uint64)
    for world in (2, 3, 8):
        cuts = [adist.shard_range_by_nnz(rp, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        for (b0, e0), (b1, e1) in zip(cuts, cuts[1:]):
            assert e0 == b1 and b0 <= e0
        nnz = np.array([int(rp[e]) - int(rp[b]) for b, e in cuts], np.float64)
        assert nnz.max() / nnz.mean() <= 1.01, (world, nnz)
        # equal row counts on the same data are far off
        rows = [adist.shard_range(n, r, world) for r in range(world)]
        nnz_rows = np.array([int(rp[e]) - int(rp[b]) for b, e in rows], np.float64)
        assert nnz_rows.max() / nnz_rows.mean() > 1.3
    # degenerate inputs: one rank, empty set, more ranks than rows
    assert adist.shard_range_by_nnz(rp, 0, 1) == (0, n)
    assert adist.shard_range_by_nnz(np.zeros(1, np.uint64), 0, 4) == (0, 0)
    tiny = np.array([0, 5, 6, 7], np.uint64)
    got = [adist.shard_range_by_nnz(tiny, r, 8) for r in range(8)]
    assert got[0][0] == 0 and got[-1][1] == 3 and all(b <= e for b, e in got)
    assert sum(e - b for b, e in got) == 3
