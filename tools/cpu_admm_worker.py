#!/usr/bin/env python3
"""One CPU worker of the reference-shaped ADMM run that bench.py times beside the GPU
(cpu_baseline.multi_process): started N times by tools/dmlc_local.py, one process per host core.

    tools/dmlc_local.py -n N python tools/cpu_admm_worker.py <dir> <dim> resident|reparse <iterations>

Worker r (DMLC_TASK_ID) owns <dir>/<r>_aggregated/part-00000 (libsvm text; written by the worker
itself, before the clock starts, from the arrays bench.py left in <dir>/shard_<r>.npz) and runs the
loop of src/admm_allreduce.cpp:55-91 of the reference with the arithmetic of oracle/admm_oracle.c:
one row-by-row FTRL pass (src/workers.cc:51-87), payload, sum all-reduce, z-update, dual update.
`reparse` parses the text again in every iteration, as the reference does (a new SampleSet per
iteration, src/admm_allreduce.cpp:57-58, src/workers.cc:73-77); `resident` parses it once.
rabit's all-reduce is replaced by a shared-memory file under ADMM_RENDEZVOUS_DIR: every rank writes
its payload into its slot, sums one slice over all slots, and reads the assembled sum.

TEST / MEASUREMENT INFRASTRUCTURE: this is the CPU side-by-side leg, never the product path.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def write_shards(csr, directory, nworkers, rows_per):
    """bench.py's side: cut `nworkers` row ranges out of a bias-translated sample set and leave each as
    raw arrays (without the bias column, indices shifted back).  Returns the stored entries the job
    sweeps per iteration, bias column included (the reference's nnz bookkeeping)."""
    total = 0
    for k in range(nworkers):
        r0 = (k * rows_per) % max(1, csr.nrows - rows_per + 1)
        b, e = int(csr.rowptr[r0]), int(csr.rowptr[r0 + rows_per])
        col = csr.col[b:e].astype(np.int64)
        rowptr = (csr.rowptr[r0:r0 + rows_per + 1] - csr.rowptr[r0]).astype(np.int64)
        keep = col > 0                                            # drop the bias entries
        rows = np.repeat(np.arange(rows_per), np.diff(rowptr))
        newptr = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=rows_per))])
        np.savez(os.path.join(directory, "shard_%d.npz" % k), rowptr=newptr, col=col[keep] - 1,
                 label=csr.label[r0:r0 + rows_per])
        total += int(e - b)
    return total


def _write_text(npz, path):
    d = np.load(npz)
    rowptr, col, label = d["rowptr"], d["col"], d["label"]
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        for r in range(len(label)):
            f.write("%g %s\n" % (label[r], " ".join(["%d:1" % c for c in col[rowptr[r]:rowptr[r + 1]]])))


class ShmSum:
    """sum all-reduce of `dim` floats over the job's processes through one shared file"""

    def __init__(self, path, rank, world, dim):
        self.rank, self.world, self.dim = rank, world, dim
        nbytes = 8 * world + 4 * dim * (world + 1)
        if rank == 0:
            with open(path + ".tmp", "wb") as f:
                f.truncate(nbytes)
            os.replace(path + ".tmp", path)
        else:
            for _ in range(60000):
                if os.path.exists(path) and os.path.getsize(path) == nbytes:
                    break
                time.sleep(0.01)
        self.gen = np.memmap(path, np.int64, "r+", 0, (world,))
        self.slots = np.memmap(path, np.float32, "r+", 8 * world, (world + 1, dim))
        self.count = 0
        step = (dim + world - 1) // world
        self.lo, self.hi = min(dim, rank * step), min(dim, (rank + 1) * step)

    def barrier(self):
        self.count += 1
        self.gen[self.rank] = self.count
        spins = 0
        while int(self.gen.min()) < self.count:
            spins += 1
            time.sleep(0 if spins < 50 else 0.0005)

    def allreduce(self, v):
        self.slots[self.rank] = v
        self.barrier()
        self.slots[self.world, self.lo:self.hi] = self.slots[:self.world, self.lo:self.hi].sum(axis=0, dtype=np.float32)
        self.barrier()
        out = np.array(self.slots[self.world])
        self.barrier()          # nobody overwrites its slot before every rank has read the sum
        return out


def main():
    directory, dim, mode, iters = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    rank = int(os.environ.get("DMLC_TASK_ID", "0"))
    world = int(os.environ.get("DMLC_NUM_WORKER", "1"))
    from oracle import oracle as orc
    text = os.path.join(directory, "%d_aggregated" % rank, "part-00000")
    if not os.path.exists(text):
        _write_text(os.path.join(directory, "shard_%d.npz" % rank), text)
    shm = ShmSum(os.path.join(os.environ.get("ADMM_RENDEZVOUS_DIR", directory), "cpu_allreduce_" + mode), rank, world, dim)
    conf = orc.AdmmConf(step_size=0.01, global_var=4.0, ftrl_alpha=0.1, passes=iters, dim=dim)
    wk, zg = orc.WorkerState(dim), np.zeros(dim, np.float32)

    def load():
        part = orc.translate(orc.parse_libsvm(text))
        part.dim = dim
        return part

    part = load()
    shm.barrier()
    t0 = time.perf_counter()
    for it in range(iters):
        if mode == "reparse":
            part = load()
        wk.load = it == iters - 1
        orc.base_update(wk, [part], conf, zg)
        total = shm.allreduce(orc.payload(wk.x, wk.y, conf.step_size))
        zg = orc.global_update(total, world, conf.step_size, conf.global_var)
        orc.dual_update(wk.y, wk.x, zg, conf.step_size)
    shm.barrier()
    loop_s = time.perf_counter() - t0
    if rank == 0 and os.environ.get("ADMM_CPU_DUMP"):
        np.save(os.environ["ADMM_CPU_DUMP"], zg)
    if rank == 0:
        print("CPU_ADMM " + json.dumps({"iterations": iters, "loop_s": loop_s, "world": world, "mode": mode,
                                        "rows": part.nrows, "nnz": part.nnz}), flush=True)


if __name__ == "__main__":
    main()
