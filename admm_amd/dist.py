"""One process per GPU: rank bookkeeping, row sharding and the per-iteration sum all-reduce
(rabit::Allreduce<op::Sum> at src/admm_allreduce.cpp:69 of the reference) on torch.distributed --
RCCL over xGMI for device tensors (backend "nccl"), gloo for host tensors in CPU tests."""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    """(rank, world, local_rank) from torch.distributed.run's environment, else the reference
    launcher's (DMLC_TASK_ID / DMLC_NUM_WORKER of tracker.py / dmlc_local.py), else (0, 1, 0)."""
    rank = int(os.environ.get("RANK", os.environ.get("DMLC_TASK_ID", "0")))
    world = int(os.environ.get("WORLD_SIZE", os.environ.get("DMLC_NUM_WORKER", "1")))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    return rank, world, local


def init(backend=None, device=None):
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:  # (a mixed "cpu:gloo,cuda:nccl" group takes no device id)
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def _launch_id():
    """what tells this launch from an earlier one on the same port: the launcher's own id when it gives one,
    else the parent process (the ranks of one launch are children of one launcher process)"""
    for k in ("ADMM_JOB_ID", "TORCHELASTIC_RUN_ID"):
        v = os.environ.get(k)
        if v and v != "none":
            return "".join(ch if ch.isalnum() else "_" for ch in v)[:64] + "_%d" % os.getppid()
    return "ppid%d" % os.getppid()


def uid_file_path():
    job = os.environ.get("MASTER_PORT", os.environ.get("DMLC_TRACKER_PORT", "local"))
    return os.path.join(os.environ.get("ADMM_RENDEZVOUS_DIR", "/tmp"), "admm_uid_py_%s_%s" % (job, _launch_id()))


def exchange_uid_by_file(rank, uid, timeout_s=600.0, max_age_s=900.0):
    """Rank 0 publishes its 128-byte communicator id in a file named after the job's port AND this launch;
    the other ranks wait for it.  A file older than max_age_s is a crashed launch's leftover and is not
    read.  Returns (uid, path); rank 0 removes the file once the communicator is up (lib_comm_init)."""
    import time
    path = uid_file_path()
    if rank == 0:
        assert uid is not None and len(uid) == 128
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(path + ".tmp", path)
        return uid, path
    deadline = time.time() + timeout_s
    while True:
        try:
            st = os.stat(path)
            if st.st_size == 128 and time.time() - st.st_mtime <= max_age_s:
                with open(path, "rb") as f:
                    data = f.read()
                if len(data) == 128:
                    return data, path
        except OSError:
            pass
        if time.time() > deadline:
            raise TimeoutError("rank %d: no communicator id from rank 0 in %s within %.0f s"
                               % (rank, path, timeout_s))
        time.sleep(0.05)


def lib_comm_init(ctx):
    """The communicator of libadmm_hip.so over all ranks (the product's all-reduce path: RCCL over
    xGMI behind admm_allreduce_sum_f32).  Rank 0 draws the 128-byte id; the other ranks receive it
    through the torch.distributed process group when one is up (host data, any backend), else through a
    file under ADMM_RENDEZVOUS_DIR keyed by MASTER_PORT / DMLC_TRACKER_PORT and by the launch, which rank 0
    removes when the communicator is up (every rank has read it by then: the bring-up is collective)."""
    rank, world, _ = env_rank_world()
    if world == 1:
        ctx.comm_init(0, 1)
        return rank, world
    uid = ctx.comm_unique_id() if rank == 0 else None
    path = None
    if dist.is_initialized():
        box = [uid]
        dist.broadcast_object_list(box, src=0, device=torch.device("cpu"))
        uid = box[0]
    else:
        uid, path = exchange_uid_by_file(rank, uid, float(os.environ.get("ADMM_RENDEZVOUS_TIMEOUT", "600")))
    try:
        ctx.comm_init(rank, world, uid)
    finally:
        if path and rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass
    return rank, world


def shard_range_by_nnz(rowptr, rank, world):
    """Rows [begin, end) of rank `rank` when the shards are to hold equal numbers of stored entries rather
    than equal numbers of rows (heavy-tailed row lengths): the cut before rank k is the first row whose
    prefix of entries reaches k / world of the total.  rowptr: the n+1 prefix sums of the row lengths
    (a CSR row pointer, or np.cumsum of the lengths with a leading 0).  Contiguous, covers every row once,
    monotone in rank."""
    import numpy as np
    rp = np.asarray(rowptr)
    n = len(rp) - 1
    base, total = int(rp[0]), int(rp[-1]) - int(rp[0])

    def cut(k):
        if k <= 0:
            return 0
        if k >= world:
            return n
        target = base + (total * k + world // 2) // world
        # the row boundary nearest to the target prefix
        i = min(n, int(np.searchsorted(rp, target, side="left")))    # first boundary at or past the target
        if i > 0 and target - int(rp[i - 1]) < int(rp[i]) - target:
            i -= 1
        return i

    b, e = cut(rank), cut(rank + 1)
    return b, max(b, e)


def shard_range(total_rows, rank, world):
    """Rows [begin, end) of rank `rank`: contiguous, sizes differ by at most one."""
    return total_rows * rank // world, total_rows * (rank + 1) // world


def allreduce_sum_(t):
    """In-place sum over ranks; a no-op for a single process (as rabit without a tracker)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def gather_slices_(buf, offset, count):
    """All ranks' slices of `buf` together on every rank: each rank has filled buf[offset:offset+count]
    (its own rows) and left the rest zero, so a sum over ranks is the concatenation.  Works with any
    backend that has a sum all-reduce (the same exchange serves the C++ programs through
    admm_allreduce_sum_f32).  Returns buf."""
    assert 0 <= offset and offset + count <= buf.numel()
    return allreduce_sum_(buf)


def slice_of_rank(count):
    """(offset, total) of this rank's `count` rows in the concatenation over ranks, rank order."""
    rank, world, _ = env_rank_world()
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return 0, int(count)
    counts = torch.zeros(world, dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    counts[rank] = int(count)
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    counts = counts.tolist()
    return int(sum(counts[:rank])), int(sum(counts))


def allreduce_max_(t):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def barrier():
    """all ranks meet; on a group with a gloo half the meeting is a sum of one host float (dist.barrier
    would pick the device backend and build an RCCL communicator just for that)"""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if "gloo" in str(dist.get_backend()):
            dist.all_reduce(torch.zeros(1))
        else:
            dist.barrier()


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()
