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


def lib_comm_init(ctx):
    """The communicator of libadmm_hip.so over all ranks (the product's all-reduce path: RCCL over
    xGMI behind admm_allreduce_sum_f32).  Rank 0 draws the 128-byte id; the other ranks receive it
    through the torch.distributed process group when one is up (host data, any backend), else through a
    file under ADMM_RENDEZVOUS_DIR keyed by MASTER_PORT / DMLC_TRACKER_PORT, as the C++ programs do."""
    rank, world, _ = env_rank_world()
    if world == 1:
        ctx.comm_init(0, 1)
        return rank, world
    uid = ctx.comm_unique_id() if rank == 0 else None
    if dist.is_initialized():
        box = [uid]
        dist.broadcast_object_list(box, src=0, device=torch.device("cpu"))
        uid = box[0]
    else:
        import time
        job = os.environ.get("MASTER_PORT", os.environ.get("DMLC_TRACKER_PORT", "local"))
        path = os.path.join(os.environ.get("ADMM_RENDEZVOUS_DIR", "/tmp"), "admm_uid_py_%s" % job)
        if rank == 0:
            with open(path + ".tmp", "wb") as f:
                f.write(uid)
            os.replace(path + ".tmp", path)
        else:
            for _ in range(6000):
                if os.path.exists(path) and os.path.getsize(path) == 128:
                    break
                time.sleep(0.1)
            with open(path, "rb") as f:
                uid = f.read()
    ctx.comm_init(rank, world, uid)
    return rank, world


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
