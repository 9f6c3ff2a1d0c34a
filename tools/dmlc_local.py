#!/usr/bin/env python3
"""Local launcher for the admm / ftrl programs of this repository, with the command line and the
process contract of the reference's yarn-cluster/tracker/dmlc_local.py:

    tools/dmlc_local.py -n 3 admm_amd/bin/admm admm.conf train_dir test_dir out_dir

Starts `-n` worker processes on this machine.  Every worker gets DMLC_NUM_WORKER, DMLC_TASK_ID,
DMLC_ROLE=worker, DMLC_NUM_ATTEMPT, DMLC_TRACKER_URI/PORT (the port number only names the job: there
is no tracker process, the ranks meet through files under ADMM_RENDEZVOUS_DIR), and RANK / WORLD_SIZE
/ LOCAL_RANK.  A worker that exits with 254 is started again with DMLC_NUM_ATTEMPT + 1, as the
reference's keep-alive loop does; any other non-zero exit stops the job and becomes the launcher's
exit code.  One worker drives one GPU; with more workers than GPUs (or --allreduce shm) the
all-reduce goes through host shared memory instead of RCCL, which cannot put two ranks on one GPU.
"""
import argparse
import logging
import os
import shutil
import subprocess
import sys
import tempfile
import time


def visible_gpus():
    env = os.environ.get("ADMM_NUM_GPUS")
    if env:
        return max(1, int(env))
    try:
        return max(1, len([d for d in os.listdir("/sys/class/kfd/kfd/topology/nodes")
                           if _is_gpu_node(d)]))
    except OSError:
        return 1


def _is_gpu_node(node):
    try:
        with open("/sys/class/kfd/kfd/topology/nodes/%s/properties" % node) as f:
            for line in f:
                if line.startswith("simd_count"):
                    return int(line.split()[1]) > 0
    except OSError:
        pass
    return False


def main():
    ap = argparse.ArgumentParser(description="start the workers of a job as local processes")
    ap.add_argument("-n", "--nworker", required=True, type=int, help="number of worker processes")
    ap.add_argument("-s", "--server-nodes", default=0, type=int, help="accepted and ignored (no servers)")
    ap.add_argument("--log-level", default="INFO", choices=["INFO", "DEBUG"])
    ap.add_argument("--log-file", type=str)
    ap.add_argument("--allreduce", default="auto", choices=["auto", "rccl", "shm"],
                    help="auto: RCCL when every worker has a GPU of its own, else host shared memory")
    ap.add_argument("command", nargs=argparse.REMAINDER, help="program and its arguments")
    args = ap.parse_args()
    if not args.command:
        ap.error("no command given")
    logging.basicConfig(level=getattr(logging, args.log_level), filename=args.log_file,
                        format="%(asctime)s %(levelname)s %(message)s")
    cmd = list(args.command)
    if "/" not in cmd[0] and os.path.exists(cmd[0]):
        cmd[0] = "./" + cmd[0]

    ngpu = visible_gpus()
    mode = args.allreduce
    if mode == "auto":
        mode = "rccl" if args.nworker <= ngpu else "shm"
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    rdv = tempfile.mkdtemp(prefix="admm_job_", dir=base)
    job = str(os.getpid())
    logging.info("%d workers, %d GPU(s), all-reduce over %s, rendezvous in %s", args.nworker, ngpu, mode, rdv)

    def start(rank, attempt):
        env = os.environ.copy()
        env.update({
            "DMLC_NUM_WORKER": str(args.nworker), "DMLC_NUM_SERVER": str(args.server_nodes),
            "DMLC_TASK_ID": str(rank), "DMLC_ROLE": "worker", "DMLC_NUM_ATTEMPT": str(attempt),
            "DMLC_TRACKER_URI": "127.0.0.1", "DMLC_TRACKER_PORT": job,
            "RANK": str(rank), "WORLD_SIZE": str(args.nworker), "LOCAL_RANK": str(rank % ngpu),
            "ADMM_RENDEZVOUS_DIR": rdv, "ADMM_ALLREDUCE": mode,
            "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
        })
        return subprocess.Popen(cmd, env=env)

    procs = {r: start(r, 0) for r in range(args.nworker)}
    attempts = {r: 0 for r in range(args.nworker)}
    rc = 0
    try:
        while procs:
            time.sleep(0.05)
            for r, p in list(procs.items()):
                ret = p.poll()
                if ret is None:
                    continue
                if ret == 254:
                    attempts[r] += 1
                    logging.info("worker %d asked for a restart (attempt %d)", r, attempts[r])
                    procs[r] = start(r, attempts[r])
                elif ret == 0:
                    logging.debug("worker %d finished", r)
                    del procs[r]
                else:
                    logging.error("worker %d exited with %d", r, ret)
                    rc = ret if ret > 0 else 1
                    del procs[r]
                    for q in procs.values():
                        q.terminate()
                    for q in procs.values():
                        try:
                            q.wait(timeout=10)
                        except subprocess.TimeoutExpired:
                            q.kill()
                    procs = {}
                    break
    except KeyboardInterrupt:
        for q in procs.values():
            q.terminate()
        rc = 130
    finally:
        shutil.rmtree(rdv, ignore_errors=True)
    return rc


if __name__ == "__main__":
    sys.exit(main())
