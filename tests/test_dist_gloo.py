"""The N>1 path on CPU: two processes under gloo run the reference's iteration order
(local update -> payload -> sum all-reduce -> consensus -> dual update) with the product's
distributed helpers carrying the exchange; the per-rank numerics come from the CPU restatement.
The result must equal the single-process run of both ranks."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import os, sys, pickle
import numpy as np, torch
sys.path.insert(0, os.environ["ADMM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["ADMM_ROOT"], "tests"))
from admm_amd import dist as adist
from oracle import oracle as orc
from conftest import make_synth_csr
rank, world, _ = adist.init("gloo")
assert world == 2
dim, total_rows = 97, 1001
b, e = adist.shard_range(total_rows, rank, world)
raw = make_synth_csr(e - b, dim, 8, seed=200 + rank)
tr = orc.translate(raw)
conf = orc.AdmmConf(step_size=0.05, global_var=2.0, ftrl_alpha=0.1, passes=3, dim=dim)
wk, zg = orc.WorkerState(dim), np.zeros(dim, np.float32)
hist = []
for it in range(conf.passes):
    wk.load = it == conf.passes - 1
    orc.base_update(wk, [tr], conf, zg)
    payload = torch.from_numpy(orc.payload(wk.x, wk.y, conf.step_size))
    adist.allreduce_sum_(payload)                      # the one collective of the iteration
    total = payload.numpy()
    zg = orc.global_update(total, world, conf.step_size, conf.global_var)
    orc.dual_update(wk.y, wk.x, zg, conf.step_size)
    hist.append((total.copy(), zg.copy(), wk.x.copy(), wk.y.copy()))
# the exchange behind the all-ranks test metrics: every rank fills its slice of a zero buffer
mine = np.arange(3 + 2 * rank, dtype=np.float32) + 100 * (rank + 1)
off, tot = adist.slice_of_rank(mine.size)
buf = torch.zeros(tot, dtype=torch.float32)
buf[off:off + mine.size] = torch.from_numpy(mine)
adist.gather_slices_(buf, off, mine.size)
hist.append((off, tot, buf.numpy().copy()))
adist.barrier()
with open(os.environ["ADMM_OUT"] + ".%d" % rank, "wb") as f:
    pickle.dump(hist, f)
adist.finalize()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_gloo(tmp_path, oracle):
    import pickle
    from conftest import make_synth_csr
    from admm_amd import dist as adist
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ADMM_ROOT=ROOT, ADMM_OUT=str(tmp_path / "hist"), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out
    hists = [pickle.load(open(str(tmp_path / "hist") + ".%d" % r, "rb")) for r in range(2)]
    # slices gathered over ranks: rank order, every rank sees the whole
    want = np.concatenate([np.arange(3, dtype=np.float32) + 100, np.arange(5, dtype=np.float32) + 200])
    for r in range(2):
        off, tot, got = hists[r].pop()
        assert (off, tot) == ((0, 8) if r == 0 else (3, 8))
        np.testing.assert_array_equal(got, want)
    # single-process run of the same two ranks
    dim, total_rows = 97, 1001
    parts = []
    for r in range(2):
        b, e = adist.shard_range(total_rows, r, 2)
        parts.append([oracle.translate(make_synth_csr(e - b, dim, 8, seed=200 + r))])
    conf = oracle.AdmmConf(step_size=0.05, global_var=2.0, ftrl_alpha=0.1, passes=3, dim=dim)
    _, _, rec = oracle.admm_run(parts, [p[0] for p in parts], conf)
    for it in range(3):
        for r in range(2):
            total, zg, x, y = hists[r][it]
            np.testing.assert_allclose(total, rec[it]["total"], rtol=1e-6, atol=1e-9)
            np.testing.assert_allclose(zg, rec[it]["z"], rtol=1e-6, atol=1e-9)
            np.testing.assert_array_equal(x, rec[it]["x"][r])
            np.testing.assert_allclose(y, rec[it]["y"][r], rtol=1e-6, atol=1e-9)
        # every rank holds the same consensus vector
        np.testing.assert_array_equal(hists[0][it][1], hists[1][it][1])
