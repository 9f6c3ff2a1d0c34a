"""The C++ host programs (admm, ftrl, evan) end to end on small libsvm directories, against the
CPU restatement of the reference's main loops: report lines, and the state files they write."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, make_synth_csr, write_libsvm

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "admm_amd", "bin")


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=110, env=e)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout


def _read_sparse_sections(path):
    sections, cur = [], None
    for line in open(path):
        line = line.strip()
        if not line:
            continue
        if ":" not in line:
            cur = np.zeros(int(line), np.float32)
            sections.append(cur)
        else:
            i, v = line.split(":")
            cur[int(i)] = np.float32(v)
    return sections


def test_admm_program_matches_reference_loop(tmp_path, oracle):
    dim = 300
    raw_train = make_synth_csr(1500, dim, 12, seed=101, binary=False)
    raw_test = make_synth_csr(400, dim, 12, seed=102, binary=False)
    write_libsvm(str(tmp_path / "train" / "A_aggregated" / "part-00000"), raw_train)
    write_libsvm(str(tmp_path / "test" / "A_aggregated" / "part-00000"), raw_test)
    os.makedirs(tmp_path / "out")
    conf = tmp_path / "admm.conf"
    conf.write_text("step_size = 0.01\nbias_var = 0.01\nglobal_var = 4\nftrl_alpha = 0.1\npasses = 3\n"
                    "dim = %d\nA : 2\n" % dim)
    out = _run([os.path.join(BIN, "admm"), str(conf), str(tmp_path / "train"), str(tmp_path / "test"),
                str(tmp_path / "out")])
    # the reference: 2 data parts of the same file, split by byte range
    text = open(tmp_path / "train" / "A_aggregated" / "part-00000", "rb").read()
    parts = [oracle.translate(oracle.parse_libsvm_text(text, p, 2)) for p in range(2)]
    for p in parts:
        p.dim = dim
    test = oracle.translate(oracle.parse_libsvm(str(tmp_path / "test" / "A_aggregated" / "part-00000")))
    cfg = oracle.AdmmConf(step_size=0.01, global_var=4, ftrl_alpha=0.1, passes=3, dim=dim, bias_var=0.01)
    workers, zg, rec = oracle.admm_run([parts], [test], cfg)
    lines = re.findall(r"^A \| ([0-9.]+) \| ([0-9.]+) \| (\d+)$", out, flags=re.M)
    assert len(lines) == 3
    for (l, a, n), r in zip(lines, rec):
        assert int(n) == 400
        assert float(l) == pytest.approx(r["lines"][0][0], rel=2e-5)
        assert float(a) == pytest.approx(r["lines"][0][1], rel=5e-5)
    assert [int(x) for x in re.findall(r"^Finish (\d+)-th iteration$", out, flags=re.M)] == [0, 1, 2]
    # state files (6 significant digits in the text)
    x, bias, y, nt, zt = _read_sparse_sections(tmp_path / "out" / "admm_weight_A")
    for got, ref in ((x, workers[0].x), (y, workers[0].y), (nt, workers[0].nt), (zt, workers[0].zt)):
        assert got.shape[0] == dim
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-6 * max(1.0, np.abs(ref).max()))
    assert not bias.any()
    g = open(tmp_path / "out" / "global_params_A").read().split("\n")
    assert [float(v) for v in g[:4]] == [0.01, 4.0, 0.01, float(dim)]
    zfile = np.zeros(dim, np.float32)
    for line in g[4:]:
        if line:
            i, v = line.split(":")
            zfile[int(i)] = np.float32(v)
    np.testing.assert_allclose(zfile, zg, rtol=2e-5, atol=1e-9)
    b = open(tmp_path / "out" / "A.bin").read().split("\n")
    assert b[0].startswith("-1:")
    thr = np.exp(np.float32(-6.0))
    expect = [i - 1 for i in range(1, dim) if abs(workers[0].x[i]) > thr]
    assert [int(l.split(":")[0]) for l in b[1:] if l] == expect
    # warm start: a second run that loads the written state continues from it
    out2 = _run([os.path.join(BIN, "admm"), str(conf), str(tmp_path / "train"), str(tmp_path / "test"),
                 str(tmp_path / "out"), str(tmp_path / "out")])
    wk = [oracle.WorkerState(dim)]
    wk[0].x, wk[0].y, wk[0].nt, wk[0].zt = x.copy(), y.copy(), nt.copy(), zt.copy()
    _, _, rec2 = oracle.admm_run([parts], [test], cfg, workers=wk, zg=zfile.copy())
    lines2 = re.findall(r"^A \| ([0-9.]+) \| ([0-9.]+) \| (\d+)$", out2, flags=re.M)
    for (l, a, n), r in zip(lines2, rec2):
        assert float(l) == pytest.approx(r["lines"][0][0], rel=5e-5)


def test_lbfgs_solver_takes_a_partitioned_file_as_one_set(tmp_path):
    """`psid : 3` with the lbfgs local solver: the batch objective is a sum over all rows, so the
    three parts of the file are one resident set and the run equals the one with `psid : 1`"""
    dim = 300
    write_libsvm(str(tmp_path / "train" / "A_aggregated" / "part-00000"), make_synth_csr(1200, dim, 12, seed=111))
    write_libsvm(str(tmp_path / "test" / "A_aggregated" / "part-00000"), make_synth_csr(300, dim, 12, seed=112))
    outs = []
    for nparts in (1, 3):
        os.makedirs(tmp_path / ("out%d" % nparts))
        conf = tmp_path / ("admm%d.conf" % nparts)
        conf.write_text("step_size = 0.05\nbias_var = 0.01\nglobal_var = 4\nftrl_alpha = 0.1\npasses = 3\n"
                        "local_solver = lbfgs\ndim = %d\nA : %d\n" % (dim, nparts))
        out = _run([os.path.join(BIN, "admm"), str(conf), str(tmp_path / "train"), str(tmp_path / "test"),
                    str(tmp_path / ("out%d" % nparts))])
        lines = re.findall(r"^A \| ([0-9.]+) \| ([0-9.]+) \| (\d+)$", out, flags=re.M)
        assert len(lines) == 3 and lines[0] != lines[2]
        outs.append((lines, open(tmp_path / ("out%d" % nparts) / "admm_weight_A").read()))
    assert outs[0] == outs[1]


def test_admm_program_on_wtest_fixture(tmp_path):
    """the program on the reference's own one-row fixture (test/wtest.libsvm), one rank"""
    for d in ("train", "test"):
        os.makedirs(tmp_path / d / "A_aggregated")
        with open(tmp_path / d / "A_aggregated" / "part-00000", "w") as f:
            f.write(open(os.path.join(GOLDEN, "wtest.libsvm")).read())
    os.makedirs(tmp_path / "out")
    conf = tmp_path / "admm.conf"
    conf.write_text("step_size = 0.01\nbias_var = 0.01\nglobal_var = 4\nftrl_alpha = 0.1\npasses = 3\ndim = 6\n"
                    "train_path = %s\ntest_path = %s\noutput_path = %s\nA : 1\n"
                    % (tmp_path / "train", tmp_path / "test", tmp_path / "out"))
    out = _run([os.path.join(BIN, "admm"), str(conf)])          # paths from the conf: `./admm admm.conf`
    assert out.count("A | 0.6931472 | ") == 3
    x, bias, y, nt, zt = _read_sparse_sections(tmp_path / "out" / "admm_weight_A")
    assert not x.any() and not y.any()
    assert nt[[0, 1, 3, 5]].tolist() == [0.25] * 4 and zt[[0, 1, 3, 5]].tolist() == [-0.5] * 4
    assert open(tmp_path / "out" / "A.bin").read() == "-1:0\n"


def test_ftrl_and_evan_programs(tmp_path, oracle):
    dim = 200
    raw_train = make_synth_csr(1200, dim, 10, seed=111)
    raw_test = make_synth_csr(300, dim, 10, seed=112)
    write_libsvm(str(tmp_path / "train" / "B_aggregated" / "part-00000"), raw_train)
    write_libsvm(str(tmp_path / "test" / "B_aggregated" / "part-00000"), raw_test)
    os.makedirs(tmp_path / "out")
    conf = tmp_path / "ftrl.conf"
    conf.write_text("train_path = %s/\ntest_path = %s/\noutput_path = %s/\nftrl_alpha = 0.01\nftrl_beta = 1\n"
                    "l_1 = 0.4\nl_2 = 2.5\npasses = 2\ndim = %d\nB : 1\n"
                    % (tmp_path / "train", tmp_path / "test", tmp_path / "out", dim))
    out = _run([os.path.join(BIN, "ftrl"), str(conf)])
    tr, te = oracle.translate(raw_train), oracle.translate(raw_test)
    w, z, n, zero = (np.zeros(dim, np.float32) for _ in range(4))
    expect = []
    for _ in range(2):
        oracle.ftrl_run(tr, w, z, n, zero, zero, 0.01, 1.0, 0.4, 2.5)
        expect.append((oracle.logloss(te, [w]), oracle.auc(te, [w])))
    lines = re.findall(r"^B \| ([0-9.]+) \| ([0-9.]+) \| (\d+)$", out, flags=re.M)
    assert len(lines) == 2
    for (l, a, nrow), (el, ea) in zip(lines, expect):
        assert int(nrow) == 300
        assert float(l) == pytest.approx(el, rel=2e-5) and float(a) == pytest.approx(ea, rel=5e-5)
    state = np.loadtxt(tmp_path / "out" / "ftrl_weight_B", dtype=np.float32)
    assert state.shape == (dim, 3)
    for k, ref in enumerate((w, z, n)):
        np.testing.assert_allclose(state[:, k], ref, rtol=2e-5, atol=1e-6 * max(1.0, np.abs(ref).max()))
    first = open(tmp_path / "out" / "ftrl_weight_B").readline()
    assert len(first.rstrip("\n")) == 51                           # three left-aligned 17-wide columns
    b = open(tmp_path / "out" / "B.bin").read().split("\n")
    assert b[0].startswith("1:") and len([l for l in b if l]) == dim
    # evan on the dense weight file
    wfile = tmp_path / "w.txt"
    wfile.write_text(" ".join("%.9g" % v for v in w))
    out = _run([os.path.join(BIN, "evan"), str(wfile), str(tmp_path / "train") + "/", str(tmp_path / "test") + "/",
                "B_aggregated/part-00000"])
    m = re.search(r"^test \| ([0-9.]+) \| ([0-9.]+) \| (\d+)$", out, flags=re.M)
    assert m and float(m.group(1)) == pytest.approx(expect[-1][0], rel=2e-5)
    assert float(m.group(2)) == pytest.approx(expect[-1][1], rel=5e-5)


def test_three_workers_under_the_local_launcher(tmp_path, oracle):
    """`dmlc_local.py -n 3 ./admm admm.conf ...`: three ranks (sharing the one GPU of the test box, so
    the all-reduce goes through host shared memory), one psid each, against the reference loop"""
    dim = 200
    names = ["A", "B", "C"]
    for k, name in enumerate(names):
        write_libsvm(str(tmp_path / "train" / (name + "_aggregated") / "part-00000"),
                     make_synth_csr(700 + 50 * k, dim, 10, seed=201 + k, binary=(k != 1)))
        write_libsvm(str(tmp_path / "test" / (name + "_aggregated") / "part-00000"),
                     make_synth_csr(200, dim, 10, seed=211 + k))
    os.makedirs(tmp_path / "out")
    conf = tmp_path / "admm.conf"
    conf.write_text("step_size = 0.01\nbias_var = 0.01\nglobal_var = 4\nftrl_alpha = 0.1\npasses = 3\n"
                    "global_metrics = 1\ndim = %d\nA : 1\nB : 1\nC : 1\n" % dim)
    out = _run([os.path.join(ROOT, "tools", "dmlc_local.py"), "-n", "3", "--allreduce", "shm",
                os.path.join(BIN, "admm"), str(conf), str(tmp_path / "train"), str(tmp_path / "test"),
                str(tmp_path / "out")])
    parts = [[oracle.translate(oracle.parse_libsvm(str(tmp_path / "train" / (n + "_aggregated") / "part-00000")))]
             for n in names]
    tests = [oracle.translate(oracle.parse_libsvm(str(tmp_path / "test" / (n + "_aggregated") / "part-00000")))
             for n in names]
    for p in parts:
        p[0].dim = dim
    cfg = oracle.AdmmConf(step_size=0.01, global_var=4, ftrl_alpha=0.1, passes=3, dim=dim, bias_var=0.01)
    workers, zg, rec = oracle.admm_run(parts, tests, cfg)
    for k, name in enumerate(names):
        lines = re.findall(r"^%s \| ([0-9.]+) \| ([0-9.]+) \| (\d+)$" % name, out, flags=re.M)
        assert len(lines) == 3
        for (l, a, n), r in zip(lines, rec):
            assert int(n) == 200
            assert float(l) == pytest.approx(r["lines"][k][0], rel=2e-5)
            assert float(a) == pytest.approx(r["lines"][k][1], rel=5e-5)
        x = _read_sparse_sections(tmp_path / "out" / ("admm_weight_" + name))[0]
        np.testing.assert_allclose(x, workers[k].x, rtol=2e-5, atol=1e-6 * max(1.0, np.abs(workers[k].x).max()))
    assert [int(x) for x in re.findall(r"^Finish (\d+)-th iteration$", out, flags=re.M)] == [0, 1, 2]
    # the residuals over all ranks ride in the tail of the payload, one iteration behind, and the last
    # pair takes one small exchange of its own: primal = sqrt(sum over ranks |x_r - z|^2), dual = rho |z - z_prev|
    res = re.findall(r"^\[residual-all\] (\d+) (\S+) (\S+)$", out, flags=re.M)
    assert [int(a[0]) for a in res] == [0, 1, 2]
    for (_, pr, du), r in zip(res, rec):
        assert float(pr) == pytest.approx(np.sqrt(sum(r2 for r2, _ in r["residuals"])), rel=1e-4)
        assert float(du) == pytest.approx(np.sqrt(r["residuals"][0][1]), rel=1e-4, abs=1e-12)
    # global_metrics = 1: rank 0 also reports logloss and AUC over the test rows of all three ranks
    alls = re.findall(r"^all \| ([0-9.]+) \| ([0-9.]+) \| (\d+)$", out, flags=re.M)
    assert len(alls) == 3 and all(int(n) == 600 for _, _, n in alls)
    for (l, _, _), r in zip(alls, rec):      # equal test sizes: the mean of the per-rank loglosses
        assert float(l) == pytest.approx(np.mean([r["lines"][k][0] for k in range(3)]), rel=2e-5)
    scores, labels = [], []
    for k in range(3):                       # last iteration: the saved x is the one that was scored
        _, p = oracle.logloss_sum_f64(tests[k], [workers[k].x, workers[k].bias], want_p=True)
        scores.append(p)
        labels.append(tests[k].label)
    want = oracle.auc_from_scores(np.concatenate(scores), np.concatenate(labels), f64=True)
    assert float(alls[-1][1]) == pytest.approx(want, rel=5e-5)


def test_wtest_in_three_psid_directories_under_the_launcher(tmp_path):
    """the reference's plumbing case as it is worded: test/wtest.libsvm replicated into the
    directories of three psids, `dmlc_local.py -n 3 ./admm admm.conf <train> <test> <out>`.  Every
    rank reports the zero-weight logloss on its one test row in each of the three iterations (one
    positive row: the reference's AUC formula divides by zero there, src/metrics.cc:80, so that
    field is not a number), and ends in the state a single FTRL step from zero leaves."""
    names = ["A", "B", "C"]
    fixture = open(os.path.join(GOLDEN, "wtest.libsvm")).read()
    for name in names:
        for d in ("train", "test"):
            os.makedirs(tmp_path / d / (name + "_aggregated"))
            (tmp_path / d / (name + "_aggregated") / "part-00000").write_text(fixture)
    os.makedirs(tmp_path / "out")
    conf = tmp_path / "admm.conf"
    conf.write_text("step_size = 0.01\nbias_var = 0.01\nglobal_var = 4\nftrl_alpha = 0.1\npasses = 3\ndim = 6\n"
                    "A : 1\nB : 1\nC : 1\n")
    out = _run([os.path.join(ROOT, "tools", "dmlc_local.py"), "-n", "3", "--allreduce", "shm",
                os.path.join(BIN, "admm"), str(conf), str(tmp_path / "train"), str(tmp_path / "test"),
                str(tmp_path / "out")])
    for name in names:
        lines = re.findall(r"^%s \| 0\.6931472 \| (\S+) \| 1$" % name, out, flags=re.M)
        assert len(lines) == 3, out
        assert all(not np.isfinite(float(a)) for a in lines)
        x, bias, y, nt, zt = _read_sparse_sections(tmp_path / "out" / ("admm_weight_" + name))
        assert not x.any() and not y.any() and not bias.any()
        assert nt[[0, 1, 3, 5]].tolist() == [0.25] * 4 and zt[[0, 1, 3, 5]].tolist() == [-0.5] * 4
        assert not nt[[2, 4]].any() and not zt[[2, 4]].any()
        assert open(tmp_path / "out" / (name + ".bin")).read() == "-1:0\n"
        g = open(tmp_path / "out" / ("global_params_" + name)).read().split("\n")
        assert [float(v) for v in g[:4]] == [0.01, 4.0, 0.01, 6.0] and not [l for l in g[4:] if l]
    assert [int(x) for x in re.findall(r"^Finish (\d+)-th iteration$", out, flags=re.M)] == [0, 1, 2]


def test_launcher_restarts_on_254_and_reports_failures(tmp_path):
    """the keep-alive contract of the reference's launcher: exit code 254 means "start me again"; this
    part needs no GPU work, the workers are shell one-liners"""
    marker = tmp_path / "seen"
    script = tmp_path / "w.sh"
    script.write_text("#!/bin/bash\n"
                      "echo \"$DMLC_TASK_ID/$DMLC_NUM_WORKER/$DMLC_ROLE/$DMLC_NUM_ATTEMPT\" >> %s\n"
                      "if [ \"$DMLC_TASK_ID\" = 1 ] && [ \"$DMLC_NUM_ATTEMPT\" = 0 ]; then exit 254; fi\n"
                      "exit 0\n" % marker)
    os.chmod(script, 0o755)
    _run([os.path.join(ROOT, "tools", "dmlc_local.py"), "-n", "2", str(script)])
    seen = sorted(open(marker).read().split())
    assert seen == ["0/2/worker/0", "1/2/worker/0", "1/2/worker/1"]
    bad = tmp_path / "bad.sh"
    bad.write_text("#!/bin/bash\nif [ \"$DMLC_TASK_ID\" = 0 ]; then exit 3; fi\nsleep 30\n")
    os.chmod(bad, 0o755)
    r = subprocess.run([os.path.join(ROOT, "tools", "dmlc_local.py"), "-n", "2", str(bad)],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 3


def test_row_cursor_reads_the_rows_back_from_the_device(tmp_path):
    """SampleSet keeps no host copy of a loaded part; Next / GetData (src/sample_set.cc:32-68) fetch the rows
    from HBM again: parsecheck -c prints them through the cursor, and they are the file's rows"""
    import subprocess
    raw = make_synth_csr(300, 500, 7, seed=11, binary=False)
    path = str(tmp_path / "part-00000")
    write_libsvm(path, raw)
    exe = os.path.join(ROOT, "admm_amd", "bin", "parsecheck")
    out = {}
    for flag in ("-c", "-d"):
        p = subprocess.run([exe, path, "0", "1", flag], capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr[-1000:]
        out[flag] = p.stdout.splitlines()
    assert out["-c"][0].split() == [str(raw.nrows), str(len(raw.col) + raw.nrows)]   # Size(), nnz with the bias column
    assert out["-c"][1:] == out["-d"][1:]       # the same rows as the plain dump of the parsed text
    assert len(out["-c"]) == raw.nrows + 1
