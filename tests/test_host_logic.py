"""Host logic that needs no GPU: the conf grammar, the report line, row sharding."""
import os

import numpy as np

from admm_amd import dist as adist
from admm_amd import driver


def test_admm_conf_grammar(tmp_path):
    conf = tmp_path / "admm.conf"
    conf.write_text("step_size = 0.01\nbias_var = 0.01 global_var = 4\n  ftrl_alpha = 0.1x\npasses = 3\n"
                    "dim = 6\nstray token\nPDPS000000005427 : 2\nB : 1\nload_path = /tmp/w\n"
                    "local_solver = lbfgs\nglobal_l1 = 0.5\nlbfgs_max_iter = 7\n")
    ap, prm = driver.ArgParser(), driver.AdmmConfig()
    ap.ADMMParse(str(conf), prm)
    assert prm.step_size == np.float32(0.01) and prm.global_var == 4 and prm.bias_var == np.float32(0.01)
    assert prm.ftrl_alpha == np.float32(0.1)          # atof stops at the first bad character
    assert (prm.passes, prm.dim) == (3, 6)
    assert ap.psid == ["PDPS000000005427", "B"] and ap.num_part == [2, 1]
    assert ap.load_path == "/tmp/w"
    assert prm.local_solver == "lbfgs" and prm.global_l1 == np.float32(0.5) and prm.lbfgs["max_iter"] == 7
    # keys the reference does not know keep their defaults when absent
    conf.write_text("step_size = 1\nglobal_var = 1\nftrl_alpha = 1\npasses = 1\ndim = 2\nA : 1\n")
    ap, prm = driver.ArgParser(), driver.AdmmConfig()
    ap.ADMMParse(str(conf), prm)
    assert prm.local_solver == "ftrl" and prm.global_l1 == 0 and prm.ftrl_batch == 1 and prm.bias_var == 0


def test_ftrl_conf_grammar(tmp_path):
    conf = tmp_path / "ftrl.conf"
    conf.write_text("train_path = /a/\ntest_path = /b/\noutput_path = /c/\nftrl_alpha = 0.01\nftrl_beta = 1\n"
                    "l_1 = 0.4\nl_2 = 2.5\npasses = 4\ndim = 2000000\nX : 3\n")
    ap, prm = driver.ArgParser(), driver.FtrlConfig()
    ap.FTRLParse(str(conf), prm)
    assert (prm.train_path, prm.test_path, prm.output_path) == ("/a/", "/b/", "/c/")
    assert prm.alpha == np.float32(0.01) and prm.beta == 1 and prm.l_1 == np.float32(0.4)
    assert prm.l_2 == np.float32(2.5) and prm.passes == 4 and prm.dim == 2000000
    assert ap.psid == ["X"] and ap.num_part == [3] and ap.load_path == ""


def test_missing_conf_is_reported_not_raised(tmp_path, capsys):
    ap, prm = driver.ArgParser(), driver.AdmmConfig()
    ap.ADMMParse(str(tmp_path / "nope.conf"), prm)
    assert "config file open failed" in capsys.readouterr().out
    assert prm.dim == 0 and ap.psid == []


def test_report_line_format():
    assert driver.format_line("A", 0.69314718, 0.5, 3) == "A | 0.6931472 | 0.5000000 | 3"


def test_shard_ranges_partition_the_rows():
    for total in (0, 1, 7, 100_000_000):
        for world in (1, 2, 3, 8):
            spans = [adist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _parsecheck(path, part, nparts, env=None):
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "admm_amd", "bin", "parsecheck")
    if not os.path.exists(exe):
        from admm_amd import build
        build.build_lib()
        build.build_host()
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([exe, str(path), str(part), str(nparts)], capture_output=True, text=True, timeout=60, env=e)
    assert out.returncode == 0, out.stderr
    return out.stdout


def _expected_dump(csr):
    lines = ["%d %d %d" % (csr.nrows, csr.nnz, int(csr.col.max()) if csr.nnz else 0)]
    for r in range(csr.nrows):
        b, e = int(csr.rowptr[r]), int(csr.rowptr[r + 1])
        lines.append(" ".join(["%.9g" % csr.label[r]] +
                              ["%d:%.9g" % (int(csr.col[k]), float(csr.val[k])) for k in range(b, e)]))
    return "\n".join(lines) + "\n"


def test_threaded_libsvm_loader_matches_the_oracle_reader(tmp_path, oracle):
    """the C++ loader (several threads, byte-range parts, the float fast path, the binary cache)
    against the oracle's reading of the same text"""
    rng = np.random.default_rng(5)
    path = tmp_path / "part-00000"
    with open(path, "w") as f:
        for r in range(3000):
            n = int(rng.integers(0, 30))
            ids = np.sort(rng.choice(5000, n, replace=False))
            toks = []
            for i in ids:
                kind = rng.integers(0, 5)
                if kind == 0:
                    toks.append("%d" % i)                       # index alone: value 1
                elif kind == 1:
                    toks.append("%d:1" % i)
                elif kind == 2:
                    toks.append("%d:%.4f" % (i, rng.uniform(-3, 3)))
                elif kind == 3:
                    toks.append("%d:%.9g" % (i, rng.normal()))   # long mantissa -> strtof
                else:
                    toks.append("%d:%.3e" % (i, rng.normal()))   # exponent -> strtof
            f.write("%s %s%s\n" % (rng.choice(["1", "0", "-1", "+1", "0.5"]), " ".join(toks),
                                   "\r" if r % 7 == 0 else ""))
            if r % 500 == 499:
                f.write("\n")                                    # blank lines are skipped
    text = open(path, "rb").read()
    for nparts in (1, 3):
        for part in range(nparts):
            want = _expected_dump(oracle.parse_libsvm_text(text, part, nparts))
            for threads in ("1", "5"):
                got = _parsecheck(path, part, nparts, {"ADMM_PARSE_THREADS": threads, "ADMM_PARSE_MIN_BYTES": "4096"})
                assert got == want
    # the binary cache: written by the first run, read by the second, dropped when the text changes
    cache = tmp_path / "cache"
    cache.mkdir()
    env = {"ADMM_CSR_CACHE_DIR": str(cache)}
    want = _expected_dump(oracle.parse_libsvm_text(text, 1, 3))
    assert _parsecheck(path, 1, 3, env) == want
    files = list(cache.iterdir())
    assert len(files) == 1
    stamp = files[0].stat().st_mtime_ns
    assert _parsecheck(path, 1, 3, env) == want and files[0].stat().st_mtime_ns == stamp
    with open(path, "a") as f:
        f.write("1 7:2.5\n")
    text2 = open(path, "rb").read()
    assert _parsecheck(path, 2, 3, env) == _expected_dump(oracle.parse_libsvm_text(text2, 2, 3))
    assert _parsecheck(path, 1, 3, env) == _expected_dump(oracle.parse_libsvm_text(text2, 1, 3))


def test_cpu_side_by_side_job_under_the_launcher(tmp_path, oracle):
    """the CPU leg that bench.py times beside the GPU: three oracle workers under tools/dmlc_local.py
    with the shared-memory sum; the consensus vector they end on equals the single-process loop's"""
    import json
    import subprocess
    import sys
    from conftest import ROOT, make_synth_csr
    sys.path.insert(0, ROOT)
    from tools import cpu_admm_worker as caw
    dim, world, rows_per = 400, 3, 150
    sample = oracle.translate(make_synth_csr(600, dim, 9, seed=5))
    nnz = caw.write_shards(sample, str(tmp_path), world, rows_per)
    assert nnz == int(sample.rowptr[rows_per * world])
    outs = {}
    for mode in ("resident", "reparse"):
        env = dict(os.environ, ADMM_CPU_DUMP=str(tmp_path / ("z_%s.npy" % mode)), OMP_NUM_THREADS="1")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dmlc_local.py"), "-n", str(world),
                            sys.executable, os.path.join(ROOT, "tools", "cpu_admm_worker.py"), str(tmp_path), str(dim),
                            mode, "3"], capture_output=True, text=True, timeout=120, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("CPU_ADMM ")][-1][9:])
        assert rec["iterations"] == 3 and rec["world"] == world and rec["rows"] == rows_per and rec["loop_s"] > 0
        outs[mode] = np.load(tmp_path / ("z_%s.npy" % mode))
    np.testing.assert_array_equal(outs["resident"], outs["reparse"])
    parts = []
    for k in range(world):
        b, e = int(sample.rowptr[k * rows_per]), int(sample.rowptr[(k + 1) * rows_per])
        part = oracle.Csr(sample.rowptr[k * rows_per:(k + 1) * rows_per + 1] - sample.rowptr[k * rows_per],
                          sample.col[b:e], sample.val[b:e], sample.label[k * rows_per:(k + 1) * rows_per], dim)
        parts.append([part])
    conf = oracle.AdmmConf(step_size=0.01, global_var=4.0, ftrl_alpha=0.1, passes=3, dim=dim)
    _, zg, _ = oracle.admm_run(parts, [p[0] for p in parts], conf)
    np.testing.assert_allclose(outs["resident"], zg, rtol=1e-6, atol=1e-9)
