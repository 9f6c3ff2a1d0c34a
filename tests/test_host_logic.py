"""Host logic that needs no GPU: the conf grammar, the report line, row sharding."""
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
