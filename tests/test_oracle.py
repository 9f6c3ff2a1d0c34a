"""The CPU restatement against every known answer available for the reference's hot path, plus
internal consistency checks.  No GPU needed."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, make_synth_csr


@pytest.fixture(scope="module")
def known():
    with open(os.path.join(GOLDEN, "wtest_known_answers.json")) as f:
        return json.load(f)


def _wtest(oracle):
    raw = oracle.parse_libsvm(os.path.join(GOLDEN, "wtest.libsvm"))
    return raw, oracle.translate(raw)


def test_wtest_bookkeeping(oracle, known):
    raw, tr = _wtest(oracle)
    assert raw.nrows == 1 and raw.nnz == 3
    assert raw.col.tolist() == [0, 2, 4] and raw.label.tolist() == [1.0]
    # bias prepend + shift: nnz bookkeeping = file nnz + rows (src/sample_set.cc:70-88)
    assert tr.nnz == raw.nnz + raw.nrows
    assert tr.col.tolist() == known["touched"] == [0, 1, 3, 5]
    assert tr.val.tolist() == [1.0, 1.0, 1.0, 1.0]
    assert tr.rowptr.tolist() == [0, 4]


def test_zero_weight_prediction(oracle, known):
    # test/ftrl_test.cc:37: prediction at zero weights is 0.5
    _, tr = _wtest(oracle)
    dim = 6
    w, z, n, off, reg = (np.zeros(dim, np.float32) for _ in range(5))
    p = oracle.lib().orc_ftrl_predict(tr.col, tr.val.ctypes.data, tr.nnz, w, z, n, off, reg,
                                      1.0, 1.0, 0.1, 0.2)
    assert p == known["zero_weight_prediction"]
    ll = oracle.logloss(tr, [w, off])
    assert "%.7f" % ll == known["zero_weight_logloss"]


def test_prediction_ignores_assigned_weights(oracle):
    """test/ftrl_test.cc:28-49: FtrlSolver(0.1, 0.2, 1, 1, 1, 5) predicts 0.5 on the row {1:1.0, 3:2.0, 4:1.4}
    at zero state, and still 0.5 after Assign(all 2.0): Predict takes the lazy weights from z and n
    (src/ftrl.cc:54-75), not from weight_"""
    col = np.array([1, 3, 4], np.uint32)
    val = np.array([1.0, 2.0, 1.4], np.float32)
    dim = 5
    z, n, off, reg = (np.zeros(dim, np.float32) for _ in range(4))
    for w in (np.zeros(dim, np.float32), np.full(dim, 2.0, np.float32)):
        p = oracle.lib().orc_ftrl_predict(col, val.ctypes.data, 3, w, z, n, off, reg, 0.1, 0.2, 1.0, 1.0)
        assert p == 0.5


def test_ftrl_passes_known_answers(oracle, known):
    _, tr = _wtest(oracle)
    prm = known["ftrl_params"]
    dim = prm["dim"]
    w, z, n, off, reg = (np.zeros(dim, np.float32) for _ in range(5))
    for expect in known["ftrl_passes"]:
        oracle.ftrl_run(tr, w, z, n, off, reg, prm["alpha"], prm["beta"], prm["l_1"], prm["l_2"])
        for j in known["touched"]:
            # recorded with 9 significant digits: exact float32 agreement
            assert float("%.9g" % w[j]) == expect["w"], (j, w[j])
            assert float("%.9g" % z[j]) == expect["z"], (j, z[j])
            assert float("%.9g" % n[j]) == expect["n"], (j, n[j])
        untouched = [j for j in range(dim) if j not in known["touched"]]
        assert not w[untouched].any() and not z[untouched].any() and not n[untouched].any()


def test_admm_wtest_state(oracle, known):
    _, tr = _wtest(oracle)
    prm = known["admm_params"]
    conf = oracle.AdmmConf(prm["step_size"], prm["global_var"], prm["ftrl_alpha"], prm["passes"],
                           prm["dim"], bias_var=prm["bias_var"])
    workers, zg, rec = oracle.admm_run([[tr]], [tr], conf)
    st = known["admm_state"]
    for j in known["touched"]:
        assert workers[0].zt[j] == np.float32(st["z"]) and workers[0].nt[j] == np.float32(st["n"])
        assert workers[0].x[j] == st["w"]
    assert not zg.any()
    for r in rec:
        assert "%.7f" % r["lines"][0][0] == known["zero_weight_logloss"]
        assert r["lines"][0][2] == 1


def test_reference_worker_test_expectations(oracle):
    """test/worker_test.cc:47-63 of the reference: after one local update on test/wtest.libsvm the
    payload is all zero, the consensus vector computed from it is all zero, and the payload after the
    dual update is all zero again (Init(1, 1, 1, ...): step_size = bias_var = global_var = 1).  The
    reference's test sizes the vectors with 5, one short of what its own bias shift needs for the
    fixture's index 4; 6 is used here."""
    _, tr = _wtest(oracle)
    dim = 6
    conf = oracle.AdmmConf(step_size=1.0, global_var=1.0, ftrl_alpha=1.0, passes=1, dim=dim, bias_var=1.0)
    wk, zg = oracle.WorkerState(dim), np.zeros(dim, np.float32)
    oracle.base_update(wk, [tr], conf, zg)                                   # :39
    result = oracle.payload(wk.x, wk.y, conf.step_size)                      # :44
    assert not result.any()                                                  # :46-47
    zg = oracle.global_update(result, 1, conf.step_size, conf.global_var)    # :50
    assert not zg.any()                                                      # :53-54
    oracle.dual_update(wk.y, wk.x, zg, conf.step_size)                       # :57
    assert not oracle.payload(wk.x, wk.y, conf.step_size).any()              # :59-62


def test_batch1_equals_sequential(oracle):
    raw = make_synth_csr(300, 64, 6, seed=3)
    tr = oracle.translate(raw)
    dim = 64
    rng = np.random.default_rng(0)
    reg = rng.normal(0, 0.01, dim).astype(np.float32)
    off = np.zeros(dim, np.float32)
    a = [np.zeros(dim, np.float32) for _ in range(3)]
    b = [np.zeros(dim, np.float32) for _ in range(3)]
    oracle.ftrl_run(tr, *a, off, reg, 0.05, 1.0, 0.05, 1.5, batch=1)
    g1 = np.zeros(dim)
    g2 = np.zeros(dim)
    oracle.lib().orc_ftrl_run_batch(*tr.args(), *b, off, reg, 0.05, 1.0, 0.05, 1.5, 1, g1, g2)
    for u, v in zip(a, b):
        np.testing.assert_allclose(u, v, rtol=2e-6, atol=1e-7)
    assert not g1.any() and not g2.any()


def test_tied_rank_and_auc_edges(oracle):
    x = np.array([0.3, 0.1, 0.3, 0.9, 0.1, 0.3], np.float32)
    r = oracle.tied_rank(x)
    assert r.tolist() == [4.0, 1.5, 4.0, 6.0, 1.5, 4.0]
    # labels: only (int)label == 1 counts as positive (src/metrics.cc:60,72)
    lab = np.array([1, 0, 2, 1, -1, 0], np.float32)
    a = oracle.auc_from_scores(x, lab)
    # positives at ranks 4 and 6: (10)/(2*4) - 3/(8)
    assert a == pytest.approx(10.0 / 8.0 - 3.0 / 8.0)
    assert oracle.auc_from_scores(x, lab, f64=True) == pytest.approx(a)
    # independent O(n^2) pair count (the definition script/auc.py uses)
    rng = np.random.default_rng(1)
    s = rng.integers(0, 20, 400).astype(np.float32) / 20
    y = (rng.random(400) < 0.4).astype(np.float32)
    pos, neg = s[y == 1], s[y != 1]
    pairs = (pos[:, None] > neg[None, :]).sum() + 0.5 * (pos[:, None] == neg[None, :]).sum()
    assert oracle.auc_from_scores(s, y) == pytest.approx(pairs / (len(pos) * len(neg)), rel=1e-5)
    # all one class: division by zero, as in the reference (nan or inf, not an exception)
    assert not np.isfinite(oracle.auc_from_scores(s, np.ones(400, np.float32)))


def test_admm_algebra(oracle):
    rng = np.random.default_rng(2)
    dim = 50
    x = rng.normal(size=dim).astype(np.float32)
    y = rng.normal(size=dim).astype(np.float32)
    rho = np.float32(0.01)
    p = oracle.payload(x, y, rho)
    np.testing.assert_array_equal(p, x * rho + y)
    z = oracle.global_update(p, 3, rho, np.float32(4.0))
    np.testing.assert_array_equal(z, p / np.float32(3 * rho + np.float32(4.0)))
    zc = oracle.global_update_commented(p, 3, rho, np.float32(0.004))
    crit = np.float32(0.004) / rho
    ref = np.where(p > crit, (p - crit) / 3, np.where(p < -crit, (p + crit) / 3, 0)).astype(np.float32)
    np.testing.assert_allclose(zc, ref, rtol=1e-6)
    y0 = y.copy()
    oracle.dual_update(y, x, z, rho)
    np.testing.assert_allclose(y, y0 + rho * (x - z), rtol=1e-6)


def test_fg_gradient_is_consistent(oracle):
    raw = make_synth_csr(200, 40, 5, seed=5, binary=False)
    tr = oracle.translate(raw)
    dim = 40
    rng = np.random.default_rng(7)
    x = rng.normal(0, 0.2, dim).astype(np.float32)
    z = rng.normal(0, 0.2, dim).astype(np.float32)
    yd = rng.normal(0, 0.1, dim).astype(np.float32)
    f0, g = oracle.logistic_fg(tr, dim, x, None, z, yd, 0.5)
    for j in (0, 3, 17):
        e = np.zeros(dim, np.float32)
        h = np.float32(1e-2)
        e[j] = h
        fp, _ = oracle.logistic_fg(tr, dim, x + e, None, z, yd, 0.5)
        fm, _ = oracle.logistic_fg(tr, dim, x - e, None, z, yd, 0.5)
        assert (fp - fm) / (2 * float(h)) == pytest.approx(g[j], rel=2e-3, abs=1e-4)


def test_lbfgs_decreases_and_converges(oracle):
    raw = make_synth_csr(400, 30, 5, seed=9)
    tr = oracle.translate(raw)
    dim = 30
    z = np.zeros(dim, np.float32)

    def fg(xx):
        return oracle.logistic_fg(tr, dim, xx, None, z, None, 1.0)

    f0, _ = fg(np.zeros(dim, np.float32))
    x, f, evals, iters = oracle.lbfgs_solve(np.zeros(dim, np.float32), fg, max_iter=60)
    assert f < f0
    _, g = fg(x)
    assert np.linalg.norm(g) / max(1.0, np.linalg.norm(x)) < 1e-3


def test_libsvm_parts_cover_all_rows(oracle):
    text = "".join("%d %s\n" % (i % 2, " ".join("%d:%g" % (j, 0.5 + j) for j in range(i % 5)))
                   for i in range(57)) + "\n\n1 7 9:2\n"
    whole = oracle.parse_libsvm_text(text)
    assert whole.nrows == 58
    assert whole.col[-2:].tolist() == [7, 9] and whole.val[-2:].tolist() == [1.0, 2.0]
    for nparts in (1, 2, 3, 7):
        parts = [oracle.parse_libsvm_text(text, p, nparts) for p in range(nparts)]
        assert sum(p.nrows for p in parts) == whole.nrows
        assert sum(p.nnz for p in parts) == whole.nnz
        np.testing.assert_array_equal(np.concatenate([p.label for p in parts]), whole.label)
        np.testing.assert_array_equal(np.concatenate([p.col for p in parts]), whole.col)
