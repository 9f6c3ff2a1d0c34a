"""The ADMM outer loop on the GPU (admm_amd.driver over the C ABI) against the CPU restatement of
src/admm_allreduce.cpp:55-91, for 1 and 3 ranks held in one process."""
import numpy as np
import pytest

from conftest import make_synth_csr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from admm_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def _run_gpu(ctx, raws_train, raws_test, conf, dim, multi=False):
    from admm_amd import driver
    nrank = len(raws_train)
    prm = driver.AdmmConfig()
    prm.step_size, prm.global_var, prm.ftrl_alpha = conf.step_size, conf.global_var, conf.ftrl_alpha
    prm.passes, prm.dim, prm.global_l1 = conf.passes, dim, conf.global_l1
    prm.local_solver, prm.ftrl_batch = conf.local_solver, conf.ftrl_batch
    prm.lbfgs.update(conf.lbfgs)
    workers, trains, tests, zs, tmps = [], [], [], [], []
    for r in range(nrank):
        w = driver.Worker(ctx)
        w.InitWorker(dim)
        workers.append(w)
        t = ctx.csr_upload(raws_train[r].rowptr, raws_train[r].col, raws_train[r].val,
                           raws_train[r].label, dim, add_bias=True)
        if prm.local_solver == "lbfgs" or prm.ftrl_batch > 1:
            t.build_transposed(panel_rows=prm.ftrl_batch if prm.ftrl_batch > 1 else 0, col_block=128)
        trains.append([t])
        tests.append(ctx.csr_upload(raws_test[r].rowptr, raws_test[r].col, raws_test[r].val,
                                    raws_test[r].label, dim, add_bias=True))
        zs.append(ctx.vec(dim))      # every rank holds its own replica of the global weights
        tmps.append(ctx.vec(dim))
    master = driver.Master(ctx)
    records = []
    for it in range(prm.passes):
        last = it == prm.passes - 1
        # phase 1 on every rank: local solve + payload
        stats = []
        for r in range(nrank):
            if last:
                workers[r].load = True
        if multi:   # the workers share the GPU: their row-by-row passes go out as one launch
            st = driver.base_update_multi(workers, trains, prm, zs[0])
            stats = [dict(evals=st["evals"] // nrank)] * nrank
        for r in range(nrank):
            if not multi:
                stats.append(workers[r].BaseUpdate(trains[r], prm, zs[r]))
            workers[r].GetWeights(prm, tmps[r])
        # the all-reduce: host sum in rank order, as the oracle does
        total = np.zeros(dim, np.float32)
        for r in range(nrank):
            total = (total + tmps[r].download()).astype(np.float32)
        lines, res = [], []
        for r in range(nrank):
            tmps[r].upload(total)
            lines.append(tests[r].metrics([workers[r].base_vec_, workers[r].bias_vec_]))
            res.append(ctx.consensus(tmps[r], nrank, prm.step_size, prm.global_var, prm.global_l1,
                                     workers[r].base_vec_, workers[r].langr_vec_, zs[r]))
        records.append(dict(total=total, z=zs[0].download(),
                            x=[w.base_vec_.download() for w in workers],
                            y=[w.langr_vec_.download() for w in workers],
                            lines=lines, residuals=res, stats=stats))
    final = dict(nt=[w.nt_vec_.download() for w in workers], zt=[w.zt_vec_.download() for w in workers])
    for t in trains:
        t[0].destroy()
    for t in tests:
        t.destroy()
    return records, final


def _compare(rec_gpu, rec_ref, rtol, atol, line_rel=1e-5, auc_rel=1e-5, res_rel=1e-5, stats=False):
    """the tolerances are the path's stated one, 1e-5 relative: logloss and AUC against the restatement's sums in
    double (the device accumulates in double; the reference's float sums are looser than that by themselves),
    the residual norms against the restatement's, which come from vectors that agree to rtol themselves"""
    assert len(rec_gpu) == len(rec_ref)
    for a, b in zip(rec_gpu, rec_ref):
        if stats:   # the local solves took the same path: as many objective evaluations and iterations
            for sa, sb in zip(a["stats"], b["stats"]):
                assert (sa["evals"], sa["iters"]) == (sb["evals"], sb["iters"]), (sa, sb)
                assert sa["f"] == pytest.approx(sb["f"], rel=1e-5)
        np.testing.assert_allclose(a["total"], b["total"], rtol=rtol, atol=atol)
        np.testing.assert_allclose(a["z"], b["z"], rtol=rtol, atol=atol)
        for xa, xb in zip(a["x"], b["x"]):
            np.testing.assert_allclose(xa, xb, rtol=rtol, atol=atol)
        for ya, yb in zip(a["y"], b["y"]):
            np.testing.assert_allclose(ya, yb, rtol=rtol, atol=atol)
        for (la, aa, na), (lb, ab, nb), (lf, af, _) in zip(a["lines"], b["lines64"], b["lines"]):
            assert na == nb
            assert la == pytest.approx(lb, rel=line_rel)
            assert aa == pytest.approx(ab, rel=auc_rel)
            # and the reference's own float sums are where float sums of this length can be
            assert la == pytest.approx(lf, rel=2e-5) and aa == pytest.approx(af, rel=5e-5)
        for (r2a, s2a), (r2b, s2b) in zip(a["residuals"], b["residuals"]):
            assert r2a == pytest.approx(r2b, rel=res_rel, abs=1e-12)
            assert s2a == pytest.approx(s2b, rel=res_rel, abs=1e-14)


@pytest.mark.parametrize("nrank,multi", [(1, False), (3, False), (3, True)])
def test_admm_ftrl_mode_matches_reference_loop(ctx, oracle, nrank, multi):
    dim = 257
    raws_train = [make_synth_csr(1200, dim, 14, seed=50 + r) for r in range(nrank)]
    raws_test = [make_synth_csr(300, dim, 14, seed=60 + r) for r in range(nrank)]
    conf = oracle.AdmmConf(step_size=0.01, global_var=4.0, ftrl_alpha=0.1, passes=4, dim=dim, bias_var=0.01)
    _, _, rec_ref = oracle.admm_run([[oracle.translate(t)] for t in raws_train],
                                    [oracle.translate(t) for t in raws_test], conf)
    rec_gpu, final = _run_gpu(ctx, raws_train, raws_test, conf, dim, multi=multi)
    _compare(rec_gpu, rec_ref, rtol=1e-5, atol=1e-7)
    # the FTRL state kept on the last iteration (src/workers.cc:81-85) matches too
    wk, _, _ = oracle.admm_run([[oracle.translate(t)] for t in raws_train],
                               [oracle.translate(t) for t in raws_test], conf)
    for r in range(nrank):
        np.testing.assert_allclose(final["nt"][r], wk[r].nt, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(final["zt"][r], wk[r].zt, rtol=1e-5, atol=1e-6)


def test_admm_lbfgs_mode(ctx, oracle):
    dim, nrank = 129, 2
    raws_train = [make_synth_csr(900, dim, 10, seed=70 + r) for r in range(nrank)]
    raws_test = [make_synth_csr(250, dim, 10, seed=80 + r) for r in range(nrank)]
    # a few L-BFGS iterations per ADMM iteration, as the benchmark runs them: the two solvers must take the same
    # trajectory (equal numbers of evaluations and iterations, objective to 1e-5, iterates to 1e-4), not merely
    # end near the same minimum
    conf = oracle.AdmmConf(step_size=5.0, global_var=1.0, ftrl_alpha=0.1, passes=3, dim=dim, global_l1=0.05,
                           local_solver="lbfgs", lbfgs=dict(max_iter=4, epsilon=1e-7))
    _, _, rec_ref = oracle.admm_run([[oracle.translate(t)] for t in raws_train],
                                    [oracle.translate(t) for t in raws_test], conf)
    rec_gpu, _ = _run_gpu(ctx, raws_train, raws_test, conf, dim)
    _compare(rec_gpu, rec_ref, rtol=1e-4, atol=1e-6, res_rel=2e-4, stats=True)


def test_rccl_binding_selftest_and_loop_with_residual_tail(ctx, oracle):
    """admm_comm_selftest on one GPU (a one-rank communicator of its own: library loading, unique id,
    ncclCommInitRank, ncclAllReduce's type and operator codes), and driver.AdmmLoop -- the loop that
    carries the residuals in the payload's tail -- against the reference loop for one rank"""
    from admm_amd import driver
    ctx.comm_init(0, 1)
    ctx.comm_selftest()
    dim = 211
    raw_train, raw_test = make_synth_csr(900, dim, 11, seed=90), make_synth_csr(200, dim, 11, seed=91)
    conf = oracle.AdmmConf(step_size=0.01, global_var=4.0, ftrl_alpha=0.1, passes=3, dim=dim)
    _, _, rec = oracle.admm_run([[oracle.translate(raw_train)]], [oracle.translate(raw_test)], conf)
    prm = driver.AdmmConfig()
    prm.step_size, prm.global_var, prm.ftrl_alpha, prm.passes, prm.dim = 0.01, 4.0, 0.1, 3, dim
    wk = driver.Worker(ctx)
    wk.InitWorker(dim)
    train = ctx.csr_upload(raw_train.rowptr, raw_train.col, raw_train.val, raw_train.label, dim)
    test = ctx.csr_upload(raw_test.rowptr, raw_test.col, raw_test.val, raw_test.label, dim)
    zg, payload = ctx.vec(dim), ctx.vec(dim + driver.RESIDUAL_TAIL)
    loop = driver.AdmmLoop(wk, driver.Master(ctx), [train], test, prm, zg, payload, lambda buf: ctx.allreduce_sum(buf), 1)
    for it in range(3):
        out = loop.iteration(last=it == 2)
        np.testing.assert_allclose(zg.download(), rec[it]["z"], rtol=1e-5, atol=1e-9)
        assert out["line"][0] == pytest.approx(rec[it]["lines64"][0][0], rel=1e-5)
        assert out["line"][1] == pytest.approx(rec[it]["lines64"][0][1], rel=1e-5)
        assert out["r2"] == pytest.approx(rec[it]["residuals"][0][0], rel=1e-5, abs=1e-12)
    last = loop.flush_residuals()
    assert last["iterations_recorded"] == 3
    for (r2, s2), r in zip(loop.global_residuals, rec):     # one rank: the sums over ranks are the rank's own
        assert r2 == pytest.approx(r["residuals"][0][0], rel=1e-5, abs=1e-12)
        assert s2 == pytest.approx(r["residuals"][0][1], rel=1e-5, abs=1e-14)
    train.destroy()
    test.destroy()


@pytest.mark.parametrize("batch", [1, 512])
def test_bias_update_is_an_ftrl_pass_over_the_base_model(ctx, oracle, batch):
    """Worker::BiasUpdate (src/workers.cc:89-113): zero state, l_2 = 200, l_1 = 1000, offset = the base
    model, no proximal centre; the result replaces bias_vec_"""
    from admm_amd import driver
    dim = 400
    raw = make_synth_csr(3000, dim, 12, seed=71)
    raw.label[:] = 1.0                 # with the bias weight below every gradient is close to -1, so that the
    tr = oracle.translate(raw)         # frequent columns' z leave the [-1000, 1000] band in which the weight is 0
    rng = np.random.default_rng(5)
    base = rng.normal(0, 0.3, dim).astype(np.float32)
    base[0] = -6.0
    prm = driver.AdmmConfig()
    prm.dim, prm.ftrl_alpha, prm.ftrl_batch, prm.local_solver = dim, 0.05, batch, "ftrl"
    wk = driver.Worker(ctx)
    wk.InitWorker(dim)
    wk.base_vec_.upload(base)
    dev = ctx.csr_upload(raw.rowptr, raw.col, raw.val, raw.label, dim, add_bias=True)
    if batch > 1:
        dev.build_transposed(panel_rows=batch, col_block=128)
    wk.BiasUpdate([dev], prm)
    hw, hz, hn = (np.zeros(dim, np.float32) for _ in range(3))
    oracle.ftrl_run(tr, hw, hz, hn, base, np.zeros(dim, np.float32), 0.05, 1.0, 1000.0, 200.0, batch=batch)
    assert np.count_nonzero(hw) >= 1
    np.testing.assert_allclose(wk.bias_vec_.download(), hw, rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(wk.base_vec_.download(), base)     # the base model is only read
    dev.destroy()
