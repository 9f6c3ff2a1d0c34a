"""Seeded cases on and around the layouts' boundaries -- dims at 2 .. 65 (fewer columns than one mask word),
255 .. 257 (the mask tier's 256 columns), 8191 .. 8193 (the 16-bit streams' ids) and beyond; row counts at 0, 1
and around 64, 256 and 4096 (a wave, a short panel, a row panel); empty rows, single-entry rows, rows in the
rarest columns only, one long row; sets with and without stored values, relabelled before or after the copies are
built, lean or not -- each through the objective / gradient / predictions, the mini-batch FTRL pass and the
row-by-row FTRL pass, against the CPU restatement.  ADMM_FUZZ_SEEDS widens the run."""
import os

import numpy as np
import pytest

from conftest import make_synth_csr

pytestmark = pytest.mark.gpu

DIMS = [2, 3, 17, 63, 64, 65, 255, 256, 257, 300, 2500, 8191, 8192, 8193, 9000, 20000, 70000]
ROWS = [0, 1, 2, 63, 64, 65, 255, 256, 257, 700, 4095, 4096, 4097, 6000]


@pytest.fixture(scope="module")
def ctx():
    from admm_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def close(a, b, rtol, atol):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def ragged(orc, base, rng, binary, dups):
    """base with some rows emptied, cut to one entry, moved into the rarest columns, shuffled, with a repeated
    column (dups), and one made long"""
    nrows, dim = base.nrows, base.dim
    nfeat = dim - 1
    rows = [base.col[int(base.rowptr[r]):int(base.rowptr[r + 1])] for r in range(nrows)]
    vals = [base.val[int(base.rowptr[r]):int(base.rowptr[r + 1])] for r in range(nrows)]
    counts = np.bincount(base.col, minlength=nfeat) if nfeat else np.zeros(0, np.int64)
    rare = np.argsort(counts, kind="stable")[:max(1, nfeat // 2)]

    def values(k):
        return np.ones(k, np.float32) if binary else rng.uniform(0.25, 2.0, k).astype(np.float32)

    for r in range(nrows):
        kind = rng.integers(0, 12)
        if kind == 0:
            rows[r], vals[r] = rows[r][:0], vals[r][:0]
        elif kind == 1:
            rows[r], vals[r] = rows[r][:1], vals[r][:1]
        elif kind == 2 and nfeat:
            k = int(min(len(rare), rng.integers(1, 6)))
            rows[r] = np.sort(rng.choice(rare, k, replace=False)).astype(np.uint32)
            vals[r] = values(k)
        elif kind == 3 and dups and len(rows[r]):          # a column stored twice (libsvm rows may do that)
            at = int(rng.integers(0, len(rows[r])))
            rows[r] = np.insert(rows[r], at, rows[r][at])
            vals[r] = np.insert(vals[r], at, values(1)[0])
        elif kind == 4 and len(rows[r]) > 1:               # ids out of order
            order = rng.permutation(len(rows[r]))
            rows[r], vals[r] = rows[r][order], vals[r][order]
    if nrows and nfeat:
        k = int(min(nfeat, rng.choice([70, 300, 5000])))
        r = int(rng.integers(0, nrows))
        rows[r] = np.sort(rng.choice(nfeat, k, replace=False)).astype(np.uint32)
        vals[r] = values(k)
    rowptr = np.concatenate([[0], np.cumsum([len(x) for x in rows])]).astype(np.uint64)
    col = np.concatenate(rows).astype(np.uint32) if rows else np.zeros(0, np.uint32)
    val = np.concatenate(vals).astype(np.float32) if vals else np.zeros(0, np.float32)
    return orc.Csr(rowptr, col, val, base.label, dim)


def draw(orc, seed):
    rng = np.random.default_rng(77000 + seed)
    dim = int(rng.choice(DIMS))
    nrows = int(rng.choice(ROWS))
    mean = float(rng.choice([0.6, 3, 12, 40]))
    binary = bool(rng.integers(0, 2))
    base = make_synth_csr(nrows, dim, min(mean, max(0.6, (dim - 1) / 2)), seed=seed, binary=binary,
                          label01=bool(rng.integers(0, 2)))
    raw = ragged(orc, base, rng, binary, dups=bool(rng.integers(0, 4) == 0)) if rng.integers(0, 3) else base
    return rng, raw, dim, nrows, binary


def build(ctx, rng, raw, dim, panel_rows, may_be_lean=True):
    """uploads, relabels before or after the copies (or not at all), lean or not; returns the device set"""
    dev = ctx.csr_upload(raw.rowptr, raw.col, raw.val, raw.label, dim, add_bias=True)
    lean = bool(rng.integers(0, 2)) and may_be_lean
    order = int(rng.integers(0, 3))      # 0: no relabelling, 1: before the copies, 2: after
    cb = int(rng.choice([64, 256, 2048, 16384 if panel_rows == 0 else 8192]))   # a mini-batch pass: at most 8192
    cb = min(cb, max(64, (dim // 64) * 64))
    if lean:
        dev.set_lean()
    if order == 1:
        dev.relabel_by_frequency()
    dev.build_transposed(panel_rows=panel_rows, col_block=cb)
    if order == 2:
        dev.relabel_by_frequency()
    return dev, lean


@pytest.mark.parametrize("seed", range(int(os.environ.get("ADMM_FUZZ_SEEDS", "24"))))
def test_objective_gradient_predictions(ctx, oracle, seed):
    rng, raw, dim, nrows, binary = draw(oracle, seed)
    tr = oracle.translate(raw)
    dev, _ = build(ctx, rng, raw, dim, int(rng.choice([0, 0, 700])))
    x = rng.normal(0, 0.2, dim).astype(np.float32)
    off = rng.normal(0, 0.05, dim).astype(np.float32)
    z = rng.normal(0, 0.2, dim).astype(np.float32)
    yd = rng.normal(0, 0.02, dim).astype(np.float32)
    f_ref, g_ref = oracle.logistic_fg(tr, dim, x, off, z, yd, 0.3)
    g = ctx.vec(dim)
    for _ in range(2):
        f = dev.logistic_fg(ctx.vec(dim, x), g, offset=ctx.vec(dim, off), z=ctx.vec(dim, z), ydual=ctx.vec(dim, yd), rho=0.3)
        close(f, f_ref, 1e-5, 1e-7)
        # (a set of a few columns has hundreds of rows with one and the same margin; their float residuals round
        # the same way and the gradient adds them up)
        close(g.download(), g_ref, 2e-5, 5e-5 * max(np.abs(g_ref).max(), 1e-30))
    if nrows:
        _, p_ref = oracle.logloss_sum_f64(tr, [x, off], want_p=True)
        # (the restatement adds a row's products in float, one after the other, as the reference does: a row of
        # 5000 entries carries that rounding into its margin)
        close(dev.predict([ctx.vec(dim, x), ctx.vec(dim, off)]), p_ref, 1e-5, 2e-5)
    dev.destroy()


@pytest.mark.parametrize("seed", range(int(os.environ.get("ADMM_FUZZ_SEEDS", "24"))))
def test_ftrl_passes(ctx, oracle, seed):
    rng, raw, dim, nrows, binary = draw(oracle, 500 + seed)
    tr = oracle.translate(raw)
    batch = int(rng.choice([1, 1, 256, 512, 2048]))
    dev, _ = build(ctx, rng, raw, dim, batch if batch > 1 else 0, may_be_lean=False)   # the passes read rows
    reg = rng.normal(0, 0.02, dim).astype(np.float32)
    off = rng.normal(0, 0.02, dim).astype(np.float32)
    hw, hz, hn = (np.zeros(dim, np.float32) for _ in range(3))
    w, z, n = ctx.vec(dim), ctx.vec(dim), ctx.vec(dim)
    doff, dreg = ctx.vec(dim, off), ctx.vec(dim, reg)
    # The first pass is held to the stated tolerance.  A second pass starts from the first one's state, and a
    # row-by-row pass is a long chain: a row's float sum in another order (a wavefront's tree here, one product
    # after the other in the reference) moves a coordinate across its l_1 threshold a row earlier or later now and
    # then, and every later row sees that -- a looser bound there, far below what a wrong entry would cause.
    for k in range(2):
        oracle.ftrl_run(tr, hw, hz, hn, off, reg, 0.05, 1.0, 0.2, 1.5, batch=batch)
        dev.ftrl_pass(w, z, n, doff, dreg, 0.05, 1.0, 0.2, 1.5, batch=batch)
        for got, ref in ((w, hw), (z, hz), (n, hn)):
            top = max(1.0, np.abs(ref).max())
            if k == 0:
                close(got.download(), ref, 5e-5 if batch > 1 else 1e-5, 2e-5 * top)
            else:
                close(got.download(), ref, 2e-4, 2e-4 * top)
    dev.destroy()


@pytest.mark.parametrize("seed", range(int(os.environ.get("ADMM_FUZZ_SEEDS", "24"))))
def test_solver_steps_and_metrics(ctx, oracle, seed):
    """three L-BFGS iterations follow the restatement's solver evaluation for evaluation in every line search mode,
    and logloss / AUC of the result agree with its double sums"""
    rng, raw, dim, nrows, binary = draw(oracle, 900 + seed)
    tr = oracle.translate(raw)
    dev, _ = build(ctx, rng, raw, dim, 0)
    z = rng.normal(0, 0.2, dim).astype(np.float32)
    off = rng.normal(0, 0.05, dim).astype(np.float32)
    mode = int(rng.integers(0, 3))

    def fg(xx):
        return oracle.logistic_fg(tr, dim, xx, off, z, None, 1.5)

    x_ref, f_ref, ev_ref, it_ref = oracle.lbfgs_solve(np.zeros(dim, np.float32), fg, max_iter=3)
    _, f_two, _, _ = oracle.lbfgs_solve(np.zeros(dim, np.float32), fg, max_iter=2)
    x = ctx.vec(dim)
    res = dev.lbfgs_solve(x, offset=ctx.vec(dim, off), z=ctx.vec(dim, z), rho=1.5, max_iter=3, linesearch=mode)
    if f_two - f_ref < 1e-2 * abs(f_ref):
        # the third step gained less than a hundredth: its direction comes from differences of nearly equal
        # iterates, and how many trial points its line search takes is the rounding's choice
        close(res["f"], f_ref, 1e-3, 1e-6)
        dev.destroy()
        return
    assert (res["evals"], res["iters"]) == (ev_ref, it_ref)
    close(res["f"], f_ref, 1e-4, 1e-6)        # after three steps; a single evaluation is held to 1e-5 above
    # (three steps from float residuals: a few hundred rows with one and the same margin round the same way, and
    # the bias coordinate, whose gradient is their sum, carries that)
    close(x.download(), x_ref, 1e-4, 1e-4 * max(1.0, np.abs(x_ref).max()))
    if nrows:
        ll, auc, n = dev.metrics([x, ctx.vec(dim, off)])
        s64 = oracle.logloss_sum_f64(tr, [x.download(), off])
        assert n == nrows
        close(ll, s64 / nrows, 1e-5, 1e-7)
        # the AUC of the device's own predictions through the restatement's tie-aware ranks (scores one ulp apart
        # tie or not: predictions from another order of float additions move the AUC by whole pairs)
        p = dev.predict([x, ctx.vec(dim, off)])
        a_ref = oracle.auc_from_scores(p, tr.label, f64=True)
        if np.isfinite(a_ref):                        # (one class only: the reference divides by zero)
            close(auc, a_ref, 1e-5, 1e-6)
    dev.destroy()


@pytest.mark.parametrize("dim", [17, 200, 3000])
def test_relabelling_after_the_copies_rebuilds_every_copy(ctx, oracle, dim):
    """a set without row masks (fewer than 256 columns, or a row that stores a column twice) whose copies were built
    with a named panel size has neither masks nor a column stream; relabelling it afterwards still has to rebuild
    the row-panel copy and the cells, which hold ids in the old numbering"""
    raw = make_synth_csr(900, dim, 6, seed=3)
    if dim == 3000:                              # one repeated column: no row masks at any width
        b = int(raw.rowptr[1])
        col = np.insert(raw.col, b, raw.col[b])
        val = np.insert(raw.val, b, np.float32(1.0))
        rowptr = raw.rowptr.copy()
        rowptr[2:] += 1
        raw = oracle.Csr(rowptr, col, val, raw.label, dim)
    tr = oracle.translate(raw)
    rng = np.random.default_rng(1)
    x = rng.normal(0, 0.3, dim).astype(np.float32)
    z = rng.normal(0, 0.3, dim).astype(np.float32)
    f_ref, g_ref = oracle.logistic_fg(tr, dim, x, None, z, None, 0.4)
    dev = ctx.csr_upload(raw.rowptr, raw.col, raw.val, raw.label, dim, add_bias=True)
    dev.build_transposed(panel_rows=256, col_block=64).relabel_by_frequency()
    g = ctx.vec(dim)
    f = dev.logistic_fg(ctx.vec(dim, x), g, z=ctx.vec(dim, z), rho=0.4)
    close(f, f_ref, 1e-5, 1e-7)
    close(g.download(), g_ref, 2e-5, 2e-5 * np.abs(g_ref).max())
    dev.destroy()


def _loop_on_the_device(ctx, parts_train, raws_test, conf, dim, relabel, cb, linesearch=0):
    """the loop of src/admm_allreduce.cpp:55-91 for len(parts_train) ranks held in one process, the all-reduce a
    host sum in rank order (as the restatement does); parts_train[r] lists rank r's data parts"""
    from admm_amd import driver
    nrank = len(parts_train)
    prm = driver.AdmmConfig()
    prm.step_size, prm.global_var, prm.ftrl_alpha = conf.step_size, conf.global_var, conf.ftrl_alpha
    prm.passes, prm.dim, prm.global_l1 = conf.passes, dim, conf.global_l1
    prm.local_solver, prm.ftrl_batch = conf.local_solver, conf.ftrl_batch
    prm.lbfgs.update(conf.lbfgs)
    prm.lbfgs["linesearch"] = linesearch       # the restatement's solver is the one trajectory all modes follow
    workers, trains, tests, zs, tmps = [], [], [], [], []
    for r in range(nrank):
        w = driver.Worker(ctx)
        w.InitWorker(dim)
        workers.append(w)
        sets = []
        for raw in parts_train[r]:
            t = ctx.csr_upload(raw.rowptr, raw.col, raw.val, raw.label, dim, add_bias=True)
            if relabel == 1:
                t.relabel_by_frequency()
            if prm.local_solver == "lbfgs" or prm.ftrl_batch > 1:
                t.build_transposed(panel_rows=prm.ftrl_batch if prm.ftrl_batch > 1 else 0, col_block=cb)
            if relabel == 2:
                t.relabel_by_frequency()
            sets.append(t)
        trains.append(sets)
        tests.append(ctx.csr_upload(raws_test[r].rowptr, raws_test[r].col, raws_test[r].val, raws_test[r].label, dim,
                                    add_bias=True))
        zs.append(ctx.vec(dim))
        tmps.append(ctx.vec(dim))
    records = []
    for it in range(prm.passes):
        stats = []
        for r in range(nrank):
            if it == prm.passes - 1:
                workers[r].load = True
            stats.append(workers[r].BaseUpdate(trains[r], prm, zs[r]))
            workers[r].GetWeights(prm, tmps[r])
        total = np.zeros(dim, np.float32)
        for r in range(nrank):
            total = (total + tmps[r].download()).astype(np.float32)
        lines, res = [], []
        for r in range(nrank):
            tmps[r].upload(total)
            lines.append(tests[r].metrics([workers[r].base_vec_, workers[r].bias_vec_]))
            res.append(ctx.consensus(tmps[r], nrank, prm.step_size, prm.global_var, prm.global_l1,
                                     workers[r].base_vec_, workers[r].langr_vec_, zs[r]))
        records.append(dict(z=zs[0].download(), x=[w.base_vec_.download() for w in workers],
                            y=[w.langr_vec_.download() for w in workers], lines=lines, residuals=res, stats=stats))
    for sets in trains:
        for t in sets:
            t.destroy()
    for t in tests:
        t.destroy()
    return records


@pytest.mark.parametrize("seed", range(int(os.environ.get("ADMM_FUZZ_SEEDS", "24")) // 2))
def test_admm_loop(ctx, oracle, seed):
    """three ADMM iterations -- one to three ranks, one to three data parts per rank, the reference's row-by-row
    FTRL x-update, its mini-batch form, or the L-BFGS solver in any line search mode -- against the restatement of
    the loop: consensus weights, every rank's x and dual, the printed logloss, the residuals"""
    rng = np.random.default_rng(31000 + seed)
    dim = int(rng.choice([5, 64, 257, 300, 2500, 9000]))
    nrank = int(rng.integers(1, 4))
    solver = str(rng.choice(["ftrl", "ftrl", "ftrl-batch", "lbfgs", "lbfgs"]))
    nparts = 1 if solver == "lbfgs" else int(rng.integers(1, 4))
    mean = float(rng.choice([3, 12, 30]))
    binary = bool(rng.integers(0, 2))
    parts = [[make_synth_csr(int(rng.integers(40, 1500)), dim, min(mean, (dim - 1) / 2), seed=1000 * seed + 10 * r + k,
                             binary=binary) for k in range(nparts)] for r in range(nrank)]
    raws_test = [make_synth_csr(int(rng.integers(20, 400)), dim, min(mean, (dim - 1) / 2), seed=1000 * seed + 500 + r,
                                binary=binary) for r in range(nrank)]
    kw = dict(step_size=float(rng.choice([0.01, 0.5, 5.0])), global_var=float(rng.choice([1.0, 4.0])), ftrl_alpha=0.1,
              passes=3, dim=dim, bias_var=0.01, global_l1=float(rng.choice([0.0, 0.05])))
    if solver == "ftrl-batch":
        kw.update(ftrl_batch=int(rng.choice([256, 512])))
    if solver == "lbfgs":
        kw.update(local_solver="lbfgs", lbfgs=dict(max_iter=int(rng.integers(1, 5)), epsilon=1e-7))
    conf = oracle.AdmmConf(**kw)
    _, _, ref = oracle.admm_run([[oracle.translate(t) for t in ps] for ps in parts],
                                [oracle.translate(t) for t in raws_test], conf)
    cb = min(int(rng.choice([64, 128, 2048])), max(64, (dim // 64) * 64))
    got = _loop_on_the_device(ctx, parts, raws_test, conf, dim, int(rng.integers(0, 3)), cb, int(rng.integers(0, 3)))
    # later iterations inherit the earlier ones' rounding: the stated tolerance on the first, a looser one after
    for it, (a, b) in enumerate(zip(got, ref)):
        rtol, scale = (1e-5, 2e-6) if it == 0 and solver != "lbfgs" else (1e-3, 1e-3)
        for va, vb in [(a["z"], b["z"])] + list(zip(a["x"], b["x"])) + list(zip(a["y"], b["y"])):
            close(va, vb, rtol, scale * max(1.0, np.abs(vb).max()))
        for (la, aa, na), (lb, ab, nb) in zip(a["lines"], b["lines64"]):
            assert na == nb
            assert la == pytest.approx(lb, rel=1e-5 if it == 0 and solver != "lbfgs" else 1e-3)
        for (r2a, s2a), (r2b, s2b) in zip(a["residuals"], b["residuals"]):
            assert r2a == pytest.approx(r2b, rel=1e-2, abs=1e-10)
            assert s2a == pytest.approx(s2b, rel=1e-2, abs=1e-12)
        # the first solve starts far from its minimum and follows the restatement's evaluation for evaluation; a
        # later one may end where the objective moves by less than its float per-row terms resolve, and a line
        # search there takes as many trial points as the rounding decides
        if solver == "lbfgs" and it == 0:
            for sa, sb in zip(a["stats"], b["stats"]):
                assert (sa["evals"], sa["iters"]) == (sb["evals"], sb["iters"]), (it, sa, sb)


def _sections(path):
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


@pytest.mark.parametrize("seed", range(max(2, int(os.environ.get("ADMM_FUZZ_SEEDS", "24")) // 8)))
def test_admm_program_over_configurations(tmp_path, oracle, seed):
    """the C++ program (conf grammar, byte-range parts of one file, report lines, state files) on seeded
    configurations against the restatement's loop"""
    import re
    import subprocess
    from conftest import ROOT, write_libsvm
    rng = np.random.default_rng(52000 + seed)
    dim = int(rng.choice([6, 65, 257, 300, 3000]))
    nparts = int(rng.integers(1, 4))
    solver = str(rng.choice(["ftrl", "ftrl", "ftrl-batch", "lbfgs"]))
    binary = bool(rng.integers(0, 2))
    mean = min(float(rng.choice([3, 12])), (dim - 1) / 2)
    raw_train = make_synth_csr(int(rng.integers(60, 2000)), dim, mean, seed=7000 + seed, binary=binary)
    raw_test = make_synth_csr(int(rng.integers(30, 500)), dim, mean, seed=8000 + seed, binary=binary)
    write_libsvm(str(tmp_path / "train" / "A_aggregated" / "part-00000"), raw_train)
    write_libsvm(str(tmp_path / "test" / "A_aggregated" / "part-00000"), raw_test)
    os.makedirs(tmp_path / "out")
    step, gvar, passes = float(rng.choice([0.01, 0.5])), float(rng.choice([1, 4])), int(rng.integers(1, 4))
    l1 = float(rng.choice([0.0, 0.05]))
    text = "step_size = %g\nbias_var = 0.01\nglobal_var = %g\nftrl_alpha = 0.1\npasses = %d\ndim = %d\n" % (step, gvar, passes, dim)
    kw = dict(step_size=step, global_var=gvar, ftrl_alpha=0.1, passes=passes, dim=dim, bias_var=0.01, global_l1=l1)
    if l1:
        text += "global_l1 = %g\n" % l1
    if solver == "ftrl-batch":
        text += "ftrl_batch = 256\n"
        kw.update(ftrl_batch=256)
    if solver == "lbfgs":
        it = int(rng.integers(1, 4))
        text += "local_solver = lbfgs\nlbfgs_max_iter = %d\nlbfgs_linesearch = %d\n" % (it, int(rng.integers(0, 3)))
        kw.update(local_solver="lbfgs", lbfgs=dict(max_iter=it))
    conf = tmp_path / "admm.conf"
    conf.write_text(text + "A : %d\n" % nparts)
    run = subprocess.run([os.path.join(ROOT, "admm_amd", "bin", "admm"), str(conf), str(tmp_path / "train"),
                          str(tmp_path / "test"), str(tmp_path / "out")], capture_output=True, text=True, timeout=110)
    assert run.returncode == 0, run.stdout + run.stderr
    data = open(tmp_path / "train" / "A_aggregated" / "part-00000", "rb").read()
    if solver == "lbfgs":     # the batch objective takes the file's parts as one set
        parts = [oracle.translate(oracle.parse_libsvm_text(data, 0, 1))]
    else:
        parts = [oracle.translate(oracle.parse_libsvm_text(data, p, nparts)) for p in range(nparts)]
    for p in parts:
        p.dim = dim
    test = oracle.translate(oracle.parse_libsvm(str(tmp_path / "test" / "A_aggregated" / "part-00000")))
    workers, zg, rec = oracle.admm_run([parts], [test], oracle.AdmmConf(**kw))
    lines = re.findall(r"^A \| ([0-9.]+) \| ([0-9.]+|-?nan|-?inf) \| (\d+)$", run.stdout, flags=re.M)
    assert len(lines) == passes, run.stdout
    for it, ((l, a, n), r) in enumerate(zip(lines, rec)):
        assert int(n) == test.nrows
        assert float(l) == pytest.approx(r["lines64"][0][0], rel=2e-5 if it == 0 else 1e-3)
    x, bias, y, nt, zt = _sections(tmp_path / "out" / "admm_weight_A")
    tol = 2e-5 if passes == 1 and solver != "lbfgs" else 1e-3
    for got, ref in ((x, workers[0].x), (y, workers[0].y)):
        np.testing.assert_allclose(got, ref, rtol=tol, atol=tol * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("seed", range(int(os.environ.get("ADMM_FUZZ_SEEDS", "24"))))
def test_tied_ranks_and_auc_on_arrays(ctx, oracle, seed):
    """Metrics::Sort / Metrics::Auc (src/metrics.cc:14-89) on bare arrays: sizes around a wave, a workgroup and the
    sort's tile, scores with few or many ties, signed zeros, infinities, labels the reference treats as positive
    (> 0) or not"""
    rng = np.random.default_rng(64000 + seed)
    n = int(rng.choice([1, 2, 3, 63, 64, 65, 255, 256, 257, 1000, 4095, 4097, 65535, 65537, 200_001]))
    decimals = int(rng.choice([0, 1, 2, 4, 7]))
    x = np.round(rng.normal(0, 1, n), decimals).astype(np.float32)
    if n > 8 and rng.integers(0, 2):
        x[rng.choice(n, 6, replace=False)] = [0.0, -0.0, np.inf, -np.inf, 0.0, np.inf]
    if rng.integers(0, 4) == 0:
        x[:] = x[0]                                   # every score the same: one tie group
    ranks = ctx.vec(n)
    ctx.tied_ranks(ctx.vec(n, x), ranks)
    np.testing.assert_array_equal(ranks.download(), oracle.tied_rank(x))
    labels = rng.choice(np.array([-1, 0, 1, 2], np.float32), n).astype(np.float32)
    ref = oracle.auc_from_scores(x, labels, f64=True)
    got = ctx.auc_scores(ctx.vec(n, x), ctx.vec(n, labels), n)
    if np.isfinite(ref):
        close(got, ref, 1e-9, 1e-12)
    else:
        assert not np.isfinite(got)                   # one class only: the reference divides by zero


@pytest.mark.parametrize("seed", range(max(2, int(os.environ.get("ADMM_FUZZ_SEEDS", "24")) // 6)))
def test_generated_sets_of_many_panels(ctx, oracle, seed):
    """generated sets large enough for several row panels, column-stream panels and (the largest) more than one
    panel of the column-block copy, short or heavy-tailed rows, with or without stored values: objective,
    gradient and a mini-batch FTRL pass against the restatement on the downloaded rows; and the lean form of the
    same set gives the same bits"""
    rng = np.random.default_rng(87000 + seed)
    rows = int(rng.choice([3000, 70_000, 300_000, 1_300_000, 2_200_000], p=[0.3, 0.3, 0.2, 0.1, 0.1]))
    dim = int(rng.choice([4096, 1 << 16, 1 << 20, 1 << 24]))
    mean = float(rng.choice([8, 40]))
    kw = dict(len_dist=1, tail_sigma=float(rng.choice([0.8, 1.2])), max_len=1024) if rng.integers(0, 2) else {}
    explicit = bool(rng.integers(0, 2))
    order = int(rng.integers(0, 3))
    cb = int(rng.choice([1024, 4096, 16384]))
    cb = max(min(cb, dim), dim // 8192)
    # (at most nine batches: over tens of batches of this size the pass amplifies last-bit differences -- the
    # restatement run twice, the second time with offsets perturbed in their last bit, then disagrees with itself
    # as much as with the device)
    named = int(rng.choice([0, 0, 65536 if rows <= 300_000 else 262144])) if rows >= 70_000 else 0

    def make(lean):
        s = ctx.csr_synth(rows, dim, mean, seed=seed, explicit_values=explicit, **kw)
        if lean:
            s.set_lean()
        if order == 1:
            s.relabel_by_frequency()
        s.build_transposed(panel_rows=named, col_block=cb)
        if order == 2:
            s.relabel_by_frequency()
        return s

    dev = make(False)
    tr = oracle.Csr(*dev.download(), dim)
    x = rng.normal(0, 0.1, dim).astype(np.float32)
    off = rng.normal(0, 0.02, dim).astype(np.float32)
    z = rng.normal(0, 0.1, dim).astype(np.float32)
    f_ref, g_ref = oracle.logistic_fg(tr, dim, x, off, z, None, 0.05)
    dx, doff, dz, g = ctx.vec(dim, x), ctx.vec(dim, off), ctx.vec(dim, z), ctx.vec(dim)
    f = dev.logistic_fg(dx, g, offset=doff, z=dz, rho=0.05)
    close(f, f_ref, 1e-5, 1e-7)
    close(g.download(), g_ref, 2e-5, 2e-5 * np.abs(g_ref).max())
    lean = make(True)
    g2 = ctx.vec(dim)
    assert lean.logistic_fg(dx, g2, offset=doff, z=dz, rho=0.05) == f
    np.testing.assert_array_equal(g2.download(), g.download())
    lean.destroy()
    if named and cb <= 8192:
        hw, hz, hn = (np.zeros(dim, np.float32) for _ in range(3))
        oracle.ftrl_run(tr, hw, hz, hn, off, z, 0.05, 1.0, 0.1, 1.0, batch=named)
        w, zz, n = ctx.vec(dim), ctx.vec(dim), ctx.vec(dim)
        dev.ftrl_pass(w, zz, n, doff, dz, 0.05, 1.0, 0.1, 1.0, batch=named)
        for got, ref in ((zz, hz), (n, hn)):
            close(got.download(), ref, 5e-5, 2e-6 * max(1.0, np.abs(ref).max()))
        # the reference's lazy weight jumps where |z| crosses l_1 (src/ftrl.cc:61-70 with its sign convention): of a
        # million columns one may sit within a last bit of that threshold when it is last looked at
        dw = np.abs(w.download() - hw)
        assert int((dw > 5e-5 * np.abs(hw) + 2e-6 * max(1.0, np.abs(hw).max())).sum()) <= 2
        for v in (w, zz, n):
            v.free()
    dev.destroy()
    for v in (dx, doff, dz, g, g2):
        v.free()
