"""Host-side mirror of the reference's ADMM classes on top of the C ABI (Python flavour; the C++
flavour used by the admm / ftrl / evan programs lives in admm_amd/host/).

Names, argument meaning and call order follow src/workers.{h,cc}, src/master.{h,cc},
src/config.h, src/arg_parser.{h,cc} and the main loop of src/admm_allreduce.cpp of the reference;
all arithmetic happens in libadmm_hip.so on the GPU.
"""
import numpy as np

from . import capi


class AdmmConfig:
    """src/config.h:11-34 plus the keys this build adds (all optional in a conf file):
    global_l1 (soft threshold of the consensus update, 0 = the reference's ridge update),
    local_solver (ftrl | lbfgs), ftrl_batch, lbfgs_* ."""

    def __init__(self):
        self.step_size = 0.0
        self.global_var = 0.0
        self.bias_var = 0.0
        self.ftrl_alpha = 0.0
        self.dim = 0
        self.passes = 0
        self.train_path = ""
        self.test_path = ""
        self.output_path = ""
        self.global_l1 = 0.0
        self.local_solver = "ftrl"
        self.ftrl_batch = 1
        self.lbfgs = dict(m=6, max_iter=10, max_linesearch=20, epsilon=1e-5, ftol=1e-4, wolfe=0.9, linesearch=0)


class FtrlConfig:
    """src/config.h:36-58"""

    def __init__(self, admm_params=None):
        self.alpha = admm_params.ftrl_alpha if admm_params else 0.0
        self.beta = 1.0 if admm_params else 0.0
        self.l_1 = 0.0
        self.l_2 = 0.0
        self.dim = admm_params.dim if admm_params else 0
        self.passes = 0
        self.train_path = self.test_path = self.output_path = ""


def _tokens(path):
    with open(path) as f:
        return f.read().split()


class ArgParser:
    """src/arg_parser.cc:10-90: whitespace-separated tokens; `key = value` fills the dictionary,
    `psid : num_part` appends one worker line (rank r uses line r)."""

    def __init__(self):
        self.psid, self.num_part, self.load_path = [], [], ""

    def _parse(self, path):
        d = {}
        try:
            tok = _tokens(path)
        except OSError:
            print("config file open failed. ")
            return d
        i = 0
        while i + 1 < len(tok):
            key, val = tok[i], tok[i + 1]
            i += 2
            if val == "=" and i < len(tok):
                d[key] = tok[i]
                i += 1
            elif val == ":" and i < len(tok):
                self.psid.append(key)
                self.num_part.append(_atoi(tok[i]))
                i += 1
        return d

    def ADMMParse(self, path, admm_params):
        d = self._parse(path)
        admm_params.step_size = np.float32(_atof(d.get("step_size", "")))
        admm_params.bias_var = np.float32(_atof(d.get("bias_var", "")))
        admm_params.global_var = np.float32(_atof(d.get("global_var", "")))
        admm_params.ftrl_alpha = np.float32(_atof(d.get("ftrl_alpha", "")))
        admm_params.passes = _atoi(d.get("passes", ""))
        admm_params.dim = _atoi(d.get("dim", ""))
        # optional keys of this build
        for k in ("train_path", "test_path", "output_path"):
            if k in d:
                setattr(admm_params, k, d[k])
        if "global_l1" in d:
            admm_params.global_l1 = np.float32(_atof(d["global_l1"]))
        if "local_solver" in d:
            admm_params.local_solver = d["local_solver"]
        if "ftrl_batch" in d:
            admm_params.ftrl_batch = max(1, _atoi(d["ftrl_batch"]))
        for k in ("m", "max_iter", "max_linesearch", "linesearch"):
            if "lbfgs_" + k in d:
                admm_params.lbfgs[k] = _atoi(d["lbfgs_" + k])
        for k in ("epsilon", "ftol", "wolfe"):
            if "lbfgs_" + k in d:
                admm_params.lbfgs[k] = _atof(d["lbfgs_" + k])
        if "load_path" in d:
            self.load_path = d["load_path"]

    def FTRLParse(self, path, ftrl_params):
        d = self._parse(path)
        ftrl_params.train_path = d.get("train_path", "")
        ftrl_params.test_path = d.get("test_path", "")
        ftrl_params.output_path = d.get("output_path", "")
        ftrl_params.alpha = np.float32(_atof(d.get("ftrl_alpha", "")))
        ftrl_params.beta = np.float32(_atof(d.get("ftrl_beta", "")))
        ftrl_params.l_1 = np.float32(_atof(d.get("l_1", "")))
        ftrl_params.l_2 = np.float32(_atof(d.get("l_2", "")))
        ftrl_params.passes = _atoi(d.get("passes", ""))
        ftrl_params.dim = _atoi(d.get("dim", ""))
        if "load_path" in d:
            self.load_path = d["load_path"]


def _atof(s):
    """C atof: longest valid numeric prefix, 0 when there is none."""
    import re
    m = re.match(r"\s*[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?)", s or "")
    return float(m.group(0)) if m else 0.0


def _atoi(s):
    import re
    m = re.match(r"\s*[+-]?\d+", s or "")
    return int(m.group(0)) if m else 0


class Worker:
    """src/workers.h:16-82.  base_vec_ (x), bias_vec_, langr_vec_ (y), nt_vec_, zt_vec_ are device
    vectors; the global weights z live in the caller's `global_weights` device vector."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.psid_, self.num_part_, self.partid_ = "", 1, 0
        self.load = False

    def InitWorker(self, fdim):
        c = self.ctx
        self.dim = int(fdim)
        self.base_vec_ = c.vec(fdim)
        self.bias_vec_ = c.vec(fdim)
        self.langr_vec_ = c.vec(fdim)
        self.nt_vec_ = c.vec(fdim)
        self.zt_vec_ = c.vec(fdim)
        # FTRL solver state of one x-update (FtrlSolver weight_/mid_weight_/squared_sum_)
        self._n = c.vec(fdim)
        self._z = c.vec(fdim)
        self.load = False

    def BaseUpdate(self, train_parts, admm_params, global_weights):
        """src/workers.cc:51-87.  train_parts: the worker's data parts as capi.SampleSet."""
        if admm_params.local_solver == "lbfgs":
            # the batch objective is a sum over all rows: a file's parts are loaded as ONE resident set
            # for this solver (host/solvers.cc does the same), whatever num_part says
            assert len(train_parts) == 1, "the L-BFGS local solver takes the shard as one resident set"
            return train_parts[0].lbfgs_solve(self.base_vec_, offset=self.bias_vec_, z=global_weights,
                                              ydual=self.langr_vec_, rho=admm_params.step_size,
                                              **admm_params.lbfgs)
        # FtrlSolver::Init zeroes w; z, n start from zt_vec_, nt_vec_ (src/workers.cc:64-67)
        self.base_vec_.zero()
        self._n.copy_from(self.nt_vec_)
        self._z.copy_from(self.zt_vec_)
        for part in train_parts:  # src/workers.cc:73-77, l_2 = 200, l_1 = 1, beta = 1
            part.ftrl_pass(self.base_vec_, self._z, self._n, self.bias_vec_, global_weights,
                           admm_params.ftrl_alpha, 1.0, 1.0, 200.0, batch=admm_params.ftrl_batch)
        if self.load:  # src/workers.cc:81-85
            self.nt_vec_.copy_from(self._n)
            self.zt_vec_.copy_from(self._z)
            self.load = False
        return dict(evals=len(train_parts))

    def BiasUpdate(self, train_parts, admm_params):
        """src/workers.cc:89-113 (its call is commented out in the reference's main loop,
        src/admm_allreduce.cpp:65): one FTRL pass from zero state with l_2 = 200, l_1 = 1000, the base
        model as the fixed offset, no proximal centre; bias_vec_ <- the pass's weights."""
        assert admm_params.local_solver != "lbfgs", "the bias model is an FTRL model"
        fdim = self.base_vec_.n
        w, z, n, zero = (self.ctx.vec(fdim) for _ in range(4))
        for part in train_parts:
            part.ftrl_pass(w, z, n, self.base_vec_, zero, admm_params.ftrl_alpha, 1.0, 1000.0, 200.0,
                           batch=admm_params.ftrl_batch)
        self.bias_vec_.copy_from(w)

    def LangrangeUpdate(self, admm_params, global_weights):
        """src/workers.cc:115-119"""
        self.ctx.dual_update(self.langr_vec_, self.base_vec_, global_weights, admm_params.step_size)

    def GetWeights(self, admm_params, out):
        """src/workers.cc:121-125"""
        self.ctx.payload(self.base_vec_, self.langr_vec_, admm_params.step_size, out)


def base_update_multi(workers, parts_of, admm_params, global_weights):
    """Worker.BaseUpdate (FTRL local solver, row-by-row passes) of several workers that live on one
    GPU, with the workers' passes over their k-th parts launched together.  parts_of[i] lists
    worker i's data parts.  Every worker ends in exactly the state its own BaseUpdate gives."""
    assert admm_params.local_solver == "ftrl" and admm_params.ftrl_batch == 1
    ctx = workers[0].ctx
    for wk in workers:
        wk.base_vec_.zero()
        wk._n.copy_from(wk.nt_vec_)
        wk._z.copy_from(wk.zt_vec_)
    for k in range(max(len(p) for p in parts_of)):
        idx = [i for i, p in enumerate(parts_of) if k < len(p)]
        ctx.ftrl_pass_multi([parts_of[i][k] for i in idx], [workers[i].base_vec_ for i in idx],
                            [workers[i]._z for i in idx], [workers[i]._n for i in idx],
                            [workers[i].bias_vec_ for i in idx], [global_weights] * len(idx),
                            admm_params.ftrl_alpha, 1.0, 1.0, 200.0)
    for wk in workers:
        if wk.load:
            wk.nt_vec_.copy_from(wk._n)
            wk.zt_vec_.copy_from(wk._z)
            wk.load = False
    return dict(evals=sum(len(p) for p in parts_of))


class Master:
    """src/master.h:12-17"""

    def __init__(self, ctx):
        self.ctx = ctx

    def GlobalUpdate(self, workers_sum, admm_params, global_weights, num_worker):
        self.ctx.global_update(workers_sum, num_worker, admm_params.step_size, admm_params.global_var,
                               admm_params.global_l1, global_weights)
        return True


def admm_iteration(worker, master, train_parts, test_set, admm_params, global_weights, tmp,
                   allreduce, world, last, fused_consensus=True):
    """One trip of the loop in src/admm_allreduce.cpp:55-91 for one rank.
    `allreduce(tmp)` sums the payload over ranks in place (rabit::Allreduce<op::Sum>, :69)."""
    if last:
        worker.load = True                                        # :60-61
    stats = worker.BaseUpdate(train_parts, admm_params, global_weights)   # :66
    worker.GetWeights(admm_params, tmp)                           # :67
    allreduce(tmp)                                                # :69
    line = None
    if test_set is not None:                                      # :76-81
        loss, auc, n = test_set.metrics([worker.base_vec_, worker.bias_vec_])
        line = (loss, auc, n)
    if fused_consensus:                                           # :84-85 in one kernel
        r2, s2 = worker.ctx.consensus(tmp, world, admm_params.step_size, admm_params.global_var,
                                      admm_params.global_l1, worker.base_vec_, worker.langr_vec_,
                                      global_weights)
    else:
        master.GlobalUpdate(tmp, admm_params, global_weights, world)
        worker.LangrangeUpdate(admm_params, global_weights)
        r2 = s2 = None
    return dict(line=line, r2=r2, s2=s2, stats=stats)


# floats appended to the payload buffer of AdmmLoop: this rank's r2 and s2 of the previous iteration, each
# as a (high, low) pair of floats, so that the iteration's one all-reduce also sums the residuals
RESIDUAL_TAIL = 4


def _split(v):
    hi = np.float32(v)
    return hi, np.float32(v - float(hi))


class AdmmLoop:
    """The loop of src/admm_allreduce.cpp:55-91 for one rank, with the primal and dual residuals of ALL
    ranks as a by-product of its one all-reduce: `payload` holds dim + RESIDUAL_TAIL floats, the tail
    carries this rank's r2 = |x - z|^2 and s2 = rho^2 |z - z_prev|^2 of the PREVIOUS iteration (they
    need the z that the all-reduce produces, so they travel one iteration behind), and after the
    all-reduce the tail holds their sums over ranks.  `allreduce(payload)` sums the whole buffer in
    place.  global_residuals[k] = (sum over ranks of r2, s2) of iteration k."""

    def __init__(self, worker, master, train_parts, test_set, admm_params, global_weights, payload, allreduce, world):
        self.worker, self.master, self.parts, self.test = worker, master, train_parts, test_set
        self.prm, self.zg, self.payload, self.allreduce, self.world = admm_params, global_weights, payload, allreduce, world
        self.dim = worker.dim
        assert payload.n >= self.dim + RESIDUAL_TAIL
        self.tail = capi.DeviceVec.from_ptr(worker.ctx, payload.ptr.value + 4 * self.dim, RESIDUAL_TAIL)
        self.pending = None
        self.global_residuals = []

    def _tail_up(self):
        t = np.zeros(RESIDUAL_TAIL, np.float32)
        if self.pending is not None:
            t[0], t[1] = _split(self.pending[0])
            t[2], t[3] = _split(self.pending[1])
        self.tail.upload(t)

    def _tail_down(self):
        if self.pending is None:
            return
        t = self.tail.download().astype(np.float64)
        self.global_residuals.append((t[0] + t[1], (t[2] + t[3]) / self.world))   # every rank holds the same s2
        self.pending = None

    def iteration(self, last=False):
        wk, prm = self.worker, self.prm
        if last:
            wk.load = True                                        # :60-61
        stats = wk.BaseUpdate(self.parts, prm, self.zg)           # :66
        wk.GetWeights(prm, self.payload)                          # :67 (the first dim floats)
        self._tail_up()
        self.allreduce(self.payload)                              # :69
        self._tail_down()
        line = None
        if self.test is not None:                                 # :76-81
            line = self.test.metrics([wk.base_vec_, wk.bias_vec_])
        r2, s2 = wk.ctx.consensus(self.payload, self.world, prm.step_size, prm.global_var, prm.global_l1,
                                  wk.base_vec_, wk.langr_vec_, self.zg)        # :84-85 in one kernel
        self.pending = (r2, s2)
        return dict(line=line, r2=r2, s2=s2, stats=stats)

    def flush_residuals(self):
        """one more (small) exchange for the last iteration's pair; returns the last iteration's
        residual norms over all ranks"""
        if self.pending is not None:
            self.payload.zero()
            self._tail_up()
            self.allreduce(self.payload)
            self._tail_down()
        if not self.global_residuals:
            return None
        r2, s2 = self.global_residuals[-1]
        return {"primal_r": float(np.sqrt(r2)), "dual_s": float(np.sqrt(s2)), "r2_sum_over_ranks": r2, "s2": s2,
                "iterations_recorded": len(self.global_residuals)}


def global_test_metrics(ctx, test_set, groups):
    """Logloss and AUC over the test rows of ALL ranks (the reference reports them per rank only,
    src/admm_allreduce.cpp:76-81), on the library's own all-reduce (ctx.comm_init; one rank: no
    exchange): row counts and loss sums travel as floats in two halves each, every rank predicts into
    its slice of a zero-filled buffer and copies its labels into a second one, a sum all-reduce
    concatenates them, and every rank then runs the device AUC (src/metrics.cc:14-89) once over the
    whole.  Collective: every rank calls it.  Returns (logloss, auc, rows).  Same steps as
    Metrics::AllRanks of the C++ programs."""
    rank, world = ctx.rank, ctx.world
    n = test_set.nrows
    lsum = 0.0
    if n:
        loss_local, _, _ = test_set.metrics(groups, want_auc=False)
        lsum = loss_local * n
    meta = np.zeros(4 * world, np.float32)
    meta[4 * rank + 0] = n >> 20
    meta[4 * rank + 1] = n & 0xFFFFF
    meta[4 * rank + 2] = np.float32(lsum)
    meta[4 * rank + 3] = np.float32(lsum - float(meta[4 * rank + 2]))
    dmeta = ctx.vec(4 * world, meta)
    ctx.allreduce_sum(dmeta)
    meta = dmeta.download().astype(np.float64)
    dmeta.free()
    counts = [(int(meta[4 * r]) << 20) + int(meta[4 * r + 1]) for r in range(world)]
    total, off = sum(counts), sum(counts[:rank])
    ltotal = float(sum(meta[4 * r + 2] + meta[4 * r + 3] for r in range(world)))
    half = max(total, 1)
    buf = ctx.vec(2 * half)                      # zero-filled on the context's stream: [scores | labels]
    base = buf.ptr.value
    if n:
        test_set.predict_into(groups, capi.DeviceVec.from_ptr(ctx, base + 4 * off, n))
        test_set.copy_labels(capi.DeviceVec.from_ptr(ctx, base + 4 * (half + off), n))
    ctx.allreduce_sum(buf)
    auc = ctx.auc_scores(capi.DeviceVec.from_ptr(ctx, base, total),
                         capi.DeviceVec.from_ptr(ctx, base + 4 * half, total), total) if total else float("nan")
    buf.free()
    return (ltotal / total if total else 0.0), auc, total


def ftrl_exact_vs_minibatch(ctx, rows, dim, mean_nnz, seed, prm, batches=(4096, 65536), test_rows=0, col_block=2048,
                            **synth_kw):
    """One FTRL pass (src/ftrl.cc:101-145) over the same generated rows in the reference's row-by-row form and in
    the mini-batch form for each size in `batches`, each from zero state; test logloss / AUC of every resulting
    model on one generated test set.  prm = (alpha, beta, l_1, l_2).  Returns {"exact": {...}, "batch": {size:
    {..., "dlogloss", "dauc"}}}: how much of the model's quality the batched semantics cost."""
    test = ctx.csr_synth(test_rows or max(1, rows // 20), dim, mean_nnz, seed=seed + 1, **synth_kw)
    off, reg = ctx.vec(dim), ctx.vec(dim)

    def one(batch):
        train = ctx.csr_synth(rows, dim, mean_nnz, seed=seed, **synth_kw)
        if batch > 1:
            train.relabel_by_frequency()
            train.build_transposed(panel_rows=batch, col_block=col_block)
        w, z, n = (ctx.vec(dim) for _ in range(3))
        train.ftrl_pass(w, z, n, off, reg, *prm, batch=batch)
        ll, auc, nr = test.metrics([w])
        train.destroy()
        for v in (w, z, n):
            v.free()
        return {"test_logloss": ll, "test_auc": auc}

    out = {"rows": rows, "test_rows": test.nrows, "exact": one(1), "batch": {}}
    for b in batches:
        rec = one(int(b))
        rec["dlogloss"] = rec["test_logloss"] - out["exact"]["test_logloss"]
        rec["dauc"] = rec["test_auc"] - out["exact"]["test_auc"]
        out["batch"][int(b)] = rec
    test.destroy()
    off.free()
    reg.free()
    return out


def format_line(psid, loss, auc, n):
    """the per-iteration report of src/admm_allreduce.cpp:79"""
    return "%s | %.7f | %.7f | %d" % (psid, loss, auc, n)
