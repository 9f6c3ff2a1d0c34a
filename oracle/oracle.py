"""ctypes front end of the CPU restatement in admm_oracle.c, plus the host-level loops of the
reference restated in Python on top of it.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under admm_amd/ imports this module.

Reference citations are relative to /root/reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libadmm_oracle.so")

_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile libadmm_oracle.so next to its source (gcc, plain C)."""
    src = os.path.join(_HERE, "admm_oracle.c")
    hdr = os.path.join(_HERE, "admm_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libadmm_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _opt(arr):
    """numpy float32 array or None -> void pointer"""
    if arr is None:
        return None
    assert arr.dtype == np.float32 and arr.flags.c_contiguous
    return arr.ctypes.data_as(C.c_void_p)


def _declare(L):
    vp = C.c_void_p
    L.orc_translate.argtypes = [_u64p, _u32p, vp, C.c_uint64, _u64p, _u32p, _f32p]
    L.orc_translate.restype = None
    L.orc_ftrl_run.argtypes = [_u64p, _u32p, vp, _f32p, C.c_uint64, _f32p, _f32p, _f32p, _f32p,
                               _f32p, C.c_float, C.c_float, C.c_float, C.c_float]
    L.orc_ftrl_run.restype = None
    L.orc_ftrl_run_batch.argtypes = [_u64p, _u32p, vp, _f32p, C.c_uint64, _f32p, _f32p, _f32p,
                                     _f32p, _f32p, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.c_uint64, _f64p, _f64p]
    L.orc_ftrl_run_batch.restype = None
    L.orc_ftrl_predict.argtypes = [_u32p, vp, C.c_uint64, _f32p, _f32p, _f32p, _f32p, _f32p,
                                   C.c_float, C.c_float, C.c_float, C.c_float]
    L.orc_ftrl_predict.restype = C.c_float
    L.orc_ftrl_update.argtypes = [C.c_float, C.c_float, _u32p, C.c_uint64, _f32p, _f32p, _f32p,
                                  C.c_float]
    L.orc_ftrl_update.restype = None
    pp = C.POINTER(C.c_void_p)
    L.orc_logloss.argtypes = [_u64p, _u32p, vp, _f32p, C.c_uint64, pp, C.c_int]
    L.orc_logloss.restype = C.c_float
    L.orc_logloss_sum_f64.argtypes = [_u64p, _u32p, vp, _f32p, C.c_uint64, pp, C.c_int, vp]
    L.orc_logloss_sum_f64.restype = C.c_double
    L.orc_auc.argtypes = [_u64p, _u32p, vp, _f32p, C.c_uint64, pp, C.c_int]
    L.orc_auc.restype = C.c_float
    L.orc_tied_rank.argtypes = [_f32p, C.c_int, _f32p]
    L.orc_tied_rank.restype = None
    L.orc_auc_from_scores.argtypes = [_f32p, _f32p, C.c_int]
    L.orc_auc_from_scores.restype = C.c_float
    L.orc_auc_from_scores_f64.argtypes = [_f32p, _f32p, C.c_int]
    L.orc_auc_from_scores_f64.restype = C.c_double
    L.orc_payload.argtypes = [_f32p, _f32p, C.c_float, _f32p, C.c_uint64]
    L.orc_payload.restype = None
    L.orc_global_update.argtypes = [_f32p, C.c_int, C.c_float, C.c_float, C.c_float, _f32p,
                                    C.c_uint64]
    L.orc_global_update.restype = None
    L.orc_global_update_commented.argtypes = [_f32p, C.c_int, C.c_float, C.c_float, _f32p,
                                              C.c_uint64]
    L.orc_global_update_commented.restype = None
    L.orc_dual_update.argtypes = [_f32p, _f32p, _f32p, C.c_float, C.c_uint64]
    L.orc_dual_update.restype = None
    L.orc_residuals.argtypes = [_f32p, _f32p, _f32p, C.c_float, C.c_uint64,
                                C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.orc_residuals.restype = None
    L.orc_logistic_fg.argtypes = [_u64p, _u32p, vp, _f32p, C.c_uint64, C.c_uint64, _f32p, vp, vp,
                                  vp, C.c_float, _f64p]
    L.orc_logistic_fg.restype = C.c_double
    L.orc_part_range.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32,
                                 C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.orc_part_range.restype = None
    L.orc_parse_libsvm.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint64), vp, vp, vp, vp]
    L.orc_parse_libsvm.restype = C.c_int


class Csr:
    """Host CSR sample set: rowptr u64, col u32, val f32 (or None = implicit 1), label f32."""

    def __init__(self, rowptr, col, val, label, dim=None):
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
        self.col = np.ascontiguousarray(col, dtype=np.uint32)
        self.val = None if val is None else np.ascontiguousarray(val, dtype=np.float32)
        self.label = np.ascontiguousarray(label, dtype=np.float32)
        self.nrows = int(self.rowptr.shape[0] - 1)
        self.nnz = int(self.rowptr[-1]) if self.nrows >= 0 else 0
        self.dim = dim

    def args(self):
        return (self.rowptr, self.col if self.col.size else np.zeros(1, np.uint32),
                _opt(self.val), self.label if self.label.size else np.zeros(1, np.float32),
                self.nrows)


def _groups(groups):
    arr = (C.c_void_p * len(groups))()
    for i, g in enumerate(groups):
        assert g.dtype == np.float32 and g.flags.c_contiguous
        arr[i] = g.ctypes.data
    return arr, len(groups)


# ---------------------------------------------------------------------------------------------
# sample set
# ---------------------------------------------------------------------------------------------

def parse_libsvm_text(text, part=0, nparts=1):
    """libsvm text (bytes) -> raw Csr (no bias column yet).  dmlc-core's parser is absent from the
    reference tree, so tokenisation and the part/nparts byte split are this build's definition."""
    if isinstance(text, str):
        text = text.encode()
    L = lib()
    b, e = C.c_uint64(), C.c_uint64()
    L.orc_part_range(text, len(text), part, nparts, C.byref(b), C.byref(e))
    nr, nz = C.c_uint64(), C.c_uint64()
    L.orc_parse_libsvm(text, b.value, e.value, C.byref(nr), C.byref(nz), None, None, None, None)
    rowptr = np.zeros(nr.value + 1, np.uint64)
    col = np.zeros(max(nz.value, 1), np.uint32)
    val = np.zeros(max(nz.value, 1), np.float32)
    label = np.zeros(max(nr.value, 1), np.float32)
    L.orc_parse_libsvm(text, b.value, e.value, C.byref(nr), C.byref(nz), rowptr.ctypes.data,
                       col.ctypes.data, val.ctypes.data, label.ctypes.data)
    return Csr(rowptr, col[:nz.value], val[:nz.value], label[:nr.value])


def parse_libsvm(path, part=0, nparts=1):
    with open(path, "rb") as f:
        return parse_libsvm_text(f.read(), part, nparts)


def translate(raw):
    """src/sample_set.cc:70-88: bias column 0 with value 1, indices +1."""
    L = lib()
    rowptr = np.zeros(raw.nrows + 1, np.uint64)
    nnz = raw.nnz + raw.nrows
    col = np.zeros(max(nnz, 1), np.uint32)
    val = np.zeros(max(nnz, 1), np.float32)
    L.orc_translate(raw.rowptr, raw.col if raw.col.size else np.zeros(1, np.uint32),
                    _opt(raw.val), raw.nrows, rowptr, col, val)
    return Csr(rowptr, col[:nnz], val[:nnz], raw.label, raw.dim)


# ---------------------------------------------------------------------------------------------
# FTRL
# ---------------------------------------------------------------------------------------------

def ftrl_run(csr, w, z, n, offset, reg_offset, alpha, beta, l1, l2, batch=1):
    """src/ftrl.cc:101-145 on a bias-translated Csr; w, z, n are updated in place."""
    L = lib()
    if batch == 1:
        L.orc_ftrl_run(*csr.args(), w, z, n, offset, reg_offset, alpha, beta, l1, l2)
    else:
        g1 = np.zeros(w.shape[0], np.float64)
        g2 = np.zeros(w.shape[0], np.float64)
        L.orc_ftrl_run_batch(*csr.args(), w, z, n, offset, reg_offset, alpha, beta, l1, l2,
                             batch, g1, g2)


# ---------------------------------------------------------------------------------------------
# metrics
# ---------------------------------------------------------------------------------------------

def logloss(csr, groups):
    arr, k = _groups(groups)
    return float(lib().orc_logloss(*csr.args(), arr, k))


def logloss_sum_f64(csr, groups, want_p=False):
    arr, k = _groups(groups)
    p = np.zeros(max(csr.nrows, 1), np.float32) if want_p else None
    s = float(lib().orc_logloss_sum_f64(*csr.args(), arr, k, _opt(p)))
    return (s, p[:csr.nrows]) if want_p else s


def auc(csr, groups):
    arr, k = _groups(groups)
    return float(lib().orc_auc(*csr.args(), arr, k))


def tied_rank(x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros_like(x)
    lib().orc_tied_rank(x, x.shape[0], out)
    return out


def auc_from_scores(scores, label, f64=False):
    scores = np.ascontiguousarray(scores, np.float32)
    label = np.ascontiguousarray(label, np.float32)
    if f64:
        return float(lib().orc_auc_from_scores_f64(scores, label, scores.shape[0]))
    return float(lib().orc_auc_from_scores(scores, label, scores.shape[0]))


# ---------------------------------------------------------------------------------------------
# ADMM algebra
# ---------------------------------------------------------------------------------------------

def payload(x, y, rho):
    out = np.zeros_like(x)
    lib().orc_payload(x, y, rho, out, x.shape[0])
    return out


def global_update(total, num_worker, rho, global_var, l1=0.0):
    z = np.zeros_like(total)
    lib().orc_global_update(total, num_worker, rho, global_var, l1, z, total.shape[0])
    return z


def global_update_commented(total, num_worker, rho, global_var):
    z = np.zeros_like(total)
    lib().orc_global_update_commented(total, num_worker, rho, global_var, z, total.shape[0])
    return z


def dual_update(y, x, z, rho):
    lib().orc_dual_update(y, x, z, rho, y.shape[0])


def residuals(x, z, z_prev, rho):
    r2, s2 = C.c_double(), C.c_double()
    lib().orc_residuals(x, z, z_prev, rho, x.shape[0], C.byref(r2), C.byref(s2))
    return r2.value, s2.value


def logistic_fg(csr, dim, x, offset, z, ydual, rho):
    g = np.zeros(dim, np.float64)
    f = lib().orc_logistic_fg(*csr.args(), dim, x, _opt(offset), _opt(z), _opt(ydual), rho, g)
    return float(f), g


# ---------------------------------------------------------------------------------------------
# L-BFGS (float storage, double reductions) -- the published two-loop recursion with a
# backtracking Armijo + weak-Wolfe line search (liblbfgs' LBFGS_LINESEARCH_BACKTRACKING_WOLFE
# rule; liblbfgs itself is linked by the reference but never called, Makefile:14).
# ---------------------------------------------------------------------------------------------

LBFGS_DEFAULTS = dict(m=6, max_iter=10, epsilon=1e-5, ftol=1e-4, wolfe=0.9, max_linesearch=20)


def _dot(a, b):
    return float(np.dot(a.astype(np.float64), b.astype(np.float64)))


def lbfgs_solve(x0, fg, m=6, max_iter=10, epsilon=1e-5, ftol=1e-4, wolfe=0.9,
                max_linesearch=20):
    """Minimise fg(x) -> (f, g float64) from x0 (float32).  Returns (x, f, n_evals, n_iters)."""
    x = x0.astype(np.float32).copy()
    f, g64 = fg(x)
    g = g64.astype(np.float32)
    evals = 1
    xnorm = max(1.0, np.sqrt(_dot(x, x)))
    gnorm = np.sqrt(_dot(g, g))
    if gnorm / xnorm <= epsilon:
        return x, f, evals, 0
    d = (-g).astype(np.float32)
    step = 1.0 / np.sqrt(_dot(d, d))
    S, Y, YS = [], [], []
    iters = 0
    for _ in range(max_iter):
        xp, gp, fp = x.copy(), g.copy(), f
        dginit = _dot(g, d)
        if dginit > 0:
            break
        dgtest = ftol * dginit
        count = 0
        ok = False
        while True:
            x = (xp + np.float32(step) * d).astype(np.float32)
            f, g64 = fg(x)
            g = g64.astype(np.float32)
            evals += 1
            count += 1
            if f > fp + step * dgtest:
                width = 0.5
            else:
                dg = _dot(g, d)
                if dg < wolfe * dginit:
                    width = 2.1
                else:
                    ok = True
                    break
            if count >= max_linesearch:
                break
            step *= width
        if not ok:
            if f > fp:
                x, g, f = xp, gp, fp
            break
        iters += 1
        xnorm = max(1.0, np.sqrt(_dot(x, x)))
        gnorm = np.sqrt(_dot(g, g))
        if gnorm / xnorm <= epsilon:
            break
        s = (x - xp).astype(np.float32)
        y = (g - gp).astype(np.float32)
        ys = _dot(y, s)
        yy = _dot(y, y)
        if ys > 1e-30:
            S.append(s)
            Y.append(y)
            YS.append(ys)
            if len(S) > m:
                S.pop(0)
                Y.pop(0)
                YS.pop(0)
        d = (-g).astype(np.float32)
        alphas = []
        for s_i, y_i, ys_i in zip(reversed(S), reversed(Y), reversed(YS)):
            a = _dot(s_i, d) / ys_i
            alphas.append(a)
            d = (d - np.float32(a) * y_i).astype(np.float32)
        if ys > 1e-30:
            d = (d * np.float32(ys / yy)).astype(np.float32)
        for (s_i, y_i, ys_i), a in zip(zip(S, Y, YS), reversed(alphas)):
            b = _dot(y_i, d) / ys_i
            d = (d + np.float32(a - b) * s_i).astype(np.float32)
        step = 1.0
    return x, f, evals, iters


# ---------------------------------------------------------------------------------------------
# The ADMM outer loop of src/admm_allreduce.cpp:55-91 for N ranks inside one process.
# Payloads are summed in rank order (rabit's tree order is not recoverable; float sums over
# ranks differ by ~1e-7 between orders).
# ---------------------------------------------------------------------------------------------

class AdmmConf:
    def __init__(self, step_size, global_var, ftrl_alpha, passes, dim, bias_var=0.0,
                 global_l1=0.0, local_solver="ftrl", ftrl_batch=1, lbfgs=None):
        self.step_size = np.float32(step_size)
        self.global_var = np.float32(global_var)
        self.bias_var = np.float32(bias_var)
        self.ftrl_alpha = np.float32(ftrl_alpha)
        self.passes = int(passes)
        self.dim = int(dim)
        self.global_l1 = np.float32(global_l1)
        self.local_solver = local_solver
        self.ftrl_batch = int(ftrl_batch)
        self.lbfgs = dict(LBFGS_DEFAULTS)
        if lbfgs:
            self.lbfgs.update(lbfgs)


class WorkerState:
    """src/workers.h:61-81: base_vec_ (x), bias_vec_, langr_vec_ (y), nt_vec_, zt_vec_."""

    def __init__(self, dim):
        self.x = np.zeros(dim, np.float32)
        self.bias = np.zeros(dim, np.float32)
        self.y = np.zeros(dim, np.float32)
        self.nt = np.zeros(dim, np.float32)
        self.zt = np.zeros(dim, np.float32)
        self.load = False


def base_update(worker, parts, conf, zg):
    """src/workers.cc:51-87 (Worker::BaseUpdate).  parts = translated Csr per data part."""
    if conf.local_solver == "ftrl":
        w = np.zeros(conf.dim, np.float32)           # FtrlSolver::Init, src/ftrl.cc:47-51
        n = worker.nt.copy()                         # src/workers.cc:66
        z = worker.zt.copy()                         # src/workers.cc:67
        reg = zg.copy()                              # src/workers.cc:58-60
        for part in parts:                           # src/workers.cc:73-77
            ftrl_run(part, w, z, n, worker.bias, reg, conf.ftrl_alpha, np.float32(1.0),
                     np.float32(1.0), np.float32(200.0), batch=conf.ftrl_batch)
        worker.x = w                                 # src/workers.cc:78
        if worker.load:                              # src/workers.cc:81-85
            worker.nt, worker.zt, worker.load = n, z, False
        return dict(evals=len(parts))
    # batch objective solved with L-BFGS (no counterpart in the reference)
    assert len(parts) == 1
    csr = parts[0]
    rho = conf.step_size

    def fg(xx):
        return logistic_fg(csr, conf.dim, xx, worker.bias, zg, worker.y, rho)

    x, f, evals, iters = lbfgs_solve(worker.x, fg, **conf.lbfgs)
    worker.x = x
    return dict(evals=evals, iters=iters, f=f)


def admm_run(train_parts, tests, conf, workers=None, zg=None):
    """train_parts[r] = list of translated Csr parts of rank r; tests[r] = translated test Csr.
    Returns (workers, zg, per-iteration records)."""
    nrank = len(train_parts)
    if workers is None:
        workers = [WorkerState(conf.dim) for _ in range(nrank)]
    if zg is None:
        zg = np.zeros(conf.dim, np.float32)
    records = []
    for it in range(conf.passes):
        if it == conf.passes - 1:                    # src/admm_allreduce.cpp:60-61
            for wk in workers:
                wk.load = True
        total = np.zeros(conf.dim, np.float32)
        payloads, stats = [], []
        for r in range(nrank):                       # lazy_ftrl, src/admm_allreduce.cpp:63-68
            stats.append(base_update(workers[r], train_parts[r], conf, zg))
            payloads.append(payload(workers[r].x, workers[r].y, conf.step_size))
        for r in range(nrank):                       # rabit::Allreduce<Sum>, :69
            total = (total + payloads[r]).astype(np.float32)
        lines, lines64 = [], []
        for r in range(nrank):                       # :71-81
            groups = [workers[r].x, workers[r].bias]
            lines.append((logloss(tests[r], groups), auc(tests[r], groups), tests[r].nrows))
            # the same two numbers with the sums kept in double (the reference keeps them in float, which loses
            # digits beyond some 10^4 rows: src/metrics.cc:68-74,92,105)
            s64, p = logloss_sum_f64(tests[r], groups, want_p=True)
            lines64.append((s64 / max(tests[r].nrows, 1), auc_from_scores(p, tests[r].label, f64=True), tests[r].nrows))
        z_prev = zg
        zg = global_update(total, nrank, conf.step_size, conf.global_var, conf.global_l1)  # :84
        res = []
        for r in range(nrank):                       # :85
            dual_update(workers[r].y, workers[r].x, zg, conf.step_size)
            res.append(residuals(workers[r].x, zg, z_prev, conf.step_size))
        records.append(dict(total=total.copy(), z=zg.copy(),
                            x=[w.x.copy() for w in workers], y=[w.y.copy() for w in workers],
                            lines=lines, lines64=lines64, residuals=res, stats=stats))
    return workers, zg, records
