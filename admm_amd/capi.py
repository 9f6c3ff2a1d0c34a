"""ctypes binding of libadmm_hip.so (the C ABI declared in include/admm_hip.h).

Thin by design: every method is one C call.  There is no CPU implementation behind this module:
if the library is missing it raises at load time, and without a GPU the library's context
creation fails.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libadmm_hip.so")

# every symbol include/admm_hip.h declares
SYMBOLS = [
    "admm_last_error", "admm_abi_version", "admm_device_count", "admm_ctx_create", "admm_ctx_destroy", "admm_sync",
    "admm_timing_enable", "admm_timing_classes", "admm_timing_every", "admm_timing_reset", "admm_timing_get",
    "admm_vec_alloc", "admm_vec_free", "admm_vec_zero", "admm_vec_copy", "admm_vec_upload",
    "admm_vec_download",
    "admm_csr_upload", "admm_csr_synth", "admm_csr_shape", "admm_csr_download",
    "admm_csr_build_transposed", "admm_csr_relabel_by_frequency", "admm_csr_set_lean", "admm_synth_row_draws", "admm_csr_bytes", "admm_csr_layout_bytes", "admm_csr_layout_counts",
    "admm_csr_destroy",
    "admm_predict", "admm_metrics", "admm_auc_scores", "admm_tied_ranks", "admm_csr_copy_labels", "admm_ftrl_pass", "admm_ftrl_predict", "admm_ftrl_update", "admm_ftrl_pass_multi", "admm_logistic_fg", "admm_lbfgs_solve",
    "admm_payload", "admm_global_update", "admm_dual_update", "admm_consensus", "admm_consensus_static",
    "admm_comm_unique_id", "admm_comm_init", "admm_allreduce_sum_f32", "admm_comm_selftest",
]

K_FORWARD, K_TRANSPOSED, K_FTRL_SEQ, K_ELEMENTWISE, K_FORWARD_EVAL = 0, 1, 2, 3, 4


class AdmmError(RuntimeError):
    pass


class SynthSpec(C.Structure):
    _fields_ = [("nrows", C.c_uint64), ("dim", C.c_uint64), ("seed", C.c_uint64),
                ("model_seed", C.c_uint64), ("row_offset", C.c_uint64), ("mean_nnz", C.c_float), ("zipf_s", C.c_float),
                ("len_dist", C.c_int), ("tail_sigma", C.c_float), ("max_len", C.c_uint32),
                ("explicit_values", C.c_int)]


class LbfgsParams(C.Structure):
    _fields_ = [("m", C.c_int), ("max_iter", C.c_int), ("max_linesearch", C.c_int),
                ("epsilon", C.c_float), ("ftol", C.c_float), ("wolfe", C.c_float), ("linesearch", C.c_int)]


class LbfgsResult(C.Structure):
    _fields_ = [("f", C.c_double), ("evals", C.c_int), ("iters", C.c_int),
                ("forward_sweeps", C.c_int), ("transposed_sweeps", C.c_int)]


_lib = None


def load():
    """Load libadmm_hip.so; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AdmmError("%s not found: build it with `python -m admm_amd.build` "
                        "(there is no fallback implementation)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u64, f32, i32 = C.c_void_p, C.c_uint64, C.c_float, C.c_int
    L.admm_last_error.restype = C.c_char_p
    L.admm_last_error.argtypes = []
    L.admm_abi_version.restype = i32
    sig = {
        "admm_device_count": [C.POINTER(i32)],
        "admm_ctx_create": [i32, vp, C.POINTER(vp)],
        "admm_ctx_destroy": [vp],
        "admm_sync": [vp],
        "admm_timing_enable": [vp, i32],
        "admm_timing_classes": [vp, C.c_uint32],
        "admm_timing_every": [vp, C.c_uint32],
        "admm_timing_reset": [vp],
        "admm_timing_get": [vp, i32, C.POINTER(C.c_double), C.POINTER(u64)],
        "admm_vec_alloc": [vp, u64, C.POINTER(vp)],
        "admm_vec_free": [vp, vp],
        "admm_vec_zero": [vp, vp, u64],
        "admm_vec_copy": [vp, vp, vp, u64],
        "admm_vec_upload": [vp, vp, vp, u64],
        "admm_vec_download": [vp, vp, vp, u64],
        "admm_csr_upload": [vp, vp, vp, vp, vp, u64, u64, u64, i32, C.POINTER(vp)],
        "admm_csr_synth": [vp, C.POINTER(SynthSpec), C.POINTER(vp)],
        "admm_csr_shape": [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)],
        "admm_csr_download": [vp, vp, vp, vp, vp],
        "admm_csr_build_transposed": [vp, u64, C.c_uint32],
        "admm_csr_relabel_by_frequency": [vp],
        "admm_csr_bytes": [vp, C.POINTER(u64), C.POINTER(u64)],
        "admm_csr_set_lean": [vp, i32],
        "admm_synth_row_draws": [vp, vp, u64, C.POINTER(u64)],
        "admm_csr_layout_bytes": [vp, C.POINTER(u64)],
        "admm_csr_layout_counts": [vp, C.POINTER(u64)],
        "admm_csr_destroy": [vp],
        "admm_predict": [vp, C.POINTER(vp), i32, vp],
        "admm_metrics": [vp, C.POINTER(vp), i32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                         C.POINTER(u64)],
        "admm_auc_scores": [vp, vp, vp, u64, C.POINTER(C.c_double)],
        "admm_tied_ranks": [vp, vp, u64, vp],
        "admm_csr_copy_labels": [vp, vp],
        "admm_ftrl_pass": [vp, vp, vp, vp, vp, vp, f32, f32, f32, f32, u64],
        "admm_ftrl_pass_multi": [vp, vp, vp, vp, vp, vp, C.c_int, f32, f32, f32, f32],
        "admm_ftrl_predict": [vp, vp, vp, C.c_uint32, vp, vp, vp, vp, vp, f32, f32, f32, f32, C.POINTER(C.c_float)],
        "admm_ftrl_update": [vp, vp, C.c_uint32, f32, f32, vp, vp, vp, f32],
        "admm_logistic_fg": [vp, vp, vp, vp, vp, f32, C.POINTER(C.c_double), vp],
        "admm_lbfgs_solve": [vp, vp, vp, vp, vp, f32, C.POINTER(LbfgsParams),
                             C.POINTER(LbfgsResult)],
        "admm_payload": [vp, vp, vp, f32, vp, u64],
        "admm_global_update": [vp, vp, i32, f32, f32, f32, vp, u64],
        "admm_dual_update": [vp, vp, vp, vp, f32, u64],
        "admm_consensus": [vp, vp, i32, f32, f32, f32, vp, vp, vp, u64, C.POINTER(C.c_double),
                           C.POINTER(C.c_double)],
        "admm_consensus_static": [vp, vp, i32, f32, f32, f32, vp, vp, vp, u64, vp],
        "admm_comm_unique_id": [vp],
        "admm_comm_init": [vp, i32, i32, vp],
        "admm_allreduce_sum_f32": [vp, vp, u64],
        "admm_comm_selftest": [vp],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = i32
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise AdmmError("libadmm_hip error %d: %s" % (rc, load().admm_last_error().decode()))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class DeviceVec:
    """dim floats in HBM."""

    def __init__(self, ctx, n):
        self.ctx, self.n = ctx, int(n)
        p = C.c_void_p()
        _check(ctx.L.admm_vec_alloc(ctx.h, self.n, C.byref(p)))
        self.ptr = p

    @classmethod
    def from_ptr(cls, ctx, ptr, n):
        """Wrap device memory owned by someone else (e.g. a torch tensor's data_ptr)."""
        v = cls.__new__(cls)
        v.ctx, v.n, v.ptr, v.borrowed = ctx, int(n), C.c_void_p(int(ptr)), True
        return v

    def upload(self, host):
        host = np.ascontiguousarray(host, np.float32)
        assert host.shape == (self.n,)
        _check(self.ctx.L.admm_vec_upload(self.ctx.h, self.ptr, _ptr(host), self.n))
        return self

    def download(self):
        out = np.empty(self.n, np.float32)
        _check(self.ctx.L.admm_vec_download(self.ctx.h, _ptr(out), self.ptr, self.n))
        return out

    def zero(self):
        _check(self.ctx.L.admm_vec_zero(self.ctx.h, self.ptr, self.n))
        return self

    def copy_from(self, other):
        _check(self.ctx.L.admm_vec_copy(self.ctx.h, self.ptr, other.ptr, self.n))
        return self

    def free(self):
        if getattr(self, "borrowed", False):
            self.ptr = None
        if self.ptr:
            _check(self.ctx.L.admm_vec_free(self.ctx.h, self.ptr))
            self.ptr = None


class Context:
    def __init__(self, device=0, stream=None):
        self.L = load()
        h = C.c_void_p()
        _check(self.L.admm_ctx_create(int(device), C.c_void_p(stream) if stream else None,
                                      C.byref(h)))
        self.h = h
        self.rank, self.world = 0, 1
        self.exchange = None

    def close(self):
        if self.h:
            self.L.admm_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        _check(self.L.admm_sync(self.h))

    def vec(self, n, host=None):
        v = DeviceVec(self, n)
        if host is not None:
            v.upload(host)
        return v

    # timing -------------------------------------------------------------------------------
    def timing_enable(self, on=True, classes=None, every=1):
        """HIP-event timing of the library's kernels; `classes` (a list of K_* ids) restricts it, and of
        each class every `every`-th launch is timed"""
        mask = 0xFFFFFFFF if classes is None else sum(1 << k for k in classes)
        _check(self.L.admm_timing_classes(self.h, mask))
        _check(self.L.admm_timing_every(self.h, max(1, int(every))))
        _check(self.L.admm_timing_enable(self.h, 1 if on else 0))

    def timing_reset(self):
        _check(self.L.admm_timing_reset(self.h))

    def timing_get(self, kernel):
        ms, cnt = C.c_double(), C.c_uint64()
        _check(self.L.admm_timing_get(self.h, kernel, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    # sample sets --------------------------------------------------------------------------
    def csr_upload(self, rowptr, col, val, label, dim, add_bias=True):
        rowptr = np.ascontiguousarray(rowptr, np.uint64)
        col = np.ascontiguousarray(col, np.uint32)
        val = None if val is None else np.ascontiguousarray(val, np.float32)
        label = np.ascontiguousarray(label, np.float32)
        nrows = rowptr.shape[0] - 1
        nnz = int(rowptr[-1])
        h = C.c_void_p()
        _check(self.L.admm_csr_upload(self.h, _ptr(rowptr), _ptr(col), _ptr(val), _ptr(label),
                                      nrows, nnz, int(dim), 1 if add_bias else 0, C.byref(h)))
        return SampleSet(self, h)

    def csr_synth(self, nrows, dim, mean_nnz, seed=20251111, row_offset=0, zipf_s=1.1,
                  len_dist=0, tail_sigma=1.0, max_len=None, model_seed=7, explicit_values=False):
        if max_len is None:
            max_len = int(min(1024, max(8, 4 * mean_nnz)))
        spec = SynthSpec(int(nrows), int(dim), int(seed), int(model_seed), int(row_offset), float(mean_nnz),
                         float(zipf_s), int(len_dist), float(tail_sigma), int(max_len),
                         1 if explicit_values else 0)
        h = C.c_void_p()
        _check(self.L.admm_csr_synth(self.h, C.byref(spec), C.byref(h)))
        return SampleSet(self, h)

    def synth_row_draws(self, nrows, dim, mean_nnz, block_rows, seed=20251111, row_offset=0, zipf_s=1.1, len_dist=0,
                        tail_sigma=1.0, max_len=None, model_seed=7, explicit_values=False):
        """entries the generator draws per block of block_rows rows (bias entry included), as a numpy array"""
        if max_len is None:
            max_len = int(min(1024, max(8, 4 * mean_nnz)))
        spec = SynthSpec(int(nrows), int(dim), int(seed), int(model_seed), int(row_offset), float(mean_nnz),
                         float(zipf_s), int(len_dist), float(tail_sigma), int(max_len), 1 if explicit_values else 0)
        nblocks = (int(nrows) + int(block_rows) - 1) // int(block_rows)
        out = np.zeros(max(nblocks, 1), np.uint64)
        _check(self.L.admm_synth_row_draws(self.h, C.byref(spec), int(block_rows), out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out[:nblocks]

    def ftrl_pass_multi(self, sets, w, z, n, offset, reg_offset, alpha, beta, l_1, l_2):
        """admm_ftrl_pass_multi: pass k runs sets[k] with state w[k], z[k], n[k]"""
        cnt = len(sets)
        assert all(len(v) == cnt for v in (w, z, n, offset, reg_offset))
        def arr(items, get):
            return (C.c_void_p * cnt)(*[getattr(get(i), "value", get(i)) for i in items])
        _check(self.L.admm_ftrl_pass_multi(arr(sets, lambda s: s.h), arr(w, lambda v: v.ptr), arr(z, lambda v: v.ptr),
                                           arr(n, lambda v: v.ptr), arr(offset, lambda v: v.ptr),
                                           arr(reg_offset, lambda v: v.ptr), cnt, alpha, beta, l_1, l_2))

    def ftrl_predict(self, index, value, w, z, n, offset, reg_offset, alpha, beta, l_1, l_2):
        """FtrlSolver::Predict (src/ftrl.cc:54-73) on one translated row held in host arrays (value None: all
        ones): refreshes w at the row's coordinates and returns P(y = 1)"""
        idx = np.ascontiguousarray(index, np.uint32)
        val = None if value is None else np.ascontiguousarray(value, np.float32)
        p = C.c_float()
        _check(self.L.admm_ftrl_predict(self.h, _ptr(idx), None if val is None else _ptr(val), idx.shape[0], w.ptr,
                                        z.ptr, n.ptr, offset.ptr, reg_offset.ptr, alpha, beta, l_1, l_2, C.byref(p)))
        return float(p.value)

    def ftrl_update(self, index, label, predict, w, z, n, alpha):
        """FtrlSolver::Update (src/ftrl.cc:75-88) on the same row"""
        idx = np.ascontiguousarray(index, np.uint32)
        _check(self.L.admm_ftrl_update(self.h, _ptr(idx), idx.shape[0], float(label), float(predict), w.ptr, z.ptr,
                                       n.ptr, alpha))

    # consensus algebra --------------------------------------------------------------------
    def payload(self, x, ydual, rho, out):
        _check(self.L.admm_payload(self.h, x.ptr, ydual.ptr, rho, out.ptr, x.n))

    def global_update(self, total, num_worker, rho, global_var, l1, z):
        _check(self.L.admm_global_update(self.h, total.ptr, num_worker, rho, global_var, l1,
                                         z.ptr, z.n))

    def dual_update(self, ydual, x, z, rho):
        _check(self.L.admm_dual_update(self.h, ydual.ptr, x.ptr, z.ptr, rho, x.n))

    def tied_ranks(self, scores, ranks, n=None):
        """Metrics::Sort (src/metrics.cc:14-46) of a device vector of scores into a device vector of ranks"""
        _check(self.L.admm_tied_ranks(self.h, scores.ptr, scores.n if n is None else int(n), ranks.ptr))

    def auc_scores(self, scores, labels, n):
        """tie-aware rank-sum AUC (src/metrics.cc:14-89) of n (score, label) pairs held in device vectors"""
        auc = C.c_double()
        _check(self.L.admm_auc_scores(self.h, scores.ptr, labels.ptr, int(n), C.byref(auc)))
        return auc.value

    def consensus(self, total, num_worker, rho, global_var, l1, x, ydual, z, want_residuals=True):
        r2, s2 = C.c_double(), C.c_double()
        _check(self.L.admm_consensus(self.h, total.ptr, num_worker, rho, global_var, l1, x.ptr,
                                     ydual.ptr, z.ptr, x.n,
                                     C.byref(r2) if want_residuals else None,
                                     C.byref(s2) if want_residuals else None))
        return (r2.value, s2.value) if want_residuals else None

    def consensus_static(self, total, num_worker, rho, global_var, l1, x, ydual, z, residuals=None, stream=None,
                         dim=None):
        """admm_consensus_static: the same step without a context, a host wait or an allocation, on `stream`
        (raw hipStream_t value, None = the default stream) of the current device; `residuals` is a device
        buffer of two doubles or None"""
        _check(self.L.admm_consensus_static(C.c_void_p(stream) if stream else None, total.ptr, num_worker, rho,
                                            global_var, l1, x.ptr, ydual.ptr, z.ptr, x.n if dim is None else int(dim),
                                            residuals.ptr if residuals is not None else None))

    def allreduce_sum(self, buf, n=None):
        """rabit::Allreduce<op::Sum> (src/admm_allreduce.cpp:69): in-place float sum over ranks on RCCL.
        `exchange` (set by a rehearsal in which several ranks share one GPU, where RCCL cannot run)
        replaces the library call by a caller-supplied function of (vector, count)."""
        cnt = buf.n if n is None else int(n)
        if self.exchange is not None:
            return self.exchange(buf, cnt)
        _check(self.L.admm_allreduce_sum_f32(self.h, buf.ptr, cnt))

    def comm_unique_id(self):
        """128 bytes that rank 0 hands to every rank (admm_comm_unique_id)"""
        buf = C.create_string_buffer(128)
        _check(self.L.admm_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank, world, unique_id=None):
        self.rank, self.world = int(rank), int(world)
        _check(self.L.admm_comm_init(self.h, self.rank, self.world,
                                     C.create_string_buffer(unique_id, 128) if unique_id else None))

    def comm_selftest(self):
        _check(self.L.admm_comm_selftest(self.h))


class SampleSet:
    """One sample set resident in HBM (CSR, optionally with its column-block copy)."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(ctx.L.admm_csr_shape(self.h, C.byref(a), C.byref(b), C.byref(c)))
        self.nrows, self.nnz, self.dim = a.value, b.value, c.value

    def destroy(self):
        if self.h:
            self.ctx.L.admm_csr_destroy(self.h)
            self.h = None

    def download(self):
        rowptr = np.empty(self.nrows + 1, np.uint64)
        col = np.empty(max(self.nnz, 1), np.uint32)
        val = np.empty(max(self.nnz, 1), np.float32)
        label = np.empty(max(self.nrows, 1), np.float32)
        _check(self.ctx.L.admm_csr_download(self.h, _ptr(rowptr), _ptr(col), _ptr(val),
                                            _ptr(label)))
        return rowptr, col[:self.nnz], val[:self.nnz], label[:self.nrows]

    def build_transposed(self, panel_rows=0, col_block=0):
        _check(self.ctx.L.admm_csr_build_transposed(self.h, int(panel_rows), int(col_block)))
        return self

    def set_lean(self, on=True):
        """whole-set sweeps only from here on: the builders free the row-major forms once their copies exist"""
        _check(self.ctx.L.admm_csr_set_lean(self.h, 1 if on else 0))
        return self

    def relabel_by_frequency(self):
        _check(self.ctx.L.admm_csr_relabel_by_frequency(self.h))
        return self

    def bytes(self):
        a, b = C.c_uint64(), C.c_uint64()
        _check(self.ctx.L.admm_csr_bytes(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def layout_bytes(self):
        """dict of HBM bytes by layout: row_major, forward_form, column_block_copy, row_vectors"""
        out = (C.c_uint64 * 4)()
        _check(self.ctx.L.admm_csr_layout_bytes(self.h, out))
        return dict(zip(("row_major", "forward_form", "column_block_copy", "row_vectors"), [int(v) for v in out]))

    def layout_counts(self):
        """dict of entry counts by layout (admm_csr_layout_counts)"""
        out = (C.c_uint64 * 6)()
        _check(self.ctx.L.admm_csr_layout_counts(self.h, out))
        return dict(zip(("mask_entries", "hot_stream_words", "list_entries", "column_stream_words", "cell_entries",
                         "cells"), [int(v) for v in out]))

    def _groups(self, groups):
        arr = (C.c_void_p * len(groups))(*[g.ptr.value for g in groups])
        return C.cast(arr, C.POINTER(C.c_void_p)), len(groups)

    def predict_into(self, groups, out):
        """per-row P(y=1|x) into a device vector of at least nrows floats"""
        arr, k = self._groups(groups)
        _check(self.ctx.L.admm_predict(self.h, arr, k, out.ptr))

    def copy_labels(self, out):
        _check(self.ctx.L.admm_csr_copy_labels(self.h, out.ptr))

    def predict(self, groups):
        out = DeviceVec(self.ctx, max(self.nrows, 1))
        arr, k = self._groups(groups)
        _check(self.ctx.L.admm_predict(self.h, arr, k, out.ptr))
        p = out.download()[:self.nrows]
        out.free()
        return p

    def metrics(self, groups, want_auc=True):
        arr, k = self._groups(groups)
        ll, auc, n = C.c_double(), C.c_double(), C.c_uint64()
        _check(self.ctx.L.admm_metrics(self.h, arr, k, C.byref(ll),
                                       C.byref(auc) if want_auc else None, C.byref(n)))
        return ll.value, (auc.value if want_auc else None), n.value

    def ftrl_pass(self, w, z, n, offset, reg_offset, alpha, beta, l1, l2, batch=1):
        _check(self.ctx.L.admm_ftrl_pass(self.h, w.ptr, z.ptr, n.ptr, offset.ptr, reg_offset.ptr,
                                         alpha, beta, l1, l2, int(batch)))

    def logistic_fg(self, x, g_out, offset=None, z=None, ydual=None, rho=0.0, want_f=True):
        f = C.c_double()
        _check(self.ctx.L.admm_logistic_fg(self.h, x.ptr, offset.ptr if offset else None,
                                           z.ptr if z else None, ydual.ptr if ydual else None,
                                           rho, C.byref(f) if want_f else None, g_out.ptr))
        return f.value if want_f else None

    def lbfgs_solve(self, x, offset=None, z=None, ydual=None, rho=0.0, **params):
        p = LbfgsParams(params.get("m", 6), params.get("max_iter", 10),
                        params.get("max_linesearch", 20), params.get("epsilon", 1e-5),
                        params.get("ftol", 1e-4), params.get("wolfe", 0.9), int(params.get("linesearch", 0)))
        res = LbfgsResult()
        _check(self.ctx.L.admm_lbfgs_solve(self.h, x.ptr, offset.ptr if offset else None,
                                           z.ptr if z else None, ydual.ptr if ydual else None,
                                           rho, C.byref(p), C.byref(res)))
        return dict(f=res.f, evals=res.evals, iters=res.iters, forward_sweeps=res.forward_sweeps,
                    transposed_sweeps=res.transposed_sweeps)
