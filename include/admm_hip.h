/*
 * admm_hip.h -- C ABI of libadmm_hip.so: the ADMM / FTRL sparse logistic-regression hot path on
 * MI355X (gfx950).  Plain C, opaque handles, plain pointers and sizes.
 *
 * Every entry point names the reference interface it stands in for (paths relative to the
 * wuyunhao/admm tree).  The reference has no FFI layer of its own: its seam is the C++ class
 * surface of src/ (SampleSet, FtrlSolver, Worker, Master, Metrics), which the host classes under
 * admm_amd/host/ rebuild on top of these calls.
 *
 * Conventions
 *   - return value 0 = success, negative = ADMM_ERR_*; admm_last_error() gives the message of
 *     the calling thread's last failure.  No exceptions cross the boundary.
 *   - one host thread per context; work is queued on the context's HIP stream; calls that
 *     return a host scalar synchronise that stream, all others are asynchronous.
 *   - `float *` vector arguments are DEVICE pointers of `dim` floats unless the name says host.
 *   - there is no CPU implementation behind these calls: without a GPU they fail.
 */
#ifndef ADMM_HIP_H_
#define ADMM_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADMM_OK 0
#define ADMM_ERR_INVALID (-1) /* bad argument / shape mismatch */
#define ADMM_ERR_HIP (-2)     /* a HIP runtime call failed */
#define ADMM_ERR_NOMEM (-3)
#define ADMM_ERR_STATE (-4) /* e.g. transposed copy not built */
#define ADMM_ERR_COMM (-5)  /* RCCL missing or failed */

typedef struct admm_ctx admm_ctx; /* one GPU + one stream */
typedef struct admm_csr admm_csr; /* one sample set resident in HBM */

const char *admm_last_error(void);
int admm_abi_version(void);

/* ---- context ----------------------------------------------------------------------------- */

/* Stands in for rabit::Init / rabit::Finalize as far as device set-up goes
 * (src/admm_allreduce.cpp:9,113).  `stream` is a hipStream_t to queue on, or NULL to let the
 * library create its own. */
int admm_device_count(int *count);
int admm_ctx_create(int device, void *stream, admm_ctx **out);
int admm_ctx_destroy(admm_ctx *ctx);
int admm_sync(admm_ctx *ctx);

/* Per-kernel device timing with HIP events on the context's stream (profiling aid, no
 * counterpart in the reference).  kernel ids: ADMM_K_*. */
#define ADMM_K_FORWARD 0    /* row-parallel CSR sweep of a training step (margins, residual, loss) */
#define ADMM_K_TRANSPOSED 1 /* column-block sweep (X^T r) */
#define ADMM_K_FTRL_SEQ 2   /* exact sequential FTRL pass */
#define ADMM_K_ELEMENTWISE 3
#define ADMM_K_FORWARD_EVAL 4 /* the same sweep in its prediction / logloss form */
#define ADMM_K_COUNT 5
int admm_timing_enable(admm_ctx *ctx, int on);
/* restricts the timing to the classes whose bit (1u << ADMM_K_*) is set; all classes by default.  Every
 * timed launch costs two event records on the stream, which a step of many small kernels feels. */
int admm_timing_classes(admm_ctx *ctx, unsigned mask);
/* of each class only every n-th launch is timed (n >= 1; 1, the default, times all of them): admm_timing_get
 * then returns the time and the count of the launches that were timed, a sample of the class. */
int admm_timing_every(admm_ctx *ctx, unsigned n);
int admm_timing_reset(admm_ctx *ctx);
int admm_timing_get(admm_ctx *ctx, int kernel, double *total_ms, uint64_t *launches);

/* ---- dim-sized vectors (Worker::InitWorker, src/workers.cc:32-46) ------------------------- */

int admm_vec_alloc(admm_ctx *ctx, uint64_t n, float **out); /* zero-filled */
int admm_vec_free(admm_ctx *ctx, float *v);
int admm_vec_zero(admm_ctx *ctx, float *v, uint64_t n);
int admm_vec_copy(admm_ctx *ctx, float *dst, const float *src, uint64_t n);
int admm_vec_upload(admm_ctx *ctx, float *dst, const float *src_host, uint64_t n);
int admm_vec_download(admm_ctx *ctx, float *dst_host, const float *src, uint64_t n);

/* ---- sample sets (SampleSet, src/sample_set.cc:13-92) ------------------------------------- */

/* SampleSet::Initialize + TranslateData: copy a parsed row block (host arrays, borrowed for the
 * call) into HBM as CSR: rowptr u64[nrows+1], col u32[nnz], val f32[nnz] (NULL = implicit 1, as
 * dmlc::Row with value == NULL), label f32[nrows].  With add_bias != 0 every row gets the
 * (index 0, value 1) entry in front and all indices move up by one, on the device
 * (src/sample_set.cc:70-88); `dim` is the dimension AFTER that shift.  Column indices must be
 * < dim (checked, as dmlc's SDot CHECKs).  Limits: dim < 2^30, nrows < 2^32 - 1 per sample set. */
int admm_csr_upload(admm_ctx *ctx, const uint64_t *rowptr, const uint32_t *col, const float *val,
                    const float *label, uint64_t nrows, uint64_t nnz, uint64_t dim, int add_bias,
                    admm_csr **out);

/* Seeded synthetic power-law sample set generated on the device (no counterpart in the
 * reference; benchmark input).  Feature ids are Zipf(zipf_s) ranks scattered over [0, dim-1) by
 * a fixed bijection, de-duplicated and sorted inside each row; values are 1; labels are
 * Bernoulli(sigmoid(x . w*)) for a hidden sparse w*.  Rows are numbered from row_offset so that
 * row shards of one data set can be generated independently.  The bias column is always added. */
typedef struct admm_synth_spec {
  uint64_t nrows;
  uint64_t dim;        /* including the bias column */
  uint64_t seed;       /* row contents */
  uint64_t model_seed; /* hidden model w* behind the labels (share it between train and test sets) */
  uint64_t row_offset; /* global index of this shard's first row */
  float mean_nnz;      /* mean feature draws per row; duplicates are dropped, the bias entry is added */
  float zipf_s;        /* > 1 */
  int len_dist;        /* 0: ~Poisson(mean_nnz); 1: log-normal heavy tail with the same mean */
  float tail_sigma;    /* log-normal sigma for len_dist 1 */
  uint32_t max_len;    /* raw row length cap, <= 1024 */
  int explicit_values; /* the generated values are all 1.0: 0 keeps the set as a pattern matrix
                          (no value array, as an uploaded all-ones set is kept), 1 stores them */
} admm_synth_spec;
int admm_csr_synth(admm_ctx *ctx, const admm_synth_spec *spec, admm_csr **out);
/* The generator's feature draws per row plus the bias entry, summed over consecutive blocks of block_rows rows of
 * the set `spec` describes, without generating it: sums_host receives ceil(nrows / block_rows) numbers.  A row's
 * stored entries are its draws minus its repeats, so these sums let a job cut a generated set into row shards of
 * (nearly) equal entry counts before any rank generates its shard -- the reference's byte-range split of a file
 * (src/sample_set.cc:15,24-25) balances entries the same way, bytes being proportional to them.  Benchmark
 * input, no counterpart in the reference. */
int admm_synth_row_draws(admm_ctx *ctx, const admm_synth_spec *spec, uint64_t block_rows, uint64_t *sums_host);

int admm_csr_shape(const admm_csr *csr, uint64_t *nrows, uint64_t *nnz, uint64_t *dim);
/* Copy the device CSR (after the bias shift) back to host arrays sized by admm_csr_shape. */
int admm_csr_download(const admm_csr *csr, uint64_t *rowptr, uint32_t *col, float *val,
                      float *label);
/* Build the column-block copy that the transposed sweep reads (X^T r, mini-batch FTRL):
 * rows are cut into panels of `panel_rows`, columns into blocks of `col_block` (a multiple of 64,
 * <= 16384; the mini-batch FTRL pass needs <= 8192), and each (panel, block) cell stores its
 * entries contiguously.  0 picks defaults (about a million rows, 8192 columns). */
int admm_csr_build_transposed(admm_csr *csr, uint64_t panel_rows, uint32_t col_block);
/* Renumber the stored column ids by decreasing frequency inside this sample set (an internal
 * layout change that speeds up the forward sweep on power-law data: the gathers of frequent
 * columns are served from LDS).  Invisible at this boundary: every vector argument stays in the
 * caller's numbering and admm_csr_download returns the original ids.  Copies built before the call
 * (admm_csr_build_transposed) are rebuilt in the new numbering with the same panel and block sizes, so calling
 * this first saves a build.  No counterpart in the reference. */
int admm_csr_relabel_by_frequency(admm_csr *csr);
/* Declares that this sample set serves whole-set sweeps only (admm_logistic_fg, admm_lbfgs_solve, admm_predict,
 * admm_metrics): the builders then free every row-major form as soon as the copies that replace it exist -- the CSR
 * once admm_csr_relabel_by_frequency has made the row masks and the compact rows, those rows once
 * admm_csr_build_transposed has made the row-panel and column-block copies -- which leaves about 1.4 times the
 * CSR's bytes in HBM instead of 3 times, and lets a set be built whose forms would not fit side by side.
 * Afterwards whatever reads rows (admm_csr_download, admm_ftrl_pass, a rebuild of the copies) returns
 * ADMM_ERR_STATE.  A set that cannot have row masks (fewer than 256 columns, or a row that stores a column twice)
 * keeps all its forms.  No counterpart in the reference, whose SampleSet re-reads its file for every pass
 * (src/workers.cc:73-77). */
int admm_csr_set_lean(admm_csr *csr, int on);
/* Bytes of HBM held by this sample set. */
int admm_csr_bytes(const admm_csr *csr, uint64_t *csr_bytes, uint64_t *transposed_bytes);
/* The same, by layout: out[0] the row-major forms still held (the CSR; beside a row-panel copy also the compact
 * rows of a relabelled set), out[1] the form the forward sweep streams (for a
 * relabelled set the compact form: row masks of the most frequent columns + the remaining entries
 * as CSR + labels; else the CSR itself), out[2] the column-block copy the transposed sweep streams,
 * out[3] the per-row work vectors allocated so far. */
int admm_csr_layout_bytes(const admm_csr *csr, uint64_t out[4]);
/* Entry counts behind those layouts (what a bytes-moved estimate is built from): out[0] entries held
 * as bits of the row masks, out[1] 16-bit words of the forward sweep's hot stream, out[2] entries of
 * its (column, row) lists, out[3] 16-bit words of the transposed sweep's column stream, out[4] entries
 * of the column-block cells, out[5] cells.  No counterpart in the reference. */
int admm_csr_layout_counts(const admm_csr *csr, uint64_t out[6]);
int admm_csr_destroy(admm_csr *csr);

/* ---- metrics (Metrics, src/metrics.cc:45-117) --------------------------------------------- */

/* Per-row P(y=1|x) with the margin summed over `ngroups` (<= 4) weight vectors and clamped to
 * [-35, 35] (src/metrics.cc:54-58).  p_out: DEVICE buffer of nrows floats. */
int admm_predict(admm_csr *csr, const float *const *groups, int ngroups, float *p_out);
/* Metrics::LogLoss and Metrics::Auc (single-model form) in one call.  logloss_mean is the mean
 * of -log p / -log(1-p) over rows with label > 0 / <= 0 (double accumulation, fixed order);
 * auc is the tie-aware rank-sum AUC with positives = rows whose label truncates to 1, computed
 * on the device in double.  Either output pointer may be NULL. */
int admm_metrics(admm_csr *csr, const float *const *groups, int ngroups, double *logloss_mean,
                 double *auc, uint64_t *nrows);
/* Metrics::Sort (src/metrics.cc:14-46): the tied ranks (1-based, a tie group shares the mean of its ranks,
 * computed in float as there) of n scores; scores and ranks are DEVICE arrays, n < 2^31. */
int admm_tied_ranks(admm_ctx *ctx, const float *scores, uint64_t n, float *ranks);
/* Metrics::Auc + Metrics::Sort (src/metrics.cc:14-89) on bare arrays: the same tie-aware rank-sum
 * AUC over n (score, label) pairs in DEVICE memory.  The reference reports one AUC per rank; a
 * job that wants the AUC of all ranks' test rows together gathers the ranks' scores and labels
 * (admm_predict into one rank's slice of a zero-filled buffer + a sum all-reduce does it) and
 * calls this once.  n < 2^31 (the sort's 32-bit positions); more rows are rejected. */
int admm_auc_scores(admm_ctx *ctx, const float *scores, const float *labels, uint64_t n,
                    double *auc);
/* Copies the nrows labels of the sample set (as stored: the file's labels, src/sample_set.cc:41)
 * to dst, a DEVICE buffer. */
int admm_csr_copy_labels(admm_csr *csr, float *dst);

/* ---- FTRL-proximal (FtrlSolver, src/ftrl.cc:36-145) --------------------------------------- */

/* FtrlSolver::Run: one pass over the sample set, then one more Predict on the last row.
 * w, z, n are the solver state (weight_, mid_weight_, squared_sum_), updated in place;
 * offset / reg_offset as in Run's arguments.  batch == 1 is the reference's sequential
 * semantics row by row.  batch > 1 (needs the transposed copy, panel_rows == batch) predicts
 * all rows of a batch from the state at its start and then applies the summed update per
 * coordinate. */
int admm_ftrl_pass(admm_csr *csr, float *w, float *z, float *n, const float *offset,
                   const float *reg_offset, float alpha, float beta, float l1, float l2,
                   uint64_t batch);

/* FtrlSolver::Predict (src/ftrl.cc:54-73) on one translated row given as HOST arrays (index[length],
 * value[length] or NULL for all ones): refreshes w at the row's coordinates from z, n, reg_offset and
 * returns P(y = 1) through p_out (host).  FtrlSolver::Update (src/ftrl.cc:75-88): z, n at the row's
 * coordinates from the prediction and the label.  Run is the two in turn for every row; these exist for
 * callers that drive the solver row by row as the reference's tests do -- one launch and one host wait per
 * call, nothing to measure. */
int admm_ftrl_predict(admm_ctx *ctx, const uint32_t *index, const float *value, uint32_t length, float *w,
                      const float *z, const float *n, const float *offset, const float *reg_offset, float alpha,
                      float beta, float l1, float l2, float *p_out);
int admm_ftrl_update(admm_ctx *ctx, const uint32_t *index, uint32_t length, float label, float predict,
                     const float *w, float *z, float *n, float alpha);

/* `count` independent sequential passes (admm_ftrl_pass with batch == 1) in one launch: pass k
 * runs sample set sets[k] with its own state w[k], z[k], n[k] and its own offset[k] /
 * reg_offset[k] (which may be shared between passes; the state vectors may not).  This is
 * Worker::BaseUpdate (src/workers.cc:51-87) of many workers at once: the reference runs one
 * worker per process, a GPU holds the shards of hundreds and runs their passes side by side, one
 * wavefront each.  Every pass gives exactly the result of its own admm_ftrl_pass call. */
int admm_ftrl_pass_multi(admm_csr *const *sets, float *const *w, float *const *z, float *const *n,
                         const float *const *offset, const float *const *reg_offset, int count,
                         float alpha, float beta, float l1, float l2);

/* ---- batch local objective for the L-BFGS local solver (no counterpart in the reference; the
 *      x-update of src/workers.cc:51-87 solved as a smooth problem) ------------------------- */

/* f = sum_rows [softplus(m) - t m] + y^T (x - z) + rho/2 ||x - z||^2 with m = X (x + offset),
 * g = X^T (sigmoid(m) - t) + y + rho (x - z).  offset, z, ydual may be NULL (= 0).
 * f_out: host double.  g_out: device, dim floats.  Needs the transposed copy. */
int admm_logistic_fg(admm_csr *csr, const float *x, const float *offset, const float *z,
                     const float *ydual, float rho, double *f_out, float *g_out);

typedef struct admm_lbfgs_params {
  int m;              /* history pairs, default 6 */
  int max_iter;       /* default 10 */
  int max_linesearch; /* default 20 */
  float epsilon;      /* ||g|| / max(1, ||x||) stop, default 1e-5 */
  float ftol;         /* Armijo, default 1e-4 */
  float wolfe;        /* curvature, default 0.9 */
  int linesearch;     /* 0 (default): trial points are evaluated from the margins X(x+offset) and
                         X d, one pass over per-row data each, and only the accepted point takes
                         a full evaluation; 1: every trial point is a full evaluation; 2: as 0, and the
                         accepted point is evaluated from the margins too (residuals, loss and the
                         mask columns' sums by per-row passes, then the transposed sweep), with a
                         full evaluation every third accepted point and at a solve's last iteration:
                         fastest, but its iterates differ from modes 0/1 in the last bits */
} admm_lbfgs_params;
typedef struct admm_lbfgs_result {
  double f;
  int evals; /* objective evaluations, trial points of the line search included */
  int iters;
  int forward_sweeps;    /* sweeps of the row-major form (X w) */
  int transposed_sweeps; /* sweeps of the column-block copy (X^T r) */
} admm_lbfgs_result;
/* Minimise the objective above over x (in place) with L-BFGS (two-loop recursion, backtracking
 * Armijo + weak Wolfe line search).  params == NULL uses the defaults. */
int admm_lbfgs_solve(admm_csr *csr, float *x, const float *offset, const float *z,
                     const float *ydual, float rho, const admm_lbfgs_params *params,
                     admm_lbfgs_result *result);

/* ---- consensus algebra -------------------------------------------------------------------- */

/* Worker::GetWeights (src/workers.cc:121-125): out_i = x_i * rho + y_i */
int admm_payload(admm_ctx *ctx, const float *x, const float *ydual, float rho, float *out,
                 uint64_t dim);
/* Master::GlobalUpdate (src/master.cc:5-20): z_i = S_i / (num_worker * rho + global_var); with
 * l1 > 0, S_i is soft-thresholded by l1 first. */
int admm_global_update(admm_ctx *ctx, const float *sum, int num_worker, float rho,
                       float global_var, float l1, float *z, uint64_t dim);
/* Worker::LangrangeUpdate (src/workers.cc:115-119): y_i += rho * (x_i - z_i) */
int admm_dual_update(admm_ctx *ctx, float *ydual, const float *x, const float *z, float rho,
                     uint64_t dim);
/* GlobalUpdate + LangrangeUpdate + residual norms as ONE elementwise pass:
 * z (in: previous consensus, out: new), ydual updated in place,
 * r2 = sum (x - z_new)^2, s2 = rho^2 sum (z_new - z_old)^2 (host doubles, may be NULL). */
int admm_consensus(admm_ctx *ctx, const float *sum, int num_worker, float rho, float global_var,
                   float l1, const float *x, float *ydual, float *z, uint64_t dim, double *r2,
                   double *s2);

/* admm_consensus for callers that cannot allocate device memory or wait for the GPU: code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart
 * in the reference.  Same update of z and ydual (bit for bit) and the same two residuals, but:
 *   - no context: the work is queued on `stream` (a hipStream_t, NULL = the default stream) of the
 *     calling thread's CURRENT device, and the call never synchronises;
 *   - residuals_out is a DEVICE pointer to two doubles {r2, s2} written by the last kernel (NULL:
 *     not wanted);
 *   - the per-workgroup partial sums live in a fixed 1 MiB array inside the library's code object
 *     (one copy per GPU, held until the process ends) instead of a context's scratch allocation,
 *     which limits dim to ADMM_CONSENSUS_STATIC_MAX_DIM; a larger dim is rejected with
 *     ADMM_ERR_INVALID and nothing is launched.
 * The call makes no hipMalloc / hipFree and calls no allocating routine.  All calls on one GPU
 * share that GPU's array: two calls that overlap on one GPU (different streams) race on it.
 * Precondition for callers under stream capture: the FIRST call on a device resolves that device's copy of the
 * array (one hipGetSymbolAddress, which may load the code object and is not legal while capturing); so make one
 * admm_ctx_create on the device, or one call of this function outside the capture, beforehand.  Every later call
 * reads the resolved address without a lock or a runtime query. */
#define ADMM_CONSENSUS_STATIC_MAX_DIM (1ull << 28)
int admm_consensus_static(void *stream, const float *sum, int num_worker, float rho,
                          float global_var, float l1, const float *x, float *ydual, float *z,
                          uint64_t dim, double *residuals_out);

/* ---- collective (rabit::Allreduce<op::Sum>, src/admm_allreduce.cpp:69) -------------------- */

/* RCCL communicator over the context's device.  unique_id: 128 bytes from
 * admm_comm_unique_id() on rank 0, handed to every rank by the launcher. */
int admm_comm_unique_id(void *unique_id_128);
int admm_comm_init(admm_ctx *ctx, int rank, int world, const void *unique_id_128);
/* In-place fp32 sum over ranks on the context's stream (world 1 without a communicator is a
 * no-op, as rabit's single-process mode). */
int admm_allreduce_sum_f32(admm_ctx *ctx, float *buf, uint64_t n);
/* Self-test of the RCCL binding, in the place of the reference's rabit smoke program
 * (src/admm_main.cpp:37-52).  Creates a one-rank communicator of its own on the context's device
 * and runs a sum all-reduce through it (works on a single GPU); with a communicator over more than
 * one rank it also checks a sum of rank-dependent values through that one (collective: every rank
 * calls it).  Synchronises the stream. */
int admm_comm_selftest(admm_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* ADMM_HIP_H_ */
