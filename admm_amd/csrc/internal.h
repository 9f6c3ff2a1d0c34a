// Shared internals of libadmm_hip.so (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "admm_hip.h"

namespace admm_impl {

void set_error(const char *fmt, ...);

#define ADMM_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      ::admm_impl::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),       \
                             __FILE__, __LINE__);                                         \
      return e_ == hipErrorOutOfMemory ? ADMM_ERR_NOMEM : ADMM_ERR_HIP;                   \
    }                                                                                     \
  } while (0)

#define ADMM_REQUIRE(cond, ...)               \
  do {                                        \
    if (!(cond)) {                            \
      ::admm_impl::set_error(__VA_ARGS__);    \
      return ADMM_ERR_INVALID;                \
    }                                         \
  } while (0)

#define ADMM_TRY(expr)          \
  do {                          \
    int rc_ = (expr);           \
    if (rc_ != ADMM_OK) return rc_; \
  } while (0)

// Relabelled sets keep their unit-valued entries in the most frequent columns as bits per row:
// one 64-bit word for columns 0..63 (`hmask`) and kWarmWords more for the columns after them
// (`hwarm`, row-major).  Those entries are stored nowhere else.
constexpr int kWarmWords = 3;
constexpr uint32_t kMaskCols = 64u * (1 + kWarmWords);

// leading entries of the gathered vector that the row-panel forward sweep keeps in LDS (a multiple
// of every bin width the copy's builder can choose)
constexpr uint32_t kFpHot = 8192;
constexpr int kHpLanes = 512;  // lanes that share a panel's hot stream = threads of the sweep's workgroup
// column stream of the transposed sweep: rows per panel (row ids take 13 bits of a word), lanes per panel
constexpr uint32_t kTcRows = 8192;
constexpr int kTcLanes = 1024;

constexpr int kMaxPartialBlocks = 4096;  // reduction partial slots per output scalar
constexpr int kPartialScalars = 8;       // scalars a reducing kernel may emit

struct TimingSlot {
  hipEvent_t start, stop;
  int kernel;
};

}  // namespace admm_impl

struct admm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int cu_count = 256;
  // reduction scratch: kPartialScalars x kMaxPartialBlocks doubles on the device, plus a pinned
  // host mirror for the final few scalars
  double *d_partials = nullptr;
  double *d_scalars = nullptr;  // kPartialScalars doubles (deterministic final sums)
  double *h_scalars = nullptr;  // pinned, mapped into the device: kPartialScalars doubles, then the sequence number
  double *h_scalars_dev = nullptr;             // the device's address of h_scalars
  unsigned long long *h_seq = nullptr, *h_seq_dev = nullptr;
  unsigned long long seq_issued = 0;            // reductions queued so far
  // timing
  bool timing = false;
  unsigned timing_classes = ~0u;  // bit k: kernels of class ADMM_K_<k> are timed while `timing` is on
  std::vector<admm_impl::TimingSlot> pending;
  std::vector<admm_impl::TimingSlot> free_slots;
  double k_ms[ADMM_K_COUNT] = {};
  uint64_t k_launches[ADMM_K_COUNT] = {};
  unsigned timing_every = 1;  // of each class, every timing_every-th launch is timed
  uint64_t k_seen[ADMM_K_COUNT] = {};
  // L-BFGS workspace (lazily sized)
  float *lbfgs_ws = nullptr;
  uint64_t lbfgs_ws_floats = 0;
  // AUC workspace (sort keys, flags, rank starts, the sort's own scratch); grows, never shrinks
  void *row_ws = nullptr;  // one row's entries for admm_ftrl_predict / admm_ftrl_update (grow-only)
  size_t row_ws_bytes = 0;
  void *auc_ws = nullptr;
  size_t auc_ws_bytes = 0;
  // device scalars of the L-BFGS two-loop recursion (see admm_lbfgs_solve)
  double *d_lb = nullptr;
  // argument table of a multi-pass FTRL launch
  void *d_multi = nullptr;
  size_t multi_bytes = 0;
  // barrier words and running maxima of the one-launch mini-batch FTRL pass
  void *d_ps = nullptr;
  size_t ps_bytes = 0;
  // RCCL
  void *comm = nullptr;
  int rank = 0, world = 1;
};

struct admm_csr {
  admm_ctx *ctx = nullptr;
  uint64_t nrows = 0, nnz = 0, dim = 0;
  bool has_val = false;       // the set stores values (else every value is 1: a pattern matrix)
  // lean: the caller uses the set for whole-set sweeps only, so the builders free every row-major form as soon as
  // the copies that replace it exist; rows_released: that has happened (d_rowptr, d_col, d_val, d_crowptr, d_ccol,
  // d_cval are gone), and whatever reads rows fails with ADMM_ERR_STATE
  bool lean = false, rows_released = false;
  bool has_bias = false;      // column 0 is the all-ones bias column (present in every row)
  bool has_dup_cols = false;  // some row stores a column twice
  uint64_t *d_rowptr = nullptr;
  uint32_t *d_col = nullptr;
  float *d_val = nullptr;  // nullptr = implicit 1
  float *d_label = nullptr;
  float *d_rowbuf = nullptr;  // nrows floats: p / residual / per-row gradient scale
  float *d_margin = nullptr;  // nrows floats each: margins X(x+offset) at the L-BFGS iterate, and X d
  float *d_dmargin = nullptr;
  // what the last L-BFGS solve ended on: its x and offset (to recognise them), the loss part of the
  // gradient there and the loss; d_margin belongs to the same point when cache_margin is set
  float *d_cache_x = nullptr, *d_cache_off = nullptr, *d_cache_gl = nullptr;
  double cache_loss = 0;
  bool cache_valid = false, cache_has_off = false, cache_margin = false;
  int margin_age = 0;  // accepted steps added to d_margin since a forward sweep last recomputed it
  float *d_wsum = nullptr;    // dim floats: the vector the forward sweep gathers (relabelled order)
  // columns relabelled by decreasing frequency: d_col holds NEW ids; perm: new -> old, inv: old -> new
  uint32_t *d_perm = nullptr, *d_inv = nullptr;
  // compact row-major form used by the forward sweep of a relabelled set: per row a 64-bit mask of
  // its unit-valued entries in the 64 most frequent columns, and the remaining entries as CSR
  bool hot_dense = false;
  uint64_t *d_hmask = nullptr, *d_hwarm = nullptr, *d_crowptr = nullptr;
  // d_hmask transposed in tiles of 64 rows: word 64 t + c holds, for rows 64 t .. 64 t + 63, who stores column c
  uint64_t *d_hmaskT = nullptr;
  uint32_t *d_ccol = nullptr;
  float *d_cval = nullptr;
  uint64_t cnnz = 0;
  uint32_t fwd_lpr = 16;      // lanes per row of the forward sweep
  // sequential FTRL pass: per stored entry, how many rows back its column was last stored (0 =
  // never or more than 65535 rows back) and at which position of that row: (rows_back << 16) | pos
  uint32_t *d_dep = nullptr;
  uint32_t *d_seqcol = nullptr;  // relabelled sets: the stored columns as original ids, for that pass
  uint64_t max_row_len = 0;
  // transposed (column-block) copy
  bool has_t = false;
  bool t_named = false;  // the caller named the panel size: the copy serves sweeps over panel ranges (mini-batches)
  uint64_t panel_rows = 0;
  uint32_t col_block = 0, nblocks = 0;
  uint64_t npanels = 0, tnnz = 0;
  uint64_t *d_tptr = nullptr;  // npanels*nblocks + 1
  uint32_t *d_trow = nullptr;
  uint16_t *d_tcol = nullptr;
  float *d_tval = nullptr;  // nullptr when d_val is
  std::vector<uint64_t> h_tptr;
  // Column stream (whole-set sweeps of a copy built with automatic panels): the entries whose stored
  // column id is below tc_cols are not in the cells above.  Rows are cut into panels of kTcRows; a
  // panel's entries are 16-bit words in column order (row inside the panel, bit 15 set on the last
  // word of its column; a column without entries in the panel holds one filler word with row
  // kTcRows), cut into kTcLanes equal runs and interleaved like the forward sweep's hot stream:
  // word i of panel p is at d_tcbase[p] + ((k / 4) kTcLanes + t) 4 + k % 4, S = words per lane,
  // t = i / S, k = i % S; d_tcfirst[p kTcLanes + t] is the column lane t starts in.
  bool has_tc = false;
  uint32_t tc_cols = 0;
  uint16_t *d_tcw = nullptr;
  float *d_tcv = nullptr;
  uint64_t *d_tcbase = nullptr;
  uint16_t *d_tcfirst = nullptr;
  uint64_t tc_words = 0, tc_entries = 0;
  // Row-panel copy read by the forward sweep of a whole set: rows cut into panels of fp_rows rows.  The
  // entries the row masks do not hold (and, in a set without masks, not the bias column) are in one
  // of two forms, by stored column id (a frequency rank in a relabelled set):
  // * id < fp_hot, the "hot stream": the panel's rows one after another, every entry a 16-bit word
  //   (id, bit 15 set on the last entry of its row; a row without such entries holds one filler word
  //   with id fp_hot), cut into kHpLanes equal runs, one per lane of the sweep's workgroup, and
  //   stored so that the lanes' k-th groups of four entries are adjacent: entry i of the panel is at
  //   d_hbase[p] + ((k / 4) kHpLanes + t) 4 + k % 4 with S = (d_hbase[p+1] - d_hbase[p]) / kHpLanes,
  //   t = i / S, k = i % S; d_hfirst[p kHpLanes + t] is the row lane t starts in.
  // * the rest: contiguous per panel from d_fptr[p], in the order of their ids, as (id u32, row
  //   inside the panel u16 [, value f32]).
  bool has_fp = false;
  uint32_t fp_rows = 0, fp_bin_cols = 0, fp_hot = 0;
  uint32_t *d_fcol = nullptr;
  uint16_t *d_frow = nullptr;
  float *d_fval = nullptr;
  uint64_t *d_fptr = nullptr;
  uint16_t *d_hcol = nullptr;
  float *d_hval = nullptr;
  uint64_t *d_hbase = nullptr;
  uint16_t *d_hfirst = nullptr;
  uint64_t fp_nnz = 0, hp_words = 0;  // entries of the second form; words of the first (fillers included)
  int fp_lenbits = 0;         // ceil(log2(longest stored row + 1))
  // per-coordinate accumulators of the transposed sweep: 64-bit fixed point (value * 2^fx_shift),
  // so that sums are exact integer adds (order-independent, reproducible)
  long long *d_g1 = nullptr, *d_g2 = nullptr;
  // X^T r of the first kFpHot STORED ids (row masks' columns and the column stream's), by stored id: those
  // sums come from every workgroup of their kernels, and adds to adjacent words share a memory request
  // where adds to their scattered caller's ids take one each; k_fg_finish picks them up and clears them
  long long *d_gs = nullptr;
  long long *d_mpart = nullptr;  // mini-batch FTRL: the mask columns' two sums, spread over a few copies by the forward sweep
  int fx_shift = 40;
  float vmax = 1.0f;  // max |val| of the set
};

namespace admm_impl {

// RAII-less helpers -------------------------------------------------------------------------
int timing_begin(admm_ctx *ctx, int kernel);  // returns slot index or -1
void timing_end(admm_ctx *ctx, int slot);
int timing_flush(admm_ctx *ctx);

struct KernelTimer {
  admm_ctx *ctx;
  int slot;
  KernelTimer(admm_ctx *c, int kernel) : ctx(c), slot(c->timing && ((c->timing_classes >> kernel) & 1u) && (c->k_seen[kernel]++ % c->timing_every) == 0 ? timing_begin(c, kernel) : -1) {}
  ~KernelTimer() {
    if (slot >= 0) timing_end(ctx, slot);
  }
};

// sums `nscalars` rows of `nblocks` partials (row-major [scalar][block]) into ctx->d_scalars in
// a fixed order, copies them to ctx->h_scalars and synchronises the stream
int finish_partials(admm_ctx *ctx, int nscalars, int nblocks);
// queues the sums of partial rows first .. first + count - 1 (count <= 4; rows below `split` hold
// nblocks_a partials, the others nblocks_b); wait_scalars returns once ctx->h_scalars holds the
// results of everything queued
int reduce_rows_queue(admm_ctx *ctx, int first, int count, int split, int nblocks_a, int nblocks_b);
int wait_scalars(admm_ctx *ctx);
// After a wait for the stream: has a kernel reported through the pinned block that it gave up (the one-launch
// mini-batch FTRL pass at a device-wide barrier)?  The report is sticky until it has been returned once.
int check_device_reports(admm_ctx *ctx);

int ensure_rowbuf(admm_csr *csr);
int build_forward_panels(admm_csr *csr);  // copies.hip
int build_dep(admm_csr *csr);             // copies.hip: previous-occurrence table of the sequential FTRL pass
int build_transposed(admm_csr *csr, uint64_t panel_rows, uint32_t col_block, bool column_stream);  // copies.hip
int ensure_taccum(admm_csr *csr);
// fails (ADMM_ERR_STATE) when the set no longer holds its rows (admm_csr_set_lean); `what` names the caller
int require_rows(const admm_csr *csr, const char *what);
// what a lean set frees once the forms that replace it exist (copies.hip): the row-major CSR, the compact rows
int release_row_major(admm_csr *csr);
int release_compact(admm_csr *csr);
// mini-batch FTRL pass in one launch (ftrl_batch.hip)
bool ftrl_persistent_eligible(const admm_csr *csr, uint64_t batch);
int ftrl_pass_persistent(admm_csr *csr, float *w, float *z, float *n, const float *offset, const float *reg_offset,
                         float alpha, float beta, float l1, float l2, const float *folded_max);

// sweeps (forward_panels.hip, forward_rows.hip, transposed.hip)
enum FwdMode { FWD_EVAL = 0, FWD_LBFGS = 1, FWD_FTRL = 2, FWD_MARGIN = 3 };
struct FwdResult {
  double s0, s1, s2;
};
// margins over rows [r0, r1) with up to 4 summed weight vectors; writes csr->d_rowbuf[r]
// (or out_rows when given) and returns mode-dependent sums through ctx->h_scalars[0..2]
// (after finish_partials; the caller decides when to read them).
// FWD_MARGIN writes the plain row sums X w and nothing else; FWD_LBFGS also writes them to
// margins_out when that is given.
int launch_forward(admm_csr *csr, FwdMode mode, uint64_t r0, uint64_t r1, const float *const *w,
                   int nw, float *out_rows, int *nblocks_out, float *margins_out = nullptr);
// transposed sweep over panels [p0, p1): accumulates into csr->d_g1 (and d_g2 in FTRL
// mode), which must be zero on entry
int launch_transposed(admm_csr *csr, bool ftrl_mode, uint64_t p0, uint64_t p1);

// vector helpers (ctx.hip)
int vec_dot(admm_ctx *ctx, const float *a, const float *b, uint64_t n, double *out_host);
int vec_axpy(admm_ctx *ctx, float *y, float a, const float *x, uint64_t n);         // y += a x
int vec_scale(admm_ctx *ctx, float *y, float a, uint64_t n);                        // y *= a
int vec_lincomb(admm_ctx *ctx, float *out, const float *a, float s, const float *b,
                uint64_t n);  // out = a + s b
int vec_sub(admm_ctx *ctx, float *out, const float *a, const float *b, uint64_t n);
void comm_release(admm_ctx *ctx);  // comm.hip

inline int grid_for(uint64_t work_items, int block, int max_blocks) {
  uint64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > (uint64_t)max_blocks) g = max_blocks;
  return (int)g;
}

}  // namespace admm_impl
