// Exact sequential FTRL-proximal pass: the row-by-row semantics of FtrlSolver::Run
// (src/ftrl.cc:101-145).  Row t+1's prediction depends on row t's update through every shared
// coordinate (the bias column is shared by all rows), so rows are walked in order by ONE
// wavefront; the parallelism is across the coordinates of a row (one lane each).
#include "internal.h"

using namespace admm_impl;

namespace admm_impl {

struct SeqArgs {
  const uint64_t *rowptr;
  const uint32_t *col;
  const uint32_t *perm;  // relabelled sets: stored id -> original id, else nullptr
  const float *val;
  const float *label;
  uint64_t nrows;
  float *w, *z, *n;
  const float *offset, *reg;
  float alpha, beta, l1, l2;
};

__device__ inline float seq_weight(float z, float n, float c, float alpha, float beta, float l1, float l2) {
  // src/ftrl.cc:61-66: threshold on l1, shrink by the literal 1, double denominator
  const float sign = z < 0.0f ? -1.0f : 1.0f;
  if (sign * z <= l1) return 0.0f;
  const float num = sign * 1 - z + l2 * c;
  const double den = ((double)beta + sqrt((double)n)) / (double)alpha + (double)l2;
  return (float)((double)num / den);
}

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ inline float seq_sigmoid(float margin) {
  // src/ftrl.cc:72
  const float m = fmaxf(fminf(margin, 35.0f), -35.0f);
  return (float)(1.0 / (1 + exp(-(double)m)));
}

// Lane-serial variant for sets in which some row stores a column twice: the reference then applies
// the update twice in sequence, so lane 0 walks every row element by element.
__global__ void __launch_bounds__(64) k_ftrl_seq_serial(SeqArgs a) {
  if (threadIdx.x != 0) return;
  float *w = a.w, *z = a.z, *n = a.n;
  for (uint64_t r = 0; r <= a.nrows; ++r) {
    const bool tail = r == a.nrows;  // trailing Predict on the last row (src/ftrl.cc:140-141)
    const uint64_t row = tail ? a.nrows - 1 : r;
    const uint64_t b = a.rowptr[row];
    const uint32_t len = (uint32_t)(a.rowptr[row + 1] - b);
    for (uint32_t k = 0; k < len; ++k) {
      const uint32_t j = a.perm ? a.perm[a.col[b + k]] : a.col[b + k];
      w[j] = seq_weight(z[j], n[j], a.reg[j], a.alpha, a.beta, a.l1, a.l2);
    }
    if (tail) break;
    float sw = 0.0f, so = 0.0f;
    for (uint32_t k = 0; k < len; ++k) {
      const uint32_t j = a.perm ? a.perm[a.col[b + k]] : a.col[b + k];
      const float v = a.val ? a.val[b + k] : 1.0f;
      sw += w[j] * v;
      so += a.offset[j] * v;
    }
    const float p = seq_sigmoid(sw + so);
    const int y = a.label[row] > 0 ? 1 : 0;
    const float g = p - y;
    for (uint32_t k = 0; k < len; ++k) {
      const uint32_t j = a.perm ? a.perm[a.col[b + k]] : a.col[b + k];
      const double sigma = (sqrt((double)(n[j] + g * g)) - sqrt((double)n[j])) / (double)a.alpha;
      z[j] = (float)((double)z[j] + ((double)g - sigma * (double)w[j]));
      n[j] = n[j] + g * g;
    }
  }
}

constexpr int kSeqAhead = 4;  // rows whose stored entries are fetched ahead of the row being solved

// One wavefront walks the rows in order; lane k owns the k-th stored coordinate of the row.
// Everything that does not depend on the solver state is fetched ahead: row pointers and labels a
// block of 64 rows at a time (one coalesced load, then lane shuffles), column ids and values
// kSeqAhead rows ahead into registers.  What remains on the per-row chain is the gather of the
// touched state, the closed-form weights, one wave reduction, the sigmoid, and the scatter.
__global__ void __launch_bounds__(64) k_ftrl_seq(SeqArgs a) {
  const int lane = threadIdx.x;
  float *w = a.w, *z = a.z, *n = a.n;
  const uint64_t total = a.nrows + 1;  // + the trailing Predict on the last row
  unsigned long long rp_lo = 0, rp_hi = 0;
  float lab_blk = 0.0f;
  uint32_t pcol[kSeqAhead];
  float pval[kSeqAhead];
  auto row_of = [&](uint64_t t) { return t < a.nrows ? t : a.nrows - 1; };
  // direct (not block-cached) fetch used for the look-ahead rows
  auto fetch = [&](uint64_t t, uint32_t &c, float &v) {
    c = 0;
    v = 0.0f;
    if (t >= total) return;
    const uint64_t row = row_of(t);
    const uint64_t b = a.rowptr[row];
    const uint32_t len = (uint32_t)(a.rowptr[row + 1] - b);
    if ((uint32_t)lane < len) {
      c = a.col[b + lane];
      v = a.val ? a.val[b + lane] : 1.0f;
    }
  };
#pragma unroll
  for (int u = 0; u < kSeqAhead; ++u) fetch(u, pcol[u], pval[u]);
  for (uint64_t t0 = 0; t0 < total; t0 += kSeqAhead) {
#pragma unroll
    for (int u = 0; u < kSeqAhead; ++u) {
      const uint64_t t = t0 + u;
      if (t >= total) break;
      const bool tail = t == a.nrows;
      const uint64_t row = row_of(t);
      if ((row & 63) == 0 || t == 0 || tail) {
        const uint64_t rb = row & ~63ull;
        const uint64_t rr = rb + lane < a.nrows ? rb + lane : a.nrows - 1;
        rp_lo = a.rowptr[rr];
        rp_hi = a.rowptr[rr + 1];
        lab_blk = a.label[rr];
      }
      const int rl = (int)(row & 63);
      const unsigned long long b = __shfl(rp_lo, rl, 64);
      const uint32_t len = (uint32_t)(__shfl(rp_hi, rl, 64) - b);
      const float lab = __shfl(lab_blk, rl, 64);
      // this row's first 64 entries arrived kSeqAhead rows ago; start the fetch for a later row
      const uint32_t c0 = pcol[u];
      const float v0 = pval[u];
      fetch(t + kSeqAhead, pcol[u], pval[u]);
      float sw = 0.0f, so = 0.0f;
      uint32_t j0 = 0;
      float z0 = 0.0f, n0 = 0.0f, w0 = 0.0f;
      for (uint32_t base = 0; base < len; base += 64) {
        const uint32_t k = base + lane;
        if (k < len) {
          uint32_t j = base == 0 ? c0 : a.col[b + k];
          if (a.perm) j = a.perm[j];
          const float v = base == 0 ? v0 : (a.val ? a.val[b + k] : 1.0f);
          const float zj = z[j], nj = n[j];
          const float wj = seq_weight(zj, nj, a.reg[j], a.alpha, a.beta, a.l1, a.l2);
          w[j] = wj;
          sw += wj * v;
          so += a.offset[j] * v;
          if (base == 0) {
            j0 = j;
            z0 = zj;
            n0 = nj;
            w0 = wj;
          }
        }
      }
      if (!tail) {
        const float p = seq_sigmoid(wave_sum(sw) + wave_sum(so));
        const int y = lab > 0 ? 1 : 0;
        const float g = p - y;
        const float g2 = g * g;
        for (uint32_t base = 0; base < len; base += 64) {
          const uint32_t k = base + lane;
          if (k < len) {
            uint32_t j;
            float zj, nj, wj;
            if (base == 0) {
              j = j0;
              zj = z0;
              nj = n0;
              wj = w0;
            } else {
              j = a.col[b + k];
              if (a.perm) j = a.perm[j];
              zj = z[j];
              nj = n[j];
              wj = w[j];
            }
            const double sigma = (sqrt((double)(nj + g2)) - sqrt((double)nj)) / (double)a.alpha;
            z[j] = (float)((double)zj + ((double)g - sigma * (double)wj));
            n[j] = nj + g2;
          }
        }
      }
      // the next row's gathers (other lanes) must observe this row's stores
      __threadfence_block();
    }
  }
}

int ftrl_pass_sequential(admm_csr *csr, float *w, float *z, float *n, const float *offset, const float *reg_offset,
                         float alpha, float beta, float l1, float l2) {
  admm_ctx *ctx = csr->ctx;
  if (csr->nrows == 0) return ADMM_OK;
  SeqArgs a{};
  a.rowptr = csr->d_rowptr;
  a.col = csr->d_col;
  a.perm = csr->d_perm;
  a.val = csr->d_val;
  a.label = csr->d_label;
  a.nrows = csr->nrows;
  a.w = w;
  a.z = z;
  a.n = n;
  a.offset = offset;
  a.reg = reg_offset;
  a.alpha = alpha;
  a.beta = beta;
  a.l1 = l1;
  a.l2 = l2;
  KernelTimer t(ctx, ADMM_K_FTRL_SEQ);
  if (csr->has_dup_cols)
    hipLaunchKernelGGL(k_ftrl_seq_serial, dim3(1), dim3(64), 0, ctx->stream, a);
  else
    hipLaunchKernelGGL(k_ftrl_seq, dim3(1), dim3(64), 0, ctx->stream, a);
  ADMM_HIP_CHECK(hipGetLastError());
  return ADMM_OK;
}

}  // namespace admm_impl
