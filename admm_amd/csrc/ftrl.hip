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

// SERIAL = false: one lane per stored coordinate (requires rows without repeated columns).
// SERIAL = true : lane 0 walks the row element by element, for rows that store a column twice
//                 (the reference then applies the update twice in sequence).
template <bool SERIAL>
__global__ void __launch_bounds__(64) k_ftrl_seq(SeqArgs a) {
  const int lane = threadIdx.x;
  float *w = a.w, *z = a.z, *n = a.n;
  for (uint64_t r = 0; r <= a.nrows; ++r) {
    // the extra trip (r == nrows) is the trailing Predict on the last row (src/ftrl.cc:140-141)
    const bool tail = r == a.nrows;
    const uint64_t row = tail ? a.nrows - 1 : r;
    const uint64_t b = a.rowptr[row];
    const uint32_t len = (uint32_t)(a.rowptr[row + 1] - b);
    if (SERIAL) {
      if (lane == 0) {
        for (uint32_t k = 0; k < len; ++k) {
          const uint32_t j = a.perm ? a.perm[a.col[b + k]] : a.col[b + k];
          w[j] = seq_weight(z[j], n[j], a.reg[j], a.alpha, a.beta, a.l1, a.l2);
        }
        if (!tail) {
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
    } else {
      float sw = 0.0f, so = 0.0f;
      // first (usually only) chunk stays in registers for the update
      uint32_t j0 = 0;
      float z0 = 0.0f, n0 = 0.0f, w0 = 0.0f;
      for (uint32_t base = 0; base < len; base += 64) {
        const uint32_t k = base + lane;
        if (k < len) {
          const uint32_t j = a.perm ? a.perm[a.col[b + k]] : a.col[b + k];
          const float v = a.val ? a.val[b + k] : 1.0f;
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
        const int y = a.label[row] > 0 ? 1 : 0;
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
              j = a.perm ? a.perm[a.col[b + k]] : a.col[b + k];
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
    }
    // the next row's gathers (other lanes) must observe this row's stores
    __threadfence_block();
    if (a.nrows == 0) break;
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
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ftrl_seq<true>), dim3(1), dim3(64), 0, ctx->stream, a);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ftrl_seq<false>), dim3(1), dim3(64), 0, ctx->stream, a);
  ADMM_HIP_CHECK(hipGetLastError());
  return ADMM_OK;
}

}  // namespace admm_impl
