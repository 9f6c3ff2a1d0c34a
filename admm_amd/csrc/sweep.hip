// The two sweeps over a sample set and everything built from them:
//   forward    row-parallel CSR sweep: margins, sigmoid, per-row residual, loss sums
//   transposed column-block sweep: X^T r with per-column accumulators staged in LDS
// plus prediction / logloss, the batch objective f,g, the L-BFGS driver and the mini-batch FTRL
// pass.
#include "internal.h"

#include <algorithm>
#include <cmath>

using namespace admm_impl;

namespace admm_impl {

// ---------------------------------------------------------------------------------------------
// forward sweep
// ---------------------------------------------------------------------------------------------

struct FwdArgs {
  const uint64_t *rowptr;
  const uint32_t *col;
  const float *val;
  const float *label;
  uint64_t r0, r1;
  const float *w[4];
  int nw;
  float *out;
  double *partials;
  int mode;
};

__device__ inline float clamp35(float m) { return fmaxf(fminf(m, 35.0f), -35.0f); }

// A wave owns a tile of 64 consecutive rows.  LPR lanes cooperate on one row (64/LPR rows at a
// time); the 64 margins of the tile are parked in LDS so that the sigmoid / loss epilogue runs
// once per tile with one row per lane and the per-row output is one coalesced 256-B store.
template <int LPR>
__global__ void __launch_bounds__(256) k_forward(FwdArgs a) {
  __shared__ float sm[4][64];
  __shared__ double shred[4];
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, grp = lane / LPR;
  const uint64_t nrows = a.r1 - a.r0;
  const uint64_t ntiles = (nrows + 63) / 64;
  double acc0 = 0, acc1 = 0;
  // block-uniform trip count (the tile loop holds a workgroup barrier)
  for (uint64_t tb = (uint64_t)blockIdx.x * 4; tb < ntiles; tb += (uint64_t)gridDim.x * 4) {
    const uint64_t tile = tb + wave;
    const uint64_t rbase = a.r0 + tile * 64;
    const uint64_t myrow = rbase + lane;
    const bool live = tile < ntiles && myrow < a.r1;
    unsigned long long lo = 0, hi = 0;
    if (live) {
      lo = a.rowptr[myrow];
      hi = a.rowptr[myrow + 1];
    }
    for (int i = 0; i < LPR; ++i) {
      const int rit = i * RPW + grp;
      const unsigned long long b = __shfl(lo, rit, 64), e = __shfl(hi, rit, 64);
      float s = 0.0f;
      for (unsigned long long k = b + sub; k < e; k += LPR) {
        const uint32_t j = a.col[k];
        float wj = a.w[0][j];
        if (a.nw > 1) wj += a.w[1][j];
        if (a.nw > 2) wj += a.w[2][j];
        if (a.nw > 3) wj += a.w[3][j];
        s += wj * (a.val ? a.val[k] : 1.0f);
      }
#pragma unroll
      for (int off = LPR / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if (sub == 0) sm[wave][rit] = s;
    }
    __syncthreads();
    if (live) {
      const float m = sm[wave][lane];
      const float lab = a.label[myrow];
      if (a.mode == FWD_EVAL) {
        // src/metrics.cc:104-105: double predict, -log p / -log(1-p)
        const double pd = 1.0f / (1 + exp(-(double)clamp35(m)));
        a.out[myrow] = (float)pd;
        acc0 += lab > 0 ? -log(pd) : -log(1 - pd);
      } else if (a.mode == FWD_LBFGS) {
        const float t = lab > 0 ? 1.0f : 0.0f;
        const float loss = log1pf(expf(-fabsf(m))) + (t > 0 ? fmaxf(-m, 0.0f) : fmaxf(m, 0.0f));
        const float res = 1.0f / (1.0f + expf(-m)) - t;
        a.out[myrow] = res;
        acc0 += (double)loss;
        acc1 += (double)res;
      } else {
        // src/ftrl.cc:72,76-78
        const float p = (float)(1.0 / (1 + exp(-(double)clamp35(m))));
        const int y = lab > 0 ? 1 : 0;
        const float g = p - y;
        a.out[myrow] = g;
        acc0 += (double)g;
        acc1 += (double)(g * g);
      }
    }
    __syncthreads();
  }
  // block reduction of the two sums
  for (int off = 32; off > 0; off >>= 1) {
    acc0 += __shfl_down(acc0, off, 64);
    acc1 += __shfl_down(acc1, off, 64);
  }
  if (lane == 0) shred[wave] = acc0;
  __syncthreads();
  if (threadIdx.x == 0) a.partials[0 * kMaxPartialBlocks + blockIdx.x] = shred[0] + shred[1] + shred[2] + shred[3];
  __syncthreads();
  if (lane == 0) shred[wave] = acc1;
  __syncthreads();
  if (threadIdx.x == 0) a.partials[1 * kMaxPartialBlocks + blockIdx.x] = shred[0] + shred[1] + shred[2] + shred[3];
}

int launch_forward(admm_csr *csr, FwdMode mode, uint64_t r0, uint64_t r1, const float *const *w, int nw,
                   float *out_rows, int *nblocks_out) {
  admm_ctx *ctx = csr->ctx;
  ADMM_REQUIRE(nw >= 1 && nw <= 4, "forward sweep: 1..4 weight vectors, got %d", nw);
  ADMM_REQUIRE(r0 <= r1 && r1 <= csr->nrows, "forward sweep: bad row range");
  FwdArgs a{};
  a.rowptr = csr->d_rowptr;
  a.col = csr->d_col;
  a.val = csr->d_val;
  a.label = csr->d_label;
  a.r0 = r0;
  a.r1 = r1;
  for (int i = 0; i < nw; ++i) a.w[i] = w[i];
  a.nw = nw;
  a.out = out_rows;
  a.partials = ctx->d_partials;
  a.mode = (int)mode;
  const uint64_t ntiles = (r1 - r0 + 63) / 64;
  const int grid = grid_for(ntiles * 64, 256, ctx->cu_count * 8);
  *nblocks_out = grid;
  KernelTimer t(ctx, ADMM_K_FORWARD);
  switch (csr->fwd_lpr) {
    case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward<4>), dim3(grid), dim3(256), 0, ctx->stream, a); break;
    case 8: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward<8>), dim3(grid), dim3(256), 0, ctx->stream, a); break;
    case 16: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward<16>), dim3(grid), dim3(256), 0, ctx->stream, a); break;
    case 32: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward<32>), dim3(grid), dim3(256), 0, ctx->stream, a); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward<64>), dim3(grid), dim3(256), 0, ctx->stream, a); break;
  }
  ADMM_HIP_CHECK(hipGetLastError());
  return ADMM_OK;
}

// ---------------------------------------------------------------------------------------------
// transposed sweep
// ---------------------------------------------------------------------------------------------

struct TArgs {
  const uint64_t *tptr;
  const uint32_t *trow;
  const uint16_t *tcol;
  const float *tval;
  const float *rvec;  // one float per row
  float *g1, *g2;
  uint32_t *mask;
  uint32_t nblocks, col_block;
  uint64_t cell0;
};

constexpr int kTThreads = 512;

// One workgroup per (panel, column block) cell.  The cell's per-column sums are accumulated with
// LDS atomics (the rows of a cell are visited almost in order, so the gather of rvec is close to
// a sequential read), then added to the global per-column vector with one coalesced atomic pass.
template <bool FTRL>
__global__ void __launch_bounds__(kTThreads) k_transposed(TArgs a) {
  extern __shared__ float acc[];  // [CB] (+ [CB] + [CB/32] bitmap in FTRL mode)
  const uint64_t cell = a.cell0 + blockIdx.x;
  const uint64_t beg = a.tptr[cell], end = a.tptr[cell + 1];
  if (beg == end) return;
  const uint32_t CB = a.col_block;
  const uint32_t blk = (uint32_t)(cell % a.nblocks);
  float *acc2 = acc + CB;
  uint32_t *bits = reinterpret_cast<uint32_t *>(acc + 2 * CB);
  for (uint32_t c = threadIdx.x; c < CB; c += kTThreads) {
    acc[c] = 0.0f;
    if (FTRL) acc2[c] = 0.0f;
  }
  if (FTRL)
    for (uint32_t c = threadIdx.x; c < CB / 32; c += kTThreads) bits[c] = 0u;
  __syncthreads();
  uint64_t k = beg + threadIdx.x;
  // four independent entries per thread per trip keep more gathers in flight
  for (; k + 3 * kTThreads < end; k += 4 * kTThreads) {
    uint32_t r[4];
    uint16_t c[4];
    float v[4], x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      r[u] = a.trow[k + u * kTThreads];
      c[u] = a.tcol[k + u * kTThreads];
      v[u] = (!FTRL && a.tval) ? a.tval[k + u * kTThreads] : 1.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = a.rvec[r[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (FTRL) {
        atomicAdd(&acc[c[u]], x[u]);
        atomicAdd(&acc2[c[u]], x[u] * x[u]);
        atomicOr(&bits[c[u] >> 5], 1u << (c[u] & 31));
      } else {
        atomicAdd(&acc[c[u]], x[u] * v[u]);
      }
    }
  }
  for (; k < end; k += kTThreads) {
    const uint32_t r = a.trow[k];
    const uint16_t c = a.tcol[k];
    const float x = a.rvec[r];
    if (FTRL) {
      atomicAdd(&acc[c], x);
      atomicAdd(&acc2[c], x * x);
      atomicOr(&bits[c >> 5], 1u << (c & 31));
    } else {
      atomicAdd(&acc[c], x * (a.tval ? a.tval[k] : 1.0f));
    }
  }
  __syncthreads();
  const uint64_t gbase = (uint64_t)blk * CB;
  for (uint32_t c = threadIdx.x; c < CB; c += kTThreads) {
    const float s = acc[c];
    if (FTRL) {
      const uint32_t touched = (bits[c >> 5] >> (c & 31)) & 1u;
      if (touched) {
        atomicAdd(&a.g1[gbase + c], s);
        atomicAdd(&a.g2[gbase + c], acc2[c]);
      }
    } else if (s != 0.0f) {
      atomicAdd(&a.g1[gbase + c], s);
    }
  }
  if (FTRL)
    for (uint32_t c = threadIdx.x; c < CB / 32; c += kTThreads)
      if (bits[c]) atomicOr(&a.mask[gbase / 32 + c], bits[c]);
}

int launch_transposed(admm_csr *csr, bool ftrl_mode, uint64_t p0, uint64_t p1) {
  admm_ctx *ctx = csr->ctx;
  ADMM_REQUIRE(csr->has_t, "transposed sweep: call admm_csr_build_transposed first");
  if (p0 >= p1) return ADMM_OK;
  TArgs a{};
  a.tptr = csr->d_tptr;
  a.trow = csr->d_trow;
  a.tcol = csr->d_tcol;
  a.tval = csr->d_tval;
  a.rvec = csr->d_rowbuf;
  a.g1 = csr->d_g1;
  a.g2 = csr->d_g2;
  a.mask = csr->d_mask;
  a.nblocks = csr->nblocks;
  a.col_block = csr->col_block;
  const uint64_t ncells = (p1 - p0) * csr->nblocks;
  const size_t lds = ftrl_mode ? sizeof(float) * 2 * csr->col_block + sizeof(uint32_t) * (csr->col_block / 32)
                               : sizeof(float) * csr->col_block;
  KernelTimer t(ctx, ADMM_K_TRANSPOSED);
  // the grid's x dimension is limited to 2^31-1 blocks; cut very long cell ranges
  const uint64_t kMaxGrid = 1u << 30;
  for (uint64_t c0 = 0; c0 < ncells; c0 += kMaxGrid) {
    a.cell0 = p0 * csr->nblocks + c0;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(kMaxGrid, ncells - c0);
    if (ftrl_mode)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_transposed<true>), dim3(grid), dim3(kTThreads), lds, ctx->stream, a);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_transposed<false>), dim3(grid), dim3(kTThreads), lds, ctx->stream, a);
    ADMM_HIP_CHECK(hipGetLastError());
  }
  return ADMM_OK;
}

// ---------------------------------------------------------------------------------------------
// f, g assembly
// ---------------------------------------------------------------------------------------------

// g = acc + y + rho (x - z), acc reset to 0; column 0 also receives the bias-column sum the
// forward sweep produced (scalars[1]); partials row 2 <- y.(x-z) + rho/2 |x-z|^2
__global__ void __launch_bounds__(256) k_fg_finish(float *__restrict__ accum, const float *__restrict__ x,
                                                   const float *__restrict__ z, const float *__restrict__ y,
                                                   float rho, const double *__restrict__ scalars, int add_bias,
                                                   float *__restrict__ g, uint64_t n, double *__restrict__ partials) {
  __shared__ double sh[4];
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  double reg = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float a = accum[i];
    accum[i] = 0.0f;
    if (add_bias && i == 0) a += (float)scalars[1];
    const float d = x[i] - (z ? z[i] : 0.0f);
    const float yd = y ? y[i] : 0.0f;
    g[i] = a + yd + rho * d;
    reg += (double)yd * (double)d + 0.5 * (double)rho * (double)d * (double)d;
  }
  for (int off = 32; off > 0; off >>= 1) reg += __shfl_down(reg, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = reg;
  __syncthreads();
  if (threadIdx.x == 0) partials[2 * kMaxPartialBlocks + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ void __launch_bounds__(256) k_finish_row(const double *partials, int row, int nblocks, double *scalars) {
  __shared__ double sh[256];
  double acc = 0;
  for (int i = threadIdx.x; i < nblocks; i += 256) acc += partials[(size_t)row * kMaxPartialBlocks + i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) scalars[row] = sh[0];
}

static int reduce_rows(admm_ctx *ctx, int first, int count, int nblocks) {
  for (int r = first; r < first + count; ++r) {
    hipLaunchKernelGGL(k_finish_row, dim3(1), dim3(256), 0, ctx->stream, ctx->d_partials, r, nblocks, ctx->d_scalars);
    ADMM_HIP_CHECK(hipGetLastError());
  }
  return ADMM_OK;
}

static int fetch_scalars(admm_ctx *ctx, int count) {
  ADMM_HIP_CHECK(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
  ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ADMM_OK;
}

static int logistic_fg(admm_csr *csr, const float *x, const float *offset, const float *z, const float *ydual,
                       float rho, double *f_out, float *g_out) {
  admm_ctx *ctx = csr->ctx;
  ADMM_REQUIRE(csr->has_t, "admm_logistic_fg: call admm_csr_build_transposed first");
  ADMM_TRY(ensure_rowbuf(csr));
  const float *w[2] = {x, offset};
  int nblk = 0;
  ADMM_TRY(launch_forward(csr, FWD_LBFGS, 0, csr->nrows, w, offset ? 2 : 1, csr->d_rowbuf, &nblk));
  ADMM_TRY(reduce_rows(ctx, 0, 2, nblk));
  ADMM_TRY(launch_transposed(csr, false, 0, csr->npanels));
  const int grid = grid_for(csr->dim, 256, ctx->cu_count * 8);
  {
    KernelTimer t(ctx, ADMM_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_fg_finish, dim3(grid), dim3(256), 0, ctx->stream, csr->d_g1, x, z, ydual, rho,
                       ctx->d_scalars, csr->has_bias ? 1 : 0, g_out, csr->dim, ctx->d_partials);
    ADMM_HIP_CHECK(hipGetLastError());
  }
  ADMM_TRY(reduce_rows(ctx, 2, 1, grid));
  if (f_out) {
    ADMM_TRY(fetch_scalars(ctx, 3));
    *f_out = ctx->h_scalars[0] + ctx->h_scalars[2];
  }
  return ADMM_OK;
}

// ---------------------------------------------------------------------------------------------
// mini-batch FTRL
// ---------------------------------------------------------------------------------------------

__device__ inline float ftrl_weight(float z, float n, float c, float alpha, float beta, float l1, float l2) {
  // src/ftrl.cc:61-66 (threshold on l1, shrink by the literal 1)
  const float sign = z < 0.0f ? -1.0f : 1.0f;
  if (sign * z <= l1) return 0.0f;
  const float num = sign * 1 - z + l2 * c;
  const double den = ((double)beta + sqrt((double)n)) / (double)alpha + (double)l2;
  return (float)((double)num / den);
}

__global__ void __launch_bounds__(256) k_ftrl_wcur(const float *__restrict__ z, const float *__restrict__ n,
                                                   const float *__restrict__ c, float alpha, float beta, float l1,
                                                   float l2, float *__restrict__ wcur, uint64_t dim) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < dim; i += stride)
    wcur[i] = ftrl_weight(z[i], n[i], c[i], alpha, beta, l1, l2);
}

// per coordinate touched in the batch: w <- wcur (lazy weight), n' = n + G2,
// z += G1 - wcur (sqrt(n') - sqrt(n)) / alpha ; accumulators and bitmap are cleared
__global__ void __launch_bounds__(256) k_ftrl_apply(float *__restrict__ g1, float *__restrict__ g2,
                                                    uint32_t *__restrict__ mask, const float *__restrict__ wcur,
                                                    float *__restrict__ w, float *__restrict__ z,
                                                    float *__restrict__ n, float alpha,
                                                    const double *__restrict__ scalars, int bias_rows,
                                                    uint64_t dim) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < dim; i += stride) {
    bool touched = (mask[i >> 5] >> (i & 31)) & 1u;
    double G1 = g1[i], G2 = g2[i];
    if (bias_rows && i == 0) {
      touched = true;
      G1 += scalars[0];
      G2 += scalars[1];
    }
    if (touched) {
      const float wc = wcur[i];
      const float nold = n[i];
      const float nnew = (float)((double)nold + G2);
      const double sigma = (sqrt((double)nnew) - sqrt((double)nold)) / (double)alpha;
      w[i] = wc;
      z[i] = (float)((double)z[i] + (G1 - sigma * (double)wc));
      n[i] = nnew;
      g1[i] = 0.0f;
      g2[i] = 0.0f;
    }
  }
}

__global__ void __launch_bounds__(256) k_clear_mask(uint32_t *mask, uint64_t words) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) mask[i] = 0u;
}

// the trailing Predict of FtrlSolver::Run on the last row (src/ftrl.cc:140-141): refresh the
// lazy weights of its coordinates
__global__ void __launch_bounds__(64) k_ftrl_last_row(const uint64_t *__restrict__ rowptr,
                                                      const uint32_t *__restrict__ col, uint64_t last,
                                                      const float *__restrict__ z, const float *__restrict__ n,
                                                      const float *__restrict__ c, float alpha, float beta,
                                                      float l1, float l2, float *__restrict__ w) {
  const uint64_t b = rowptr[last], e = rowptr[last + 1];
  for (uint64_t k = b + threadIdx.x; k < e; k += 64) {
    const uint32_t j = col[k];
    w[j] = ftrl_weight(z[j], n[j], c[j], alpha, beta, l1, l2);
  }
}

int ftrl_pass_batched(admm_csr *csr, float *w, float *z, float *n, const float *offset, const float *reg_offset,
                      float alpha, float beta, float l1, float l2, uint64_t batch) {
  admm_ctx *ctx = csr->ctx;
  ADMM_REQUIRE(csr->has_t, "admm_ftrl_pass: batch > 1 needs admm_csr_build_transposed");
  ADMM_REQUIRE(csr->panel_rows == batch || (csr->npanels <= 1 && batch >= csr->nrows),
               "admm_ftrl_pass: batch (%llu) must equal the transposed copy's panel_rows (%llu)",
               (unsigned long long)batch, (unsigned long long)csr->panel_rows);
  ADMM_TRY(ensure_rowbuf(csr));
  // the per-coordinate weight snapshot lives in the second half of the L-BFGS workspace slot
  if (ctx->lbfgs_ws_floats < csr->dim) {
    if (ctx->lbfgs_ws) {
      ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      ADMM_HIP_CHECK(hipFree(ctx->lbfgs_ws));
      ctx->lbfgs_ws = nullptr;
      ctx->lbfgs_ws_floats = 0;
    }
    ADMM_HIP_CHECK(hipMalloc(&ctx->lbfgs_ws, sizeof(float) * csr->dim));
    ctx->lbfgs_ws_floats = csr->dim;
  }
  float *wcur = ctx->lbfgs_ws;
  const int egrid = grid_for(csr->dim, 256, ctx->cu_count * 8);
  for (uint64_t p = 0; p < csr->npanels; ++p) {
    const uint64_t r0 = p * csr->panel_rows, r1 = std::min(csr->nrows, r0 + csr->panel_rows);
    {
      KernelTimer t(ctx, ADMM_K_ELEMENTWISE);
      hipLaunchKernelGGL(k_ftrl_wcur, dim3(egrid), dim3(256), 0, ctx->stream, z, n, reg_offset, alpha, beta, l1,
                         l2, wcur, csr->dim);
      ADMM_HIP_CHECK(hipGetLastError());
    }
    const float *wv[2] = {wcur, offset};
    int nblk = 0;
    ADMM_TRY(launch_forward(csr, FWD_FTRL, r0, r1, wv, 2, csr->d_rowbuf, &nblk));
    ADMM_TRY(reduce_rows(ctx, 0, 2, nblk));
    ADMM_TRY(launch_transposed(csr, true, p, p + 1));
    {
      KernelTimer t(ctx, ADMM_K_ELEMENTWISE);
      hipLaunchKernelGGL(k_ftrl_apply, dim3(egrid), dim3(256), 0, ctx->stream, csr->d_g1, csr->d_g2, csr->d_mask,
                         wcur, w, z, n, alpha, ctx->d_scalars, csr->has_bias ? 1 : 0, csr->dim);
      ADMM_HIP_CHECK(hipGetLastError());
      const uint64_t words = ((uint64_t)csr->nblocks * csr->col_block) / 32 + 1;
      hipLaunchKernelGGL(k_clear_mask, dim3(grid_for(words, 256, ctx->cu_count * 8)), dim3(256), 0, ctx->stream,
                         csr->d_mask, words);
      ADMM_HIP_CHECK(hipGetLastError());
    }
  }
  if (csr->nrows) {
    hipLaunchKernelGGL(k_ftrl_last_row, dim3(1), dim3(64), 0, ctx->stream, csr->d_rowptr, csr->d_col,
                       csr->nrows - 1, z, n, reg_offset, alpha, beta, l1, l2, w);
    ADMM_HIP_CHECK(hipGetLastError());
  }
  return ADMM_OK;
}

}  // namespace admm_impl

// =============================================================================================
// C ABI
// =============================================================================================

extern "C" int admm_predict(admm_csr *csr, const float *const *groups, int ngroups, float *p_out) {
  ADMM_REQUIRE(csr && groups && p_out, "admm_predict: NULL argument");
  int nblk = 0;
  ADMM_TRY(launch_forward(csr, FWD_EVAL, 0, csr->nrows, groups, ngroups, p_out, &nblk));
  return ADMM_OK;
}

namespace admm_impl {
int auc_device(admm_csr *csr, const float *scores, double *auc_out);  // auc.hip
}

extern "C" int admm_metrics(admm_csr *csr, const float *const *groups, int ngroups, double *logloss_mean,
                            double *auc, uint64_t *nrows) {
  ADMM_REQUIRE(csr && groups, "admm_metrics: NULL argument");
  admm_ctx *ctx = csr->ctx;
  if (nrows) *nrows = csr->nrows;
  if (csr->nrows == 0) {
    if (logloss_mean) *logloss_mean = NAN;
    if (auc) *auc = NAN;
    return ADMM_OK;
  }
  ADMM_TRY(ensure_rowbuf(csr));
  int nblk = 0;
  ADMM_TRY(launch_forward(csr, FWD_EVAL, 0, csr->nrows, groups, ngroups, csr->d_rowbuf, &nblk));
  if (logloss_mean) {
    ADMM_TRY(reduce_rows(ctx, 0, 1, nblk));
    ADMM_TRY(fetch_scalars(ctx, 1));
    *logloss_mean = ctx->h_scalars[0] / (double)csr->nrows;
  }
  if (auc) ADMM_TRY(auc_device(csr, csr->d_rowbuf, auc));
  return ADMM_OK;
}

extern "C" int admm_logistic_fg(admm_csr *csr, const float *x, const float *offset, const float *z,
                                const float *ydual, float rho, double *f_out, float *g_out) {
  ADMM_REQUIRE(csr && x && g_out, "admm_logistic_fg: NULL argument");
  return logistic_fg(csr, x, offset, z, ydual, rho, f_out, g_out);
}

extern "C" int admm_lbfgs_solve(admm_csr *csr, float *x, const float *offset, const float *z, const float *ydual,
                                float rho, const admm_lbfgs_params *params, admm_lbfgs_result *result) {
  ADMM_REQUIRE(csr && x, "admm_lbfgs_solve: NULL argument");
  admm_ctx *ctx = csr->ctx;
  admm_lbfgs_params P = {6, 10, 20, 1e-5f, 1e-4f, 0.9f};
  if (params) P = *params;
  ADMM_REQUIRE(P.m >= 1 && P.m <= 32 && P.max_iter >= 0 && P.max_linesearch >= 1, "admm_lbfgs_solve: bad params");
  const uint64_t dim = csr->dim;
  const uint64_t need = (uint64_t)(4 + 2 * P.m) * dim;
  if (ctx->lbfgs_ws_floats < need) {
    ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->lbfgs_ws) ADMM_HIP_CHECK(hipFree(ctx->lbfgs_ws));
    ctx->lbfgs_ws = nullptr;
    ctx->lbfgs_ws_floats = 0;
    ADMM_HIP_CHECK(hipMalloc(&ctx->lbfgs_ws, sizeof(float) * need));
    ctx->lbfgs_ws_floats = need;
  }
  float *g = ctx->lbfgs_ws, *xp = g + dim, *gp = xp + dim, *d = gp + dim;
  float *S = d + dim, *Y = S + (uint64_t)P.m * dim;
  std::vector<double> ys_hist(P.m, 0.0), alpha(P.m, 0.0);
  int head = 0, count = 0;  // ring: newest at (head-1) mod m

  double f = 0, xx = 0, gg = 0;
  int evals = 0, iters = 0;
  ADMM_TRY(logistic_fg(csr, x, offset, z, ydual, rho, &f, g));
  evals = 1;
  ADMM_TRY(vec_dot(ctx, x, x, dim, &xx));
  ADMM_TRY(vec_dot(ctx, g, g, dim, &gg));
  auto done = [&]() {
    if (result) {
      result->f = f;
      result->evals = evals;
      result->iters = iters;
    }
    return ADMM_OK;
  };
  if (std::sqrt(gg) / std::max(1.0, std::sqrt(xx)) <= P.epsilon) return done();
  ADMM_TRY(vec_lincomb(ctx, d, g, -2.0f, g, dim));  // d = g - 2g = -g
  double step = 1.0 / std::sqrt(gg);
  for (int it = 0; it < P.max_iter; ++it) {
    ADMM_TRY(admm_vec_copy(ctx, xp, x, dim));
    ADMM_TRY(admm_vec_copy(ctx, gp, g, dim));
    const double fp = f;
    double dginit = 0;
    ADMM_TRY(vec_dot(ctx, g, d, dim, &dginit));
    if (dginit > 0) break;
    const double dgtest = P.ftol * dginit;
    int ls = 0;
    bool ok = false;
    for (;;) {
      ADMM_TRY(vec_lincomb(ctx, x, xp, (float)step, d, dim));
      ADMM_TRY(logistic_fg(csr, x, offset, z, ydual, rho, &f, g));
      ++evals;
      ++ls;
      double width;
      if (f > fp + step * dgtest) {
        width = 0.5;
      } else {
        double dg = 0;
        ADMM_TRY(vec_dot(ctx, g, d, dim, &dg));
        if (dg < P.wolfe * dginit) {
          width = 2.1;
        } else {
          ok = true;
          break;
        }
      }
      if (ls >= P.max_linesearch) break;
      step *= width;
    }
    if (!ok) {
      if (f > fp) {
        ADMM_TRY(admm_vec_copy(ctx, x, xp, dim));
        ADMM_TRY(admm_vec_copy(ctx, g, gp, dim));
        f = fp;
      }
      break;
    }
    ++iters;
    ADMM_TRY(vec_dot(ctx, x, x, dim, &xx));
    ADMM_TRY(vec_dot(ctx, g, g, dim, &gg));
    if (std::sqrt(gg) / std::max(1.0, std::sqrt(xx)) <= P.epsilon) break;
    float *s_new = S + (uint64_t)head * dim, *y_new = Y + (uint64_t)head * dim;
    ADMM_TRY(vec_sub(ctx, s_new, x, xp, dim));
    ADMM_TRY(vec_sub(ctx, y_new, g, gp, dim));
    double ys = 0, yy = 0;
    ADMM_TRY(vec_dot(ctx, y_new, s_new, dim, &ys));
    ADMM_TRY(vec_dot(ctx, y_new, y_new, dim, &yy));
    if (ys > 1e-30) {
      ys_hist[head] = ys;
      head = (head + 1) % P.m;
      if (count < P.m) ++count;
    }
    ADMM_TRY(vec_lincomb(ctx, d, g, -2.0f, g, dim));  // d = -g
    for (int i = 0; i < count; ++i) {  // newest -> oldest
      const int idx = (head - 1 - i + 2 * P.m) % P.m;
      double sd = 0;
      ADMM_TRY(vec_dot(ctx, S + (uint64_t)idx * dim, d, dim, &sd));
      alpha[i] = sd / ys_hist[idx];
      ADMM_TRY(vec_axpy(ctx, d, (float)(-alpha[i]), Y + (uint64_t)idx * dim, dim));
    }
    if (ys > 1e-30) ADMM_TRY(vec_scale(ctx, d, (float)(ys / yy), dim));
    for (int i = count - 1; i >= 0; --i) {  // oldest -> newest
      const int idx = (head - 1 - i + 2 * P.m) % P.m;
      double yd = 0;
      ADMM_TRY(vec_dot(ctx, Y + (uint64_t)idx * dim, d, dim, &yd));
      const double beta = yd / ys_hist[idx];
      ADMM_TRY(vec_axpy(ctx, d, (float)(alpha[i] - beta), S + (uint64_t)idx * dim, dim));
    }
    step = 1.0;
  }
  return done();
}

namespace admm_impl {
int ftrl_pass_sequential(admm_csr *csr, float *w, float *z, float *n, const float *offset, const float *reg_offset,
                         float alpha, float beta, float l1, float l2);  // ftrl.hip
}

extern "C" int admm_ftrl_pass(admm_csr *csr, float *w, float *z, float *n, const float *offset,
                              const float *reg_offset, float alpha, float beta, float l1, float l2,
                              uint64_t batch) {
  ADMM_REQUIRE(csr && w && z && n && offset && reg_offset, "admm_ftrl_pass: NULL argument");
  ADMM_REQUIRE(alpha > 0, "admm_ftrl_pass: alpha must be > 0");
  if (batch <= 1) return ftrl_pass_sequential(csr, w, z, n, offset, reg_offset, alpha, beta, l1, l2);
  return ftrl_pass_batched(csr, w, z, n, offset, reg_offset, alpha, beta, l1, l2, batch);
}
