// Objective / gradient assembly on top of the two sweeps, the L-BFGS local solver, prediction and metrics entry points.
#include "internal.h"
#include "sweep_common.h"

#include <algorithm>
#include <cmath>

using namespace admm_impl;

namespace admm_impl {

// ---------------------------------------------------------------------------------------------
// f, g assembly
// ---------------------------------------------------------------------------------------------

// g = acc + y + rho (x - z), acc reset to 0 (acc: the column's word of `accum`, plus its word of `gs` when its
// stored id has one); column 0 also receives the bias-column sum the
// forward sweep produced (scalars[1]); partials row 2 <- y.(x-z) + rho/2 |x-z|^2.
// gl_out (optional) keeps the loss part `acc` of the gradient as a float vector; with gl_in the
// kernel takes that part from such a vector instead of the accumulators (same arithmetic after
// it, so the result is bit for bit the one of a fresh evaluation at the same x).  With dot_with, partials
// row 3 <- g . dot_with (same order of additions as k_dot_row on the finished g).
__global__ void __launch_bounds__(256) k_fg_finish(long long *__restrict__ accum, double inv_scale,
                                                   const float *__restrict__ x,
                                                   const float *__restrict__ z, const float *__restrict__ y,
                                                   float rho, const double *__restrict__ scalars, int add_bias,
                                                   float *__restrict__ g, uint64_t n, double *__restrict__ partials,
                                                   float *__restrict__ gl_out, const float *__restrict__ gl_in,
                                                   const float *__restrict__ dot_with,
                                                   const uint32_t *__restrict__ inv, long long *__restrict__ gs) {
  __shared__ double sh[4], sh2[4];
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  double reg = 0, dot = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float a;
    if (gl_in) {
      a = gl_in[i];
    } else {
      long long q = accum[i];
      accum[i] = 0;
      if (gs) {  // the sums kept by stored id (inv: caller's id -> stored id, nullptr: identity)
        const uint32_t sid = inv ? inv[i] : (uint32_t)i;
        if (sid < kFpHot) {
          q += gs[sid];
          gs[sid] = 0;
        }
      }
      double ad = (double)q * inv_scale;
      if (add_bias && i == 0) ad += scalars[1];
      a = (float)ad;
      if (gl_out) gl_out[i] = a;
    }
    const float d = x[i] - (z ? z[i] : 0.0f);
    const float yd = y ? y[i] : 0.0f;
    const float gi = a + yd + rho * d;
    g[i] = gi;
    reg += (double)yd * (double)d + 0.5 * (double)rho * (double)d * (double)d;
    if (dot_with) dot += (double)gi * (double)dot_with[i];
  }
  for (int off = 32; off > 0; off >>= 1) {
    reg += __shfl_down(reg, off, 64);
    dot += __shfl_down(dot, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sh[threadIdx.x >> 6] = reg;
    sh2[threadIdx.x >> 6] = dot;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * kMaxPartialBlocks + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
    if (dot_with) partials[3 * kMaxPartialBlocks + blockIdx.x] = (sh2[0] + sh2[1]) + (sh2[2] + sh2[3]);
  }
}

static int reduce_rows2(admm_ctx *ctx, int first, int count, int split, int nblocks_a, int nblocks_b) {
  return reduce_rows_queue(ctx, first, count, split, nblocks_a, nblocks_b);
}

static int reduce_rows(admm_ctx *ctx, int first, int count, int nblocks) {
  return reduce_rows_queue(ctx, first, count, first + count, nblocks, nblocks);
}

static int fetch_scalars(admm_ctx *ctx, int) { return wait_scalars(ctx); }

// dot products without a host round trip each: `count` (at most 4) pairs go to partial rows 3.. and
// d_scalars[3..]; the caller fetches once
struct DotReq {
  const float *a, *b;
};

__global__ void __launch_bounds__(256) k_dot_row(const float *__restrict__ a, const float *__restrict__ b, uint64_t n,
                                                 double *__restrict__ partials) {
  __shared__ double sh[4];
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  double acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc += (double)a[i] * (double)b[i];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

static int dots_async(admm_ctx *ctx, const DotReq *req, int count, uint64_t n) {
  const int grid = grid_for(n, 256, ctx->cu_count * 8);
  for (int i = 0; i < count; ++i) {
    hipLaunchKernelGGL(k_dot_row, dim3(grid), dim3(256), 0, ctx->stream, req[i].a, req[i].b, n,
                       ctx->d_partials + (size_t)(3 + i) * kMaxPartialBlocks);
    ADMM_HIP_CHECK(hipGetLastError());
  }
  return reduce_rows(ctx, 3, count, grid);
}

static int dots(admm_ctx *ctx, const DotReq *req, int count, uint64_t n, double *out) {
  ADMM_TRY(dots_async(ctx, req, count, n));
  ADMM_TRY(fetch_scalars(ctx, 3 + count));
  for (int i = 0; i < count; ++i) out[i] = ctx->h_scalars[3 + i];
  return ADMM_OK;
}

// One step of the two-loop recursion with its coefficient computed on the device.  The dot product
// it divides arrives as the per-workgroup partials of the launch before (num_parts, nparts of them);
// every workgroup adds them up in k_finish_row's order.  mode 0: alpha = num / den is stored and
// y -= alpha x; mode 1: y += (*alpha - num / den) x.  Then y *= post, and when `next` is given the
// partials of y . next (the product the following step divides) go to next_parts, in k_dot_row's
// order.  Same double arithmetic and float rounding as separate launches (and as on the host).
__global__ void __launch_bounds__(256) k_axpy_coef(float *__restrict__ y, const float *__restrict__ x, uint64_t n,
                                                   const double *__restrict__ num_parts, int nparts, double den,
                                                   double *__restrict__ alpha, int mode, float post,
                                                   const float *__restrict__ next, double *__restrict__ next_parts) {
  __shared__ double sh[256];
  double num = 0;
  for (int i = threadIdx.x; i < nparts; i += 256) num += num_parts[i];
  sh[threadIdx.x] = num;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  const double q = sh[0] / den;
  __syncthreads();
  const float a = mode == 0 ? (float)(-q) : (float)(*alpha - q);
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  double acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float v = y[i];
    v += a * x[i];
    v *= post;
    y[i] = v;
    if (next) acc += (double)v * (double)next[i];
  }
  if (next) {
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) next_parts[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  }
  if (mode == 0 && blockIdx.x == 0 && threadIdx.x == 0) *alpha = q;
}

// two vector copies in one launch (b_dst may be nullptr)
__global__ void __launch_bounds__(256) k_copy2(float *__restrict__ a_dst, const float *__restrict__ a_src,
                                               float *__restrict__ b_dst, const float *__restrict__ b_src, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    a_dst[i] = a_src[i];
    if (b_dst) b_dst[i] = b_src[i];
  }
}

static int copy2(admm_ctx *ctx, float *a_dst, const float *a_src, float *b_dst, const float *b_src, uint64_t n) {
  hipLaunchKernelGGL(k_copy2, dim3(grid_for(n, 256, ctx->cu_count * 8)), dim3(256), 0, ctx->stream, a_dst, a_src, b_dst, b_src, n);
  ADMM_HIP_CHECK(hipGetLastError());
  return ADMM_OK;
}

// The newest correction pair and the four products the solver wants of it, in one pass:
// s = x - xp, y = g - gp; partial rows 3..6 <- x.x, g.g, y.s, y.y (k_dot_row's order of additions).
__global__ void __launch_bounds__(256) k_pair(const float *__restrict__ x, const float *__restrict__ xp,
                                              const float *__restrict__ g, const float *__restrict__ gp,
                                              float *__restrict__ s, float *__restrict__ y, uint64_t n,
                                              double *__restrict__ partials) {
  __shared__ double sh[4][4];
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  double acc[4] = {0, 0, 0, 0};
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float xv = x[i], gv = g[i];
    const float sv = xv + -1.0f * xp[i], yv = gv + -1.0f * gp[i];
    s[i] = sv;
    y[i] = yv;
    acc[0] += (double)xv * (double)xv;
    acc[1] += (double)gv * (double)gv;
    acc[2] += (double)yv * (double)sv;
    acc[3] += (double)yv * (double)yv;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_down(acc[k], off, 64);
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int k = threadIdx.x;
    partials[(size_t)(3 + k) * kMaxPartialBlocks + blockIdx.x] = (sh[k][0] + sh[k][1]) + (sh[k][2] + sh[k][3]);
  }
}

// number of positions where two vectors (and, when given, two more) differ bit for bit
__global__ void __launch_bounds__(256) k_count_diff(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b,
                                                    const uint32_t *__restrict__ c, const uint32_t *__restrict__ d,
                                                    uint64_t n, double *__restrict__ partials) {
  __shared__ double sh[4];
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  double diff = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    diff += a[i] != b[i] ? 1.0 : 0.0;
    if (c) diff += c[i] != d[i] ? 1.0 : 0.0;
  }
  for (int off = 32; off > 0; off >>= 1) diff += __shfl_down(diff, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = diff;
  __syncthreads();
  if (threadIdx.x == 0) partials[0 * kMaxPartialBlocks + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// One trial point of the line search evaluated from the margins: with m0 = X(x+offset) and
// md = X d per row, the loss part of the objective at x + s d is the sum over rows of
// loss(m0 + s md), and its slope along d the sum of (sigma(m0 + s md) - t) md.  Same per-row
// formulas as the forward sweep's L-BFGS epilogue.
__global__ void __launch_bounds__(256) k_row_trial(const float *__restrict__ m0, const float *__restrict__ md,
                                                   const float *__restrict__ label, float step, uint64_t n,
                                                   double *__restrict__ partials) {
  __shared__ double sh[2][4];
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  double loss = 0, slope = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float d = md[i];
    const float m = m0[i] + step * d;
    const float t = label[i] > 0 ? 1.0f : 0.0f;
    loss += (double)(log1pf(expf(-fabsf(m))) + (t > 0 ? fmaxf(-m, 0.0f) : fmaxf(m, 0.0f)));
    slope += (double)((1.0f / (1.0f + expf(-m)) - t) * d);
  }
  for (int off = 32; off > 0; off >>= 1) {
    loss += __shfl_down(loss, off, 64);
    slope += __shfl_down(slope, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = loss;
    sh[1][threadIdx.x >> 6] = slope;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[0 * kMaxPartialBlocks + blockIdx.x] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    partials[1 * kMaxPartialBlocks + blockIdx.x] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
}

// The accepted point of a line search from the margins: m0 <- m0 + s md, and what the forward
// sweep's L-BFGS epilogue would have written for those margins (residual per row, loss and
// residual sums).  Same per-row arithmetic as k_row_trial, so the loss equals the trial's.
__global__ void __launch_bounds__(256) k_row_accept(float *__restrict__ m0, const float *__restrict__ md,
                                                    const float *__restrict__ label, float step, uint64_t n,
                                                    float *__restrict__ res_out, double *__restrict__ partials) {
  __shared__ double sh[2][4];
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  double loss = 0, rsum = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float m = m0[i] + step * md[i];
    m0[i] = m;
    const float t = label[i] > 0 ? 1.0f : 0.0f;
    const float res = 1.0f / (1.0f + expf(-m)) - t;
    res_out[i] = res;
    loss += (double)(log1pf(expf(-fabsf(m))) + (t > 0 ? fmaxf(-m, 0.0f) : fmaxf(m, 0.0f)));
    rsum += (double)res;
  }
  for (int off = 32; off > 0; off >>= 1) {
    loss += __shfl_down(loss, off, 64);
    rsum += __shfl_down(rsum, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = loss;
    sh[1][threadIdx.x >> 6] = rsum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[0 * kMaxPartialBlocks + blockIdx.x] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    partials[1 * kMaxPartialBlocks + blockIdx.x] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
}

// X^T r restricted to the kMaskCols columns that a compact set keeps as per-row bit masks, from the
// residuals in memory (the forward sweep adds these sums itself when it computes the residuals).
// One wave takes 64 rows at a time, a row per lane.  Columns 0..63 are dense: the wave walks its
// rows, broadcasts a row's mask and fixed-point residual, and lane k adds the residual when bit k
// is set.  Columns 64.. are sparse: each lane walks the set bits of its own row and adds to
// per-column LDS accumulators.  Same 64-bit fixed point as everywhere else: order-independent.
__global__ void __launch_bounds__(256) k_mask_colsums(const unsigned long long *__restrict__ hmask,
                                                      const unsigned long long *__restrict__ hwarm,
                                                      const float *__restrict__ res, uint64_t nrows, float scale,
                                                      const uint32_t *__restrict__ perm,
                                                      unsigned long long *__restrict__ g1) {
  __shared__ unsigned long long acc[kMaskCols];
  const int tid = threadIdx.x, lane = tid & 63;
  acc[tid] = 0ull;
  static_assert(kMaskCols == 256, "one accumulator per thread");
  __syncthreads();
  unsigned long long h = 0;  // lane k: column k over the rows this wave has seen
  const uint64_t nwaves = (uint64_t)gridDim.x * 4, wave = (uint64_t)blockIdx.x * 4 + (tid >> 6);
  auto bcast = [](unsigned long long v, int src) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
  };
  for (uint64_t base = wave * 64; base < nrows; base += nwaves * 64) {
    const uint64_t r = base + lane;
    const bool on = r < nrows;
    const unsigned long long m0 = on ? hmask[r] : 0ull;
    const unsigned long long rv = on ? (unsigned long long)(long long)(res[r] * scale) : 0ull;
    unsigned long long mw[kWarmWords];
#pragma unroll
    for (int w = 0; w < kWarmWords; ++w) mw[w] = on ? hwarm[r * kWarmWords + w] : 0ull;
    // dense word: four rows per trip, independent partial sums
    unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0;
#pragma unroll 4
    for (int k = 0; k < 64; k += 4) {
      p0 += __builtin_amdgcn_inverse_ballot_w64(bcast(m0, k)) ? bcast(rv, k) : 0ull;
      p1 += __builtin_amdgcn_inverse_ballot_w64(bcast(m0, k + 1)) ? bcast(rv, k + 1) : 0ull;
      p2 += __builtin_amdgcn_inverse_ballot_w64(bcast(m0, k + 2)) ? bcast(rv, k + 2) : 0ull;
      p3 += __builtin_amdgcn_inverse_ballot_w64(bcast(m0, k + 3)) ? bcast(rv, k + 3) : 0ull;
    }
    h += (p0 + p1) + (p2 + p3);
    // sparse words: the three words of a row are walked together
    while (mw[0] | mw[1] | mw[2]) {
#pragma unroll
      for (int w = 0; w < kWarmWords; ++w) {
        if (mw[w]) {
          atomicAdd(&acc[64 * (w + 1) + __builtin_ctzll(mw[w])], rv);
          mw[w] &= mw[w] - 1;
        }
      }
    }
  }
  static_assert(kWarmWords == 3, "the walk above names three words");
  if (h) atomicAdd(&acc[lane], h);
  __syncthreads();
  if (acc[tid]) atomicAdd(g1 + (perm ? perm[tid] : (uint32_t)tid), acc[tid]);
}

static int logistic_fg(admm_csr *csr, const float *x, const float *offset, const float *z, const float *ydual,
                       float rho, double *f_out, float *g_out, double *loss_out = nullptr,
                       float *margins_out = nullptr, float *gl_out = nullptr, const float *dot_with = nullptr,
                       double *dot_out = nullptr) {
  admm_ctx *ctx = csr->ctx;
  ADMM_REQUIRE(csr->has_t, "admm_logistic_fg: call admm_csr_build_transposed first");
  ADMM_TRY(ensure_rowbuf(csr));
  const float *w[2] = {x, offset};
  int nblk = 0;
  ADMM_TRY(launch_forward(csr, FWD_LBFGS, 0, csr->nrows, w, offset ? 2 : 1, csr->d_rowbuf, &nblk, margins_out));
  // the bias column's sum (scalar 1) enters the gradient below unless the mask columns carry it
  const bool bias_scalar = csr->has_bias && !csr->hot_dense;
  if (bias_scalar) ADMM_TRY(reduce_rows(ctx, 0, 2, nblk));
  ADMM_TRY(launch_transposed(csr, false, 0, csr->npanels));
  const int grid = grid_for(csr->dim, 256, ctx->cu_count * 8);
  {
    KernelTimer t(ctx, ADMM_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_fg_finish, dim3(grid), dim3(256), 0, ctx->stream, csr->d_g1, std::ldexp(1.0, -csr->fx_shift), x, z,
                       ydual, rho, ctx->d_scalars, bias_scalar ? 1 : 0, g_out, csr->dim, ctx->d_partials,
                       gl_out, (const float *)nullptr, dot_with, csr->d_inv, csr->d_gs);  // g . dot_with rides on the same fetch
    ADMM_HIP_CHECK(hipGetLastError());
  }
  // one launch sums the forward sweep's two rows of partials and this kernel's one or two
  if (bias_scalar)
    ADMM_TRY(reduce_rows(ctx, 2, dot_with ? 2 : 1, grid));
  else
    ADMM_TRY(reduce_rows2(ctx, 0, dot_with ? 4 : 3, 2, nblk, grid));
  if (f_out) {
    ADMM_TRY(fetch_scalars(ctx, dot_with ? 4 : 3));
    *f_out = ctx->h_scalars[0] + ctx->h_scalars[2];
    if (loss_out) *loss_out = ctx->h_scalars[0];
    if (dot_with && dot_out) *dot_out = ctx->h_scalars[3];
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
  admm_lbfgs_params P = {6, 10, 20, 1e-5f, 1e-4f, 0.9f, 0};
  if (params) P = *params;
  ADMM_REQUIRE(P.m >= 1 && P.m <= 32 && P.max_iter >= 0 && P.max_linesearch >= 1, "admm_lbfgs_solve: bad params");
  ADMM_REQUIRE(P.linesearch >= 0 && P.linesearch <= 2, "admm_lbfgs_solve: linesearch must be 0, 1 or 2, got %d", P.linesearch);
  const bool from_margins = P.linesearch != 1;
  const bool cheap_accept = P.linesearch == 2;
  const uint64_t dim = csr->dim;
  const uint64_t need = (uint64_t)(6 + 2 * P.m) * dim;
  if (ctx->lbfgs_ws_floats < need) {
    ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->lbfgs_ws) ADMM_HIP_CHECK(hipFree(ctx->lbfgs_ws));
    ctx->lbfgs_ws = nullptr;
    ctx->lbfgs_ws_floats = 0;
    ADMM_HIP_CHECK(hipMalloc(&ctx->lbfgs_ws, sizeof(float) * need));
    ctx->lbfgs_ws_floats = need;
  }
  if (from_margins && csr->nrows) {
    if (!csr->d_margin) ADMM_HIP_CHECK(hipMalloc(&csr->d_margin, sizeof(float) * csr->nrows));
    if (!csr->d_dmargin) ADMM_HIP_CHECK(hipMalloc(&csr->d_dmargin, sizeof(float) * csr->nrows));
  }
  float *g = ctx->lbfgs_ws, *xp = g + dim, *gp = xp + dim, *d = gp + dim;
  // rings of m+1 slots: the pairs in use are the `count` slots before `head`, and slot `head` takes
  // the newest pair before it is known to be usable (so it never overwrites one still in use)
  const int slots = P.m + 1;
  float *S = d + dim, *Y = S + (uint64_t)slots * dim;
  std::vector<double> ys_hist(slots, 0.0);
  int head = 0, count = 0;  // newest pair in use at (head-1) mod slots

  for (float **p : {&csr->d_cache_x, &csr->d_cache_off, &csr->d_cache_gl})
    if (!*p) ADMM_HIP_CHECK(hipMalloc(p, sizeof(float) * dim));
  double f = 0, loss = 0, xx = 0, gg = 0;
  int evals = 0, iters = 0, fwd_sweeps = 0, tr_sweeps = 0;
  bool gl_at_x = true;  // d_cache_gl is the loss gradient at the current x
  // a full evaluation: objective, gradient, and (for the line search from the margins) the margins
  double dg_eval = 0;  // g . d at the point of the last full evaluation that asked for it
  auto full_eval = [&](bool new_point, const float *dir = nullptr) -> int {
    ADMM_TRY(logistic_fg(csr, x, offset, z, ydual, rho, &f, g, &loss, from_margins ? csr->d_margin : nullptr,
                         csr->d_cache_gl, dir, &dg_eval));
    gl_at_x = true;
    if (from_margins) csr->margin_age = 0;
    if (new_point) ++evals;
    ++fwd_sweeps;
    ++tr_sweeps;
    return ADMM_OK;
  };
  // loss part of the objective and its slope along d at step s, from the margins
  auto row_trial = [&](double s, double *loss_s, double *slope_s) -> int {
    const int grid = grid_for(csr->nrows, 256, ctx->cu_count * 8);
    {
      KernelTimer t(ctx, ADMM_K_ELEMENTWISE);
      hipLaunchKernelGGL(k_row_trial, dim3(grid), dim3(256), 0, ctx->stream, csr->d_margin, csr->d_dmargin, csr->d_label,
                         (float)s, csr->nrows, ctx->d_partials);
      ADMM_HIP_CHECK(hipGetLastError());
    }
    ADMM_TRY(reduce_rows(ctx, 0, 2, grid));
    ADMM_TRY(fetch_scalars(ctx, 2));
    *loss_s = ctx->h_scalars[0];
    *slope_s = ctx->h_scalars[1];
    return ADMM_OK;
  };
  // An ADMM iteration starts its solve where the previous one ended: same x, same offset, only z and
  // the dual have moved, and the sweeps do not depend on those.  If x and offset are bit for bit the
  // ones the last solve on this sample set ended on, its loss and loss gradient are still valid and the
  // first evaluation costs no sweep.
  bool from_cache = false;
  {
    const char *env = getenv("ADMM_FG_CACHE");
    if (csr->cache_valid && !(env && env[0] == '0') && csr->cache_has_off == (offset != nullptr) &&
        (!from_margins || csr->cache_margin)) {
      const int grid = grid_for(dim, 256, ctx->cu_count * 8);
      hipLaunchKernelGGL(k_count_diff, dim3(grid), dim3(256), 0, ctx->stream, reinterpret_cast<const uint32_t *>(x),
                         reinterpret_cast<const uint32_t *>(csr->d_cache_x), reinterpret_cast<const uint32_t *>(offset),
                         reinterpret_cast<const uint32_t *>(csr->d_cache_off), dim, ctx->d_partials);
      ADMM_HIP_CHECK(hipGetLastError());
      ADMM_TRY(reduce_rows(ctx, 0, 1, grid));
      ADMM_TRY(fetch_scalars(ctx, 1));
      from_cache = ctx->h_scalars[0] == 0.0;
    }
  }
  csr->cache_valid = false;  // until this solve has ended
  if (from_cache) {
    const int grid = grid_for(dim, 256, ctx->cu_count * 8);
    hipLaunchKernelGGL(k_fg_finish, dim3(grid), dim3(256), 0, ctx->stream, (long long *)nullptr, 0.0, x, z, ydual, rho,
                       ctx->d_scalars, 0, g, dim, ctx->d_partials, (float *)nullptr, (const float *)csr->d_cache_gl,
                       (const float *)nullptr, (const uint32_t *)nullptr, (long long *)nullptr);
    ADMM_HIP_CHECK(hipGetLastError());
    ADMM_TRY(reduce_rows(ctx, 2, 1, grid));
    ADMM_TRY(fetch_scalars(ctx, 3));
    loss = csr->cache_loss;
    f = loss + ctx->h_scalars[2];
    ++evals;
  } else {
    ADMM_TRY(full_eval(true));
  }
  // The accepted point of a line search from the margins without a forward sweep: the margins
  // there are m0 + s md, so the residuals, the loss and the mask columns' sums come from per-row
  // passes, and only the transposed sweep is left.  The margins pick up one float rounding per
  // such step, so after two of them (and at the last iteration of a solve, whose margins the next
  // solve may start from) the accepted point takes a full evaluation again.  The gradient differs
  // from a full evaluation's in its last bits, which a long ADMM run amplifies (1e-4 relative in the
  // test logloss after five iterations at 100M rows), hence a mode of its own.
  auto accept_from_margins = [&](double s) -> int {
    ADMM_TRY(ensure_rowbuf(csr));
    const int rgrid = grid_for(csr->nrows, 256, ctx->cu_count * 8);
    {
      KernelTimer t(ctx, ADMM_K_ELEMENTWISE);
      hipLaunchKernelGGL(k_row_accept, dim3(rgrid), dim3(256), 0, ctx->stream, csr->d_margin, csr->d_dmargin, csr->d_label,
                         (float)s, csr->nrows, csr->d_rowbuf, ctx->d_partials);
      ADMM_HIP_CHECK(hipGetLastError());
    }
    ADMM_TRY(reduce_rows(ctx, 0, 2, rgrid));
    if (csr->hot_dense) {
      ADMM_TRY(ensure_taccum(csr));
      const int mgrid = (int)std::max<uint64_t>(1, std::min<uint64_t>((csr->nrows + 255) / 256, (uint64_t)ctx->cu_count * 8));
      KernelTimer t(ctx, ADMM_K_ELEMENTWISE);
      hipLaunchKernelGGL(k_mask_colsums, dim3(mgrid), dim3(256), 0, ctx->stream,
                         reinterpret_cast<const unsigned long long *>(csr->d_hmask),
                         reinterpret_cast<const unsigned long long *>(csr->d_hwarm), csr->d_rowbuf,
                         csr->nrows, std::ldexp(1.0f, csr->fx_shift), (const uint32_t *)nullptr,
                         reinterpret_cast<unsigned long long *>(csr->d_gs));  // by stored id, as the forward sweep leaves them
      ADMM_HIP_CHECK(hipGetLastError());
    }
    ADMM_TRY(launch_transposed(csr, false, 0, csr->npanels));
    ++tr_sweeps;
    const int grid = grid_for(dim, 256, ctx->cu_count * 8);
    {
      KernelTimer t(ctx, ADMM_K_ELEMENTWISE);
      hipLaunchKernelGGL(k_fg_finish, dim3(grid), dim3(256), 0, ctx->stream, csr->d_g1, std::ldexp(1.0, -csr->fx_shift), x, z,
                         ydual, rho, ctx->d_scalars, (csr->has_bias && !csr->hot_dense) ? 1 : 0, g, dim, ctx->d_partials,
                         csr->d_cache_gl, (const float *)nullptr, (const float *)nullptr, csr->d_inv, csr->d_gs);
      ADMM_HIP_CHECK(hipGetLastError());
    }
    ADMM_TRY(reduce_rows(ctx, 2, 1, grid));
    ADMM_TRY(fetch_scalars(ctx, 3));
    loss = ctx->h_scalars[0];
    f = loss + ctx->h_scalars[2];
    gl_at_x = true;
    ++csr->margin_age;
    return ADMM_OK;
  };
  if (!ctx->d_lb) ADMM_HIP_CHECK(hipMalloc(&ctx->d_lb, sizeof(double) * 64));
  {
    const DotReq rq[2] = {{x, x}, {g, g}};
    double out[2];
    ADMM_TRY(dots(ctx, rq, 2, dim, out));
    xx = out[0];
    gg = out[1];
  }
  auto done = [&]() -> int {
    // remember where this solve ended
    if (gl_at_x) {
      ADMM_TRY(copy2(ctx, csr->d_cache_x, x, offset ? csr->d_cache_off : nullptr, offset, dim));
      csr->cache_has_off = offset != nullptr;
      csr->cache_loss = loss;
      csr->cache_margin = from_margins;
      csr->cache_valid = true;
    }
    if (result) {
      result->f = f;
      result->evals = evals;
      result->iters = iters;
      result->forward_sweeps = fwd_sweeps;
      result->transposed_sweeps = tr_sweeps;
    }
    return ADMM_OK;
  };
  if (std::sqrt(gg) / std::max(1.0, std::sqrt(xx)) <= P.epsilon) return done();
  ADMM_TRY(vec_lincomb(ctx, d, g, -2.0f, g, dim));  // d = g - 2g = -g
  double step = 1.0 / std::sqrt(gg);
  int dg_row = -1;  // scalar slot holding g . d for the next line search, when the device has it
  for (int it = 0; it < P.max_iter; ++it) {
    ADMM_TRY(copy2(ctx, xp, x, gp, g, dim));
    const double fp = f, lossp = loss;
    double dginit = 0;
    if (dg_row >= 0) {  // the two-loop recursion's last step left g . d on the device
      ADMM_TRY(fetch_scalars(ctx, dg_row + 1));
      dginit = ctx->h_scalars[dg_row];
      dg_row = -1;
    } else {
      ADMM_TRY(vec_dot(ctx, g, d, dim, &dginit));
    }
    if (dginit > 0) break;
    const double dgtest = P.ftol * dginit;
    // Along x + s d the margins are m0 + s X d and the rest of the objective (the dual term and the
    // proximal term) is a quadratic in s: reg(s) = reg(0) + s A + s^2 B with A = g.d - (slope of the
    // loss at 0) and B = rho/2 d.d.  One forward sweep for X d then serves every trial point.
    double lin = 0, quad = 0;
    if (from_margins) {
      const float *wd[1] = {d};
      int nblk = 0;
      ADMM_TRY(launch_forward(csr, FWD_MARGIN, 0, csr->nrows, wd, 1, csr->d_dmargin, &nblk));
      ++fwd_sweeps;
      double loss0 = 0, slope0 = 0, dd = 0;
      ADMM_TRY(row_trial(0.0, &loss0, &slope0));
      ADMM_TRY(vec_dot(ctx, d, d, dim, &dd));
      lin = dginit - slope0;
      quad = 0.5 * (double)rho * dd;
    }
    int ls = 0;
    bool ok = false;
    for (;;) {
      double dg = 0;
      bool have_dg = false;
      if (from_margins) {
        double loss_s = 0, slope_s = 0;
        ADMM_TRY(row_trial(step, &loss_s, &slope_s));
        ++evals;
        f = loss_s + (fp - lossp) + step * lin + step * step * quad;
        dg = slope_s + lin + 2.0 * step * quad;
        have_dg = true;
      } else {
        ADMM_TRY(vec_lincomb(ctx, x, xp, (float)step, d, dim));
        ADMM_TRY(full_eval(true, d));  // g . d comes back with the objective: one host round trip per trial point
        dg = dg_eval;
        have_dg = true;
      }
      ++ls;
      double width;
      if (f > fp + step * dgtest) {
        width = 0.5;
      } else {
        if (!have_dg) ADMM_TRY(vec_dot(ctx, g, d, dim, &dg));
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
    if (from_margins) {
      // x and g still hold the iterate the search started from; move only to a point that is no worse
      if (ok || f <= fp) {
        ADMM_TRY(vec_lincomb(ctx, x, xp, (float)step, d, dim));
        if (cheap_accept && ok && it + 1 < P.max_iter && csr->margin_age < 2)
          ADMM_TRY(accept_from_margins(step));
        else
          ADMM_TRY(full_eval(false));  // exact objective, gradient and margins at the accepted trial point
      } else {
        f = fp;
      }
      if (!ok) break;
    } else if (!ok) {
      if (f > fp) {
        ADMM_TRY(copy2(ctx, x, xp, g, gp, dim));
        f = fp;
        loss = lossp;
        gl_at_x = false;  // the kept loss gradient belongs to the rejected trial point
      }
      break;
    }
    ++iters;
    if (it + 1 >= P.max_iter) break;  // nothing below serves the last iteration
    float *s_new = S + (uint64_t)head * dim, *y_new = Y + (uint64_t)head * dim;
    const int egrid = grid_for(dim, 256, ctx->cu_count * 8);
    double ys = 0, yy = 0;
    {
      hipLaunchKernelGGL(k_pair, dim3(egrid), dim3(256), 0, ctx->stream, x, xp, g, gp, s_new, y_new, dim, ctx->d_partials);
      ADMM_HIP_CHECK(hipGetLastError());
      ADMM_TRY(reduce_rows(ctx, 3, 4, egrid));
      ADMM_TRY(fetch_scalars(ctx, 7));
      xx = ctx->h_scalars[3];
      gg = ctx->h_scalars[4];
      ys = ctx->h_scalars[5];
      yy = ctx->h_scalars[6];
    }
    if (std::sqrt(gg) / std::max(1.0, std::sqrt(xx)) <= P.epsilon) break;
    if (ys > 1e-30) {  // only a pair with positive curvature enters the ring
      ys_hist[head] = ys;
      head = (head + 1) % slots;
      if (count < P.m) ++count;
    }
    // two-loop recursion; its coefficients stay on the device (no host round trip inside), and every
    // step also forms the product the next one divides
    ADMM_TRY(vec_lincomb(ctx, d, g, -2.0f, g, dim));  // d = -g
    if (count > 0) {
      const int total = 2 * count;
      auto slot = [&](int i) { return (uint64_t)((head - 1 - i + 2 * slots) % slots); };  // i = 0: newest
      // step j < count: pair j, newest -> oldest; then oldest -> newest
      auto pair_of = [&](int j) { return j < count ? j : total - 1 - j; };
      auto dot_vec = [&](int j) { return (j < count ? S : Y) + slot(pair_of(j)) * dim; };
      auto part_row = [&](int j) { return ctx->d_partials + (size_t)(3 + (j & 1)) * kMaxPartialBlocks; };
      hipLaunchKernelGGL(k_dot_row, dim3(egrid), dim3(256), 0, ctx->stream, dot_vec(0), d, dim, part_row(0));
      ADMM_HIP_CHECK(hipGetLastError());
      for (int j = 0; j < total; ++j) {
        const int i = pair_of(j);
        const float post = (j == count - 1 && ys > 1e-30) ? (float)(ys / yy) : 1.0f;  // the initial Hessian's scale
        const float *next = j + 1 < total ? dot_vec(j + 1) : g;  // last: g . d for the line search
        hipLaunchKernelGGL(k_axpy_coef, dim3(egrid), dim3(256), 0, ctx->stream, d, (j < count ? Y : S) + slot(i) * dim, dim,
                           part_row(j), egrid, ys_hist[slot(i)], ctx->d_lb + i, j < count ? 0 : 1, post, next,
                           part_row(j + 1));
        ADMM_HIP_CHECK(hipGetLastError());
      }
      dg_row = 3 + (total & 1);
      ADMM_TRY(reduce_rows(ctx, dg_row, 1, egrid));
    }
    step = 1.0;
  }
  return done();
}
