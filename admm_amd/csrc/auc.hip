// Tie-aware rank-sum AUC on the device (Metrics::Sort + Metrics::Auc, src/metrics.cc:14-89):
// radix sort of the scores, average 1-based rank per tie group, sum over positives in double.
#include "internal.h"

#include <rocprim/rocprim.hpp>

using namespace admm_impl;

namespace admm_impl {

// positives are rows whose label truncated to int equals 1 (src/metrics.cc:60,72)
__global__ void __launch_bounds__(256) k_auc_flags(const float *__restrict__ label, uint32_t *__restrict__ flag,
                                                   uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    flag[i] = ((int)label[i] == 1) ? 1u : 0u;
}

// start[i] = i where a tie group begins, else 0; an inclusive max-scan turns it into the group
// start of every position
__global__ void __launch_bounds__(256) k_auc_heads(const float *__restrict__ key, uint32_t *__restrict__ start,
                                                   uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    start[i] = (i > 0 && key[i] != key[i - 1]) ? (uint32_t)i : 0u;
}

__global__ void __launch_bounds__(256) k_auc_sum(const uint32_t *__restrict__ start,
                                                 const uint32_t *__restrict__ flag, uint64_t n,
                                                 double *__restrict__ partials) {
  __shared__ double sh[4];
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  double rank_sum = 0, npos = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    if (!flag[i]) continue;
    const uint32_t s = start[i];
    // first position after i whose group start differs (start[] is non-decreasing)
    uint64_t lo = i + 1, hi = n;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      if (start[mid] > s) hi = mid;
      else lo = mid + 1;
    }
    rank_sum += ((double)s + (double)lo + 1.0) * 0.5;  // mean of ranks s+1 .. lo
    npos += 1.0;
  }
  for (int off = 32; off > 0; off >>= 1) {
    rank_sum += __shfl_down(rank_sum, off, 64);
    npos += __shfl_down(npos, off, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) sh[wave] = rank_sum;
  __syncthreads();
  if (threadIdx.x == 0) partials[0 * kMaxPartialBlocks + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  if (lane == 0) sh[wave] = npos;
  __syncthreads();
  if (threadIdx.x == 0) partials[1 * kMaxPartialBlocks + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

int auc_scores(admm_ctx *ctx, const float *labels, const float *scores, uint64_t n, double *auc_out) {
  ADMM_REQUIRE(n < 0x7FFFFFFFull, "admm_metrics: AUC supports up to 2^31-1 rows");
  // one workspace kept in the context: four n-element arrays and the sort's scratch
  const size_t arr = (sizeof(uint32_t) * n + 255) & ~(size_t)255;
  size_t bytes = 0, bytes2 = 0;
#define AUC_CHECK(expr)                                                        \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess) {                                                    \
      set_error("admm_metrics(auc): %s: %s", #expr, hipGetErrorString(e_));    \
      return e_ == hipErrorOutOfMemory ? ADMM_ERR_NOMEM : ADMM_ERR_HIP;        \
    }                                                                          \
  } while (0)
  AUC_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, scores, (float *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                      (size_t)n, 0, 32, ctx->stream));
  AUC_CHECK(rocprim::inclusive_scan(nullptr, bytes2, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n,
                                    rocprim::maximum<uint32_t>(), ctx->stream));
  if (bytes2 > bytes) bytes = bytes2;
  if (bytes < 16) bytes = 16;
  const size_t need = 4 * arr + bytes;
  if (ctx->auc_ws_bytes < need) {
    AUC_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->auc_ws) AUC_CHECK(hipFree(ctx->auc_ws));
    ctx->auc_ws = nullptr;
    ctx->auc_ws_bytes = 0;
    AUC_CHECK(hipMalloc(&ctx->auc_ws, need));
    ctx->auc_ws_bytes = need;
  }
  char *base = static_cast<char *>(ctx->auc_ws);
  float *keys = reinterpret_cast<float *>(base);
  uint32_t *flag_in = reinterpret_cast<uint32_t *>(base + arr), *flag_out = reinterpret_cast<uint32_t *>(base + 2 * arr),
           *start = reinterpret_cast<uint32_t *>(base + 3 * arr);
  void *tmp = base + 4 * arr;
  const int grid = grid_for(n, 256, ctx->cu_count * 8);
  hipLaunchKernelGGL(k_auc_flags, dim3(grid), dim3(256), 0, ctx->stream, labels, flag_in, n);
  AUC_CHECK(rocprim::radix_sort_pairs(tmp, bytes, scores, keys, flag_in, flag_out, (size_t)n, 0, 32, ctx->stream));
  hipLaunchKernelGGL(k_auc_heads, dim3(grid), dim3(256), 0, ctx->stream, keys, start, n);
  AUC_CHECK(rocprim::inclusive_scan(tmp, bytes, start, start, (size_t)n, rocprim::maximum<uint32_t>(), ctx->stream));
  hipLaunchKernelGGL(k_auc_sum, dim3(grid), dim3(256), 0, ctx->stream, start, flag_out, n, ctx->d_partials);
  AUC_CHECK(hipGetLastError());
  int rc = finish_partials(ctx, 2, grid);
#undef AUC_CHECK
  if (rc != ADMM_OK) return rc;
  const double rank_sum = ctx->h_scalars[0], P = ctx->h_scalars[1], N = (double)n - P;
  // src/metrics.cc:80; with one class missing this divides by zero exactly as the reference does
  *auc_out = rank_sum / (P * N) - (P + 1.0) / (2.0 * N);
  return ADMM_OK;
}

// rank[orig[i]] = (s + e + 1) / 2 in float arithmetic as the reference has it (src/metrics.cc:32-44), for
// sorted position i in the tie group [s, e): the integer s + e is rounded to float, then 1.0f is added; the
// last group is written as (s + (n - 1)) rounded to float plus 2.0f there (src/metrics.cc:37-40), which differs
// from the former once s + n passes 2^24
__global__ void __launch_bounds__(256) k_rank_write(const uint32_t *__restrict__ start, const uint32_t *__restrict__ orig,
                                                    uint64_t n, float *__restrict__ ranks) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const uint32_t s = start[i];
    uint64_t lo = i + 1, hi = n;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      if (start[mid] > s) hi = mid;
      else lo = mid + 1;
    }
    ranks[orig[i]] = lo == n ? ((float)(int)(s + (uint32_t)n - 1u) + 2.0f) / 2 : ((float)(int)(s + (uint32_t)lo) + 1.0f) / 2;
  }
}

__global__ void __launch_bounds__(256) k_iota_u32(uint32_t *v, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) v[i] = (uint32_t)i;
}

// Metrics::Sort (src/metrics.cc:14-46): tied ranks, 1-based, of n scores
int tied_ranks(admm_ctx *ctx, const float *scores, uint64_t n, float *ranks) {
  ADMM_REQUIRE(n < 0x7FFFFFFFull, "admm_tied_ranks: up to 2^31-1 scores");
  if (n == 0) return ADMM_OK;
  const size_t arr = (sizeof(uint32_t) * n + 255) & ~(size_t)255;
  size_t bytes = 0, bytes2 = 0;
  ADMM_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, scores, (float *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                           (size_t)n, 0, 32, ctx->stream));
  ADMM_HIP_CHECK(rocprim::inclusive_scan(nullptr, bytes2, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n,
                                         rocprim::maximum<uint32_t>(), ctx->stream));
  if (bytes2 > bytes) bytes = bytes2;
  if (bytes < 16) bytes = 16;
  const size_t need = 4 * arr + bytes;
  if (ctx->auc_ws_bytes < need) {
    ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->auc_ws) ADMM_HIP_CHECK(hipFree(ctx->auc_ws));
    ctx->auc_ws = nullptr;
    ctx->auc_ws_bytes = 0;
    ADMM_HIP_CHECK(hipMalloc(&ctx->auc_ws, need));
    ctx->auc_ws_bytes = need;
  }
  char *base = static_cast<char *>(ctx->auc_ws);
  float *keys = reinterpret_cast<float *>(base);
  uint32_t *idx_in = reinterpret_cast<uint32_t *>(base + arr), *idx_out = reinterpret_cast<uint32_t *>(base + 2 * arr),
           *start = reinterpret_cast<uint32_t *>(base + 3 * arr);
  void *tmp = base + 4 * arr;
  const int grid = grid_for(n, 256, ctx->cu_count * 8);
  hipLaunchKernelGGL(k_iota_u32, dim3(grid), dim3(256), 0, ctx->stream, idx_in, n);
  ADMM_HIP_CHECK(rocprim::radix_sort_pairs(tmp, bytes, scores, keys, idx_in, idx_out, (size_t)n, 0, 32, ctx->stream));
  hipLaunchKernelGGL(k_auc_heads, dim3(grid), dim3(256), 0, ctx->stream, keys, start, n);
  ADMM_HIP_CHECK(rocprim::inclusive_scan(tmp, bytes, start, start, (size_t)n, rocprim::maximum<uint32_t>(), ctx->stream));
  hipLaunchKernelGGL(k_rank_write, dim3(grid), dim3(256), 0, ctx->stream, start, idx_out, n, ranks);
  ADMM_HIP_CHECK(hipGetLastError());
  return ADMM_OK;
}

int auc_device(admm_csr *csr, const float *scores, double *auc_out) {
  return auc_scores(csr->ctx, csr->d_label, scores, csr->nrows, auc_out);
}

}  // namespace admm_impl

extern "C" int admm_auc_scores(admm_ctx *ctx, const float *scores, const float *labels, uint64_t n, double *auc) {
  ADMM_REQUIRE(ctx && auc && (n == 0 || (scores && labels)), "admm_auc_scores: NULL argument");
  ADMM_HIP_CHECK(hipSetDevice(ctx->device));
  return admm_impl::auc_scores(ctx, labels, scores, n, auc);
}

extern "C" int admm_tied_ranks(admm_ctx *ctx, const float *scores, uint64_t n, float *ranks) {
  ADMM_REQUIRE(ctx && (n == 0 || (scores && ranks)), "admm_tied_ranks: NULL argument");
  ADMM_HIP_CHECK(hipSetDevice(ctx->device));
  return admm_impl::tied_ranks(ctx, scores, n, ranks);
}

extern "C" int admm_csr_copy_labels(admm_csr *csr, float *dst) {
  ADMM_REQUIRE(csr && (dst || csr->nrows == 0), "admm_csr_copy_labels: NULL argument");
  if (csr->nrows == 0) return ADMM_OK;
  ADMM_HIP_CHECK(hipSetDevice(csr->ctx->device));
  ADMM_HIP_CHECK(hipMemcpyAsync(dst, csr->d_label, sizeof(float) * csr->nrows, hipMemcpyDeviceToDevice, csr->ctx->stream));
  return ADMM_OK;
}
