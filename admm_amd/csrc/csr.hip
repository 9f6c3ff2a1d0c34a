// Sample sets in HBM: upload with the bias shift and its checks, the synthetic generator, relabelling by frequency with
// the row masks and compact rows, sizes, download.
#include "internal.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>

using namespace admm_impl;

namespace admm_impl {

// ---- bias shift on the device (src/sample_set.cc:70-88) --------------------------------------

// rowptr'[r] = rowptr[r] + r ; entry 0 of every row is (col 0, val 1); the rest is col+1.
__global__ void __launch_bounds__(256) k_translate(const uint64_t *__restrict__ rowptr_in,
                                                   const uint32_t *__restrict__ col_in,
                                                   const float *__restrict__ val_in,
                                                   uint64_t nrows, uint64_t *__restrict__ rowptr_out,
                                                   uint32_t *__restrict__ col_out,
                                                   float *__restrict__ val_out) {
  // one 16-lane group per row
  const int sub = threadIdx.x & 15;
  const uint64_t grp = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const uint64_t ngrp = ((uint64_t)gridDim.x * 256) >> 4;
  for (uint64_t r = grp; r < nrows; r += ngrp) {
    const uint64_t b = rowptr_in[r], e = rowptr_in[r + 1];
    const uint64_t ob = b + r;
    if (sub == 0) {
      rowptr_out[r] = ob;
      col_out[ob] = 0u;
      if (val_out) val_out[ob] = 1.0f;
      if (r == nrows - 1) rowptr_out[nrows] = e + nrows;
    }
    for (uint64_t k = b + sub; k < e; k += 16) {
      col_out[ob + 1 + (k - b)] = col_in[k] + 1u;
      if (val_out) val_out[ob + 1 + (k - b)] = val_in ? val_in[k] : 1.0f;
    }
  }
}

// ---- synthetic generator ------------------------------------------------------------------------

__device__ __host__ inline uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ inline float u01(uint64_t h) { return (float)((h >> 40) + 1) * (1.0f / 16777217.0f); }  // (0,1)
__device__ inline float normal01(uint64_t h) {
  const float u1 = u01(h), u2 = u01(mix64(h ^ 0xA5A5A5A5A5A5A5A5ull));
  return sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
}

// bijection on [0, nfeat): multiply-xorshift on kbits bits with cycle walking
__device__ inline uint32_t scatter_id(uint32_t x, uint32_t nfeat, int kbits) {
  const uint32_t mask = kbits >= 32 ? 0xFFFFFFFFu : ((1u << kbits) - 1u);
  do {
    x = (x * 0x9E3779B1u + 0x7F4A7C15u) & mask;
    x ^= x >> ((kbits + 1) / 2);
    x = (x * 0x85EBCA6Bu) & mask;
    x ^= x >> ((kbits + 1) / 2);
  } while (x >= nfeat);
  return x;
}

__device__ inline float wstar_of(uint64_t seed, uint32_t id) {
  const uint64_t h = mix64(seed ^ (0xC0FFEEull << 20) ^ id);
  if ((h & 0xFF) >= 77) return 0.0f;  // ~30 % of the features carry signal
  return 0.7f * normal01(mix64(h));
}

struct SynthArgs {
  admm_synth_spec s;
  uint32_t nfeat;
  int kbits;
  float zipf_a, zipf_top;  // a = 1 - s ; top = (nfeat+1)^a - 1
  float ln_mu;
  uint64_t *rowlen;        // pass 1 output (entries per row incl. bias)
  const uint64_t *rowptr;  // pass 2 input
  uint32_t *col;
  float *val;
  float *label;
};

// One 64-lane workgroup per row (grid-stride): draw ids into LDS, bitonic sort, drop duplicates.
template <bool WRITE>
__global__ void __launch_bounds__(64) k_synth(SynthArgs a) {
  __shared__ uint32_t ids[1024];
  __shared__ uint32_t outn;
  const int lane = threadIdx.x;
  for (uint64_t r = blockIdx.x; r < a.s.nrows; r += gridDim.x) {
    const uint64_t grow = a.s.row_offset + r;
    const uint64_t hrow = mix64(a.s.seed * 0x100000001B3ull + grow);
    // row length
    float lf;
    if (a.s.len_dist == 1) {
      lf = expf(a.ln_mu + a.s.tail_sigma * normal01(hrow));
    } else {
      lf = a.s.mean_nnz + sqrtf(a.s.mean_nnz) * normal01(hrow);
    }
    int len = (int)floorf(lf + 0.5f);
    if (len < 0) len = 0;
    if (len > (int)a.s.max_len) len = (int)a.s.max_len;
    int n2 = 64;
    while (n2 < len) n2 <<= 1;
    for (int k = lane; k < n2; k += 64) {
      uint32_t id = 0xFFFFFFFFu;
      if (k < len) {
        const float u = u01(mix64(hrow + 0x632BE59BD9B4E019ull * (uint64_t)(k + 1)));
        float x = powf(u * a.zipf_top + 1.0f, 1.0f / a.zipf_a);  // in [1, nfeat+1)
        uint32_t rank = (uint32_t)x;
        rank = rank < 1 ? 0 : rank - 1;
        if (rank >= a.nfeat) rank = a.nfeat - 1;
        id = scatter_id(rank, a.nfeat, a.kbits);
      }
      ids[k] = id;
    }
    __syncthreads();
    // bitonic sort of n2 keys by one wave
    for (int size = 2; size <= n2; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = lane; t < (n2 >> 1); t += 64) {
          const int i = 2 * t - (t & (stride - 1));
          const int j = i + stride;
          const bool up = ((i & size) == 0);
          const uint32_t x = ids[i], y = ids[j];
          if ((x > y) == up) {
            ids[i] = y;
            ids[j] = x;
          }
        }
        __syncthreads();
      }
    }
    // unique + compaction (+ hidden-model margin for the label)
    if (lane == 0) outn = 0;
    __syncthreads();
    const uint64_t ob = WRITE ? a.rowptr[r] : 0;
    float margin = 0.0f;
    for (int base = 0; base < n2; base += 64) {
      const int k = base + lane;
      const uint32_t id = ids[k];
      const bool keep = id != 0xFFFFFFFFu && (k == 0 || ids[k - 1] != id);
      const unsigned long long m = __ballot(keep);
      const uint32_t before = outn + __popcll(m & ((1ull << lane) - 1ull));
      if (WRITE && keep) {
        a.col[ob + 1 + before] = id + 1u;
        if (a.val) a.val[ob + 1 + before] = 1.0f;
        margin += wstar_of(a.s.model_seed, id);
      }
      __syncthreads();
      if (lane == 0) outn += (uint32_t)__popcll(m);
      __syncthreads();
    }
    if (WRITE) {
      for (int off = 32; off > 0; off >>= 1) margin += __shfl_down(margin, off, 64);
      if (lane == 0) {
        a.col[ob] = 0u;
        if (a.val) a.val[ob] = 1.0f;
        const float p = 1.0f / (1.0f + expf(-(margin - 0.5f)));
        a.label[r] = u01(mix64(hrow ^ 0x1234567ull)) < p ? 1.0f : 0.0f;
      }
    } else if (lane == 0) {
      a.rowlen[r] = (uint64_t)outn + 1u;
    }
    __syncthreads();
  }
}

// Feature draws per row (before de-duplication) plus the bias entry, summed over blocks of `block_rows` rows: what
// a job needs to cut a generated set into shards of equal entry counts without generating it (the draws are a
// pure function of seed and global row number; the stored entries are the draws minus a row's repeats).
__global__ void __launch_bounds__(256) k_synth_draws(SynthArgs a, uint64_t block_rows, uint64_t nblocks,
                                                     unsigned long long *__restrict__ out) {
  __shared__ unsigned long long sh[4];
  for (uint64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const uint64_t r0 = blk * block_rows, r1 = r0 + block_rows < a.s.nrows ? r0 + block_rows : a.s.nrows;
    unsigned long long acc = 0;
    for (uint64_t r = r0 + threadIdx.x; r < r1; r += 256) {
      const uint64_t hrow = mix64(a.s.seed * 0x100000001B3ull + (a.s.row_offset + r));
      float lf;
      if (a.s.len_dist == 1) lf = expf(a.ln_mu + a.s.tail_sigma * normal01(hrow));
      else lf = a.s.mean_nnz + sqrtf(a.s.mean_nnz) * normal01(hrow);
      int len = (int)floorf(lf + 0.5f);
      len = len < 0 ? 0 : (len > (int)a.s.max_len ? (int)a.s.max_len : len);
      acc += (unsigned long long)len + 1ull;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blk] = sh[0] + sh[1] + sh[2] + sh[3];
  }
}

// ---- relabelling columns by frequency ------------------------------------------------------------

__global__ void __launch_bounds__(256) k_colcount(const uint64_t *__restrict__ rowptr,
                                                  const uint32_t *__restrict__ col, uint64_t nrows,
                                                  uint64_t chunk_stride, uint32_t *__restrict__ count) {
  // chunks of 256 rows; only every chunk_stride-th chunk is counted (the ranking needs no more)
  const uint64_t nchunks = (nrows + 255) / 256;
  for (uint64_t ch = (uint64_t)blockIdx.x * chunk_stride; ch < nchunks; ch += (uint64_t)gridDim.x * chunk_stride) {
    const uint64_t r0 = ch * 256, r1 = r0 + 256 < nrows ? r0 + 256 : nrows;
    const uint64_t kb = rowptr[r0], ke = rowptr[r1];
    for (uint64_t k = kb + threadIdx.x; k < ke; k += 256) atomicAdd(&count[col[k]], 1u);
  }
}

__global__ void __launch_bounds__(256) k_iota(uint32_t *v, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) v[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(256) k_invert(const uint32_t *__restrict__ perm, uint32_t *__restrict__ inv, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) inv[perm[i]] = (uint32_t)i;
}

__global__ void __launch_bounds__(256) k_relabel(uint32_t *__restrict__ col, const uint32_t *__restrict__ inv, uint64_t nnz) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nnz; i += stride) col[i] = inv[col[i]];
}

// Split every row of a relabelled set into a 64-bit mask (unit-valued entries whose id is below 64)
// and the rest, kept in order.  FILL = false counts the rest per row, FILL = true writes it.
template <bool FILL>
__global__ void __launch_bounds__(256) k_compact(const uint64_t *__restrict__ rowptr,
                                                 const uint32_t *__restrict__ col,
                                                 const float *__restrict__ val, uint64_t nrows,
                                                 uint64_t *__restrict__ rowlen,
                                                 const uint64_t *__restrict__ crowptr,
                                                 uint32_t *__restrict__ ccol, float *__restrict__ cval,
                                                 uint64_t *__restrict__ hmask, uint64_t *__restrict__ hwarm) {
  // one 16-lane group per row
  const int lane = threadIdx.x & 63, sub = threadIdx.x & 15, gshift = lane & 48;
  const uint64_t grp = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const uint64_t ngrp = ((uint64_t)gridDim.x * 256) >> 4;
  const uint64_t nrounds = (nrows + ngrp - 1) / ngrp;  // wave-uniform trip count (ballots inside)
  for (uint64_t it = 0; it < nrounds; ++it) {
    const uint64_t r = grp + it * ngrp;
    const bool live = r < nrows;
    const uint64_t b = live ? rowptr[r] : 0, e = live ? rowptr[r + 1] : 0;
    uint64_t out = FILL && live ? crowptr[r] : 0;
    uint32_t mbits[2 * (1 + kWarmWords)] = {};
    uint32_t ncold = 0;
    const uint64_t trips = (e - b + 15) / 16;
    uint64_t maxtrips = trips;
    for (int off = 32; off > 0; off >>= 1) {
      const uint64_t o = __shfl_xor((unsigned long long)maxtrips, off, 64);
      maxtrips = o > maxtrips ? o : maxtrips;
    }
    for (uint64_t tr = 0; tr < maxtrips; ++tr) {
      const uint64_t k = b + tr * 16 + sub;
      const bool have = live && k < e;
      const uint32_t c = have ? col[k] : 0xFFFFFFFFu;
      const float v = have ? (val ? val[k] : 1.0f) : 0.0f;
      const bool hot = have && c < kMaskCols && v == 1.0f;
      const bool cold = have && !hot;
      if (hot) {
#pragma unroll
        for (int w = 0; w < 2 * (1 + kWarmWords); ++w)
          if ((c >> 5) == (uint32_t)w) mbits[w] |= 1u << (c & 31u);
      }
      const unsigned long long cm = __ballot(cold);
      const uint32_t gm = (uint32_t)((cm >> gshift) & 0xFFFFu);
      if (FILL && cold) {
        const uint64_t pos = out + __popc(gm & ((1u << sub) - 1u));
        ccol[pos] = c;
        if (cval) cval[pos] = v;
      }
      out += __popc(gm);
      ncold += __popc(gm);
    }
    if (!FILL) {
      for (int off = 8; off > 0; off >>= 1) {
#pragma unroll
        for (int w = 0; w < 2 * (1 + kWarmWords); ++w) mbits[w] |= __shfl_xor(mbits[w], off, 64);
      }
      if (live && sub == 0) {
        rowlen[r] = ncold;
        hmask[r] = ((uint64_t)mbits[1] << 32) | mbits[0];
#pragma unroll
        for (int w = 0; w < kWarmWords; ++w)
          hwarm[r * kWarmWords + w] = ((uint64_t)mbits[2 * w + 3] << 32) | mbits[2 * w + 2];
      }
    }
  }
}

static void pick_lpr(admm_csr *csr) {
  const double mean = csr->nrows ? (double)csr->nnz / (double)csr->nrows : 1.0;
  uint32_t lpr = 4;
  while (lpr < 64 && mean > 4.0 * lpr) lpr <<= 1;
  csr->fwd_lpr = lpr;
}

static void free_csr(admm_csr *c) {
  if (!c) return;
  if (c->ctx) {
    hipSetDevice(c->ctx->device);
    hipStreamSynchronize(c->ctx->stream);
  }
  void *ptrs[] = {c->d_rowptr, c->d_col, c->d_val,  c->d_label, c->d_rowbuf, c->d_margin, c->d_dmargin, c->d_cache_x, c->d_cache_off, c->d_cache_gl, c->d_tptr,
                  c->d_trow,   c->d_tcol, c->d_tval, c->d_g1,    c->d_g2,
                  c->d_wsum,   c->d_perm, c->d_inv,  c->d_hmask, c->d_hwarm, c->d_crowptr, c->d_ccol, c->d_cval, c->d_dep, c->d_seqcol,
                  c->d_fcol,   c->d_frow, c->d_fval, c->d_fptr, c->d_hcol, c->d_hval, c->d_hbase, c->d_hfirst,
                  c->d_tcw,    c->d_tcv,  c->d_tcbase, c->d_tcfirst, c->d_hmaskT, c->d_mpart, c->d_gs};
  for (void *p : ptrs)
    if (p) hipFree(p);
  delete c;
}

}  // namespace admm_impl

// =============================================================================================
// C ABI
// =============================================================================================

namespace admm_impl {

// What admm_csr_upload checks, on the device (the arrays are there anyway, and a host loop over 10^10 entries
// would take longer than their transfer): row pointers that decrease or leave [0, nnz], feature indices
// beyond `limit`, a value other than 1, a row that stores a column twice.  One 16-lane group per row.  Rows
// whose indices increase strictly (what libsvm files hold) need no pairwise comparison.
struct UploadFlags {
  unsigned long long bad_row;  // smallest offending row, ~0 if none
  unsigned int dup, nonunit;
  unsigned int long_unsorted, pad;  // some row longer than kPairwiseCap is not in increasing order: the host looks at those
};

// rows up to this length are searched for a repeated column pair by pair on the device (16 lanes per row); a
// longer row out of order -- a million entries would keep its 16 lanes comparing for minutes -- is left to the
// host, which sorts a copy of it
constexpr uint64_t kPairwiseCap = 2048;

__global__ void __launch_bounds__(256) k_validate(const uint64_t *__restrict__ rowptr, const uint32_t *__restrict__ col,
                                                  const float *__restrict__ val, uint64_t nrows, uint64_t nnz,
                                                  uint64_t limit, UploadFlags *__restrict__ f) {
  const int sub = threadIdx.x & 15;
  const uint64_t grp = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const uint64_t ngrp = ((uint64_t)gridDim.x * 256) >> 4;
  for (uint64_t r = grp; r < nrows; r += ngrp) {
    const uint64_t b = rowptr[r], e = rowptr[r + 1];
    if (e < b || e > nnz) {
      if (sub == 0) atomicMin(&f->bad_row, (unsigned long long)r);
      continue;
    }
    int unsorted = 0;
    for (uint64_t k = b + sub; k < e; k += 16) {
      const uint32_t c = col[k];
      if (c >= limit) atomicMin(&f->bad_row, (unsigned long long)r);
      if (val && val[k] != 1.0f) f->nonunit = 1u;
      if (k > b && col[k - 1] >= c) unsorted = 1;
    }
    for (int off = 8; off > 0; off >>= 1) unsorted |= __shfl_xor(unsorted, off, 16);
    if (unsorted && e - b > kPairwiseCap) {
      f->long_unsorted = 1u;
    } else if (unsorted) {
      for (uint64_t k = b + sub; k < e; k += 16) {
        const uint32_t c = col[k];
        for (uint64_t q = b; q < k; ++q)
          if (col[q] == c) {
            f->dup = 1u;
            break;
          }
      }
    }
  }
}

}  // namespace admm_impl

extern "C" int admm_csr_upload(admm_ctx *ctx, const uint64_t *rowptr, const uint32_t *col,
                               const float *val, const float *label, uint64_t nrows, uint64_t nnz,
                               uint64_t dim, int add_bias, admm_csr **out) {
  ADMM_REQUIRE(ctx && rowptr && label && out, "admm_csr_upload: NULL argument");
  ADMM_REQUIRE(nnz == 0 || col, "admm_csr_upload: col is NULL");
  // 2^30: the forward sweep addresses the gathered vector through a buffer resource whose size
  // field is 32 bits of bytes
  ADMM_REQUIRE(dim > 0 && dim < (1ull << 30), "admm_csr_upload: dim %llu out of range (below 2^30)",
               (unsigned long long)dim);
  ADMM_REQUIRE(nrows < 0xFFFFFFFFull, "admm_csr_upload: at most 2^32-1 rows per sample set");
  ADMM_REQUIRE(rowptr[0] == 0 && rowptr[nrows] == nnz, "admm_csr_upload: rowptr does not span nnz");
  const uint64_t limit = add_bias ? dim - 1 : dim;
  ADMM_HIP_CHECK(hipSetDevice(ctx->device));
  admm_csr *c = new admm_csr();
  c->ctx = ctx;
  c->nrows = nrows;
  c->dim = dim;
  c->has_bias = add_bias != 0;
  c->nnz = add_bias ? nnz + nrows : nnz;
  // the raw arrays go to the device first (with the padding the sweeps' wide loads want), are checked
  // there, and then either become the set's arrays or are translated into them (bias column)
  uint64_t *t_rowptr = nullptr;
  uint32_t *t_col = nullptr;
  float *t_val = nullptr;
  UploadFlags *d_flags = nullptr;
  auto fail = [&](int code) {
    if (t_rowptr) hipFree(t_rowptr);
    if (t_col) hipFree(t_col);
    if (t_val) hipFree(t_val);
    if (d_flags) hipFree(d_flags);
    free_csr(c);
    return code;
  };
#define UP_CHECK(expr)                                                                     \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      set_error("%s failed: %s", #expr, hipGetErrorString(e_));                            \
      return fail(e_ == hipErrorOutOfMemory ? ADMM_ERR_NOMEM : ADMM_ERR_HIP);              \
    }                                                                                      \
  } while (0)
  UP_CHECK(hipMalloc(&t_rowptr, sizeof(uint64_t) * (nrows + 1)));
  UP_CHECK(hipMalloc(&t_col, sizeof(uint32_t) * (nnz + 8)));
  if (val) UP_CHECK(hipMalloc(&t_val, sizeof(float) * (nnz + 8)));
  UP_CHECK(hipMalloc(&d_flags, sizeof(UploadFlags)));
  UP_CHECK(hipMalloc(&c->d_label, sizeof(float) * std::max<uint64_t>(nrows, 1)));
  UploadFlags flags = {~0ull, 0u, 0u, 0u, 0u};
  UP_CHECK(hipMemcpyAsync(d_flags, &flags, sizeof(flags), hipMemcpyHostToDevice, ctx->stream));
  if (nrows) UP_CHECK(hipMemcpyAsync(c->d_label, label, sizeof(float) * nrows, hipMemcpyHostToDevice, ctx->stream));
  UP_CHECK(hipMemcpyAsync(t_rowptr, rowptr, sizeof(uint64_t) * (nrows + 1), hipMemcpyHostToDevice, ctx->stream));
  if (nnz) UP_CHECK(hipMemcpyAsync(t_col, col, sizeof(uint32_t) * nnz, hipMemcpyHostToDevice, ctx->stream));
  if (val && nnz) UP_CHECK(hipMemcpyAsync(t_val, val, sizeof(float) * nnz, hipMemcpyHostToDevice, ctx->stream));
  if (nrows) {
    hipLaunchKernelGGL(k_validate, dim3(grid_for(nrows * 16, 256, ctx->cu_count * 8)), dim3(256), 0, ctx->stream, t_rowptr,
                       t_col, t_val, nrows, nnz, limit, d_flags);
    UP_CHECK(hipGetLastError());
  }
  UP_CHECK(hipMemcpyAsync(&flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, ctx->stream));
  UP_CHECK(hipStreamSynchronize(ctx->stream));
  if (flags.bad_row != ~0ull) {
    // the message names what is wrong with the first offending row (the host still holds the arrays)
    const uint64_t r = flags.bad_row;
    if (rowptr[r] > rowptr[r + 1] || rowptr[r + 1] > nnz) {
      set_error("admm_csr_upload: rowptr decreases at row %llu", (unsigned long long)r);
    } else {
      for (uint64_t k = rowptr[r]; k < rowptr[r + 1]; ++k)
        if (col[k] >= limit) {
          set_error("admm_csr_upload: feature index %u of row %llu needs dim > %llu", col[k], (unsigned long long)r,
                    (unsigned long long)(add_bias ? col[k] + 1 : col[k]));
          break;
        }
    }
    return fail(ADMM_ERR_INVALID);
  }
  if (flags.long_unsorted && !flags.dup) {
    // the few long rows out of order, on the host: sort a copy, look at neighbours
    std::vector<uint32_t> tmp_ids;
    for (uint64_t r = 0; r < nrows && !flags.dup; ++r) {
      const uint64_t b = rowptr[r], e = rowptr[r + 1];
      if (e - b <= kPairwiseCap) continue;
      bool increasing = true;
      for (uint64_t k = b + 1; k < e && increasing; ++k) increasing = col[k - 1] < col[k];
      if (increasing) continue;
      tmp_ids.assign(col + b, col + e);
      std::sort(tmp_ids.begin(), tmp_ids.end());
      if (std::adjacent_find(tmp_ids.begin(), tmp_ids.end()) != tmp_ids.end()) flags.dup = 1u;
    }
  }
  c->has_dup_cols = flags.dup != 0;
  // a set whose stored values are all 1 (binary features, the common case for this model) is kept
  // as a pattern matrix: no value array, the kernels read an implicit 1
  const bool keep_val = val && flags.nonunit != 0;
  c->has_val = keep_val;
  if (!keep_val && t_val) {
    hipFree(t_val);
    t_val = nullptr;
  }
  if (!add_bias) {
    c->d_rowptr = t_rowptr;
    c->d_col = t_col;
    c->d_val = t_val;
    t_rowptr = nullptr;
    t_col = nullptr;
    t_val = nullptr;
  } else {
    UP_CHECK(hipMalloc(&c->d_rowptr, sizeof(uint64_t) * (nrows + 1)));
    UP_CHECK(hipMalloc(&c->d_col, sizeof(uint32_t) * (c->nnz + 8)));
    if (keep_val) UP_CHECK(hipMalloc(&c->d_val, sizeof(float) * (c->nnz + 8)));
    if (nrows) {
      const int grid = grid_for(nrows * 16, 256, ctx->cu_count * 8);
      hipLaunchKernelGGL(k_translate, dim3(grid), dim3(256), 0, ctx->stream, t_rowptr, t_col, t_val,
                         nrows, c->d_rowptr, c->d_col, c->d_val);
    } else {
      UP_CHECK(hipMemsetAsync(c->d_rowptr, 0, sizeof(uint64_t), ctx->stream));
    }
    UP_CHECK(hipGetLastError());
    UP_CHECK(hipStreamSynchronize(ctx->stream));
    hipFree(t_rowptr);
    hipFree(t_col);
    if (t_val) hipFree(t_val);
    t_rowptr = nullptr;
    t_col = nullptr;
    t_val = nullptr;
  }
  hipFree(d_flags);
  d_flags = nullptr;
#undef UP_CHECK
  pick_lpr(c);
  *out = c;
  return ADMM_OK;
}

extern "C" int admm_csr_shape(const admm_csr *csr, uint64_t *nrows, uint64_t *nnz, uint64_t *dim) {
  ADMM_REQUIRE(csr, "admm_csr_shape: csr is NULL");
  if (nrows) *nrows = csr->nrows;
  if (nnz) *nnz = csr->nnz;
  if (dim) *dim = csr->dim;
  return ADMM_OK;
}

extern "C" int admm_csr_bytes(const admm_csr *csr, uint64_t *csr_bytes, uint64_t *transposed_bytes) {
  ADMM_REQUIRE(csr, "admm_csr_bytes: csr is NULL");
  if (csr_bytes)
    *csr_bytes = csr->rows_released ? 4 * csr->nrows
                                    : 8 * (csr->nrows + 1) + 4 * csr->nnz + (csr->has_val ? 4 * csr->nnz : 0) + 4 * csr->nrows;
  if (transposed_bytes)
    *transposed_bytes = csr->has_t ? 8 * (csr->npanels * csr->nblocks + 1) + 6 * csr->tnnz +
                                         (csr->d_tval ? 4 * csr->tnnz : 0) + (csr->d_tcv ? 6 : 2) * csr->tc_words
                                   : 0;
  return ADMM_OK;
}

extern "C" int admm_csr_layout_bytes(const admm_csr *csr, uint64_t out[4]) {
  ADMM_REQUIRE(csr && out, "admm_csr_layout_bytes: NULL argument");
  const uint64_t n = csr->nrows;
  // (the row-major forms: the CSR and, beside a row-panel copy, the compact rows of a set with row masks)
  out[0] = (csr->rows_released ? 0 : 8 * (n + 1) + 4 * csr->nnz + (csr->has_val ? 4 * csr->nnz : 0) + 4 * n) +
           ((csr->has_fp && csr->hot_dense && csr->d_crowptr) ? 8 * (n + 1) + 4 * csr->cnnz + (csr->d_cval ? 4 * csr->cnnz : 0) : 0);
  if (csr->has_fp)  // row-panel copy: masks + hot stream (u16 [, value]) + (column u32, row u16 [, value]) per remaining entry + labels
    out[1] = (csr->hot_dense ? 8 * n * (2 + kWarmWords) : 0) + (csr->d_hval ? 6 : 2) * csr->hp_words +
             (csr->d_fval ? 10 : 6) * csr->fp_nnz + 4 * n;
  else
    out[1] = csr->hot_dense ? 8 * (n + 1) + 8 * n * (1 + kWarmWords) + 4 * csr->cnnz + (csr->d_cval ? 4 * csr->cnnz : 0) + 4 * n
                            : out[0];
  out[2] = csr->has_t ? 8 * (csr->npanels * csr->nblocks + 1) + 6 * csr->tnnz + (csr->d_tval ? 4 * csr->tnnz : 0) +
                            (csr->d_tcv ? 6 : 2) * csr->tc_words
                      : 0;
  out[3] = (csr->d_rowbuf ? 4 * n : 0) + (csr->d_margin ? 4 * n : 0) + (csr->d_dmargin ? 4 * n : 0);
  return ADMM_OK;
}

extern "C" int admm_csr_layout_counts(const admm_csr *csr, uint64_t out[6]) {
  ADMM_REQUIRE(csr && out, "admm_csr_layout_counts: NULL argument");
  out[0] = csr->hot_dense ? csr->nnz - csr->cnnz : 0;
  out[1] = csr->has_fp ? csr->hp_words : 0;
  out[2] = csr->has_fp ? csr->fp_nnz : 0;
  out[3] = csr->has_tc ? csr->tc_words : 0;
  out[4] = csr->has_t ? csr->tnnz : 0;
  out[5] = csr->has_t ? csr->npanels * csr->nblocks : 0;
  return ADMM_OK;
}

extern "C" int admm_csr_download(const admm_csr *csr, uint64_t *rowptr, uint32_t *col, float *val,
                                 float *label) {
  ADMM_REQUIRE(csr, "admm_csr_download: csr is NULL");
  ADMM_TRY(require_rows(csr, "admm_csr_download"));
  admm_ctx *ctx = csr->ctx;
  ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (rowptr) ADMM_HIP_CHECK(hipMemcpy(rowptr, csr->d_rowptr, sizeof(uint64_t) * (csr->nrows + 1), hipMemcpyDeviceToHost));
  if (col && csr->nnz) {
    ADMM_HIP_CHECK(hipMemcpy(col, csr->d_col, sizeof(uint32_t) * csr->nnz, hipMemcpyDeviceToHost));
    if (csr->d_perm) {  // stored ids are frequency ranks: hand back the original ids
      std::vector<uint32_t> perm(csr->dim);
      ADMM_HIP_CHECK(hipMemcpy(perm.data(), csr->d_perm, sizeof(uint32_t) * csr->dim, hipMemcpyDeviceToHost));
      for (uint64_t i = 0; i < csr->nnz; ++i) col[i] = perm[col[i]];
    }
  }
  if (val && csr->nnz) {
    if (csr->d_val) {
      ADMM_HIP_CHECK(hipMemcpy(val, csr->d_val, sizeof(float) * csr->nnz, hipMemcpyDeviceToHost));
    } else {
      for (uint64_t i = 0; i < csr->nnz; ++i) val[i] = 1.0f;
    }
  }
  if (label && csr->nrows) ADMM_HIP_CHECK(hipMemcpy(label, csr->d_label, sizeof(float) * csr->nrows, hipMemcpyDeviceToHost));
  return ADMM_OK;
}

extern "C" int admm_csr_relabel_by_frequency(admm_csr *csr) {
  ADMM_REQUIRE(csr, "admm_csr_relabel_by_frequency: csr is NULL");
  if (csr->d_perm || csr->nnz == 0) return ADMM_OK;
  ADMM_TRY(require_rows(csr, "admm_csr_relabel_by_frequency"));
  admm_ctx *ctx = csr->ctx;
  ADMM_HIP_CHECK(hipSetDevice(ctx->device));
  const uint64_t dim = csr->dim;
  uint32_t *count = nullptr, *count_sorted = nullptr, *ids = nullptr;
  void *tmp = nullptr;
  auto cleanup = [&]() {
    for (void *p : {(void *)count, (void *)count_sorted, (void *)ids, tmp})
      if (p) hipFree(p);
  };
#define RL_CHECK(expr)                                                               \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess) {                                                          \
      set_error("admm_csr_relabel_by_frequency: %s: %s", #expr, hipGetErrorString(e_)); \
      cleanup();                                                                     \
      return e_ == hipErrorOutOfMemory ? ADMM_ERR_NOMEM : ADMM_ERR_HIP;              \
    }                                                                                \
  } while (0)
  RL_CHECK(hipMalloc(&count, sizeof(uint32_t) * dim));
  RL_CHECK(hipMalloc(&count_sorted, sizeof(uint32_t) * dim));
  RL_CHECK(hipMalloc(&ids, sizeof(uint32_t) * dim));
  RL_CHECK(hipMalloc(&csr->d_perm, sizeof(uint32_t) * dim));
  RL_CHECK(hipMalloc(&csr->d_inv, sizeof(uint32_t) * dim));
  RL_CHECK(hipMemsetAsync(count, 0, sizeof(uint32_t) * dim, ctx->stream));
  const uint64_t stride = std::max<uint64_t>(1, csr->nnz / 400000000ull);
  const int egrid = grid_for(dim, 256, ctx->cu_count * 8);
  hipLaunchKernelGGL(k_colcount, dim3(ctx->cu_count * 8), dim3(256), 0, ctx->stream, csr->d_rowptr, csr->d_col,
                     csr->nrows, stride, count);
  if (csr->has_bias) {  // the bias column keeps id 0
    const uint32_t top = 0xFFFFFFFFu;
    RL_CHECK(hipMemcpyAsync(count, &top, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
  }
  hipLaunchKernelGGL(k_iota, dim3(egrid), dim3(256), 0, ctx->stream, ids, dim);
  size_t bytes = 0;
  RL_CHECK(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, count, count_sorted, ids, csr->d_perm,
                                                        (int64_t)dim, 0, 32, ctx->stream));
  RL_CHECK(hipMalloc(&tmp, std::max<size_t>(bytes, 16)));
  RL_CHECK(hipcub::DeviceRadixSort::SortPairsDescending(tmp, bytes, count, count_sorted, ids, csr->d_perm,
                                                        (int64_t)dim, 0, 32, ctx->stream));
  hipLaunchKernelGGL(k_invert, dim3(egrid), dim3(256), 0, ctx->stream, csr->d_perm, csr->d_inv, dim);
  hipLaunchKernelGGL(k_relabel, dim3(grid_for(csr->nnz, 256, ctx->cu_count * 16)), dim3(256), 0, ctx->stream,
                     csr->d_col, csr->d_inv, csr->nnz);
  RL_CHECK(hipGetLastError());
  RL_CHECK(hipStreamSynchronize(ctx->stream));
  cleanup();
  count = count_sorted = ids = nullptr;
  tmp = nullptr;
  // compact form for the forward sweep (sets whose rows never repeat a column, dim >= kMaskCols)
  if (!csr->has_dup_cols && dim >= kMaskCols && csr->nrows > 0) {
    const uint64_t n = csr->nrows;
    uint64_t *rowlen = nullptr;
    RL_CHECK(hipMalloc(&rowlen, sizeof(uint64_t) * (n + 1)));
    RL_CHECK(hipMemsetAsync(rowlen, 0, sizeof(uint64_t) * (n + 1), ctx->stream));
    RL_CHECK(hipMalloc(&csr->d_hmask, sizeof(uint64_t) * n));
    RL_CHECK(hipMalloc(&csr->d_hwarm, sizeof(uint64_t) * n * kWarmWords));
    RL_CHECK(hipMalloc(&csr->d_crowptr, sizeof(uint64_t) * (n + 1)));
    const int cgrid = grid_for(n * 16, 256, ctx->cu_count * 8);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_compact<false>), dim3(cgrid), dim3(256), 0, ctx->stream, csr->d_rowptr,
                       csr->d_col, csr->d_val, n, rowlen, nullptr, nullptr, nullptr, csr->d_hmask, csr->d_hwarm);
    size_t sbytes = 0;
    hipError_t se = hipcub::DeviceScan::ExclusiveSum(nullptr, sbytes, rowlen, csr->d_crowptr, (int64_t)(n + 1), ctx->stream);
    void *stmp = nullptr;
    if (se == hipSuccess) se = hipMalloc(&stmp, std::max<size_t>(sbytes, 16));
    if (se == hipSuccess)
      se = hipcub::DeviceScan::ExclusiveSum(stmp, sbytes, rowlen, csr->d_crowptr, (int64_t)(n + 1), ctx->stream);
    uint64_t total = 0;
    if (se == hipSuccess)
      se = hipMemcpyAsync(&total, csr->d_crowptr + n, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (se == hipSuccess) se = hipStreamSynchronize(ctx->stream);
    if (stmp) hipFree(stmp);
    hipFree(rowlen);
    RL_CHECK(se);
    csr->cnnz = total;
    RL_CHECK(hipMalloc(&csr->d_ccol, sizeof(uint32_t) * (total + 8)));
    if (csr->d_val) RL_CHECK(hipMalloc(&csr->d_cval, sizeof(float) * (total + 8)));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_compact<true>), dim3(cgrid), dim3(256), 0, ctx->stream, csr->d_rowptr,
                       csr->d_col, csr->d_val, n, nullptr, csr->d_crowptr, csr->d_ccol, csr->d_cval, nullptr, nullptr);
    RL_CHECK(hipGetLastError());
    RL_CHECK(hipStreamSynchronize(ctx->stream));
    csr->hot_dense = true;
  }
  csr->cache_valid = false;
  // tables a sequential FTRL pass built before the relabelling hold (or lack) column ids in the old
  // numbering: drop them, the next pass rebuilds both
  for (void **p : {(void **)&csr->d_dep, (void **)&csr->d_seqcol}) {
    if (*p) hipFree(*p);
    *p = nullptr;
  }
#undef RL_CHECK
  // copies built earlier are in the old numbering (the row-panel copy's ids index the gathered vector in stored
  // order, the column stream's the stored ids' sums), and they still hold the entries the masks now cover: rebuild
  if (csr->has_t) return build_transposed(csr, csr->panel_rows, csr->col_block, csr->has_tc);
  ADMM_TRY(release_row_major(csr));
  return ADMM_OK;
}

extern "C" int admm_csr_set_lean(admm_csr *csr, int on) {
  ADMM_REQUIRE(csr, "admm_csr_set_lean: csr is NULL");
  ADMM_REQUIRE(!csr->rows_released || on, "admm_csr_set_lean: the rows are gone already");
  csr->lean = on != 0;
  if (csr->lean && csr->hot_dense) {
    ADMM_TRY(release_row_major(csr));
    ADMM_TRY(release_compact(csr));
  }
  return ADMM_OK;
}

extern "C" int admm_csr_destroy(admm_csr *csr) {
  free_csr(csr);
  return ADMM_OK;
}

extern "C" int admm_synth_row_draws(admm_ctx *ctx, const admm_synth_spec *spec, uint64_t block_rows, uint64_t *sums_host) {
  ADMM_REQUIRE(ctx && spec && sums_host && block_rows > 0, "admm_synth_row_draws: bad arguments");
  ADMM_REQUIRE(spec->max_len >= 1 && spec->max_len <= 1024, "admm_synth_row_draws: max_len must be in [1, 1024]");
  const uint64_t nblocks = (spec->nrows + block_rows - 1) / block_rows;
  if (nblocks == 0) return ADMM_OK;
  ADMM_HIP_CHECK(hipSetDevice(ctx->device));
  SynthArgs a{};
  a.s = *spec;
  a.ln_mu = logf(spec->mean_nnz) - 0.5f * spec->tail_sigma * spec->tail_sigma;
  unsigned long long *d = nullptr;
  ADMM_HIP_CHECK(hipMalloc(&d, sizeof(unsigned long long) * nblocks));
  hipLaunchKernelGGL(k_synth_draws, dim3((unsigned)std::min<uint64_t>(nblocks, (uint64_t)ctx->cu_count * 8)), dim3(256), 0, ctx->stream,
                     a, block_rows, nblocks, d);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(sums_host, d, sizeof(unsigned long long) * nblocks, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  ADMM_HIP_CHECK(e);
  return ADMM_OK;
}

extern "C" int admm_csr_synth(admm_ctx *ctx, const admm_synth_spec *spec, admm_csr **out) {
  ADMM_REQUIRE(ctx && spec && out, "admm_csr_synth: NULL argument");
  ADMM_REQUIRE(spec->dim >= 3 && spec->dim < (1ull << 30), "admm_csr_synth: dim out of range (3 .. 2^30 - 1)");
  ADMM_REQUIRE(spec->max_len >= 1 && spec->max_len <= 1024, "admm_csr_synth: max_len must be in [1, 1024]");
  ADMM_REQUIRE(spec->zipf_s > 1.0f, "admm_csr_synth: zipf_s must be > 1");
  ADMM_REQUIRE(spec->nrows < 0xFFFFFFFFull, "admm_csr_synth: at most 2^32-1 rows per sample set");
  ADMM_HIP_CHECK(hipSetDevice(ctx->device));
  admm_csr *c = new admm_csr();
  c->ctx = ctx;
  c->nrows = spec->nrows;
  c->dim = spec->dim;
  c->has_bias = true;
  SynthArgs a{};
  a.s = *spec;
  a.nfeat = (uint32_t)(spec->dim - 1);
  a.kbits = 1;
  while ((1ull << a.kbits) < a.nfeat) ++a.kbits;
  a.zipf_a = 1.0f - spec->zipf_s;
  a.zipf_top = powf((float)a.nfeat + 1.0f, a.zipf_a) - 1.0f;
  a.ln_mu = logf(spec->mean_nnz) - 0.5f * spec->tail_sigma * spec->tail_sigma;
  auto fail = [&](const char *what, hipError_t e) {
    set_error("admm_csr_synth: %s: %s", what, hipGetErrorString(e));
    free_csr(c);
    return e == hipErrorOutOfMemory ? ADMM_ERR_NOMEM : ADMM_ERR_HIP;
  };
  hipError_t e;
  uint64_t *d_len = nullptr;
  const uint64_t n = spec->nrows;
  if ((e = hipMalloc(&c->d_rowptr, sizeof(uint64_t) * (n + 1))) != hipSuccess) return fail("rowptr", e);
  if ((e = hipMalloc(&d_len, sizeof(uint64_t) * (n + 1))) != hipSuccess) return fail("rowlen", e);
  if ((e = hipMalloc(&c->d_label, sizeof(float) * std::max<uint64_t>(n, 1))) != hipSuccess) {
    hipFree(d_len);
    return fail("label", e);
  }
  hipMemsetAsync(d_len, 0, sizeof(uint64_t) * (n + 1), ctx->stream);
  const int grid = grid_for(n * 64, 64, ctx->cu_count * 32);
  a.rowlen = d_len;
  if (n) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_synth<false>), dim3(grid), dim3(64), 0, ctx->stream, a);
  // exclusive scan of n+1 lengths (last is 0) -> rowptr[0..n]
  void *tmp = nullptr;
  size_t tmp_bytes = 0;
  hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_len, c->d_rowptr, (int64_t)(n + 1), ctx->stream);
  if ((e = hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16))) != hipSuccess) {
    hipFree(d_len);
    return fail("scan scratch", e);
  }
  hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, d_len, c->d_rowptr, (int64_t)(n + 1), ctx->stream);
  uint64_t total = 0;
  hipMemcpyAsync(&total, c->d_rowptr + n, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream);
  e = hipStreamSynchronize(ctx->stream);
  hipFree(tmp);
  hipFree(d_len);
  if (e != hipSuccess) return fail("length pass", e);
  c->nnz = total;
  if ((e = hipMalloc(&c->d_col, sizeof(uint32_t) * (total + 8))) != hipSuccess) return fail("col", e);
  // the generated values are all 1: stored only on request
  if (spec->explicit_values && (e = hipMalloc(&c->d_val, sizeof(float) * (total + 8))) != hipSuccess) return fail("val", e);
  c->has_val = spec->explicit_values != 0;
  a.rowptr = c->d_rowptr;
  a.col = c->d_col;
  a.val = c->d_val;
  a.label = c->d_label;
  if (n) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_synth<true>), dim3(grid), dim3(64), 0, ctx->stream, a);
  if ((e = hipGetLastError()) != hipSuccess) return fail("fill pass launch", e);
  if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail("fill pass", e);
  pick_lpr(c);
  *out = c;
  return ADMM_OK;
}

