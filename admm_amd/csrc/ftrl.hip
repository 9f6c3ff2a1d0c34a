// Exact sequential FTRL-proximal pass: the row-by-row semantics of FtrlSolver::Run
// (src/ftrl.cc:101-145).  Row t+1's prediction depends on row t's update through every shared
// coordinate (the bias column is shared by all rows), so rows are walked in order by ONE
// wavefront; the parallelism is across the coordinates of a row (one lane each).
#include "internal.h"

#include <algorithm>
#include <type_traits>

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

// the same with sqrt(n) (in double) supplied by the caller, who needs it again for the update
__device__ inline float seq_weight_sn(float z, double sqrt_n, float c, float alpha, float beta, float l1, float l2) {
  const float sign = z < 0.0f ? -1.0f : 1.0f;
  const float num = sign * 1 - z + l2 * c;
  const double den = ((double)beta + sqrt_n) / (double)alpha + (double)l2;
  const float w = (float)((double)num / den);
  return sign * z <= l1 ? 0.0f : w;
}

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum over the wavefront with DPP adds (no LDS round trips); every lane returns the total.
__device__ inline float wave_sum_dpp(float v) {
  auto dpp = [](float x, auto ctrl, auto row_mask) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value,
                                           decltype(row_mask)::value, 0xF, false));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});   // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});   // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x124>{}, std::integral_constant<int, 0xF>{});  // row_ror:4
  v += dpp(v, std::integral_constant<int, 0x128>{}, std::integral_constant<int, 0xF>{});  // row_ror:8
  v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});  // row_bcast:15
  v += dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});  // row_bcast:31
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
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

// ---- pipelined exact pass ---------------------------------------------------------------------
//
// The dependence between rows goes through the coordinates they share.  Which earlier row last
// touched each stored coordinate, and at which position, depends only on the data, so it is
// computed once per sample set (d_dep).  With that, one wavefront can run several rows ahead:
// the state of row t+kAhead is gathered from memory while row t is being solved, and every
// coordinate that one of the last kRing-1 rows has updated since takes its value from a small
// LDS ring of those rows' results instead of the (possibly stale) gathered value.  Rows older
// than the ring were stored at least kRing - kAhead rows before the gather was issued.
constexpr int kRing = 8;       // rows whose results stay in LDS
constexpr int kAhead = 4;      // state gathers are issued this many rows ahead
constexpr int kStatic = 8;     // column ids / values / dependences are fetched this many rows ahead

// Every load below is unconditional (indices are clamped, results masked), and the pipeline is
// filled by running the same loop body over kStatic empty rows in front of row 0: a load under a
// branch, or a separate prologue, would make the number of requests in flight unknown to the
// compiler at the top of the loop, which then waits for ALL of them and so undoes the look-ahead.
// EPL = stored entries per lane: rows of up to 64 * EPL entries
template <int EPL, bool HASVAL>
__device__ __forceinline__ void ftrl_ring_body(const SeqArgs &a, const uint32_t *__restrict__ col,
                                               const uint32_t *__restrict__ dep, uint64_t nnz) {
  __shared__ float ring_z_[kRing][64 * EPL];
  __shared__ float ring_n_[kRing][64 * EPL];
  // plain (not volatile) LDS accesses: a wavefront's LDS operations execute in order, and a
  // volatile access would make the compiler drain every outstanding load around it
  float(*ring_z)[64 * EPL] = ring_z_;
  float(*ring_n)[64 * EPL] = ring_n_;
  const int lane = threadIdx.x;
  float *w = a.w, *z = a.z, *n = a.n;
  const long long nrows = (long long)a.nrows;
  const long long total = nrows + 1;  // + the trailing Predict on the last row
  const uint64_t last_entry = nnz - 1;
  // static per-row data, kStatic rows deep
  uint32_t s_col[kStatic][EPL], s_dep[kStatic][EPL];
  float s_val[kStatic][EPL];
  uint32_t s_len[kStatic];
  float s_lab[kStatic];
  // gathered state, kAhead rows deep
  float g_z[kAhead][EPL], g_n[kAhead][EPL], g_r[kAhead][EPL], g_o[kAhead][EPL];
#pragma unroll
  for (int u = 0; u < kStatic; ++u) {
    s_len[u] = 0;
    s_lab[u] = 0.0f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) s_col[u][e] = 0, s_dep[u][e] = 0, s_val[u][e] = 0.0f;
  }
#pragma unroll
  for (int u = 0; u < kAhead; ++u)
#pragma unroll
    for (int e = 0; e < EPL; ++e) g_z[u][e] = g_n[u][e] = g_r[u][e] = g_o[u][e] = 0.0f;
  // Row pointers and labels travel one group of kStatic rows per lane set: lane u of a group's
  // registers holds row (group * kStatic + u).  `nx` is the group whose static data is fetched
  // during the current group, `nn` the one after it (still in flight).
  unsigned long long nx_lo, nx_hi, nn_lo, nn_hi;
  float nx_lab, nn_lab;
  auto load_group = [&](long long grp, unsigned long long &lo, unsigned long long &hi, float &lab) {
    long long r = grp * kStatic + (lane & (kStatic - 1));
    r = r < 0 ? 0 : (r < nrows ? r : nrows - 1);  // the trailing Predict re-reads the last row
    lo = a.rowptr[r];
    hi = a.rowptr[r + 1];
    lab = a.label[r];
  };
  load_group(0, nx_lo, nx_hi, nx_lab);
  load_group(1, nn_lo, nn_hi, nn_lab);
  const long long ngroups = (total + kStatic - 1) / kStatic;
  for (long long grp = -1; grp < ngroups; ++grp) {
#pragma unroll
    for (int u = 0; u < kStatic; ++u) {
      const long long t = grp * kStatic + u;  // rows before 0 and past the end have length 0
      const bool tail = t == nrows;
      const uint32_t len = s_len[u];
      float sw = 0.0f, so = 0.0f;
      float zj[EPL], nj[EPL], wj[EPL];
      double sn[EPL];  // sqrt(n) of the coordinate before this row's update: used by the weight and by sigma
      asm volatile("" ::: "memory");
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const uint32_t k = lane + 64 * e;
        const uint32_t d = tail ? ((1u << 16) | k) : s_dep[u][e];
        const uint32_t back = d >> 16, pos = d & 0xFFFFu;
        const bool fwd = k < len && back >= 1 && back < (uint32_t)kRing;
        const uint32_t slot = ((uint32_t)u - back) & (kRing - 1);  // kStatic == kRing: t mod kRing == u
        const float rz = ring_z[slot][fwd ? pos : 0], rn = ring_n[slot][fwd ? pos : 0];
        zj[e] = fwd ? rz : g_z[u % kAhead][e];
        nj[e] = fwd ? rn : g_n[u % kAhead][e];
        sn[e] = sqrt((double)nj[e]);
        wj[e] = seq_weight_sn(zj[e], sn[e], g_r[u % kAhead][e], a.alpha, a.beta, a.l1, a.l2);
        if (k < len) w[s_col[u][e]] = wj[e];
        sw += k < len ? wj[e] * s_val[u][e] : 0.0f;
        so += k < len ? g_o[u % kAhead][e] * s_val[u][e] : 0.0f;
      }
      const float p = seq_sigmoid(wave_sum_dpp(sw) + wave_sum_dpp(so));
      const int y = s_lab[u] > 0 ? 1 : 0;
      const float g = p - y;
      const float g2 = g * g;
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const uint32_t k = lane + 64 * e;
        const double sigma = (sqrt((double)(nj[e] + g2)) - sn[e]) / (double)a.alpha;
        const float zn = (float)((double)zj[e] + ((double)g - sigma * (double)wj[e]));
        const float nn = nj[e] + g2;
        if (k < len && !tail) {
          const uint32_t j = s_col[u][e];
          z[j] = zn;
          n[j] = nn;
          ring_z[u][k] = zn;
          ring_n[u][k] = nn;
        }
      }
      // keep the order: this row's stores, then the gather for row t + kAhead, then the static
      // fetch for row t + kStatic
      asm volatile("" ::: "memory");
      {
        const int su = (u + kAhead) % kStatic;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          const uint32_t j = s_col[su][e];
          g_z[u % kAhead][e] = z[j];
          g_n[u % kAhead][e] = n[j];
          g_r[u % kAhead][e] = a.reg[j];
          g_o[u % kAhead][e] = a.offset[j];
        }
      }
      {
        const long long tn = t + kStatic;
        // lane u of the group's registers, read with v_readlane (scalar results, no LDS trip)
        auto lane64 = [](unsigned long long v, int src) {
          const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
          const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
          return ((unsigned long long)hi << 32) | lo;
        };
        const unsigned long long b = lane64(nx_lo, u);
        const unsigned long long e1 = lane64(nx_hi, u);
        s_lab[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nx_lab), u));
        s_len[u] = tn < total ? (uint32_t)(e1 - b) : 0u;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
          const uint32_t k = lane + 64 * e;
          const bool on = k < s_len[u];
          const uint64_t at = on ? b + k : last_entry;
          // raw values; lanes past the row's end are masked where the values are used, so that
          // nothing here waits for these loads
          s_col[u][e] = col[at];
          s_val[u][e] = HASVAL ? a.val[at] : 1.0f;
          s_dep[u][e] = dep[at];
        }
      }
    }
    nx_lo = nn_lo;
    nx_hi = nn_hi;
    nx_lab = nn_lab;
    load_group(grp + 3, nn_lo, nn_hi, nn_lab);
  }
}

template <int EPL, bool HASVAL>
__global__ void __launch_bounds__(64)
    k_ftrl_seq_ring(SeqArgs a, const uint32_t *__restrict__ col, const uint32_t *__restrict__ dep, uint64_t nnz) {
  ftrl_ring_body<EPL, HASVAL>(a, col, dep, nnz);
}

// Many independent passes in one launch, one single-wave workgroup each: the reference runs one
// worker per process (and per core); on the GPU the workers of a job are resident together and
// their passes run side by side.
struct MultiEntry {
  SeqArgs a;
  const uint32_t *col, *dep;
  uint64_t nnz;
  int epl, hasval;
};

__global__ void __launch_bounds__(64) k_ftrl_seq_multi(const MultiEntry *__restrict__ entries) {
  const MultiEntry m = entries[blockIdx.x];
  if (m.epl == 1) {
    if (m.hasval) ftrl_ring_body<1, true>(m.a, m.col, m.dep, m.nnz);
    else ftrl_ring_body<1, false>(m.a, m.col, m.dep, m.nnz);
  } else {
    if (m.hasval) ftrl_ring_body<2, true>(m.a, m.col, m.dep, m.nnz);
    else ftrl_ring_body<2, false>(m.a, m.col, m.dep, m.nnz);
  }
}

static int ensure_dep(admm_csr *csr) { return build_dep(csr); }

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
  if (!csr->has_dup_cols) ADMM_TRY(ensure_dep(csr));
  KernelTimer t(ctx, ADMM_K_FTRL_SEQ);
  if (csr->has_dup_cols)
    hipLaunchKernelGGL(k_ftrl_seq_serial, dim3(1), dim3(64), 0, ctx->stream, a);
  else if (csr->d_dep && csr->nnz && csr->max_row_len <= 128) {
    using Kern = void (*)(SeqArgs, const uint32_t *, const uint32_t *, uint64_t);
    static const Kern table[2][2] = {{k_ftrl_seq_ring<1, false>, k_ftrl_seq_ring<1, true>},
                                     {k_ftrl_seq_ring<2, false>, k_ftrl_seq_ring<2, true>}};
    const Kern kern = table[csr->max_row_len > 64][csr->d_val != nullptr];
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, ctx->stream, a, csr->d_seqcol ? csr->d_seqcol : csr->d_col,
                       csr->d_dep, csr->nnz);
  }
  else
    hipLaunchKernelGGL(k_ftrl_seq, dim3(1), dim3(64), 0, ctx->stream, a);
  ADMM_HIP_CHECK(hipGetLastError());
  return ADMM_OK;
}

int ftrl_pass_multi(admm_csr *const *sets, float *const *w, float *const *z, float *const *n,
                    const float *const *offset, const float *const *reg_offset, int count, float alpha, float beta,
                    float l1, float l2) {
  if (count <= 0) return ADMM_OK;
  admm_ctx *ctx = sets[0]->ctx;
  std::vector<MultiEntry> entries;
  std::vector<int> single;  // sets the pipelined kernel does not take: repeated columns, very long rows
  for (int k = 0; k < count; ++k) {
    admm_csr *csr = sets[k];
    ADMM_REQUIRE(csr->ctx == ctx, "admm_ftrl_pass_multi: sample set %d belongs to another context", k);
    if (csr->nrows == 0) continue;
    if (!csr->has_dup_cols) ADMM_TRY(ensure_dep(csr));
    if (csr->has_dup_cols || !csr->d_dep || csr->nnz == 0 || csr->max_row_len > 128) {
      single.push_back(k);
      continue;
    }
    MultiEntry m{};
    m.a.rowptr = csr->d_rowptr;
    m.a.col = csr->d_col;
    m.a.perm = csr->d_perm;
    m.a.val = csr->d_val;
    m.a.label = csr->d_label;
    m.a.nrows = csr->nrows;
    m.a.w = w[k];
    m.a.z = z[k];
    m.a.n = n[k];
    m.a.offset = offset[k];
    m.a.reg = reg_offset[k];
    m.a.alpha = alpha;
    m.a.beta = beta;
    m.a.l1 = l1;
    m.a.l2 = l2;
    m.col = csr->d_seqcol ? csr->d_seqcol : csr->d_col;
    m.dep = csr->d_dep;
    m.nnz = csr->nnz;
    m.epl = csr->max_row_len > 64 ? 2 : 1;
    m.hasval = csr->d_val != nullptr;
    entries.push_back(m);
  }
  if (!entries.empty()) {
    const size_t bytes = sizeof(MultiEntry) * entries.size();
    if (ctx->multi_bytes < bytes) {
      ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // an earlier launch may still read the old buffer
      if (ctx->d_multi) hipFree(ctx->d_multi);
      ctx->d_multi = nullptr;
      ctx->multi_bytes = 0;
      ADMM_HIP_CHECK(hipMalloc(&ctx->d_multi, bytes));
      ctx->multi_bytes = bytes;
    }
    // stream-ordered: an earlier launch on this stream has finished with the buffer before the copy
    ADMM_HIP_CHECK(hipMemcpyAsync(ctx->d_multi, entries.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
    ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // `entries` is pageable host memory
    KernelTimer t(ctx, ADMM_K_FTRL_SEQ);
    hipLaunchKernelGGL(k_ftrl_seq_multi, dim3((unsigned)entries.size()), dim3(64), 0, ctx->stream,
                       static_cast<const MultiEntry *>(ctx->d_multi));
    ADMM_HIP_CHECK(hipGetLastError());
  }
  for (int k : single)
    ADMM_TRY(ftrl_pass_sequential(sets[k], w[k], z[k], n[k], offset[k], reg_offset[k], alpha, beta, l1, l2));
  return ADMM_OK;
}

}  // namespace admm_impl

// ---- one row at a time (FtrlSolver::Predict / Update as calls of their own) -----------------------

namespace admm_impl {

struct RowArgs {
  const uint32_t *idx;
  const float *val;
  uint32_t len;
  float *w, *z, *n;
  const float *offset, *reg;
  float alpha, beta, l1, l2, label, predict;
  float *p_out;
};

// src/ftrl.cc:57-72, one lane, element by element as there
__global__ void __launch_bounds__(64) k_ftrl_row_predict(RowArgs a) {
  if (threadIdx.x != 0) return;
  for (uint32_t k = 0; k < a.len; ++k) {
    const uint32_t j = a.idx[k];
    a.w[j] = seq_weight(a.z[j], a.n[j], a.reg[j], a.alpha, a.beta, a.l1, a.l2);
  }
  float sw = 0.0f, so = 0.0f;
  for (uint32_t k = 0; k < a.len; ++k) {
    const uint32_t j = a.idx[k];
    const float v = a.val ? a.val[k] : 1.0f;
    sw += a.w[j] * v;
    so += a.offset[j] * v;
  }
  *a.p_out = seq_sigmoid(sw + so);
}

// src/ftrl.cc:76-87
__global__ void __launch_bounds__(64) k_ftrl_row_update(RowArgs a) {
  if (threadIdx.x != 0) return;
  const int y = a.label > 0 ? 1 : 0;
  const float g = a.predict - y;
  for (uint32_t k = 0; k < a.len; ++k) {
    const uint32_t j = a.idx[k];
    const double sigma = (sqrt((double)(a.n[j] + g * g)) - sqrt((double)a.n[j])) / (double)a.alpha;
    a.z[j] = (float)((double)a.z[j] + ((double)g - sigma * (double)a.w[j]));
    a.n[j] = a.n[j] + g * g;
  }
}

// the row's entries in the context's scratch: [p_out | idx[len] | val[len]]
static int stage_row(admm_ctx *ctx, const uint32_t *index, const float *value, uint32_t length, RowArgs *a) {
  const size_t need = 256 + (sizeof(uint32_t) + sizeof(float)) * (size_t)length;
  if (ctx->row_ws_bytes < need) {
    ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->row_ws) ADMM_HIP_CHECK(hipFree(ctx->row_ws));
    ctx->row_ws = nullptr;
    ctx->row_ws_bytes = 0;
    ADMM_HIP_CHECK(hipMalloc(&ctx->row_ws, need * 2));
    ctx->row_ws_bytes = need * 2;
  }
  char *base = static_cast<char *>(ctx->row_ws);
  a->p_out = reinterpret_cast<float *>(base);
  uint32_t *didx = reinterpret_cast<uint32_t *>(base + 256);
  float *dval = reinterpret_cast<float *>(base + 256 + sizeof(uint32_t) * (size_t)length);
  if (length) {
    ADMM_HIP_CHECK(hipMemcpyAsync(didx, index, sizeof(uint32_t) * length, hipMemcpyHostToDevice, ctx->stream));
    if (value) ADMM_HIP_CHECK(hipMemcpyAsync(dval, value, sizeof(float) * length, hipMemcpyHostToDevice, ctx->stream));
  }
  a->idx = didx;
  a->val = value ? dval : nullptr;
  a->len = length;
  return ADMM_OK;
}

}  // namespace admm_impl

extern "C" int admm_ftrl_predict(admm_ctx *ctx, const uint32_t *index, const float *value, uint32_t length, float *w,
                                 const float *z, const float *n, const float *offset, const float *reg_offset,
                                 float alpha, float beta, float l1, float l2, float *p_out) {
  ADMM_REQUIRE(ctx && w && z && n && offset && reg_offset && p_out && (index || length == 0),
               "admm_ftrl_predict: NULL argument");
  ADMM_HIP_CHECK(hipSetDevice(ctx->device));
  admm_impl::RowArgs a{};
  ADMM_TRY(admm_impl::stage_row(ctx, index, value, length, &a));
  a.w = w;
  a.z = const_cast<float *>(z);
  a.n = const_cast<float *>(n);
  a.offset = offset;
  a.reg = reg_offset;
  a.alpha = alpha;
  a.beta = beta;
  a.l1 = l1;
  a.l2 = l2;
  hipLaunchKernelGGL(admm_impl::k_ftrl_row_predict, dim3(1), dim3(64), 0, ctx->stream, a);
  ADMM_HIP_CHECK(hipGetLastError());
  ADMM_HIP_CHECK(hipMemcpyAsync(p_out, a.p_out, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // (also: the caller's arrays are free again)
  return ADMM_OK;
}

extern "C" int admm_ftrl_update(admm_ctx *ctx, const uint32_t *index, uint32_t length, float label, float predict,
                                const float *w, float *z, float *n, float alpha) {
  ADMM_REQUIRE(ctx && w && z && n && (index || length == 0), "admm_ftrl_update: NULL argument");
  ADMM_HIP_CHECK(hipSetDevice(ctx->device));
  admm_impl::RowArgs a{};
  ADMM_TRY(admm_impl::stage_row(ctx, index, nullptr, length, &a));
  a.w = const_cast<float *>(w);
  a.z = z;
  a.n = n;
  a.alpha = alpha;
  a.label = label;
  a.predict = predict;
  hipLaunchKernelGGL(admm_impl::k_ftrl_row_update, dim3(1), dim3(64), 0, ctx->stream, a);
  ADMM_HIP_CHECK(hipGetLastError());
  ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ADMM_OK;
}
