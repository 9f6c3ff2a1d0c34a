// Transposed sweep X^T r: k_transposed over the column-block cells (with the mini-batch FTRL pass's fused update)
// and k_transposed_cs over the column stream.
#include "internal.h"
#include "sweep_common.h"

#include <algorithm>
#include <cmath>

using namespace admm_impl;

namespace admm_impl {

// ---------------------------------------------------------------------------------------------
// transposed sweep
// ---------------------------------------------------------------------------------------------

// Integer LDS atomics run three times as fast as float ones on gfx950 and make the sum
// independent of the order of the adds.
template <bool FTRL>
__device__ inline void t_accumulate(fx_t *acc, fx_t *acc2, uint32_t c, float x, float v, float scale) {
  if (FTRL) {
    // the low bit of every squared-gradient addend is set, so a column's sum is non-zero exactly
    // when some row of the batch stored the column (a gradient can be 0.0f: sigmoid(35) rounds to 1)
    atomicAdd(&acc[c], to_fx(x, scale));
    atomicAdd(&acc2[c], to_fx(x * x, scale) | 1ull);
  } else {
    atomicAdd(&acc[c], to_fx(x * v, scale));
  }
}

// One workgroup per (panel, column block) cell.  The cell's entries are streamed with 16-byte
// loads (four entries per lane per load; one entry per lane per load, so that adjacent lanes gather
// adjacent rows, was 12 % slower); its per-column sums are accumulated with LDS atomics
// while the per-row vector is gathered almost sequentially (a cell keeps its rows nearly in
// order); the sums are then added to the global per-column vector in one coalesced atomic pass.
// Workgroups whose index is equal modulo 8 share an XCD under round-robin dispatch; they are
// given cells of the same panel so that the panel's slice of the per-row vector stays in that
// XCD's L2 (placement affects speed only).
template <bool FTRL, int kTThreads>
__global__ void __launch_bounds__(kTThreads) k_transposed(TArgs a) {
  extern __shared__ fx_t acc[];  // [CB] (+ [CB] squared-gradient sums in FTRL mode)
  // eight panels or more: workgroups of one XCD share a panel; fewer (the per-batch launches of the
  // mini-batch FTRL pass sweep one panel): cells are dealt to the workgroups in order, over all XCDs
  const bool grouped = a.npanels >= 8;
  const uint64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const uint64_t plocal = grouped ? (slot / a.nblocks) * 8 + xcd : blockIdx.x / a.nblocks;
  if (plocal >= a.npanels) return;
  const uint32_t blk = (uint32_t)((grouped ? slot : (uint64_t)blockIdx.x) % a.nblocks);
  const uint64_t cell = (a.panel0 + plocal) * a.nblocks + blk;
  const uint64_t beg = a.tptr[cell], end = a.tptr[cell + 1];
  const bool fused = FTRL && a.fz != nullptr;
  if (beg == end && !fused) return;  // (fused: the cell's columns may still have sums waiting from the row masks)
  const uint32_t CB = a.col_block;
  fx_t *acc2 = acc + CB;
  const float scale = a.scale;
  // head: up to 3 entries before the first 4-aligned index; tail: what is left after the body
  const uint64_t body0 = (beg + 3) & ~3ull;
  const uint64_t body1 = end & ~3ull;
  // (mini-batch form, where a cell is a few thousand entries: each lane's first two groups of entries are
  // asked for before the accumulators are cleared, one memory latency less on the way to the gathers)
  u32x4 pra = {}, prb = {};
  u32x2 pca = {}, pcb = {};
  const bool pre_body = FTRL && body0 < body1 && body0 + (uint64_t)threadIdx.x * 4 + 4 * kTThreads < body1;
  if (pre_body) {
    const uint64_t k = body0 + (uint64_t)threadIdx.x * 4, k2 = k + 4 * kTThreads;
    pra = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.trow + k));
    prb = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.trow + k2));
    pca = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(a.tcol + k));
    pcb = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(a.tcol + k2));
  }
  for (uint32_t c = threadIdx.x; c < CB; c += kTThreads) {
    acc[c] = 0;
    if (FTRL) acc2[c] = 0;
  }
  if (fused && threadIdx.x == 0) *reinterpret_cast<uint32_t *>(reinterpret_cast<uint16_t *>(acc2 + CB) + CB) = 0;  // the touched columns' count
  __syncthreads();
  if (body0 >= body1) {
    for (uint64_t k = beg + threadIdx.x; k < end; k += kTThreads)
      t_accumulate<FTRL>(acc, acc2, a.tcol[k], a.rvec[a.trow[k]], a.tval ? a.tval[k] : 1.0f, scale);
  } else {
    if (threadIdx.x < body0 - beg) {
      const uint64_t k = beg + threadIdx.x;
      t_accumulate<FTRL>(acc, acc2, a.tcol[k], a.rvec[a.trow[k]], a.tval ? a.tval[k] : 1.0f, scale);
    }
    if (threadIdx.x < end - body1) {
      const uint64_t k = body1 + threadIdx.x;
      t_accumulate<FTRL>(acc, acc2, a.tcol[k], a.rvec[a.trow[k]], a.tval ? a.tval[k] : 1.0f, scale);
    }
    const bool use_val = !FTRL && a.tval != nullptr;
    uint64_t k = body0 + (uint64_t)threadIdx.x * 4;
    // two independent 4-entry groups per trip
    for (; k + 4 * kTThreads < body1; k += 8 * kTThreads) {
      const uint64_t k2 = k + 4 * kTThreads;
      u32x4 ra, rb;
      u32x2 ca, cb;
      if (pre_body && k == body0 + (uint64_t)threadIdx.x * 4) {
        ra = pra, rb = prb, ca = pca, cb = pcb;
      } else {
        ra = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.trow + k));
        rb = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.trow + k2));
        ca = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(a.tcol + k));
        cb = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(a.tcol + k2));
      }
      f32x4 va = {1.f, 1.f, 1.f, 1.f}, vb = va;
      if (use_val) {
        va = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a.tval + k));
        vb = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a.tval + k2));
      }
      float xa0, xa1, xa2, xa3, xb0, xb1, xb2, xb3;
      if (a.no_gather & 1) {
        xa0 = __uint_as_float(ra.x), xa1 = __uint_as_float(ra.y), xa2 = __uint_as_float(ra.z), xa3 = __uint_as_float(ra.w);
        xb0 = __uint_as_float(rb.x), xb1 = __uint_as_float(rb.y), xb2 = __uint_as_float(rb.z), xb3 = __uint_as_float(rb.w);
      } else {
        xa0 = a.rvec[ra.x], xa1 = a.rvec[ra.y], xa2 = a.rvec[ra.z], xa3 = a.rvec[ra.w];
        xb0 = a.rvec[rb.x], xb1 = a.rvec[rb.y], xb2 = a.rvec[rb.z], xb3 = a.rvec[rb.w];
      }
      t_accumulate<FTRL>(acc, acc2, ca.x & 0xFFFFu, xa0, va.x, scale);
      t_accumulate<FTRL>(acc, acc2, ca.x >> 16, xa1, va.y, scale);
      t_accumulate<FTRL>(acc, acc2, ca.y & 0xFFFFu, xa2, va.z, scale);
      t_accumulate<FTRL>(acc, acc2, ca.y >> 16, xa3, va.w, scale);
      t_accumulate<FTRL>(acc, acc2, cb.x & 0xFFFFu, xb0, vb.x, scale);
      t_accumulate<FTRL>(acc, acc2, cb.x >> 16, xb1, vb.y, scale);
      t_accumulate<FTRL>(acc, acc2, cb.y & 0xFFFFu, xb2, vb.z, scale);
      t_accumulate<FTRL>(acc, acc2, cb.y >> 16, xb3, vb.w, scale);
    }
    for (; k < body1; k += 4 * kTThreads) {
      const u32x4 ra = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.trow + k));
      const u32x2 ca = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(a.tcol + k));
      f32x4 va = {1.f, 1.f, 1.f, 1.f};
      if (use_val) va = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a.tval + k));
      const float xa0 = a.rvec[ra.x], xa1 = a.rvec[ra.y], xa2 = a.rvec[ra.z], xa3 = a.rvec[ra.w];
      t_accumulate<FTRL>(acc, acc2, ca.x & 0xFFFFu, xa0, va.x, scale);
      t_accumulate<FTRL>(acc, acc2, ca.x >> 16, xa1, va.y, scale);
      t_accumulate<FTRL>(acc, acc2, ca.y & 0xFFFFu, xa2, va.z, scale);
      t_accumulate<FTRL>(acc, acc2, ca.y >> 16, xa3, va.w, scale);
    }
  }
  __syncthreads();
  const uint64_t gbase = (uint64_t)blk * CB;
  if (fused) {
    // the batch's update of this cell's columns (src/ftrl.cc:76-88 summed over the batch's rows):
    // w <- the lazy weight at the batch's start, n' = n + G2, z += G1 - w (sqrt(n') - sqrt(n)) / alpha;
    // then the column's entry of the forward sweep's vector is brought up to date
    // Two steps: every lane looks at its columns' sums (adding what the forward sweep left for a mask column)
    // and the columns the batch touched are listed in LDS; then the list is worked off by full waves -- the
    // update is double-precision arithmetic (two square roots, three divisions), and a batch touches a
    // fraction of a cell's columns only.
    uint16_t *tl = reinterpret_cast<uint16_t *>(acc2 + CB);
    uint32_t *tcnt = reinterpret_cast<uint32_t *>(tl + CB);
    const int lane = threadIdx.x & 63;
    for (uint32_t c0 = 0; c0 < CB; c0 += kTThreads) {  // (uniform trip count: the ballot below wants whole waves)
      const uint32_t c = c0 + threadIdx.x;
      const uint64_t i = gbase + c;
      bool touched = false;
      if (c < CB && i < a.fdim) {
        long long q2 = (long long)acc2[c];
        const uint32_t sid = a.finv ? a.finv[i] : (uint32_t)i;
        if (a.fmpart && sid < kMaskCols) {
          long long q1 = (long long)acc[c];
#pragma unroll
          for (int b = 0; b < kMaskSlots; ++b) {
            long long *slot = a.fmpart + (size_t)b * 2 * kMaskCols;
            q1 += slot[sid];
            q2 += slot[kMaskCols + sid];
            slot[sid] = 0;
            slot[kMaskCols + sid] = 0;
          }
          acc[c] = (fx_t)q1;
          acc2[c] = (fx_t)q2;
        }
        touched = q2 != 0 || (a.fbias_blocks && i == 0);
      }
      const unsigned long long vote = __ballot(touched);
      if (vote) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(tcnt, (uint32_t)__popcll(vote));
        base = __shfl(base, 0, 64);
        if (touched) tl[base + __popcll(vote & ((1ull << lane) - 1))] = (uint16_t)c;
      }
    }
    __syncthreads();
    const uint32_t ntouched = *tcnt;
    float vmax = 0.0f;
    for (uint32_t k = threadIdx.x; k < ntouched; k += kTThreads) {
      const uint32_t c = tl[k];
      const uint64_t i = gbase + c;
      const long long q1 = (long long)acc[c], q2 = (long long)acc2[c];
      const uint32_t sid = a.finv ? a.finv[i] : (uint32_t)i;
      double G1 = 0, G2 = 0;
      if (q2 != 0) {
        G1 = (double)q1 * a.finv_scale;
        G2 = (double)q2 * a.finv_scale;
      }
      if (a.fbias_blocks && i == 0) {
        for (int b = 0; b < a.fbias_blocks; ++b) {
          G1 += a.fpartials[0 * kMaxPartialBlocks + b];
          G2 += a.fpartials[1 * kMaxPartialBlocks + b];
        }
      }
      const float nold = a.fn[i], zold = a.fz[i], reg = a.freg[i];
      const float wc = ftrl_weight(zold, nold, reg, a.falpha, a.fbeta, a.fl1, a.fl2);
      const float nnew = (float)((double)nold + G2);
      const double sigma = (sqrt((double)nnew) - sqrt((double)nold)) / (double)a.falpha;
      const float znew = (float)((double)zold + (G1 - sigma * (double)wc));
      a.fw[i] = wc;
      a.fz[i] = znew;
      a.fn[i] = nnew;
      const float v = ftrl_weight(znew, nnew, reg, a.falpha, a.fbeta, a.fl1, a.fl2) + a.foff[i];
      a.fwsum[sid] = v;
      vmax = fmaxf(vmax, fabsf(v));
    }
    // the block's running maximum (no other workgroup writes it in this launch)
    for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, off, 64));
    __syncthreads();  // the accumulators are read: their first words now carry the waves' maxima
    float *wmax = reinterpret_cast<float *>(acc);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = a.fabsmax[blk];
      for (int w8 = 0; w8 < kTThreads / 64; ++w8) m = fmaxf(m, wmax[w8]);
      a.fabsmax[blk] = m;
    }
    return;
  }
  if (a.no_gather & 2) return;
  for (uint32_t c = threadIdx.x; c < CB; c += kTThreads) {
    const fx_t s = acc[c];
    if (FTRL) {
      const fx_t s2 = acc2[c];
      if (s2 != 0) {
        atomicAdd(&a.g1[gbase + c], s);
        atomicAdd(&a.g2[gbase + c], s2);
      }
    } else if (s != 0) {
      atomicAdd(&a.g1[gbase + c], s);
    }
  }
}

// Column stream: the part of X^T r that belongs to the most frequent columns (those outside the row
// masks).  A workgroup takes panels of kTcRows rows; the panel's slice of the per-row vector sits in
// LDS, already in fixed point; every lane walks its run of the panel's column-ordered words, sums a
// column's table entries in a register (exact: integer-valued doubles below 2^51) and adds the sum
// to the column's LDS accumulator when the column ends -- no memory gather and no atomic per entry.
// The accumulators go to the global vector once, when the workgroup has swept its panels.
struct TcArgs {
  const uint16_t *tw;
  const float *tv;
  const uint64_t *tbase;
  const uint16_t *tfirst;
  const float *rvec;
  uint64_t nrows;
  uint32_t ncols;
  const uint32_t *perm;  // stored id -> index in g1 (nullptr: identity)
  unsigned long long *g1;
  unsigned long long *gs;  // when given: the sums go to gs[stored id] (adjacent words) instead of g1[perm[..]]
  float scale;
};

template <bool HASVAL>
__global__ void __launch_bounds__(kTcLanes) k_transposed_cs(TcArgs a) {
  extern __shared__ fx_t cs_acc[];  // [ncols] column sums, then float rq[2][kTcRows + 1]
  float *rq = reinterpret_cast<float *>(cs_acc + a.ncols);
  const int tid = threadIdx.x;
  const uint64_t npanels = (a.nrows + kTcRows - 1) / kTcRows;
  const float scale = a.scale;
  constexpr int kPer = kTcRows / kTcLanes;  // rows of a slice each lane loads
  for (uint32_t c = tid; c < a.ncols; c += kTcLanes) cs_acc[c] = 0;
  if (tid < 2) rq[tid * (kTcRows + 1) + kTcRows] = 0.0f;  // the filler words' slot
  float nxt[kPer];
  auto fetch = [&](uint64_t p) {
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const uint64_t r = p * kTcRows + (uint64_t)j * kTcLanes + tid;
      const float x = r < a.nrows ? a.rvec[r] : 0.0f;
      nxt[j] = HASVAL ? x : truncf(x * scale);
    }
  };
  auto put = [&](int buf) {
#pragma unroll
    for (int j = 0; j < kPer; ++j) rq[buf * (kTcRows + 1) + j * kTcLanes + tid] = nxt[j];
  };
  int buf = 0;
  if (blockIdx.x < npanels) {
    fetch(blockIdx.x);
    put(0);
  }
  for (uint64_t p = blockIdx.x; p < npanels; p += gridDim.x) {
    __syncthreads();  // slice `buf` is complete, and nobody reads the other one any more
    const bool more = p + gridDim.x < npanels;
    if (more) fetch(p + gridDim.x);
    const float *tab = rq + buf * (kTcRows + 1);
    const uint64_t tb = a.tbase[p];
    const uint32_t nq = (uint32_t)((a.tbase[p + 1] - tb) / (4 * kTcLanes));
    uint32_t col = a.tfirst[p * kTcLanes + tid];
    const u32x2 *src = reinterpret_cast<const u32x2 *>(a.tw + tb) + tid;
    const f32x4 *vsrc = HASVAL ? reinterpret_cast<const f32x4 *>(a.tv + tb) + tid : nullptr;
    double sum = 0.0;
    auto step = [&](uint32_t e, float v) {
      float x = tab[e & 0x3FFFu];
      if (HASVAL) x = truncf((x * v) * scale);
      sum += (double)x;
      if (e & 0x8000u) {
        if (sum != 0.0) atomicAdd(&cs_acc[col], rp_d2fx(sum));
        sum = 0.0;
        ++col;
      }
    };
    uint32_t q = 0;
    // four groups of four words per trip, the next trip's loads issued before this trip's work: a wave
    // otherwise has nothing in flight while it adds
    constexpr int kAhead = 4;
    u32x2 nx[kAhead];
    f32x4 nv[kAhead];
    auto fetch = [&](uint32_t at) {
#pragma unroll
      for (int j = 0; j < kAhead; ++j) {
        nx[j] = __builtin_nontemporal_load(src + (uint64_t)(at + j) * kTcLanes);
        if (HASVAL) nv[j] = __builtin_nontemporal_load(vsrc + (uint64_t)(at + j) * kTcLanes);
      }
    };
    if (nq >= kAhead) fetch(0);
    for (; q + kAhead <= nq; q += kAhead) {
      u32x2 cx[kAhead];
      f32x4 cv[kAhead];
#pragma unroll
      for (int j = 0; j < kAhead; ++j) {
        cx[j] = nx[j];
        cv[j] = HASVAL ? nv[j] : f32x4{1.f, 1.f, 1.f, 1.f};
      }
      if (q + 2 * kAhead <= nq) fetch(q + kAhead);
#pragma unroll
      for (int j = 0; j < kAhead; ++j) {
        step(cx[j].x & 0xFFFFu, cv[j].x);
        step(cx[j].x >> 16, cv[j].y);
        step(cx[j].y & 0xFFFFu, cv[j].z);
        step(cx[j].y >> 16, cv[j].w);
      }
    }
    for (; q < nq; ++q) {
      const u32x2 c0 = __builtin_nontemporal_load(src + (uint64_t)q * kTcLanes);
      f32x4 v0 = {1.f, 1.f, 1.f, 1.f};
      if (HASVAL) v0 = __builtin_nontemporal_load(vsrc + (uint64_t)q * kTcLanes);
      step(c0.x & 0xFFFFu, v0.x);
      step(c0.x >> 16, v0.y);
      step(c0.y & 0xFFFFu, v0.z);
      step(c0.y >> 16, v0.w);
    }
    if (sum != 0.0) atomicAdd(&cs_acc[col], rp_d2fx(sum));
    if (more) put(buf ^ 1);
    buf ^= 1;
  }
  __syncthreads();
  // every workgroup adds its sums to the same ncols global words: each starts at another column, so
  // that they do not queue up at one word at a time
  const uint32_t rot = (uint32_t)(((uint64_t)blockIdx.x * 64) % a.ncols);
  for (uint32_t j = tid; j < a.ncols; j += kTcLanes) {
    const uint32_t c = j + rot < a.ncols ? j + rot : j + rot - a.ncols;
    const fx_t v = cs_acc[c];
    if (v != 0) atomicAdd(a.gs ? &a.gs[c] : &a.g1[a.perm ? a.perm[c] : c], v);
  }
}

static int launch_column_stream(admm_csr *csr) {
  admm_ctx *ctx = csr->ctx;
  TcArgs a{};
  a.tw = csr->d_tcw;
  a.tv = csr->d_tcv;
  a.tbase = csr->d_tcbase;
  a.tfirst = csr->d_tcfirst;
  a.rvec = csr->d_rowbuf;
  a.nrows = csr->nrows;
  a.ncols = csr->tc_cols;
  a.perm = csr->d_perm;
  a.g1 = reinterpret_cast<unsigned long long *>(csr->d_g1);
  a.gs = reinterpret_cast<unsigned long long *>(csr->d_gs);
  a.scale = std::ldexp(1.0f, csr->fx_shift);
  const size_t lds = sizeof(fx_t) * csr->tc_cols + sizeof(float) * 2 * (kTcRows + 1);
  static bool raised[64] = {};
  if (!raised[ctx->device & 63]) {
    for (const void *fn : {reinterpret_cast<const void *>(&k_transposed_cs<true>),
                           reinterpret_cast<const void *>(&k_transposed_cs<false>)})
      ADMM_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    raised[ctx->device & 63] = true;
  }
  const uint64_t npanels = (csr->nrows + kTcRows - 1) / kTcRows;
  const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>(npanels, (uint64_t)ctx->cu_count));
  if (a.tv) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_transposed_cs<true>), dim3(grid), dim3(kTcLanes), lds, ctx->stream, a);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_transposed_cs<false>), dim3(grid), dim3(kTcLanes), lds, ctx->stream, a);
  ADMM_HIP_CHECK(hipGetLastError());
  return ADMM_OK;
}

int launch_transposed(admm_csr *csr, bool ftrl_mode, uint64_t p0, uint64_t p1) {
  return launch_transposed_impl(csr, ftrl_mode, p0, p1, nullptr);
}

// fuse: the f* fields of a mini-batch's fused update (FTRL form, one panel), or nullptr
int launch_transposed_impl(admm_csr *csr, bool ftrl_mode, uint64_t p0, uint64_t p1, const TArgs *fuse) {
  admm_ctx *ctx = csr->ctx;
  ADMM_REQUIRE(csr->has_t, "transposed sweep: call admm_csr_build_transposed first");
  if (p0 >= p1) return ADMM_OK;
  ADMM_REQUIRE(!csr->has_tc || (!ftrl_mode && p0 == 0 && p1 == csr->npanels),
               "transposed sweep: a copy with a column stream serves whole-set sweeps only");
  TArgs a{};
  if (fuse) {
    ADMM_REQUIRE(ftrl_mode && p1 == p0 + 1, "transposed sweep: the fused update takes one panel in the FTRL form");
    a = *fuse;
  }
  a.tptr = csr->d_tptr;
  a.trow = csr->d_trow;
  a.tcol = csr->d_tcol;
  a.tval = csr->d_tval;
  a.rvec = csr->d_rowbuf;
  a.g1 = reinterpret_cast<unsigned long long *>(csr->d_g1);
  a.g2 = reinterpret_cast<unsigned long long *>(csr->d_g2);
  a.scale = std::ldexp(1.0f, csr->fx_shift);
  a.nblocks = csr->nblocks;
  a.col_block = csr->col_block;
  // measurement aid: a gather-free run moves a known number of bytes, which calibrates the
  // hardware counters; it computes nothing useful
  a.no_gather = (getenv("ADMM_CALIBRATE_NO_GATHER") != nullptr ? 1 : 0) | (getenv("ADMM_CALIBRATE_NO_FLUSH") != nullptr ? 2 : 0);
  // (the fused update also lists the touched columns: 2 bytes a column and a counter)
  const size_t lds = sizeof(fx_t) * csr->col_block * (ftrl_mode ? 2 : 1) + (fuse ? 2 * (size_t)csr->col_block + 64 : 0);
  // more than half of the CU's LDS: one workgroup per CU, so give it the full 1024 threads
  const bool big = lds > 72 * 1024;
  static bool raised[64] = {};  // per device
  if (!raised[ctx->device & 63]) {
    // the 512-thread forms run with up to 72 KB, which is above the 64 KB a kernel gets by default
    for (const void *fn : {reinterpret_cast<const void *>(&k_transposed<true, 1024>),
                           reinterpret_cast<const void *>(&k_transposed<false, 1024>),
                           reinterpret_cast<const void *>(&k_transposed<true, 512>),
                           reinterpret_cast<const void *>(&k_transposed<false, 512>)})
      ADMM_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
    raised[ctx->device & 63] = true;
  }
  KernelTimer t(ctx, ADMM_K_TRANSPOSED);
  // keep each launch's grid below 2^30 workgroups
  const uint64_t max_panels = std::max<uint64_t>(8, ((1ull << 30) / csr->nblocks) & ~7ull);
  for (uint64_t q0 = p0; q0 < p1; q0 += max_panels) {
    a.panel0 = q0;
    a.npanels = std::min<uint64_t>(max_panels, p1 - q0);
    const uint64_t groups = (a.npanels + 7) / 8;
    const uint32_t grid = (uint32_t)((a.npanels >= 8 ? groups * 8 : a.npanels) * csr->nblocks);
    if (ftrl_mode && big)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_transposed<true, 1024>), dim3(grid), dim3(1024), lds, ctx->stream, a);
    else if (ftrl_mode)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_transposed<true, 512>), dim3(grid), dim3(512), lds, ctx->stream, a);
    else if (big)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_transposed<false, 1024>), dim3(grid), dim3(1024), lds, ctx->stream, a);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_transposed<false, 512>), dim3(grid), dim3(512), lds, ctx->stream, a);
    ADMM_HIP_CHECK(hipGetLastError());
  }
  if (csr->has_tc) ADMM_TRY(launch_column_stream(csr));
  return ADMM_OK;
}

}  // namespace admm_impl
