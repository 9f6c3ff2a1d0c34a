// Forward sweep over the row-panel copy: k_forward_rp (row masks, hot stream, column-ordered lists), and the
// dispatcher launch_forward.
#include "internal.h"
#include "sweep_common.h"

#include <algorithm>
#include <cmath>

using namespace admm_impl;

namespace admm_impl {

// ---------------------------------------------------------------------------------------------
// forward sweep over the row-panel copy
// ---------------------------------------------------------------------------------------------
//
// The row-major sweep above gathers one weight per stored entry from a vector of `dim` floats in
// row order: 64 lanes, 64 different cache lines.  Here a workgroup owns a panel of fp_rows rows and
// keeps their margins in LDS as 64-bit fixed point.  The panel's entries come in two forms:
// * entries whose stored column id is below nhot (the bulk of a frequency-relabelled set) are 16-bit
//   words in row order, cut into one run per lane.  The weights of those columns sit in LDS, already
//   in fixed point; a lane sums a row's table entries in a register and adds to the row's margin
//   once per row -- no memory gather and no atomic per entry.
// * the other entries are stored in column order (up to a bin's width), so 64 consecutive entries
//   gather their weights from a few adjacent lines, and the scattered access -- one add per entry
//   to the margin of the entry's row -- goes to LDS, where an integer atomic costs one lane-clock.
// Integer adds commute: the margins do not depend on the order of the entries, and a sweep is
// bit-reproducible.  The fixed-point shift is chosen per sweep on the device from max |w|
// (k_wsum_absmax leaves it per workgroup), the largest |value| and the longest row, so that no row sum can
// reach 2^50.
//
// Entries in the most frequent kMaskCols columns of a relabelled set are not in the copy: they are
// bits of the row's mask words, summed when the panel's margins are initialised, and their part of
// X^T r is accumulated in the epilogue (as in the row-major sweep).

constexpr int kRpThreads = kHpLanes;

// mini-batch FTRL: a batch is one short panel per workgroup, and a few hundred workgroups adding to the
// same kMaskCols global words at once take turns (9 of the kernel's 32 microseconds); they are spread
// over kMaskSlots copies of those words, which the per-coordinate pass adds up and clears
struct RpArgs {
  const uint16_t *hcol;    // hot stream (ids below nhot; bit 15: last entry of its row)
  const float *hval;
  const uint64_t *hbase;   // per panel: first word of its hot stream
  const uint16_t *hfirst;  // per panel and lane: the row the lane's run starts in
  const uint32_t *fcol;    // the entries outside the hot stream, in column order per panel
  const uint16_t *frow;
  const float *fval;
  const uint64_t *fptr;    // per panel: first of those entries
  const float *label;
  const uint64_t *hmask, *hwarm;  // nullptr: no mask tier
  const uint64_t *hmaskT;  // hmask bit-transposed in tiles of 64 rows
  const uint32_t *perm;    // stored id -> caller's id (relabelled sets), for the column sums' destination
  uint32_t nhot;           // leading entries of w cached in LDS
  const float *w;          // gathered vector in the set's stored column order
  const float *blockmax;   // max |w| per workgroup of the kernel that filled w; the sweep's fixed-point shift follows from it
  int nblockmax, lenbits;  // (lenbits: ceil(log2(longest row + 1)))
  float vmax;              // max |value| of the set
  uint64_t nrows;
  uint32_t panel_rows;
  int add_bias;            // no mask tier, bias column left out of the copy: every margin starts at w[0]
  float *out, *out2;
  double *partials;
  unsigned long long *g1;  // X^T r of the mask columns goes here (fixed point, gscale), nullptr: not wanted
  unsigned long long *g2;  // FTRL form: the sums of the squared gradients of those columns
  unsigned long long *gs;  // when given (not in the FTRL form): the mask columns' sums go to gs[stored id] instead of g1[perm[..]]
  unsigned long long *mpart;  // FTRL form: [kMaskSlots][2][kMaskCols] sums by workgroup index mod kMaskSlots, instead of adds to g1 / g2
  float gscale;
  uint64_t panel0, pcount; // panels [panel0, panel0 + pcount) are swept by this launch
};

#ifndef ADMM_RP_PRELIST
#define ADMM_RP_PRELIST (MODE == FWD_FTRL)
#endif
template <int MODE, bool HASVAL>
__global__ void __launch_bounds__(kRpThreads, MODE == FWD_FTRL ? 2 : 4) k_forward_rp(RpArgs a) {
  extern __shared__ unsigned long long rp_acc[];                     // [panel_rows] margins, fixed point; then float hot[nhot + 1]
  // leading entries of w: as they are (sets with values) or times 2^shift with the fraction dropped
  // (pattern sets: an entry's addend is then the table entry itself); hot[nhot] = 0 for filler words
  float *hot = reinterpret_cast<float *>(rp_acc + a.panel_rows);
  __shared__ float raw[kMaskCols];                                   // weights of the mask columns
  __shared__ float tab[8 * 256];                                     // byte of the first mask word -> sum of its weights
  __shared__ unsigned long long wacc[kMaskCols];                     // X^T r of the mask columns, this workgroup
  __shared__ unsigned long long wacc2[MODE == FWD_FTRL ? kMaskCols : 1];  // FTRL form: squared-gradient sums of those columns
  __shared__ double shred[kRpThreads / 64], shred1[kRpThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool dense = a.hmask != nullptr;
  const bool colsums = dense && (MODE == FWD_LBFGS || MODE == FWD_FTRL) && a.g1 != nullptr;
  const uint32_t R = a.panel_rows;
  // What the workgroup's first panel starts with is asked for before anything else (its pointers, the row
  // each lane's run starts in, the masks of each lane's first row): a launch of short panels, one per
  // workgroup, is otherwise a chain of dependent loads, one memory latency per phase.  Barriers wait for
  // LDS only, so these loads stay in flight through the set-up below.
  const uint64_t p_first = a.panel0 + blockIdx.x;
  uint64_t pre_hb0 = 0, pre_hb1 = 0, pre_f0 = 0, pre_f1 = 0;
  uint32_t pre_row = 0;
  unsigned long long pre_m = 0, pre_w[kWarmWords] = {};
  if (p_first < a.panel0 + a.pcount) {
    pre_hb0 = a.hbase[p_first];
    pre_hb1 = a.hbase[p_first + 1];
    pre_f0 = a.fptr[p_first];
    pre_f1 = a.fptr[p_first + 1];
    pre_row = a.hfirst[p_first * kRpThreads + tid];
    const uint64_t r = p_first * R + tid;
    if (MODE == FWD_FTRL && dense && (uint32_t)tid < R && r < a.nrows) {
      pre_m = a.hmask[r];
#pragma unroll
      for (int w = 0; w < kWarmWords; ++w) pre_w[w] = a.hwarm[r * kWarmWords + w];
    }
  }
  // the fixed-point shift of this sweep: the largest s with (longest row) * max |w v| * 2^s < 2^50, at
  // most 40 -- a margin, and every partial sum of one, then is an integer below 2^51 in magnitude,
  // which a double holds exactly.  Every workgroup derives it from the same maxima.
  {
    float mx = 0.0f;
    for (int i = tid; i < a.nblockmax; i += kRpThreads) mx = fmaxf(mx, a.blockmax[i]);
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off, 64));
    if (lane == 0) shred[wave] = (double)mx;
    __syncthreads();
  }
  int shift;
  {
    float top = 0.0f;
    for (int w8 = 0; w8 < kRpThreads / 64; ++w8) top = fmaxf(top, (float)shred[w8]);
    top *= a.vmax;
    int e = 0;
    if (top > 0.0f && top < 3.0e38f) frexpf(top, &e);  // top < 2^e
    else if (!(top < 3.0e38f)) e = 128;
    shift = 50 - a.lenbits - e;
    shift = shift > 40 ? 40 : (shift < -60 ? -60 : shift);
  }
  const float scale = ldexpf(1.0f, shift);
  const double inv_scale = ldexp(1.0, -shift);
  for (uint32_t j = tid; j < a.nhot; j += kRpThreads) {
    const float x = a.w[j];
    hot[j] = HASVAL ? x : truncf(x * scale);
  }
  if (tid == 0) hot[a.nhot] = 0.0f;
  // the margins start at zero; every epilogue leaves the rows it has read at zero again
  for (uint32_t r = tid; r < R; r += kRpThreads) rp_acc[r] = 0ull;
  if (tid < (int)kMaskCols) {
    wacc[tid] = 0ull;
    if (MODE == FWD_FTRL) wacc2[tid] = 0ull;
    raw[tid] = dense ? a.w[tid] : 0.0f;  // a set with masks has at least kMaskCols columns
  }
  __syncthreads();
  // (short panels: building the byte table would cost more than the lookups it saves; their rows walk the
  // first word's bits instead, byte by byte, which adds the same floats in the same order)
  const bool use_tab = dense && R > 512;
  if (use_tab) {
    for (int i = tid; i < 8 * 256; i += kRpThreads) {
      const int g = i >> 8, byte = i & 255;
      float t = 0.0f;
#pragma unroll
      for (int bit = 0; bit < 8; ++bit)
        if (byte & (1 << bit)) t += raw[8 * g + bit];
      tab[i] = t;
    }
  }
  const float w0 = a.add_bias ? a.w[0] : 0.0f;
  double acc0 = 0, acc1 = 0;
  i32x4 cacc[4] = {};  // column sums of the first mask word, by digit (see the epilogue)
  bool did_tile = false;  // this wave has put rows into cacc
  __syncthreads();
  for (uint64_t p = a.panel0 + blockIdx.x; p < a.panel0 + a.pcount; p += gridDim.x) {
    const uint64_t r0 = p * R;
    const uint32_t nr = (uint32_t)(a.nrows - r0 < R ? a.nrows - r0 : R);
    // ---- margins start at the mask tier's part (or at the bias weight).  A lane takes every
    // kRpThreads-th row; the mask words of four of its rows are loaded before any of them is used
    // (each row's work hangs on its own loads, which would otherwise be waited for one row at a time)
    for (uint32_t rb4 = tid; rb4 < nr; rb4 += 4 * kRpThreads) {
      unsigned long long mm[4] = {0, 0, 0, 0}, ww[4][kWarmWords] = {};
      if (dense) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t r = rb4 + j * kRpThreads;
          if (r < nr) {
            if (MODE == FWD_FTRL && j == 0 && p == p_first && rb4 == (uint32_t)tid) {
              mm[0] = pre_m;
#pragma unroll
              for (int w = 0; w < kWarmWords; ++w) ww[0][w] = pre_w[w];
            } else {
              mm[j] = a.hmask[r0 + r];
#pragma unroll
              for (int w = 0; w < kWarmWords; ++w) ww[j][w] = a.hwarm[(r0 + r) * kWarmWords + w];
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t r = rb4 + j * kRpThreads;
        if (r >= nr) break;
        float m0 = w0;
        if (dense) {
          unsigned long long m = mm[j];
          if (use_tab) {
#pragma unroll
            for (int g = 0; g < 8; ++g) m0 += tab[g * 256 + (int)((m >> (8 * g)) & 255)];
          } else {
            float t = 0.0f;
            int cur = 0;
            while (m) {
              const int b = __builtin_ctzll(m);
              if ((b >> 3) != cur) {
                m0 += t;
                t = 0.0f;
                cur = b >> 3;
              }
              t += raw[b];
              m &= m - 1;
            }
            m0 += t;
          }
#pragma unroll
          for (int w = 0; w < kWarmWords; ++w) {
            unsigned long long mw = ww[j][w];
            while (mw) {
              m0 += raw[64 * (w + 1) + __builtin_ctzll(mw)];
              mw &= mw - 1;
            }
          }
        }
        atomicAdd(&rp_acc[r], rp_to_fx(m0, scale));  // (an add like all the others: no barrier before the streams)
      }
    }
    // the first four list entries of each lane are asked for now: they arrive while the hot stream is walked
    const bool first = p == p_first;
    const uint64_t beg = first ? pre_f0 : a.fptr[p], end = first ? pre_f1 : a.fptr[p + 1];
    uint32_t pc[4] = {0, 0, 0, 0}, pq[4] = {0, 0, 0, 0};
    const bool pre_list = ADMM_RP_PRELIST && beg + tid + 3 * kRpThreads < end;
    if (pre_list) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pc[j] = __builtin_nontemporal_load(a.fcol + beg + tid + j * kRpThreads);
        pq[j] = __builtin_nontemporal_load(a.frow + beg + tid + j * kRpThreads);
      }
    }
    // ---- hot stream: every lane walks its run of the panel's rows, sums a row's table entries in a
    // register (exact: integer-valued doubles below 2^51) and adds the sum to the row's margin when
    // the row ends; a run that ends inside a row leaves its part at the end
    {
      const uint64_t hb = first ? pre_hb0 : a.hbase[p];
      const uint32_t nq = (uint32_t)(((first ? pre_hb1 : a.hbase[p + 1]) - hb) / (4 * kRpThreads));  // groups of four words per lane
      uint32_t row = first ? pre_row : a.hfirst[p * kRpThreads + tid];
      const u32x2 *src = reinterpret_cast<const u32x2 *>(a.hcol + hb) + tid;
      const f32x4 *vsrc = HASVAL ? reinterpret_cast<const f32x4 *>(a.hval + hb) + tid : nullptr;
      double sum = 0.0;
      auto step = [&](uint32_t e, float v) {
        float x = hot[e & 0x7FFFu];
        if (HASVAL) x = truncf((x * v) * scale);
        sum += (double)x;
        if (e & 0x8000u) {
          atomicAdd(&rp_acc[row], rp_d2fx(sum));
          sum = 0.0;
          ++row;
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
          nx[j] = __builtin_nontemporal_load(src + (uint64_t)(at + j) * kRpThreads);
          if (HASVAL) nv[j] = __builtin_nontemporal_load(vsrc + (uint64_t)(at + j) * kRpThreads);
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
        const u32x2 c0 = __builtin_nontemporal_load(src + (uint64_t)q * kRpThreads);
        f32x4 v0 = {1.f, 1.f, 1.f, 1.f};
        if (HASVAL) v0 = __builtin_nontemporal_load(vsrc + (uint64_t)q * kRpThreads);
        step(c0.x & 0xFFFFu, v0.x);
        step(c0.x >> 16, v0.y);
        step(c0.y & 0xFFFFu, v0.z);
        step(c0.y >> 16, v0.w);
      }
      if (sum != 0.0) atomicAdd(&rp_acc[row], rp_d2fx(sum));
    }
    // ---- the other entries: stream, gather, add.  They come in the order of their column ids, so
    // adjacent lanes gather from the same or neighbouring cache lines.
    {
      uint64_t k = beg + tid;
      for (; k + 3 * kRpThreads < end; k += 4 * kRpThreads) {
        uint32_t c0, c1, c2, c3, q0, q1, q2, q3;
        if (pre_list && k == beg + tid) {
          c0 = pc[0], c1 = pc[1], c2 = pc[2], c3 = pc[3];
          q0 = pq[0], q1 = pq[1], q2 = pq[2], q3 = pq[3];
        } else {
          c0 = __builtin_nontemporal_load(a.fcol + k), c1 = __builtin_nontemporal_load(a.fcol + k + kRpThreads);
          c2 = __builtin_nontemporal_load(a.fcol + k + 2 * kRpThreads);
          c3 = __builtin_nontemporal_load(a.fcol + k + 3 * kRpThreads);
          q0 = __builtin_nontemporal_load(a.frow + k), q1 = __builtin_nontemporal_load(a.frow + k + kRpThreads);
          q2 = __builtin_nontemporal_load(a.frow + k + 2 * kRpThreads);
          q3 = __builtin_nontemporal_load(a.frow + k + 3 * kRpThreads);
        }
        float x0 = a.w[c0], x1 = a.w[c1], x2 = a.w[c2], x3 = a.w[c3];
        if (HASVAL) {
          x0 *= __builtin_nontemporal_load(a.fval + k);
          x1 *= __builtin_nontemporal_load(a.fval + k + kRpThreads);
          x2 *= __builtin_nontemporal_load(a.fval + k + 2 * kRpThreads);
          x3 *= __builtin_nontemporal_load(a.fval + k + 3 * kRpThreads);
        }
        atomicAdd(&rp_acc[q0], rp_to_fx(x0, scale));
        atomicAdd(&rp_acc[q1], rp_to_fx(x1, scale));
        atomicAdd(&rp_acc[q2], rp_to_fx(x2, scale));
        atomicAdd(&rp_acc[q3], rp_to_fx(x3, scale));
      }
      for (; k < end; k += kRpThreads) {
        const float x = a.w[a.fcol[k]];
        atomicAdd(&rp_acc[a.frow[k]], rp_to_fx(HASVAL ? x * a.fval[k] : x, scale));
      }
    }
    __syncthreads();
    // ---- per-row epilogue; a wave takes 64 consecutive rows per trip (uniform trip count)
    for (uint32_t rb = wave * 64; rb < nr; rb += kRpThreads) {
      const uint32_t r = rb + lane;
      const bool on = r < nr;
      unsigned long long rv = 0, rv2 = 0;
      if (on) {
        const float m = (float)((double)(long long)rp_acc[r] * inv_scale);
        const float lab = a.label[r0 + r];
        if (MODE == FWD_MARGIN) {
          a.out[r0 + r] = m;
        } else if (MODE == FWD_FTRL) {
          // src/ftrl.cc:72,76-78
          const float pr = (float)(1.0 / (1 + exp(-(double)clamp35(m))));
          const int y = lab > 0 ? 1 : 0;
          const float g = pr - y;
          a.out[r0 + r] = g;
          rv = (unsigned long long)(long long)(g * a.gscale);
          rv2 = (unsigned long long)(long long)(g * g * a.gscale) | 1ull;  // non-zero marks "touched"
          acc0 += (double)g;
          acc1 += (double)(g * g);
        } else if (MODE == FWD_EVAL) {
          // src/metrics.cc:104-105: double predict, -log p / -log(1-p)
          const double pd = 1.0f / (1 + exp(-(double)clamp35(m)));
          a.out[r0 + r] = (float)pd;
          acc0 += lab > 0 ? -log(pd) : -log(1 - pd);
        } else {
          const float t = lab > 0 ? 1.0f : 0.0f;
          const float loss = log1pf(expf(-fabsf(m))) + (t > 0 ? fmaxf(-m, 0.0f) : fmaxf(m, 0.0f));
          const float res = 1.0f / (1.0f + expf(-m)) - t;
          a.out[r0 + r] = res;
          if (a.out2) a.out2[r0 + r] = m;
          rv = (unsigned long long)(long long)(res * a.gscale);
          acc0 += (double)loss;
          acc1 += (double)res;
        }
      }
      if (colsums) {
        did_tile = true;
        // X^T r of the mask columns over these 64 rows.  First word (64 columns, about half of a row's
        // entries): a 64 x 64 bit matrix times the residuals on the matrix unit.  The residuals, offset
        // to be positive, are cut into six 7-bit digits plus a constant 1 (which counts a column's rows,
        // to take the offset out again); D[column][digit] += sum over rows of bit * digit, in i32,
        // exact.  The digits are handed over through the wave's own 64 margins in LDS, which it has
        // read.  Other words: each row's set bits walked into LDS accumulators (the same sums by
        // broadcast and ballot for all four words were slower: 1.90 against 1.64 ms per 10M rows).
        {
          const unsigned long long u = rv + (1ull << 41);  // |rv| <= 2^37
          // (FTRL form: six more digits for the squared gradient, in the same B operand; its thirteen
          // planes have a 1 KB slot per wave behind the weight table)
          unsigned char *pl = MODE == FWD_FTRL
                                  ? reinterpret_cast<unsigned char *>(hot + ((a.nhot + 1 + 3) & ~3u)) + wave * 1024
                                  : reinterpret_cast<unsigned char *>(rp_acc + rb);
#pragma unroll
          for (int n = 0; n < 6; ++n) pl[n * 64 + lane] = (unsigned char)((u >> (7 * n)) & 127u);
          pl[6 * 64 + lane] = 1;
          if (MODE == FWD_FTRL) {
#pragma unroll
            for (int n = 0; n < 6; ++n) pl[(7 + n) * 64 + lane] = (unsigned char)((rv2 >> (7 * n)) & 127u);
          }
          i32x4 bfrag = {0, 0, 0, 0};
          if ((lane & 15) < (MODE == FWD_FTRL ? 13 : 7))
            bfrag = *reinterpret_cast<const i32x4 *>(pl + (lane & 15) * 64 + 16 * (lane >> 4));
          const uint16_t *tt = reinterpret_cast<const uint16_t *>(a.hmaskT + (r0 + rb));
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const uint32_t bits = tt[(16 * m + (lane & 15)) * 4 + (lane >> 4)];  // column 16 m + (lane & 15), rows 16 (lane >> 4) ..
            i32x4 afrag;
            afrag.x = (int)((((bits >> 0) & 15u) * 0x204081u) & 0x01010101u);
            afrag.y = (int)((((bits >> 4) & 15u) * 0x204081u) & 0x01010101u);
            afrag.z = (int)((((bits >> 8) & 15u) * 0x204081u) & 0x01010101u);
            afrag.w = (int)((((bits >> 12) & 15u) * 0x204081u) & 0x01010101u);
            cacc[m] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afrag, bfrag, cacc[m], 0, 0, 0);
          }
        }
        unsigned long long mw[kWarmWords];
#pragma unroll
        for (int w = 0; w < kWarmWords; ++w) mw[w] = on ? a.hwarm[(r0 + r) * kWarmWords + w] : 0ull;
        while (mw[0] | mw[1] | mw[2]) {
#pragma unroll
          for (int w = 0; w < kWarmWords; ++w) {
            if (mw[w]) {
              atomicAdd(&wacc[64 * (w + 1) + __builtin_ctzll(mw[w])], rv);
              if (MODE == FWD_FTRL) atomicAdd(&wacc2[64 * (w + 1) + __builtin_ctzll(mw[w])], rv2);
              mw[w] &= mw[w] - 1;
            }
          }
        }
      }
      rp_acc[rb + lane] = 0ull;  // (also wipes the digit planes the column sums kept there)
    }
    __syncthreads();  // the margins are reused by the next panel
  }
  static_assert(kWarmWords == 3, "the walk above names three words");
  // the workgroup's two sums, then the column sums; the global adds come last, so that nothing waits for them
  for (int off = 32; off > 0; off >>= 1) {
    acc0 += __shfl_down(acc0, off, 64);
    acc1 += __shfl_down(acc1, off, 64);
  }
  if (lane == 0) {
    shred[wave] = acc0;
    shred1[wave] = acc1;
  }
  if (colsums && did_tile) {  // (a wave that took no tile holds zeros)
    // D of block m: this lane holds column 16 m + 4 (lane >> 4) + i, digit lane & 15, in element i.  The
    // 16 digit lanes of a column are one DPP row: they are put together there, and one lane adds the
    // column's sum to the workgroup's accumulator (sixteen lanes adding to one LDS word at once take
    // turns at a full atomic's latency each).
    const int digit = lane & 15;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long v = cacc[m][i];
        const int col = 16 * m + 4 * (lane >> 4) + i;
        long long x = digit < 6 ? (v << (7 * digit)) : (digit == 6 ? -(v << 41) : 0ll);
        x = row16_sum(x);
        if (digit == 0 && x) atomicAdd(&wacc[col], (unsigned long long)x);
        if (MODE == FWD_FTRL) {
          long long x2 = (digit > 6 && digit < 13) ? (v << (7 * (digit - 7))) : 0ll;
          x2 = row16_sum(x2);
          if (digit == 0 && x2) atomicAdd(&wacc2[col], (unsigned long long)x2);
        }
      }
  }
  __syncthreads();
  if (tid == 0) {
    double t0 = 0, t1 = 0;
    for (int w8 = 0; w8 < kRpThreads / 64; ++w8) {
      t0 += shred[w8];
      t1 += shred1[w8];
    }
    a.partials[0 * kMaxPartialBlocks + blockIdx.x] = t0;
    a.partials[1 * kMaxPartialBlocks + blockIdx.x] = t1;
  }
  if (colsums) {
    if (MODE == FWD_FTRL && a.mpart) {
      if (tid < (int)kMaskCols) {
        unsigned long long *slot = a.mpart + (size_t)(blockIdx.x % kMaskSlots) * 2 * kMaskCols;
        if (wacc[tid]) atomicAdd(slot + tid, wacc[tid]);
        if (wacc2[tid]) atomicAdd(slot + kMaskCols + tid, wacc2[tid]);
      }
    } else if (MODE != FWD_FTRL && a.gs) {
      if (tid < (int)kMaskCols && wacc[tid]) atomicAdd(a.gs + tid, wacc[tid]);
    } else {
      if (tid < (int)kMaskCols && wacc[tid]) atomicAdd(a.g1 + a.perm[tid], wacc[tid]);
      if (MODE == FWD_FTRL && tid < (int)kMaskCols && wacc2[tid]) atomicAdd(a.g2 + a.perm[tid], wacc2[tid]);
    }
  }
}

// wsum (see k_wsum), plus the largest |wsum| per workgroup.  The vectors are read in the caller's
// order (coalesced) and the sum goes to the column's stored position (inv: caller's id -> stored id;
// nullptr: identity): a scattered 4-byte store is cheaper than two or more scattered loads.
__global__ void __launch_bounds__(256) k_wsum_absmax(const float *__restrict__ w0, const float *__restrict__ w1,
                                                     const float *__restrict__ w2, const float *__restrict__ w3,
                                                     const uint32_t *__restrict__ inv, float *__restrict__ wsum,
                                                     uint64_t dim, float *__restrict__ blockmax) {
  __shared__ float sh[4];
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  float m = 0.0f;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < dim; i += stride) {
    float s = w0[i];
    if (w1) s += w1[i];
    if (w2) s += w2[i];
    if (w3) s += w3[i];
    wsum[inv ? inv[i] : i] = s;
    const float av = fabsf(s);
    m = av > m ? av : m;  // a NaN weight is not seen here: it reaches the margins through to_fx as garbage
  }
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_down(m, off, 64);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) blockmax[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// sweep over the row-panel copy: the whole set, or (r0, r1 on panel boundaries) a range of panels;
// nw == 0: the caller has filled csr->d_wsum and the per-workgroup maxima itself
static int launch_forward_panels(admm_csr *csr, FwdMode mode, uint64_t r0, uint64_t r1, const float *const *w, int nw,
                                 float *out_rows, int *nblocks_out, float *margins_out) {
  admm_ctx *ctx = csr->ctx;
  const int wgrid = wsum_grid(ctx, csr->dim);
  float *blockmax = wsum_blockmax(ctx);
  if (nw > 0) {
    KernelTimer t(ctx, ADMM_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_wsum_absmax, dim3(wgrid), dim3(256), 0, ctx->stream, w[0], nw > 1 ? w[1] : nullptr,
                       nw > 2 ? w[2] : nullptr, nw > 3 ? w[3] : nullptr, csr->d_inv, csr->d_wsum, csr->dim, blockmax);
    ADMM_HIP_CHECK(hipGetLastError());
  }
  RpArgs a{};
  a.hcol = csr->d_hcol;
  a.hval = csr->d_hval;
  a.hbase = csr->d_hbase;
  a.hfirst = csr->d_hfirst;
  a.fcol = csr->d_fcol;
  a.frow = csr->d_frow;
  a.fval = csr->d_fval;
  a.fptr = csr->d_fptr;
  a.add_bias = (csr->has_bias && !csr->hot_dense) ? 1 : 0;
  a.label = csr->d_label;
  if (csr->hot_dense) {
    a.hmask = csr->d_hmask;
    a.hwarm = csr->d_hwarm;
    a.hmaskT = csr->d_hmaskT;
    a.perm = csr->d_perm;
  }
  a.w = csr->d_wsum;
  a.nhot = csr->fp_hot;
  a.blockmax = blockmax;
  // (the mini-batch pass keeps running maxima behind the per-workgroup ones: one per column block, one more)
  a.nblockmax = wgrid + (mode == FWD_FTRL ? (int)csr->nblocks + 1 : 0);
  if (mode == FWD_FTRL && nw == 0) {
    // ... and has folded the per-workgroup maxima into the last of them (k_max_fold), so that each of its
    // many short launches reads one word for them
    a.blockmax = blockmax + wgrid - 1;
    a.nblockmax = 1 + (int)csr->nblocks + 1;
  }
  a.lenbits = csr->fp_lenbits;
  a.vmax = csr->vmax;
  a.nrows = csr->nrows;
  a.panel_rows = csr->fp_rows;
  a.out = out_rows;
  a.out2 = margins_out;
  a.partials = ctx->d_partials;
  if (mode == FWD_LBFGS || mode == FWD_FTRL) {
    ADMM_REQUIRE(csr->has_t, "forward sweep: the training forms need admm_csr_build_transposed");
    a.gscale = std::ldexp(1.0f, csr->fx_shift);
    if (csr->hot_dense) {
      a.g1 = reinterpret_cast<unsigned long long *>(csr->d_g1);
      a.g2 = reinterpret_cast<unsigned long long *>(csr->d_g2);
      if (mode == FWD_FTRL) a.mpart = reinterpret_cast<unsigned long long *>(csr->d_mpart);
      else a.gs = reinterpret_cast<unsigned long long *>(csr->d_gs);
    }
  }
  a.panel0 = r0 / csr->fp_rows;
  a.pcount = (r1 - r0 + csr->fp_rows - 1) / csr->fp_rows;
  // margins, the weight table (+ its filler slot, rounded to 16 bytes), and for the FTRL form 1 KB per wave
  const size_t lds = sizeof(unsigned long long) * csr->fp_rows + sizeof(float) * ((a.nhot + 1 + 3) & ~3u) +
                     (mode == FWD_FTRL ? (size_t)(kRpThreads / 64) * 1024 : 0);
  static bool raised[64] = {};
  if (!raised[ctx->device & 63]) {
    for (const void *fn : {reinterpret_cast<const void *>(&k_forward_rp<FWD_EVAL, true>),
                           reinterpret_cast<const void *>(&k_forward_rp<FWD_EVAL, false>),
                           reinterpret_cast<const void *>(&k_forward_rp<FWD_LBFGS, true>),
                           reinterpret_cast<const void *>(&k_forward_rp<FWD_LBFGS, false>),
                           reinterpret_cast<const void *>(&k_forward_rp<FWD_MARGIN, true>),
                           reinterpret_cast<const void *>(&k_forward_rp<FWD_MARGIN, false>),
                           reinterpret_cast<const void *>(&k_forward_rp<FWD_FTRL, true>),
                           reinterpret_cast<const void *>(&k_forward_rp<FWD_FTRL, false>)})
      ADMM_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    raised[ctx->device & 63] = true;
  }
  const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>(a.pcount, (uint64_t)ctx->cu_count * 2));  // two resident per CU
  *nblocks_out = grid;
  const bool hasval = a.hval != nullptr;
  KernelTimer t(ctx, mode == FWD_EVAL ? ADMM_K_FORWARD_EVAL : ADMM_K_FORWARD);
#define RP_LAUNCH(M)                                                                                                \
  if (hasval) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward_rp<M, true>), dim3(grid), dim3(kRpThreads), lds, ctx->stream, a); \
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward_rp<M, false>), dim3(grid), dim3(kRpThreads), lds, ctx->stream, a);
  if (mode == FWD_EVAL) {
    RP_LAUNCH(FWD_EVAL)
  } else if (mode == FWD_LBFGS) {
    RP_LAUNCH(FWD_LBFGS)
  } else if (mode == FWD_FTRL) {
    RP_LAUNCH(FWD_FTRL)
  } else {
    RP_LAUNCH(FWD_MARGIN)
  }
#undef RP_LAUNCH
  ADMM_HIP_CHECK(hipGetLastError());
  return ADMM_OK;
}

int launch_forward(admm_csr *csr, FwdMode mode, uint64_t r0, uint64_t r1, const float *const *w, int nw,
                   float *out_rows, int *nblocks_out, float *margins_out) {
  ADMM_REQUIRE(nw >= 0 && nw <= 4, "forward sweep: up to 4 weight vectors, got %d", nw);
  ADMM_REQUIRE(r0 <= r1 && r1 <= csr->nrows, "forward sweep: bad row range");
  if (!csr->d_wsum) ADMM_HIP_CHECK(hipMalloc(&csr->d_wsum, sizeof(float) * csr->dim));
  if (csr->has_fp && nw > 0 && mode != FWD_FTRL && r0 == 0 && r1 == csr->nrows)
    return launch_forward_panels(csr, mode, r0, r1, w, nw, out_rows, nblocks_out, margins_out);
  // a mini-batch of the FTRL pass: a range of whole panels, weights already in csr->d_wsum
  if (csr->has_fp && nw == 0 && mode == FWD_FTRL && r0 < r1 && r0 % csr->fp_rows == 0 &&
      (r1 % csr->fp_rows == 0 || r1 == csr->nrows))
    return launch_forward_panels(csr, mode, r0, r1, w, nw, out_rows, nblocks_out, margins_out);
  return launch_forward_rows(csr, mode, r0, r1, w, nw, out_rows, nblocks_out, margins_out);
}

}  // namespace admm_impl
