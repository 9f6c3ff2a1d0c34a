// Mini-batch FTRL pass as ONE launch: the workgroups stay resident for the whole pass and meet at a
// device-wide barrier between a batch's two phases.
//
//   phase A (rows)    every row of the batch is predicted from the state at the batch's start: margins over
//                     the row-panel copy (row masks, hot stream, lists), sigmoid, per-row gradient g; the mask
//                     columns' sums of g and g^2 go to their global copies
//   phase B (columns) every column block's cell of the batch: per-column sums of g and g^2, then the batch's
//                     update of the touched columns (src/ftrl.cc:76-88 summed over the batch's rows) and the
//                     refreshed entry of the vector phase A gathers
//
// The same arithmetic, in the same order, as the two launches per batch of ftrl_pass_batched below
// (k_forward_rp<FWD_FTRL> of forward_panels.hip and the fused k_transposed<true> of transposed.hip), so the two
// forms agree bit for bit.  A phase loads everything it reads that no phase writes (row masks, stream words, list
// and cell entries) in one batch when it starts, from pointers loaded a phase earlier, so that it pays one memory
// latency for its inputs and one for the gathers that depend on the other phase's results.
//
// Data that crosses workgroups inside the launch and how it travels (the XCDs' L2s are not coherent):
//   g per row        A -> B   write-through stores, read after the barrier's agent-scope acquire
//   mask sums        A -> B   global integer atomics (memory side), read after the acquire, cleared by
//                             write-through stores
//   gathered vector  B -> A   write-through stores, read after the acquire
//   running maxima   B -> A   one write-through store per workgroup
// The solver state (w, z, n) of a column is read and written by one workgroup only.
#include "internal.h"
#include "sweep_common.h"

#include <algorithm>
#include <cmath>

using namespace admm_impl;

namespace admm_impl {

constexpr int kPsThreads = 512;
constexpr int kPsListAhead = 8;   // list entries per lane in a phase's first batch of loads
constexpr int kPsHotAhead = 4;    // groups of four hot-stream words per lane in it
constexpr int kPsCellAhead = 8;   // cell entries per lane and cell in a batch of loads
constexpr int kPsCells = 2;       // cells a workgroup works on together when their accumulators fit the LDS
constexpr int kPsShards = 16;     // words the barrier's arrivals are spread over (a power of two, <= 64)
constexpr int kPsSyncWords = 32 * (kPsShards + 1);  // shards, then the give-up word, 128 bytes each
constexpr unsigned kPsSpinLimit = 4000000u;  // polls of the barrier's counter before a workgroup gives up (seconds)

struct PsArgs {
  // row-panel copy
  const uint16_t *hcol;
  const uint64_t *hbase;
  const uint16_t *hfirst;
  const uint32_t *fcol;
  const uint16_t *frow;
  const uint64_t *fptr;
  const float *label;
  const uint64_t *hmask, *hwarm, *hmaskT;
  const uint32_t *perm;  // stored id -> caller's id
  const uint32_t *inv;   // caller's id -> stored id
  uint32_t nhot;
  int lenbits;
  float vmax;
  uint64_t nrows;
  uint32_t fp_rows;      // rows per panel of the row-panel copy (<= kPsThreads, a multiple of 64)
  uint32_t ppb;          // panels per batch
  uint64_t fp_npanels;
  // column-block copy: panel b = batch b
  const uint64_t *tptr;
  const uint32_t *trow;
  const uint16_t *tcol;
  uint32_t nblocks, col_block;
  uint64_t nbatches;
  // solver
  float *w, *z, *n;
  const float *reg, *off;
  float *wsum;    // the gathered vector, stored order
  float *rowbuf;  // g per row
  float *pmax;    // [0]: max |wsum| when the pass starts; [1 + workgroup]: running maximum of what that workgroup has written since
  unsigned long long *mpart;  // [kMaskSlots][2][kMaskCols]
  float gscale;
  double inv_gscale;
  float alpha, beta, l1, l2;
  uint64_t dim;
  unsigned *sync;  // [32 k], k < kPsShards: arrivals at the barriers; [32 kPsShards]: set when a workgroup gave up waiting
  unsigned long long *host_abort;  // the same news for the host: a word of pinned host memory mapped into the device
  int cpg;                         // cells per group: 1 or kPsCells
  unsigned long long *trace;       // measurement aid (ADMM_PS_TRACE): per workgroup, ticks spent in [phase A, barrier, phase B, barrier]
};

typedef __attribute__((address_space(1))) unsigned gu32;

// Device-wide barrier of a launch whose workgroups are all resident.  Every wave has drained its stores and
// atomics before its workgroup arrives; one lane adds to the counter and polls it; its acquire then covers
// the workgroup (L1 invalidated before anybody loads).  Returns false when the pass is to be abandoned.
__device__ inline bool ps_grid_barrier(unsigned *sync, unsigned long long *host_abort, unsigned target, int *lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // The arrivals are counted in kPsShards words on lines of their own (adds to one word take turns, some 35 ns
  // each: 256 workgroups on one word would spend 9 us arriving); the first wave polls them, a lane a shard.
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    gu32 *abrt = (gu32 *)(sync + 32 * kPsShards);
    if (lane == 0)
      __hip_atomic_fetch_add((gu32 *)(sync + 32 * (blockIdx.x % kPsShards)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool good = true;
    for (unsigned spins = 0;; ++spins) {
      unsigned v = lane < kPsShards ? __hip_atomic_load((gu32 *)(sync + 32 * lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      for (int o = kPsShards / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      v = __shfl(v, 0, 64);
      if (v >= target) break;
      __builtin_amdgcn_s_sleep(1);
      unsigned ab = 0;
      if ((spins & 63u) == 63u) ab = __shfl(lane == 0 ? __hip_atomic_load(abrt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u, 0, 64);
      if (spins > kPsSpinLimit || ab != 0u) {
        good = false;
        break;
      }
    }
    if (lane == 0) {
      if (!good) {
        __hip_atomic_store(abrt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(host_abort, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      *lds_flag = good ? 1 : 0;
    }
  }
  __syncthreads();
  return *lds_flag != 0;
}

// Loads go through buffer resources: a wave-uniform base in scalar registers, a 32-bit offset per lane and a
// scalar offset per unrolled step, so that a batch of loads keeps no 64-bit address per load in vector
// registers, and the hardware's range check (an offset past the resource's end reads 0) stands in for a branch
// around every load.
typedef __amdgpu_buffer_rsrc_t ps_rsrc_t;
__device__ inline uint64_t ps_uni(uint64_t v) {  // the value is the same in every lane: say so
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
__device__ inline ps_rsrc_t ps_rsrc(const void *base, uint64_t bytes) {
  const uint64_t b = ps_uni(reinterpret_cast<uint64_t>(base));
  const uint64_t n = ps_uni(bytes);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(b), 0, (int)(n > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n), 0x00020000);
}
constexpr int kPsStream = 2;  // cache policy of data that is read once (non-temporal)
constexpr uint32_t kPsNowhere = 0xFFFFFFF0u;  // an offset past every resource's end

__device__ inline void ps_store_wt(float *p, float v) {  // write-through store
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// what one lane loads of its workgroup's panel when phase A starts (one batch of loads, one memory latency)
struct PsRowPtr {
  uint64_t hb0, hb1, f0, f1;      // the panel's pointers
  uint32_t row0;                  // the row this lane's run of the hot stream starts in
};
struct PsRowPre {
  unsigned long long m, wm[kWarmWords];
  float lab;
  u32x2 hq[kPsHotAhead];
  uint32_t lc[kPsListAhead], lq[kPsListAhead];
  uint32_t tbits[4];
};

// ... and of its cells when phase B starts
struct PsCellPtr {
  uint64_t beg, end;
};
struct PsCellPre {
  uint32_t row[kPsCellAhead], col[kPsCellAhead];
};

__global__ void __launch_bounds__(kPsThreads, 2) k_ftrl_persist(PsArgs a) {
  // LDS, one region used by both phases in turn:
  //   phase A  margins u64[fp_rows] | hot float[nhot + 1 -> 16 B] | digit planes 1 KB per wave
  //   phase B  acc u64[cpg CB] | acc2 u64[cpg CB] | touched list u16[cpg CB] | count      (cpg cells worked on together)
  extern __shared__ __attribute__((aligned(16))) unsigned long long ps_lds[];
  __shared__ float raw[kMaskCols];
  __shared__ unsigned long long wacc[kMaskCols], wacc2[kMaskCols];
  __shared__ float shmax[kPsThreads / 64];
  __shared__ int bar_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t R = a.fp_rows, CB = a.col_block;
  const unsigned G = gridDim.x;
  const int cpg = a.cpg;  // 1 or kPsCells
  unsigned long long *rp_acc = ps_lds;
  float *hot = reinterpret_cast<float *>(rp_acc + R);
  fx_t *acc = ps_lds, *acc2 = ps_lds + (size_t)cpg * CB;
  uint16_t *tl = reinterpret_cast<uint16_t *>(acc2 + (size_t)cpg * CB);
  uint32_t *tcnt = reinterpret_cast<uint32_t *>(tl + (size_t)cpg * CB);
  unsigned epoch = 0;
  float my_max = 0.0f;  // thread 0: running maximum of what this workgroup has written to wsum

  auto panel_of = [&](uint64_t b) { return b * a.ppb + blockIdx.x; };   // this workgroup's first panel of batch b
  auto panel_end = [&](uint64_t b) {
    const uint64_t e = (b + 1) * a.ppb;
    return e < a.fp_npanels ? e : a.fp_npanels;
  };
  auto row_ptrs = [&](PsRowPtr &q, uint64_t b) {
    const uint64_t p = panel_of(b);
    if (b < a.nbatches && p < panel_end(b)) {
      q.hb0 = a.hbase[p];
      q.hb1 = a.hbase[p + 1];
      q.f0 = a.fptr[p];
      q.f1 = a.fptr[p + 1];
      q.row0 = a.hfirst[p * kPsThreads + tid];
    }
  };
  auto row_data = [&](PsRowPre &q, const PsRowPtr &pt, uint64_t b) {
    const uint64_t p = panel_of(b);
    if (!(p < panel_end(b))) return;
    const uint64_t r0 = p * R;
    const uint32_t nr = (uint32_t)(a.nrows - r0 < R ? a.nrows - r0 : R);
    {
      const ps_rsrc_t rm = ps_rsrc(a.hmask + r0, (uint64_t)nr * 8), rw = ps_rsrc(a.hwarm + r0 * kWarmWords, (uint64_t)nr * 8 * kWarmWords),
                      rl = ps_rsrc(a.label + r0, (uint64_t)nr * 4);
      const u32x2 m = __builtin_amdgcn_raw_buffer_load_b64(rm, tid * 8, 0, 0);
      q.m = ((unsigned long long)m.y << 32) | m.x;
#pragma unroll
      for (int w = 0; w < kWarmWords; ++w) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rw, tid * (8 * kWarmWords), 8 * w, 0);
        q.wm[w] = ((unsigned long long)v.y << 32) | v.x;
      }
      q.lab = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, tid * 4, 0, 0));
    }
    {
      const ps_rsrc_t rh = ps_rsrc(a.hcol + pt.hb0, (pt.hb1 - pt.hb0) * 2);
#pragma unroll
      for (int j = 0; j < kPsHotAhead; ++j) q.hq[j] = __builtin_amdgcn_raw_buffer_load_b64(rh, tid * 8, j * kPsThreads * 8, kPsStream);
    }
    {
      const ps_rsrc_t rc = ps_rsrc(a.fcol + pt.f0, (pt.f1 - pt.f0) * 4), rr = ps_rsrc(a.frow + pt.f0, (pt.f1 - pt.f0) * 2);
#pragma unroll
      for (int j = 0; j < kPsListAhead; ++j) {
        q.lc[j] = __builtin_amdgcn_raw_buffer_load_b32(rc, tid * 4, j * kPsThreads * 4, kPsStream);
        q.lq[j] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rr, tid * 2, j * kPsThreads * 2, kPsStream);
      }
    }
    // the first mask word bit-transposed, for the column sums of this wave's 64 rows
    {
      const ps_rsrc_t rt = ps_rsrc(a.hmaskT + r0, (uint64_t)((nr + 63) / 64) * 512);
#pragma unroll
      for (int m = 0; m < 4; ++m)
        q.tbits[m] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rt, wave * 512 + ((16 * m + (lane & 15)) * 4 + (lane >> 4)) * 2, 0, 0);
    }
  };
  // cells of batch b this workgroup owns: column blocks blockIdx.x + j G
  auto cell_ptrs = [&](PsCellPtr (&c)[kPsCells], uint64_t b) {
#pragma unroll
    for (int j = 0; j < kPsCells; ++j) {
      const uint64_t blk = blockIdx.x + (uint64_t)j * G;
      c[j].beg = c[j].end = 0;
      if (b < a.nbatches && blk < a.nblocks) {
        c[j].beg = a.tptr[b * a.nblocks + blk];
        c[j].end = a.tptr[b * a.nblocks + blk + 1];
      }
    }
  };

  __shared__ unsigned long long tk[14];  // measurement aid: ticks per stage, then the two last stamps; thread 0 keeps them
  if (a.trace && tid == 0) {
    for (int k = 0; k < 12; ++k) tk[k] = 0;
    tk[12] = tk[13] = wall_clock64();
  }
  auto stamp = [&](int k) {
    if (a.trace && tid == 0) {
      const unsigned long long now = wall_clock64();
      tk[k] += now - tk[12];
      tk[12] = tk[13] = now;
    }
  };
  auto fine = [&](int k) {
    if (a.trace && tid == 0) {
      const unsigned long long now = wall_clock64();
      tk[k] += now - tk[13];
      tk[13] = now;
    }
  };

  // The pointers of a phase's static inputs are loaded one phase ahead (a few scalars); the inputs themselves
  // when the phase starts, all at once, so that a phase pays one memory latency for them and one for the
  // gathers that depend on the other phase's results.
  PsRowPtr rp{};
  PsCellPtr cp[kPsCells];
  row_ptrs(rp, 0);
  for (uint64_t b = 0; b < a.nbatches; ++b) {
    // ======== phase A: the batch's rows ========
    PsRowPre rq;
    row_data(rq, rp, b);
    cell_ptrs(cp, b);
    // the fixed-point shift of the margins: the largest s with (longest row) * max |w v| * 2^s < 2^50, at most 40
    {
      float mx = 0.0f;
      for (unsigned i = tid; i < G + 1; i += kPsThreads) mx = fmaxf(mx, a.pmax[i]);
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_down(mx, o, 64));
      if (lane == 0) shmax[wave] = mx;
    }
    // the weight table's entries of this lane, asked for before anything waits
    constexpr int kTabPer = 16;  // 8192 / 512
    float tabv[kTabPer];
    const ps_rsrc_t rtab = ps_rsrc(a.wsum, (uint64_t)a.nhot * 4), rws = ps_rsrc(a.wsum, a.dim * 4);
#pragma unroll
    for (int j = 0; j < kTabPer; ++j)
      tabv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rtab, tid * 4, j * kPsThreads * 4, 0));
    if (tid < (int)kMaskCols) {
      wacc[tid] = 0ull;
      wacc2[tid] = 0ull;
    }
    for (uint32_t r = tid; r < R; r += kPsThreads) rp_acc[r] = 0ull;
    __syncthreads();
    fine(4);
    int shift;
    {
      float top = 0.0f;
      for (int w8 = 0; w8 < kPsThreads / 64; ++w8) top = fmaxf(top, shmax[w8]);
      top *= a.vmax;
      int e = 0;
      if (top > 0.0f && top < 3.0e38f) frexpf(top, &e);
      else if (!(top < 3.0e38f)) e = 128;
      shift = 50 - a.lenbits - e;
      shift = shift > 40 ? 40 : (shift < -60 ? -60 : shift);
    }
    const float scale = ldexpf(1.0f, shift);
    const double inv_scale = ldexp(1.0, -shift);
#pragma unroll
    for (int j = 0; j < kTabPer; ++j) {
      const uint32_t c = (uint32_t)tid + (uint32_t)j * kPsThreads;
      if (c < a.nhot) hot[c] = truncf(tabv[j] * scale);
    }
    if (tid < (int)kMaskCols) raw[tid] = tabv[0];  // (a set with row masks has at least kMaskCols columns, all below nhot)
    for (uint32_t c = (uint32_t)tid + kTabPer * kPsThreads; c < a.nhot; c += kPsThreads) hot[c] = truncf(a.wsum[c] * scale);
    if (tid == 0) hot[a.nhot] = 0.0f;
    i32x4 cacc[4] = {};
    bool did_tile = false;
    __syncthreads();
    fine(5);
    for (uint64_t p = panel_of(b); p < panel_end(b); p += G) {
      const bool first = p == panel_of(b);
      const uint64_t r0 = p * R;
      const uint32_t nr = (uint32_t)(a.nrows - r0 < R ? a.nrows - r0 : R);
      // ---- the list entries' weights are asked for first; they arrive while the masks and the hot stream are walked
      const uint64_t beg = first ? rp.f0 : a.fptr[p], end = first ? rp.f1 : a.fptr[p + 1];
      uint32_t lq[kPsListAhead];
      float lx[kPsListAhead];
#pragma unroll
      for (int j = 0; j < kPsListAhead; ++j) {
        const uint64_t k = beg + tid + (uint64_t)j * kPsThreads;
        uint32_t c = 0;
        lq[j] = 0;
        if (first) {
          c = rq.lc[j];
          lq[j] = rq.lq[j];
        } else if (k < end) {
          c = __builtin_nontemporal_load(a.fcol + k);
          lq[j] = (uint32_t)__builtin_nontemporal_load(a.frow + k);
        }
        lx[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rws, k < end ? c * 4u : kPsNowhere, 0, 0));
      }
      // ---- this lane's row: its masks, and the margin's start from them
      unsigned long long my_m = 0, my_w[kWarmWords] = {};
      float my_lab = 0.0f;
      if ((uint32_t)tid < nr) {
        if (first) {
          my_m = rq.m;
          my_lab = rq.lab;
#pragma unroll
          for (int w = 0; w < kWarmWords; ++w) my_w[w] = rq.wm[w];
        } else {
          my_m = a.hmask[r0 + tid];
          my_lab = a.label[r0 + tid];
#pragma unroll
          for (int w = 0; w < kWarmWords; ++w) my_w[w] = a.hwarm[(r0 + tid) * kWarmWords + w];
        }
        // (the first word's bits byte by byte, then the other words' bits: the same floats in the same order as
        // the per-batch launches add them)
        float m0 = 0.0f;
        {
          unsigned long long m = my_m;
          float t = 0.0f;
          int cur = 0;
          while (m) {
            const int bit = __builtin_ctzll(m);
            if ((bit >> 3) != cur) {
              m0 += t;
              t = 0.0f;
              cur = bit >> 3;
            }
            t += raw[bit];
            m &= m - 1;
          }
          m0 += t;
        }
#pragma unroll
        for (int w = 0; w < kWarmWords; ++w) {
          unsigned long long mw = my_w[w];
          while (mw) {
            m0 += raw[64 * (w + 1) + __builtin_ctzll(mw)];
            mw &= mw - 1;
          }
        }
        atomicAdd(&rp_acc[tid], rp_to_fx(m0, scale));
      }
      // ---- hot stream
      {
        const uint64_t hb = first ? rp.hb0 : a.hbase[p];
        const uint32_t nq = (uint32_t)(((first ? rp.hb1 : a.hbase[p + 1]) - hb) / (4 * kPsThreads));
        uint32_t row = first ? rp.row0 : a.hfirst[p * kPsThreads + tid];
        const u32x2 *src = reinterpret_cast<const u32x2 *>(a.hcol + hb) + tid;
        double sum = 0.0;
        auto step = [&](uint32_t e) {
          sum += (double)hot[e & 0x7FFFu];
          if (e & 0x8000u) {
            atomicAdd(&rp_acc[row], rp_d2fx(sum));
            sum = 0.0;
            ++row;
          }
        };
        for (uint32_t q = 0; q < nq; ++q) {
          u32x2 c0;
          if (first && q < (uint32_t)kPsHotAhead) {
            c0 = rq.hq[0];
#pragma unroll
            for (int j = 1; j < kPsHotAhead; ++j)
              if (q == (uint32_t)j) c0 = rq.hq[j];
          } else {
            c0 = __builtin_nontemporal_load(src + (uint64_t)q * kPsThreads);
          }
          step(c0.x & 0xFFFFu);
          step(c0.x >> 16);
          step(c0.y & 0xFFFFu);
          step(c0.y >> 16);
        }
        if (sum != 0.0) atomicAdd(&rp_acc[row], rp_d2fx(sum));
      }
      // ---- lists
#pragma unroll
      for (int j = 0; j < kPsListAhead; ++j) {
        const uint64_t k = beg + tid + (uint64_t)j * kPsThreads;
        if (k < end) atomicAdd(&rp_acc[lq[j]], rp_to_fx(lx[j], scale));
      }
      for (uint64_t k = beg + tid + (uint64_t)kPsListAhead * kPsThreads; k < end; k += kPsThreads)
        atomicAdd(&rp_acc[a.frow[k]], rp_to_fx(a.wsum[a.fcol[k]], scale));
      __syncthreads();
      fine(6);
      // ---- per-row epilogue; wave w has rows 64 w .. 64 w + 63 (R <= kPsThreads)
      const uint32_t rb = (uint32_t)wave * 64u;
      if (rb < nr) {
        const uint32_t r = rb + lane;
        const bool on = r < nr;
        unsigned long long rv = 0, rv2 = 0;
        if (on) {
          const float m = (float)((double)(long long)rp_acc[r] * inv_scale);
          // src/ftrl.cc:72,76-78
          const float pr = (float)(1.0 / (1 + exp(-(double)clamp35(m))));
          const int y = my_lab > 0 ? 1 : 0;
          const float g = pr - y;
          ps_store_wt(a.rowbuf + r0 + r, g);
          rv = (unsigned long long)(long long)(g * a.gscale);
          rv2 = (unsigned long long)(long long)(g * g * a.gscale) | 1ull;  // non-zero marks "touched"
        }
        did_tile = true;
        // column sums of the first mask word on the matrix unit (see k_forward_rp): g + 2^41 in six 7-bit
        // digits and a count digit, g^2 in six more
        {
          const unsigned long long u = rv + (1ull << 41);
          unsigned char *pl = reinterpret_cast<unsigned char *>(hot + ((a.nhot + 1 + 3) & ~3u)) + wave * 1024;
#pragma unroll
          for (int n = 0; n < 6; ++n) pl[n * 64 + lane] = (unsigned char)((u >> (7 * n)) & 127u);
          pl[6 * 64 + lane] = 1;
#pragma unroll
          for (int n = 0; n < 6; ++n) pl[(7 + n) * 64 + lane] = (unsigned char)((rv2 >> (7 * n)) & 127u);
          i32x4 bfrag = {0, 0, 0, 0};
          if ((lane & 15) < 13) bfrag = *reinterpret_cast<const i32x4 *>(pl + (lane & 15) * 64 + 16 * (lane >> 4));
          const uint16_t *tt = reinterpret_cast<const uint16_t *>(a.hmaskT + (r0 + rb));
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const uint32_t bits = first ? rq.tbits[m] : (uint32_t)tt[(16 * m + (lane & 15)) * 4 + (lane >> 4)];
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
        for (int w = 0; w < kWarmWords; ++w) mw[w] = on ? my_w[w] : 0ull;
        while (mw[0] | mw[1] | mw[2]) {
#pragma unroll
          for (int w = 0; w < kWarmWords; ++w) {
            if (mw[w]) {
              atomicAdd(&wacc[64 * (w + 1) + __builtin_ctzll(mw[w])], rv);
              atomicAdd(&wacc2[64 * (w + 1) + __builtin_ctzll(mw[w])], rv2);
              mw[w] &= mw[w] - 1;
            }
          }
        }
        rp_acc[rb + lane] = 0ull;
      }
      __syncthreads();
      fine(7);
    }
    static_assert(kWarmWords == 3, "the walk above names three words");
    if (did_tile) {
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
          long long x2 = (digit > 6 && digit < 13) ? (v << (7 * (digit - 7))) : 0ll;
          x2 = row16_sum(x2);
          if (digit == 0 && x2) atomicAdd(&wacc2[col], (unsigned long long)x2);
        }
    }
    __syncthreads();
    if (tid < (int)kMaskCols) {
      unsigned long long *slot = a.mpart + (size_t)(blockIdx.x % kMaskSlots) * 2 * kMaskCols;
      if (wacc[tid]) atomicAdd(slot + tid, wacc[tid]);
      if (wacc2[tid]) atomicAdd(slot + kMaskCols + tid, wacc2[tid]);
    }
    stamp(0);
    if (!ps_grid_barrier(a.sync, a.host_abort, ++epoch * G, &bar_flag)) return;
    stamp(1);

    // ======== phase B: the batch's cells, cpg of them at a time ========
    row_ptrs(rp, b + 1);
    float vmax = 0.0f;
    const ps_rsrc_t rrow = ps_rsrc(a.rowbuf, a.nrows * 4);
    for (uint64_t j0 = 0; blockIdx.x + j0 * G < a.nblocks; j0 += cpg) {
      // the group's cells: pointers (the first group's were loaded in phase A), then all their entries at once
      PsCellPtr cc[kPsCells];
      PsCellPre cq[kPsCells];
#pragma unroll
      for (int jj = 0; jj < kPsCells; ++jj) {
        cc[jj].beg = cc[jj].end = 0;
        const uint64_t blk = blockIdx.x + (j0 + jj) * G;
        if (jj < cpg && blk < a.nblocks) {
          if (j0 == 0) {
            cc[jj] = cp[jj];
          } else {
            cc[jj].beg = a.tptr[b * a.nblocks + blk];
            cc[jj].end = a.tptr[b * a.nblocks + blk + 1];
          }
        }
        {
          const ps_rsrc_t rr = ps_rsrc(a.trow + cc[jj].beg, (cc[jj].end - cc[jj].beg) * 4), rc = ps_rsrc(a.tcol + cc[jj].beg, (cc[jj].end - cc[jj].beg) * 2);
#pragma unroll
          for (int i = 0; i < kPsCellAhead; ++i) {
            cq[jj].row[i] = __builtin_amdgcn_raw_buffer_load_b32(rr, tid * 4, i * kPsThreads * 4, kPsStream);
            cq[jj].col[i] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rc, tid * 2, i * kPsThreads * 2, kPsStream);
          }
        }
      }
      for (uint32_t c = tid; c < (uint32_t)cpg * CB; c += kPsThreads) {
        acc[c] = 0;
        acc2[c] = 0;
      }
      if (tid == 0) *tcnt = 0;
      __syncthreads();
      fine(8);
#pragma unroll
      for (int jj = 0; jj < kPsCells; ++jj) {
        float x[kPsCellAhead];
#pragma unroll
        for (int i = 0; i < kPsCellAhead; ++i) {
          const uint64_t k = cc[jj].beg + tid + (uint64_t)i * kPsThreads;
          x[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrow, k < cc[jj].end ? cq[jj].row[i] * 4u : kPsNowhere, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < kPsCellAhead; ++i) {
          const uint64_t k = cc[jj].beg + tid + (uint64_t)i * kPsThreads;
          if (k < cc[jj].end) {
            atomicAdd(&acc[jj * CB + cq[jj].col[i]], to_fx(x[i], a.gscale));
            atomicAdd(&acc2[jj * CB + cq[jj].col[i]], to_fx(x[i] * x[i], a.gscale) | 1ull);
          }
        }
        // a cell longer than what one batch of loads takes: the rest in further batches of the same kind
        for (uint64_t k0 = cc[jj].beg + (uint64_t)kPsCellAhead * kPsThreads; k0 < cc[jj].end; k0 += (uint64_t)kPsCellAhead * kPsThreads) {
          uint32_t rr[kPsCellAhead], cl[kPsCellAhead];
#pragma unroll
          for (int i = 0; i < kPsCellAhead; ++i) {
            const uint64_t k = k0 + tid + (uint64_t)i * kPsThreads;
            rr[i] = cl[i] = 0;
            if (k < cc[jj].end) {
              rr[i] = __builtin_nontemporal_load(a.trow + k);
              cl[i] = __builtin_nontemporal_load(a.tcol + k);
            }
          }
#pragma unroll
          for (int i = 0; i < kPsCellAhead; ++i) {
            const uint64_t k = k0 + tid + (uint64_t)i * kPsThreads;
            x[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrow, k < cc[jj].end ? rr[i] * 4u : kPsNowhere, 0, 0));
          }
#pragma unroll
          for (int i = 0; i < kPsCellAhead; ++i) {
            const uint64_t k = k0 + tid + (uint64_t)i * kPsThreads;
            if (k < cc[jj].end) {
              atomicAdd(&acc[jj * CB + cl[i]], to_fx(x[i], a.gscale));
              atomicAdd(&acc2[jj * CB + cl[i]], to_fx(x[i] * x[i], a.gscale) | 1ull);
            }
          }
        }
      }
      // the mask columns that live in these blocks pick up what phase A left for them
      if (tid < (int)kMaskCols) {
        const uint64_t i = a.perm[tid];
        const uint64_t blk = i / CB;
        if (blk % G == blockIdx.x && blk / G >= j0 && blk / G < j0 + (uint64_t)cpg) {
          const uint32_t slotc = (uint32_t)(blk / G - j0) * CB + (uint32_t)(i % CB);
          long long q1 = 0, q2 = 0;
#pragma unroll 1
          for (int s = 0; s < kMaskSlots; ++s) {
            unsigned long long *slot = a.mpart + (size_t)s * 2 * kMaskCols;
            q1 += (long long)slot[tid];
            q2 += (long long)slot[kMaskCols + tid];
            __hip_atomic_store(slot + tid, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(slot + kMaskCols + tid, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          atomicAdd(&acc[slotc], (fx_t)q1);
          atomicAdd(&acc2[slotc], (fx_t)q2);
        }
      }
      __syncthreads();
      fine(9);
      // the columns the batch touched, listed (cell in the group << 13 | column in the block); then the list worked off by full waves
      for (uint32_t c0 = 0; c0 < (uint32_t)cpg * CB; c0 += kPsThreads) {
        const uint32_t c = c0 + tid;
        const uint32_t jj = c / CB, cb = c % CB;
        const uint64_t gcol = (blockIdx.x + (j0 + jj) * G) * (uint64_t)CB + cb;
        const bool touched = c < (uint32_t)cpg * CB && gcol < a.dim && acc2[c] != 0;
        const unsigned long long vote = __ballot(touched);
        if (vote) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(tcnt, (uint32_t)__popcll(vote));
          base = __shfl(base, 0, 64);
          if (touched) tl[base + __popcll(vote & ((1ull << lane) - 1))] = (uint16_t)(jj << 13 | cb);
        }
      }
      __syncthreads();
      fine(10);
      const uint32_t ntouched = *tcnt;
      for (uint32_t k0 = 0; k0 < ntouched; k0 += 2 * kPsThreads) {
        // two columns per lane and trip: their state is asked for together
        uint64_t ci[2];
        uint32_t sid[2] = {0, 0}, lc[2] = {0, 0};
        float nold[2] = {0, 0}, zold[2] = {0, 0}, reg[2] = {0, 0}, offv[2] = {0, 0};
        bool on[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint32_t k = k0 + tid + u * kPsThreads;
          on[u] = k < ntouched;
          ci[u] = 0;
          if (on[u]) {
            const uint32_t e = tl[k];
            lc[u] = (e >> 13) * CB + (e & 0x1FFFu);
            ci[u] = (blockIdx.x + (j0 + (e >> 13)) * G) * (uint64_t)CB + (e & 0x1FFFu);
            sid[u] = a.inv ? a.inv[ci[u]] : (uint32_t)ci[u];
            nold[u] = a.n[ci[u]];
            zold[u] = a.z[ci[u]];
            reg[u] = a.reg[ci[u]];
            offv[u] = a.off[ci[u]];
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (!on[u]) continue;
          const uint64_t i = ci[u];
          const double G1 = (double)(long long)acc[lc[u]] * a.inv_gscale, G2 = (double)(long long)acc2[lc[u]] * a.inv_gscale;
          const float wc = ftrl_weight(zold[u], nold[u], reg[u], a.alpha, a.beta, a.l1, a.l2);
          const float nnew = (float)((double)nold[u] + G2);
          const double sigma = (sqrt((double)nnew) - sqrt((double)nold[u])) / (double)a.alpha;
          const float znew = (float)((double)zold[u] + (G1 - sigma * (double)wc));
          a.w[i] = wc;
          a.z[i] = znew;
          a.n[i] = nnew;
          const float v = ftrl_weight(znew, nnew, reg[u], a.alpha, a.beta, a.l1, a.l2) + offv[u];
          ps_store_wt(a.wsum + sid[u], v);
          vmax = fmaxf(vmax, fabsf(v));
        }
      }
      __syncthreads();  // the accumulators are zeroed again by the next group
      fine(11);
    }
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, o, 64));
    if (lane == 0) shmax[wave] = vmax;
    __syncthreads();
    if (tid == 0) {
      float m = my_max;
      for (int w8 = 0; w8 < kPsThreads / 64; ++w8) m = fmaxf(m, shmax[w8]);
      if (m > my_max) {
        my_max = m;
        ps_store_wt(a.pmax + 1 + blockIdx.x, m);
      }
    }
    stamp(2);
    if (!ps_grid_barrier(a.sync, a.host_abort, ++epoch * G, &bar_flag)) return;
    stamp(3);
  }
  if (a.trace && tid == 0)
    for (int k = 0; k < 12; ++k) a.trace[blockIdx.x * 12 + k] = tk[k];
}

// host side ---------------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256) k_ps_init_max(float *pmax, const float *folded, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) pmax[i] = i == 0 ? *folded : 0.0f;
}

// Opt-in (ADMM_FTRL_PERSISTENT=1): on MI355X the one-launch form is the slower one (9.0 against 5.7 ms per pass of
// 10M rows x 2^20 features in batches of 65536).  Both phases of a batch are bound by what a CU can do per stored
// entry -- one gather that leaves the L2 and two 64-bit LDS adds in phase B, one weight gather and one LDS add in
// phase A -- not by launches, and the per-batch launches run two workgroups per CU side by side where this form
// has one; its two device-wide barriers per batch (5 us each with the arrivals spread over 16 words, plus the wait
// for the slowest workgroup) cost more than the two launches they replace.
bool ftrl_persistent_eligible(const admm_csr *csr, uint64_t batch) {
  const char *env = getenv("ADMM_FTRL_PERSISTENT");
  if (!env || env[0] != '1') return false;
  return csr->has_t && !csr->has_tc && csr->hot_dense && csr->has_fp && !csr->has_val && csr->d_perm && csr->d_inv &&
         csr->fp_rows <= (uint32_t)kPsThreads && csr->fp_rows % 64 == 0 && csr->panel_rows == batch &&
         batch % csr->fp_rows == 0 && csr->col_block <= 8192 && csr->dim >= kMaskCols && csr->npanels > 0 &&
         csr->npanels < (1ull << 22) && csr->nrows < (1ull << 30);
}

// The pass over all batches in one launch.  The caller (ftrl_pass_batched) has filled csr->d_wsum and folded
// max |wsum| into `folded_max`, and runs the trailing Predict of the last row afterwards.
int ftrl_pass_persistent(admm_csr *csr, float *w, float *z, float *n, const float *offset, const float *reg_offset,
                         float alpha, float beta, float l1, float l2, const float *folded_max) {
  admm_ctx *ctx = csr->ctx;
  const size_t lds_a = sizeof(unsigned long long) * csr->fp_rows + sizeof(float) * ((csr->fp_hot + 1 + 3) & ~3u) +
                       (size_t)(kPsThreads / 64) * 1024;
  const size_t lds_b1 = sizeof(fx_t) * 2 * csr->col_block + 2 * (size_t)csr->col_block + 64;
  const int cpg = (kPsCells * lds_b1 <= 96 * 1024 && csr->nblocks > (uint32_t)ctx->cu_count) ? kPsCells : 1;
  const size_t lds = (std::max(lds_a, cpg * lds_b1) + 15) & ~(size_t)15;
  static bool raised[64] = {};
  if (!raised[ctx->device & 63]) {
    ADMM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ftrl_persist), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       150 * 1024));
    raised[ctx->device & 63] = true;
  }
  ADMM_REQUIRE(lds <= 150 * 1024, "admm_ftrl_pass: the one-launch form needs %zu bytes of LDS", lds);
  // every workgroup must be resident: one per CU, never more than the device can hold at once
  int per_cu = 0;
  ADMM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&k_ftrl_persist), kPsThreads, lds));
  ADMM_REQUIRE(per_cu >= 1, "admm_ftrl_pass: the one-launch form does not fit a CU");
  const uint32_t ppb = (uint32_t)(csr->panel_rows / csr->fp_rows);
  const int grid = (int)std::min<uint64_t>((uint64_t)ctx->cu_count, std::max<uint64_t>(ppb, csr->nblocks));
  // [pmax: grid + 1 floats] [sync: 64 words, zeroed before every launch]
  const size_t need = sizeof(float) * (((size_t)grid + 1 + 63) & ~(size_t)63) + kPsSyncWords * sizeof(unsigned);
  if (ctx->ps_bytes < need) {
    ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->d_ps) ADMM_HIP_CHECK(hipFree(ctx->d_ps));
    ctx->d_ps = nullptr;
    ctx->ps_bytes = 0;
    ADMM_HIP_CHECK(hipMalloc(&ctx->d_ps, need));
    ctx->ps_bytes = need;
  }
  unsigned *sync = reinterpret_cast<unsigned *>(ctx->d_ps);
  float *pmax = reinterpret_cast<float *>(sync + kPsSyncWords);
  ADMM_HIP_CHECK(hipMemsetAsync(sync, 0, kPsSyncWords * sizeof(unsigned), ctx->stream));
  hipLaunchKernelGGL(k_ps_init_max, dim3((grid + 1 + 255) / 256), dim3(256), 0, ctx->stream, pmax, folded_max, grid + 1);
  ADMM_HIP_CHECK(hipGetLastError());
  PsArgs a{};
  a.hcol = csr->d_hcol;
  a.hbase = csr->d_hbase;
  a.hfirst = csr->d_hfirst;
  a.fcol = csr->d_fcol;
  a.frow = csr->d_frow;
  a.fptr = csr->d_fptr;
  a.label = csr->d_label;
  a.hmask = csr->d_hmask;
  a.hwarm = csr->d_hwarm;
  a.hmaskT = csr->d_hmaskT;
  a.perm = csr->d_perm;
  a.inv = csr->d_inv;
  a.nhot = csr->fp_hot;
  a.lenbits = csr->fp_lenbits;
  a.vmax = csr->vmax;
  a.nrows = csr->nrows;
  a.fp_rows = csr->fp_rows;
  a.ppb = ppb;
  a.fp_npanels = (csr->nrows + csr->fp_rows - 1) / csr->fp_rows;
  a.tptr = csr->d_tptr;
  a.trow = csr->d_trow;
  a.tcol = csr->d_tcol;
  a.nblocks = csr->nblocks;
  a.col_block = csr->col_block;
  a.nbatches = csr->npanels;
  a.w = w;
  a.z = z;
  a.n = n;
  a.reg = reg_offset;
  a.off = offset;
  a.wsum = csr->d_wsum;
  a.rowbuf = csr->d_rowbuf;
  a.pmax = pmax;
  a.mpart = reinterpret_cast<unsigned long long *>(csr->d_mpart);
  a.gscale = std::ldexp(1.0f, csr->fx_shift);
  a.inv_gscale = std::ldexp(1.0, -csr->fx_shift);
  a.alpha = alpha;
  a.beta = beta;
  a.l1 = l1;
  a.l2 = l2;
  a.dim = csr->dim;
  a.cpg = cpg;
  a.sync = sync;
  a.host_abort = ctx->h_seq_dev + 1;
  const bool trace = getenv("ADMM_PS_TRACE") != nullptr;
  unsigned long long *d_trace = nullptr;
  if (trace) {
    ADMM_HIP_CHECK(hipMalloc(&d_trace, sizeof(unsigned long long) * 12 * grid));
    a.trace = d_trace;
  }
  {
    KernelTimer t(ctx, ADMM_K_FORWARD);
    void *params[] = {&a};
    ADMM_HIP_CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(&k_ftrl_persist), dim3(grid), dim3(kPsThreads), params,
                                              (unsigned)lds, ctx->stream));
  }
  if (trace) {  // 100 MHz ticks -> microseconds per batch, for the first and the slowest-in-phase-A workgroup
    std::vector<unsigned long long> h(12 * (size_t)grid);
    ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ADMM_HIP_CHECK(hipMemcpy(h.data(), d_trace, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    (void)hipFree(d_trace);
    double mean[12] = {}, mx[12] = {};
    for (int g = 0; g < grid; ++g)
      for (int k = 0; k < 12; ++k) {
        const double us = (double)h[12 * g + k] / 100.0 / (double)csr->npanels;
        mean[k] += us / grid;
        mx[k] = std::max(mx[k], us);
      }
    fprintf(stderr, "[ps-trace] per batch, us: phase A mean %.2f max %.2f | barrier %.2f / %.2f | phase B %.2f / %.2f | barrier %.2f / %.2f (grid %d, %llu batches)\n",
            mean[0], mx[0], mean[1], mx[1], mean[2], mx[2], mean[3], mx[3], grid, (unsigned long long)csr->npanels);
    fprintf(stderr, "[ps-trace]   A: maxima+zero %.2f | table %.2f | masks, streams, lists %.2f | epilogue %.2f    B: zero %.2f | entries %.2f | "
            "listing %.2f | update %.2f  (means; B summed over the workgroup's cells)\n",
            mean[4], mean[5], mean[6], mean[7], mean[8], mean[9], mean[10], mean[11]);
  }
  return ADMM_OK;
}

// ---------------------------------------------------------------------------------------------
// the pass as two launches per batch (the default)
// ---------------------------------------------------------------------------------------------

// v[n - 1] <- max of v[0 .. n)
__global__ void __launch_bounds__(256) k_max_fold(float *v, int n) {
  __shared__ float sh[4];
  float m = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, v[i]);
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) v[n - 1] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}


// The vector the forward sweep of a batch gathers: lazy weight at the batch's start plus offset, in
// the sample set's stored column order.  The state is read in its own order (coalesced) and the
// result goes to the column's stored position (inv: caller's id -> stored id; nullptr: identity).
__global__ void __launch_bounds__(256) k_ftrl_wsum(const float *__restrict__ z, const float *__restrict__ n,
                                                   const float *__restrict__ c, const float *__restrict__ offset,
                                                   const uint32_t *__restrict__ inv, float alpha, float beta, float l1,
                                                   float l2, float *__restrict__ wsum, uint64_t dim,
                                                   float *__restrict__ blockmax) {
  __shared__ float sh[4];
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  float m = 0.0f;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < dim; i += stride) {
    const float v = ftrl_weight(z[i], n[i], c[i], alpha, beta, l1, l2) + offset[i];
    wsum[inv ? inv[i] : i] = v;
    m = fmaxf(m, fabsf(v));
  }
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) blockmax[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));  // for the sweep's fixed-point shift
}

// Fallback of the fused update in the transposed sweep, for a batch whose forward sweep ran over the
// row-major form (it adds the mask columns' sums to g1 / g2): per coordinate touched in the batch (its
// squared-gradient sum is non-zero) w <- the lazy weight at the batch's start, n' = n + G2,
// z += G1 - w (sqrt(n') - sqrt(n)) / alpha; the accumulators are cleared and the column's entry of the
// forward sweep's vector is brought up to date
__global__ void __launch_bounds__(256) k_ftrl_apply(long long *__restrict__ g1, long long *__restrict__ g2,
                                                    double inv_scale, const float *__restrict__ c,
                                                    const float *__restrict__ offset, float *__restrict__ w,
                                                    float *__restrict__ z, float *__restrict__ n, float alpha, float beta,
                                                    float l1, float l2, const double *__restrict__ partials, int bias_blocks,
                                                    uint64_t dim, const uint32_t *__restrict__ inv, float *__restrict__ wsum,
                                                    unsigned int *__restrict__ absmax) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  float vmax = 0.0f;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < dim; i += stride) {
    const long long q2 = g2[i];
    bool touched = q2 != 0;
    double G1 = 0, G2 = 0;
    if (touched) {
      G1 = (double)g1[i] * inv_scale;
      G2 = (double)q2 * inv_scale;
    }
    if (bias_blocks && i == 0) {
      touched = true;
      for (int b = 0; b < bias_blocks; ++b) {
        G1 += partials[0 * kMaxPartialBlocks + b];
        G2 += partials[1 * kMaxPartialBlocks + b];
      }
    }
    if (touched) {
      const float nold = n[i], zold = z[i];
      const float wc = ftrl_weight(zold, nold, c[i], alpha, beta, l1, l2);
      const float nnew = (float)((double)nold + G2);
      const double sigma = (sqrt((double)nnew) - sqrt((double)nold)) / (double)alpha;
      const float znew = (float)((double)zold + (G1 - sigma * (double)wc));
      w[i] = wc;
      z[i] = znew;
      n[i] = nnew;
      g1[i] = 0;
      g2[i] = 0;
      const float v = ftrl_weight(znew, nnew, c[i], alpha, beta, l1, l2) + offset[i];
      wsum[inv ? inv[i] : i] = v;
      vmax = fmaxf(vmax, fabsf(v));
    }
  }
  for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_down(vmax, off, 64));
  if ((threadIdx.x & 63) == 0 && vmax > 0.0f) atomicMax(absmax, __float_as_uint(vmax));  // non-negative floats order like their bits
}

// the trailing Predict of FtrlSolver::Run on the last row (src/ftrl.cc:140-141): refresh the
// lazy weights of its coordinates
__global__ void __launch_bounds__(64) k_ftrl_last_row(const uint64_t *__restrict__ rowptr,
                                                      const uint32_t *__restrict__ col,
                                                      const uint32_t *__restrict__ perm, uint64_t last,
                                                      const float *__restrict__ z, const float *__restrict__ n,
                                                      const float *__restrict__ c, float alpha, float beta,
                                                      float l1, float l2, float *__restrict__ w) {
  const uint64_t b = rowptr[last], e = rowptr[last + 1];
  for (uint64_t k = b + threadIdx.x; k < e; k += 64) {
    const uint32_t j = perm ? perm[col[k]] : col[k];
    w[j] = ftrl_weight(z[j], n[j], c[j], alpha, beta, l1, l2);
  }
}

int ftrl_pass_batched(admm_csr *csr, float *w, float *z, float *n, const float *offset, const float *reg_offset,
                      float alpha, float beta, float l1, float l2, uint64_t batch) {
  admm_ctx *ctx = csr->ctx;
  ADMM_REQUIRE(csr->has_t, "admm_ftrl_pass: batch > 1 needs admm_csr_build_transposed");
  // mini-batches sweep panel ranges, which read cells only: a copy whose frequent columns are in the
  // column stream is rebuilt without it (once)
  if (csr->has_tc) ADMM_TRY(build_transposed(csr, csr->panel_rows, csr->col_block, false));
  ADMM_REQUIRE(csr->col_block <= 8192, "admm_ftrl_pass: batch > 1 needs col_block <= 8192 (two accumulators per column)");
  ADMM_REQUIRE(csr->panel_rows == batch || (csr->npanels <= 1 && batch >= csr->nrows),
               "admm_ftrl_pass: batch (%llu) must equal the transposed copy's panel_rows (%llu)",
               (unsigned long long)batch, (unsigned long long)csr->panel_rows);
  ADMM_TRY(ensure_rowbuf(csr));
  if (!csr->d_wsum) ADMM_HIP_CHECK(hipMalloc(&csr->d_wsum, sizeof(float) * csr->dim));
  const bool bias_partials = csr->has_bias && !csr->hot_dense;
  if (csr->hot_dense && csr->has_fp && !csr->d_mpart) {
    ADMM_HIP_CHECK(hipMalloc(&csr->d_mpart, sizeof(long long) * 2 * kMaskCols * kMaskSlots));
    ADMM_HIP_CHECK(hipMemsetAsync(csr->d_mpart, 0, sizeof(long long) * 2 * kMaskCols * kMaskSlots, ctx->stream));
  }
  // The vector the forward sweeps gather (lazy weight + offset per column, stored order) is computed
  // once for all columns; after that a column's entry changes only when a batch updates the column,
  // and the update writes it.  The maxima the sweeps derive their fixed-point shift from: one per
  // workgroup of this first pass, plus one running maximum that the updates raise.
  const int wgrid = wsum_grid(ctx, csr->dim);
  float *blockmax = wsum_blockmax(ctx);
  ADMM_REQUIRE((size_t)wgrid + csr->nblocks + 1 <= 2 * (size_t)kMaxPartialBlocks * sizeof(double) / sizeof(float),
               "admm_ftrl_pass: too many column blocks (%u)", csr->nblocks);
  ADMM_HIP_CHECK(hipMemsetAsync(blockmax + wgrid, 0, sizeof(float) * (csr->nblocks + 1), ctx->stream));
  {
    KernelTimer t(ctx, ADMM_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_ftrl_wsum, dim3(wgrid), dim3(256), 0, ctx->stream, z, n, reg_offset, offset, csr->d_inv, alpha, beta,
                       l1, l2, csr->d_wsum, csr->dim, blockmax);
    ADMM_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_max_fold, dim3(1), dim3(256), 0, ctx->stream, blockmax, wgrid);
    ADMM_HIP_CHECK(hipGetLastError());
  }
  if (ftrl_persistent_eligible(csr, batch) && csr->d_mpart) {
    // the whole pass in one launch (ftrl_batch.hip); same results bit for bit
    ADMM_TRY(ftrl_pass_persistent(csr, w, z, n, offset, reg_offset, alpha, beta, l1, l2, blockmax + wgrid - 1));
    if (csr->nrows) {
      hipLaunchKernelGGL(k_ftrl_last_row, dim3(1), dim3(64), 0, ctx->stream, csr->d_rowptr, csr->d_col,
                         csr->d_perm, csr->nrows - 1, z, n, reg_offset, alpha, beta, l1, l2, w);
      ADMM_HIP_CHECK(hipGetLastError());
    }
    return ADMM_OK;
  }
  TArgs fuse{};
  fuse.fw = w;
  fuse.fz = z;
  fuse.fn = n;
  fuse.freg = reg_offset;
  fuse.foff = offset;
  fuse.finv = csr->d_inv;
  fuse.fwsum = csr->d_wsum;
  fuse.fabsmax = blockmax + wgrid;
  fuse.falpha = alpha;
  fuse.fbeta = beta;
  fuse.fl1 = l1;
  fuse.fl2 = l2;
  fuse.finv_scale = std::ldexp(1.0, -csr->fx_shift);
  fuse.fpartials = ctx->d_partials;
  fuse.fdim = csr->dim;
  for (uint64_t p = 0; p < csr->npanels; ++p) {
    const uint64_t r0 = p * csr->panel_rows, r1 = std::min(csr->nrows, r0 + csr->panel_rows);
    int nblk = 0;
    ADMM_TRY(launch_forward(csr, FWD_FTRL, r0, r1, nullptr, 0, csr->d_rowbuf, &nblk));
    // (the sweep over the row-panel copy leaves the mask columns' sums in their copies; the row-major one adds them to g1 / g2 itself)
    const bool mpart_used = csr->d_mpart && csr->hot_dense && csr->has_fp && r0 % csr->fp_rows == 0 &&
                            (r1 % csr->fp_rows == 0 || r1 == csr->nrows);
    if (csr->hot_dense && !mpart_used) {
      // mask columns' sums sit in g1 / g2: the separate per-coordinate pass reads them there
      ADMM_TRY(launch_transposed(csr, true, p, p + 1));
      KernelTimer t(ctx, ADMM_K_ELEMENTWISE);
      hipLaunchKernelGGL(k_ftrl_apply, dim3(grid_for(csr->dim, 256, ctx->cu_count * 8)), dim3(256), 0, ctx->stream, csr->d_g1,
                         csr->d_g2, std::ldexp(1.0, -csr->fx_shift), reg_offset, offset, w, z, n, alpha, beta, l1, l2,
                         ctx->d_partials, 0, csr->dim, csr->d_inv, csr->d_wsum,
                         reinterpret_cast<unsigned int *>(blockmax + wgrid + csr->nblocks));
      ADMM_HIP_CHECK(hipGetLastError());
      continue;
    }
    fuse.fmpart = mpart_used ? csr->d_mpart : nullptr;
    fuse.fbias_blocks = bias_partials ? nblk : 0;
    ADMM_TRY(launch_transposed_impl(csr, true, p, p + 1, &fuse));
  }
  if (csr->nrows) {
    hipLaunchKernelGGL(k_ftrl_last_row, dim3(1), dim3(64), 0, ctx->stream, csr->d_rowptr, csr->d_col,
                       csr->d_perm, csr->nrows - 1, z, n, reg_offset, alpha, beta, l1, l2, w);
    ADMM_HIP_CHECK(hipGetLastError());
  }
  return ADMM_OK;
}

}  // namespace admm_impl

namespace admm_impl {
int ftrl_pass_sequential(admm_csr *csr, float *w, float *z, float *n, const float *offset, const float *reg_offset,
                         float alpha, float beta, float l1, float l2);  // ftrl.hip
int ftrl_pass_multi(admm_csr *const *sets, float *const *w, float *const *z, float *const *n,
                    const float *const *offset, const float *const *reg_offset, int count, float alpha, float beta,
                    float l1, float l2);  // ftrl.hip
}

extern "C" int admm_ftrl_pass_multi(admm_csr *const *sets, float *const *w, float *const *z, float *const *n,
                                    const float *const *offset, const float *const *reg_offset, int count,
                                    float alpha, float beta, float l1, float l2) {
  ADMM_REQUIRE(count >= 0, "admm_ftrl_pass_multi: negative count");
  ADMM_REQUIRE(count == 0 || (sets && w && z && n && offset && reg_offset), "admm_ftrl_pass_multi: NULL argument");
  ADMM_REQUIRE(alpha > 0, "admm_ftrl_pass_multi: alpha must be > 0");
  for (int k = 0; k < count; ++k) {
    ADMM_REQUIRE(sets[k] && w[k] && z[k] && n[k] && offset[k] && reg_offset[k],
                 "admm_ftrl_pass_multi: NULL argument for pass %d", k);
    ADMM_TRY(require_rows(sets[k], "admm_ftrl_pass_multi"));
    for (int j = 0; j < k; ++j)
      ADMM_REQUIRE(w[j] != w[k] && z[j] != z[k] && n[j] != n[k],
                   "admm_ftrl_pass_multi: passes %d and %d share a state vector", j, k);
  }
  return ftrl_pass_multi(sets, w, z, n, offset, reg_offset, count, alpha, beta, l1, l2);
}

extern "C" int admm_ftrl_pass(admm_csr *csr, float *w, float *z, float *n, const float *offset,
                              const float *reg_offset, float alpha, float beta, float l1, float l2,
                              uint64_t batch) {
  ADMM_REQUIRE(csr && w && z && n && offset && reg_offset, "admm_ftrl_pass: NULL argument");
  ADMM_REQUIRE(alpha > 0, "admm_ftrl_pass: alpha must be > 0");
  ADMM_TRY(require_rows(csr, "admm_ftrl_pass"));
  if (batch <= 1) return ftrl_pass_sequential(csr, w, z, n, offset, reg_offset, alpha, beta, l1, l2);
  return ftrl_pass_batched(csr, w, z, n, offset, reg_offset, alpha, beta, l1, l2, batch);
}

