// The two sweeps over a sample set and everything built from them:
//   forward    row-parallel CSR sweep: margins, sigmoid, per-row residual, loss sums
//   transposed column-block sweep: X^T r with per-column accumulators staged in LDS
// plus prediction / logloss, the batch objective f,g, the L-BFGS driver and the mini-batch FTRL
// pass.
#include "internal.h"
#include "sweep_common.h"

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
  float *out2;  // FWD_LBFGS: the row sums themselves (nullptr: not wanted)
  double *partials;
  int mode;
  uint32_t rows_per_tile;  // <= kFwRowSlots
  uint32_t dim;            // entries of w[0]
  uint32_t nhot;  // leading entries of w[0] cached in LDS
  // compact sets: per-row mask of unit entries in columns 0..63 (nullptr otherwise), and where the
  // per-column sums of those columns go (64-bit fixed point accumulators indexed by original id)
  const uint64_t *hmask;
  const uint64_t *hwarm;  // kWarmWords more mask words per row (columns 64 .. kMaskCols-1)
  const uint32_t *perm;
  unsigned long long *g1, *g2;
  float scale;
};


// Forward sweep geometry: one workgroup per CU, split into kFwConsumers reducing threads and
// kFwProducers streaming threads; a tile is up to kFwRowSlots consecutive rows (the launch picks
// the count so that a tile's stored entries normally fit one chunk of kFwChunk entries); the
// first kFwHot entries of the gathered vector are cached in LDS.
constexpr int kFwProducers = 512;
// consumer threads (512, with the leaner producers of a set without values in a 1024-thread
// workgroup, measured no faster)
constexpr int fw_consumers(bool) { return 256; }
constexpr int fw_threads(bool hasval) { return fw_consumers(hasval) + kFwProducers; }
constexpr int kFwRowSlots = 128;
constexpr int kFwChunk = 8192;
constexpr int kFwHot = 16384;
constexpr int kFwSlots = 5;                                  // tiles whose row data is resident in LDS
constexpr size_t kFwLds = sizeof(float) * (2 * kFwChunk + kFwHot + 8 * 256 + (kFwSlots + 4) * kFwRowSlots) +
                          sizeof(unsigned long long) * (kFwSlots * (kFwRowSlots + 2) + kFwSlots * kFwRowSlots +
                                                        2 * kFwRowSlots * kWarmWords + 2 * 64 * kWarmWords + kWarmWords) + 64;

// CSR-stream forward sweep.  The stored entries of a tile's rows are one contiguous range of
// col/val.  Producer threads stream it (one entry per lane per load, so that a wave-wide load
// covers 64 consecutive entries), fetch the weight of each entry, and leave the products in LDS in
// storage order; consumer threads (two lanes per row) add up each row's slice of LDS, run the
// sigmoid / loss epilogue once per row, and accumulate X^T r of the mask columns.
//
// The gather is what bounds this kernel: a wave-wide gather of scattered floats costs the memory
// system about 2 clocks per active lane when it hits L2 (every lane pulls a whole line into L1),
// more when it does not.  The sample set's columns are therefore relabelled by decreasing
// frequency (power-law data: a few thousand columns hold most entries), the gathered vector
// `wsum` is stored in that order, and its first kFwHot entries are kept in LDS, where a gather is
// an order of magnitude cheaper.
//
// A wave that issues loads stalls until the memory pipe has accepted them, so the waves that
// feed the pipe do nothing else: while the consumers reduce tile i (product buffer half i & 1),
// the producers make the products of tile i+1 in the other half, issue the stream of tile i+3
// behind that, wait for everything except that stream, and meet the consumers at the barrier that
// ends the step.  A tile's stream is thus in flight for two whole steps (three register sets per
// producer thread): its trip to HBM is never waited for.
// The weights outside the LDS copy are fetched by exec-masked `buffer_load_dword ... lds`
// instructions, which the memory pipe completes by writing LDS itself: lanes whose column is in
// the LDS copy (or past the tile's end) are masked off and cost the pipe nothing, and no register
// is written behind the compiler's back (a gather into registers under a branch makes the
// compiler wait for ALL outstanding loads at its first use, and hiding the load from the compiler
// in inline assembly is only correct while the compiler happens not to copy the destination
// register before the data lands).  Sets with values multiply the gathered weights in place
// after the wait.
template <int MODE, bool HASVAL>
__global__ void __launch_bounds__(fw_threads(HASVAL)) k_forward(FwdArgs a) {
  constexpr int kFwConsumers = fw_consumers(HASVAL), kFwThreads = fw_threads(HASVAL);
  constexpr int kFwLanesPerRow = kFwConsumers / kFwRowSlots;  // 2 or 4
  extern __shared__ unsigned long long fw_lds[];
  unsigned long long *rp0 = fw_lds;                                     // [kFwSlots][kFwRowSlots + 2] row pointers
  unsigned long long *mask0 = rp0 + kFwSlots * (kFwRowSlots + 2);       // [kFwSlots][kFwRowSlots] hot masks
  unsigned long long *warm0 = mask0 + kFwSlots * kFwRowSlots;           // [2][kFwRowSlots][kWarmWords] warm masks
  unsigned long long *wacc = warm0 + 2 * kFwRowSlots * kWarmWords;      // [64 * kWarmWords] X^T r of the warm columns
  unsigned long long *wacc2 = wacc + 64 * kWarmWords;                   // [64 * kWarmWords] same for r^2 (FTRL form)
  float *prod0 = reinterpret_cast<float *>(wacc2 + 64 * kWarmWords + kWarmWords);  // [2][kFwChunk]
  float *hot = prod0 + 2 * kFwChunk;                                    // [kFwHot]
  float *tab = hot + kFwHot;                                            // [8][256] byte -> sum of hot weights
  float *lab0 = tab + 8 * 256;                                          // [kFwSlots][kFwRowSlots] labels
  // per-row results of a tile, two halves: the consumers fill half (i & 1) for tile i while the
  // producers copy tile i-1's results out of the other half
  float *out_s0 = lab0 + kFwSlots * kFwRowSlots;                        // [2][kFwRowSlots] per-row results of a tile
  float *out2_s0 = out_s0 + 2 * kFwRowSlots;                            // [2][kFwRowSlots] row sums of a tile (FWD_LBFGS)
  __shared__ double shred[16];
  // after the tile loop the product buffer is free: per-wave column sums are combined there
  unsigned long long(*hsum)[64] = reinterpret_cast<unsigned long long(*)[64]>(prod0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const bool producer = tid >= kFwConsumers;
  const int ptid = tid - kFwConsumers;  // producers: 0 .. kFwProducers-1
  const int trow = tid / kFwLanesPerRow, q = tid % kFwLanesPerRow;  // consumers
  const uint64_t nrows = a.r1 - a.r0;
  const uint32_t rpt = a.rows_per_tile;
  const uint64_t ntiles = (nrows + rpt - 1) / rpt;
  const float *__restrict__ wsum = a.w[0];
  const uint32_t nhot = a.nhot;
  const bool dense = a.hmask != nullptr;
  constexpr bool hasval = HASVAL;
  const bool colsums = dense && MODE != FWD_EVAL && MODE != FWD_MARGIN;  // X^T r of the mask columns is wanted
  for (uint32_t j = tid; j < nhot; j += kFwThreads) hot[j] = wsum[j];
  if (tid < 2 * 64 * kWarmWords) wacc[tid] = 0ull;  // wacc, wacc2 are contiguous
  if (dense) {
    // tab[g][byte] = sum of the weights of columns 8g .. 8g+7 selected by the bits of `byte`
    __syncthreads();
    for (int i = tid; i < 8 * 256; i += kFwThreads) {
      const int g = i >> 8, byte = i & 255;
      float t = 0.0f;
#pragma unroll
      for (int bit = 0; bit < 8; ++bit)
        if (byte & (1 << bit)) t += hot[8 * g + bit];
      tab[i] = t;
    }
  }
  double acc0 = 0, acc1 = 0;
  // per-lane sums of the 64 mask columns over the rows this wave has seen (lane k <-> column k)
  unsigned long long h1 = 0, h2 = 0;

  // Producers keep three register sets of streamed entries, one dword per lane and load: entry
  // j * kFwProducers + ptid of the tile's chunk, so that the 64 entries of one wave-wide load are
  // consecutive in the product buffer.  Set t % 3 holds tile t: during step i one set holds tile
  // i+1 (arrived; its products are made now), one tile i+2 (in flight), one receives tile i+3.
  constexpr int J = kFwChunk / kFwProducers;  // loads per producer thread and chunk
  uint32_t c0[J], c1[J], c2[J];
  float v0[J], v1[J], v2[J];  // sets with values only
#pragma unroll
  for (int j = 0; j < J; ++j) {
    c0[j] = c1[j] = c2[j] = 0u;
    v0[j] = v1[j] = v2[j] = 0.0f;
  }
  // wait until only the loads of the newest stream (J, or 2 J with values) are outstanding
  auto wait_all_but_newest_stream = [] {
    if (HASVAL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * J) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(J) : "memory");
  };
  static_assert(2 * J <= 63, "vmcnt holds six bits");
  const int pwave = __builtin_amdgcn_readfirstlane(ptid >> 6);
  // the gathered vector as a buffer resource (dim * 4 bytes)
  const uint64_t wbase = reinterpret_cast<uint64_t>(wsum);
  const u32x4 wres = {(uint32_t)wbase, (uint32_t)(wbase >> 32) & 0xFFFFu, a.dim * 4u, 0x00020000u};
  // Stream of one tile: its stored entries [B, E) (at most one chunk of them) through a buffer
  // resource that starts at the tile's first entry; lanes past the tile's end read zero.
  auto issue_stream = [&](uint32_t (&pc)[J], float (&pv)[J], unsigned long long Bv, unsigned long long Ev) {
    // the extents are the same in every lane: tell the compiler, or it wraps each load in a loop
    // over the distinct resources
    auto uni = [](unsigned long long v) {
      return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const unsigned long long B = uni(Bv), E = uni(Ev);
    const unsigned long long n = E - B < (unsigned long long)kFwChunk ? E - B : (unsigned long long)kFwChunk;
    const __amdgpu_buffer_rsrc_t rc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(a.col + B), 0, (int)(n * 4), 0x00020000);
#pragma unroll
    for (int j = 0; j < J; ++j) pc[j] = __builtin_amdgcn_raw_buffer_load_b32(rc, ptid * 4, j * kFwProducers * 4, 2);
    if (HASVAL) {
      const __amdgpu_buffer_rsrc_t rv =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.val + B), 0, (int)(n * 4), 0x00020000);
#pragma unroll
      for (int j = 0; j < J; ++j)
        pv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, ptid * 4, j * kFwProducers * 4, 2));
    }
  };
  // Gathers of the weights that are not in the LDS copy: exec-masked buffer loads that the memory
  // pipe writes straight into the product buffer (lane l of a wave-wide load lands at the
  // destination base + 4 l, which is this wave's stretch of the chunk; inactive lanes leave their
  // words alone).  No register is written behind the compiler's back; the wait below retires them.
  auto gather_tail = [&](const uint32_t (&pc)[J], float *pbuf, uint32_t n) {
    const uint32_t dst0 = (uint32_t)(uintptr_t)pbuf + (uint32_t)pwave * 256u;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const uint32_t p = (uint32_t)(j * kFwProducers + ptid);
      const uint32_t off = pc[j] << 2;
      const uint32_t dst = dst0 + (uint32_t)j * (kFwProducers * 4u);
      if (p < n && pc[j] >= nhot) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(off), "s"(dst), "s"(wres)
                     : "memory");
      }
    }
  };
  auto hot_products = [&](const uint32_t (&pc)[J], const float (&pv)[J], float *pbuf, uint32_t n) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const uint32_t p = (uint32_t)(j * kFwProducers + ptid);
      if (p < n && pc[j] < nhot) pbuf[p] = HASVAL ? hot[pc[j]] * pv[j] : hot[pc[j]];
    }
  };
  // sets with values: the gathered weights are multiplied in place once they have landed
  auto scale_tail = [&](const uint32_t (&pc)[J], const float (&pv)[J], float *pbuf, uint32_t n) {
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const uint32_t p = (uint32_t)(j * kFwProducers + ptid);
      if (p < n && pc[j] >= nhot) pbuf[p] *= pv[j];
    }
  };
  auto chunk_len = [&](unsigned long long B, unsigned long long E) {
    return (uint32_t)(E - B < (unsigned long long)kFwChunk ? E - B : (unsigned long long)kFwChunk);
  };

  // this workgroup's tiles: blockIdx.x + i * gridDim.x
  const uint64_t nmine = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  auto tile_of = [&](uint64_t i) { return blockIdx.x + i * (uint64_t)gridDim.x; };
  auto tile_rows = [&](uint64_t t) {
    const uint64_t rb = a.r0 + t * rpt;
    return (int)((a.r1 - rb) < (uint64_t)rpt ? (a.r1 - rb) : (uint64_t)rpt);
  };
  // park the row data of the first tiles in their LDS slots
  for (uint64_t i = 0; i < 4 && i < nmine; ++i) {
    const int slot = (int)(i % kFwSlots);
    const uint64_t rb = a.r0 + tile_of(i) * rpt;
    const int nr = tile_rows(tile_of(i));
    if (tid <= nr) rp0[slot * (kFwRowSlots + 2) + tid] = a.rowptr[rb + tid];
    if (tid < kFwRowSlots) {
      mask0[slot * kFwRowSlots + tid] = (dense && tid < nr) ? a.hmask[rb + tid] : 0ull;
      lab0[slot * kFwRowSlots + tid] = tid < nr ? a.label[rb + tid] : 0.0f;
    }
  }
  if (dense && nmine > 0) {
    const int nr = tile_rows(tile_of(0));
    if (tid < nr * kWarmWords) warm0[tid] = a.hwarm[(a.r0 + tile_of(0) * rpt) * kWarmWords + tid];
  }
  __syncthreads();  // also orders the fill of hot[] / tab[] before their first use
  if (producer) {
    if (nmine > 0) {
      const unsigned long long B = rp0[0], E = rp0[tile_rows(tile_of(0))];
      const uint32_t n0 = chunk_len(B, E);
      issue_stream(c0, v0, B, E);
      gather_tail(c0, prod0, n0);
      hot_products(c0, v0, prod0, n0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (HASVAL) scale_tail(c0, v0, prod0, n0);
    }
    if (nmine > 1) {
      const unsigned long long *rp = rp0 + 1 * (kFwRowSlots + 2);
      issue_stream(c1, v1, rp[0], rp[tile_rows(tile_of(1))]);
    }
    if (nmine > 2) {
      const unsigned long long *rp = rp0 + 2 * (kFwRowSlots + 2);
      issue_stream(c2, v2, rp[0], rp[tile_rows(tile_of(2))]);
      wait_all_but_newest_stream();  // tile 1 has arrived when step 0 starts, tile 2 stays in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  // Row data (row pointers, masks, labels, warm mask words) is copied from global memory straight
  // into its LDS slot by LDS-direct loads, 64 consecutive dwords per wave-instruction: no register
  // is involved, so nothing makes the compiler wait for these loads, and the producers' wait at
  // the end of a step covers them (they are older than the stream that stays in flight).
  auto dma64 = [&](const void *src, uint32_t nbytes, const void *lds_dst, int chunk) {
    const uint32_t off = (uint32_t)(chunk * 64 + lane) * 4u;
    if (off < nbytes) {
      const uint64_t base = reinterpret_cast<uint64_t>(src);
      const u32x4 res = {(uint32_t)base, (uint32_t)(base >> 32) & 0xFFFFu, nbytes, 0x00020000u};
      const uint32_t dst = (uint32_t)(uintptr_t)lds_dst + (uint32_t)chunk * 256u;
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(off), "s"(dst), "s"(res)
                   : "memory");
    }
  };
  __syncthreads();
  // The two roles run separate tile loops (same trip count, one barrier per tile each), so the
  // registers a producer keeps across tiles are not live in the consumers' code and vice versa.
  // The role is wave-uniform; readfirstlane lets the compiler branch on a scalar.
#define FW_TILE_HEAD \
    const int slot = (int)(i % kFwSlots), slot1 = (int)((i + 1) % kFwSlots), slot3 = (int)((i + 3) % kFwSlots), \
              slot4 = (int)((i + 4) % kFwSlots); \
    const unsigned long long *rp = rp0 + slot * (kFwRowSlots + 2); \
    const unsigned long long *mask_s = mask0 + slot * kFwRowSlots; \
    float *prod = prod0 + (i & 1) * kFwChunk; \
    const uint64_t tile = tile_of(i); \
    const int nr = tile_rows(tile); \
    const unsigned long long B = rp[0], E = rp[nr]; \
    const unsigned long long cb0 = B; \
    const bool has1 = i + 1 < nmine, has3 = i + 3 < nmine, has4 = i + 4 < nmine;
  // One producer step: while the consumers reduce tile i, make the products of tile i+1 (held in
  // the X registers) and stream tile i+3 into the Z registers (which held tile i).
  const bool gather_first = pwave < kFwProducers / 128;
  auto produce = [&](uint64_t i, uint32_t (&pxc)[J], float (&pxv)[J], uint32_t (&pzc)[J], float (&pzv)[J]) {
    FW_TILE_HEAD
    (void)mask_s;
    (void)prod;
    // per-row results of the previous tile (the consumers left them in LDS: a store from a
    // consumer wave would queue behind everything the producers have in the memory pipe)
    if (i > 0) {
      const uint64_t ptile = tile_of(i - 1);
      const int ph = (int)((i - 1) & 1) * kFwRowSlots;
      if (ptid < tile_rows(ptile)) {
        a.out[a.r0 + ptile * rpt + ptid] = out_s0[ph + ptid];
        if (MODE == FWD_LBFGS && a.out2) a.out2[a.r0 + ptile * rpt + ptid] = out2_s0[ph + ptid];
      }
    }
    // row data of the tile four ahead into its slot: 5 wave-loads of row pointers, 4 of masks, 2 of
    // labels, dealt over the 8 producer waves (rows past the tile's end keep stale words, which no
    // consumer reads)
    if (has4) {
      const uint64_t rb4 = a.r0 + tile_of(i + 4) * rpt;
      const uint32_t nr4 = (uint32_t)tile_rows(tile_of(i + 4));
#pragma unroll
      for (int k0 = 0; k0 < 16; k0 += kFwProducers / 64) {
        const int k = k0 + pwave;
        if (k < 5) dma64(a.rowptr + rb4, (nr4 + 1) * 8u, rp0 + slot4 * (kFwRowSlots + 2), k);
        else if (k < 9) { if (dense) dma64(a.hmask + rb4, nr4 * 8u, mask0 + slot4 * kFwRowSlots, k - 5); }
        else if (k < 11) dma64(a.label + rb4, nr4 * 4u, lab0 + slot4 * kFwRowSlots, k - 9);
      }
    }
    // warm mask words of the next tile (the consumers need them one step from now): 12 wave-loads
    if (dense && has1) {
      const uint64_t rb1 = a.r0 + tile_of(i + 1) * rpt;
      const uint32_t nr1 = (uint32_t)tile_rows(tile_of(i + 1));
#pragma unroll
      for (int k0 = 0; k0 < 16; k0 += kFwProducers / 64) {
        const int k = k0 + pwave;
        if (k < 2 * kWarmWords * kFwRowSlots / 64)
          dma64(a.hwarm + rb1 * kWarmWords, nr1 * (8u * kWarmWords), warm0 + ((i + 1) & 1) * kFwRowSlots * kWarmWords, k);
      }
    }
    float *prodn = prod0 + ((i + 1) & 1) * kFwChunk;
    uint32_t n1 = 0;
    if (has1) {
      const unsigned long long *rp1 = rp0 + slot1 * (kFwRowSlots + 2);
      n1 = chunk_len(rp1[0], rp1[tile_rows(tile_of(i + 1))]);
      // Half of the producer waves (one of the two on each SIMD) issue their gathers first and
      // write the products of the cached columns afterwards, the other half the other way round:
      // the memory pipe takes gathers from one half while the other half works in LDS.
      if (gather_first) {
        gather_tail(pxc, prodn, n1);
        hot_products(pxc, pxv, prodn, n1);
      } else {
        hot_products(pxc, pxv, prodn, n1);
        gather_tail(pxc, prodn, n1);
      }
    }
    // the stream of tile i+3 goes out last, so that the wait below can leave it in flight: it has
    // this step's remainder and the two steps after it to arrive
    if (has3) {
      const unsigned long long *rp3 = rp0 + slot3 * (kFwRowSlots + 2);
      issue_stream(pzc, pzv, rp3[0], rp3[tile_rows(tile_of(i + 3))]);
    }
    // tiles whose rows are so long that they do not fit one chunk take extra passes by the
    // consumers, each bracketed by two barriers that both roles execute
    for (unsigned long long cb = cb0 + kFwChunk; cb < E; cb += kFwChunk) {
      __syncthreads();
      __syncthreads();
    }
    // the gathers have landed in LDS after this wait, and so have the row data fetched above and
    // the stream of tile i+2 (issued a step ago); only the stream issued just now stays in flight
    if (has3) wait_all_but_newest_stream();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (HASVAL && has1) scale_tail(pxc, pxv, prodn, n1);
    __syncthreads();
  };
  if (__builtin_amdgcn_readfirstlane(tid) >= kFwConsumers) {
    // three steps per trip with the register sets rotated: no copies of registers that loads in
    // flight are still writing
    for (uint64_t i = 0; i < nmine; i += 3) {
      produce(i, c1, v1, c0, v0);
      if (i + 1 < nmine) produce(i + 1, c2, v2, c1, v1);
      if (i + 2 < nmine) produce(i + 2, c0, v0, c2, v2);
    }
  } else {
  for (uint64_t i = 0; i < nmine; ++i) {
    FW_TILE_HEAD
    (void)slot1;
    (void)slot3;
    (void)slot4;
    (void)has1;
    (void)has3;
    (void)has4;
    {
      // ---- consumers --------------------------------------------------------------------------
      float rowsum = 0.0f;
      {
        // lane q of a row adds every second product of the row, four per trip; a read past the
        // row's end stays inside LDS and is discarded
        const unsigned long long myb = trow < nr ? rp[trow] : E, mye = trow < nr ? rp[trow + 1] : E;
        const uint32_t s = (uint32_t)(myb - cb0);
        const uint32_t e = (uint32_t)((mye < cb0 + kFwChunk ? mye : cb0 + kFwChunk) - cb0);
        for (uint32_t k = s + q; k < e; k += 4 * kFwLanesPerRow) {
          const float v0 = prod[k], v1 = prod[k + kFwLanesPerRow], v2 = prod[k + 2 * kFwLanesPerRow],
                      v3 = prod[k + 3 * kFwLanesPerRow];
          rowsum += v0;
          rowsum += k + kFwLanesPerRow < e ? v1 : 0.0f;
          rowsum += k + 2 * kFwLanesPerRow < e ? v2 : 0.0f;
          rowsum += k + 3 * kFwLanesPerRow < e ? v3 : 0.0f;
        }
      }
      for (unsigned long long cb = cb0 + kFwChunk; cb < E; cb += kFwChunk) {
        // the rest of an oversized tile goes through the same half of the product buffer, chunk
        // by chunk, with plain loads
        __syncthreads();
        const unsigned long long ce = E < cb + kFwChunk ? E : cb + kFwChunk;
        for (unsigned long long k = cb + tid; k < ce; k += kFwConsumers) {
          const uint32_t c = a.col[k];
          const float v = hasval ? a.val[k] : 1.0f;
          prod[k - cb] = (c < nhot ? hot[c] : wsum[c]) * v;
        }
        __syncthreads();
        const unsigned long long myb = trow < nr ? rp[trow] : E, mye = trow < nr ? rp[trow + 1] : E;
        const unsigned long long s = myb > cb ? myb : cb;
        const unsigned long long e = mye < ce ? mye : ce;
        for (unsigned long long k = s + q; k < e; k += kFwLanesPerRow) rowsum += prod[k - cb];
      }
      if (dense && trow < nr) {
        const unsigned long long m = mask_s[trow];
        constexpr int kBytesPerLane = 8 / kFwLanesPerRow;
#pragma unroll
        for (int b8 = 0; b8 < kBytesPerLane; ++b8) {
          const int g = q * kBytesPerLane + b8;
          rowsum += tab[g * 256 + (int)((m >> (8 * g)) & 255)];
        }
      }
      // warm columns: the row's lanes take every second (fourth) bit of its warm words each and walk
      // their set bits, two LDS reads per trip
      static_assert(kFwLanesPerRow == 2 || kFwLanesPerRow == 4, "every second or every fourth bit per lane");
      unsigned long long my_w[kWarmWords];
#pragma unroll
      for (int w = 0; w < kWarmWords; ++w) {
        const unsigned long long half = (kFwLanesPerRow == 2 ? 0x5555555555555555ull : 0x1111111111111111ull) << q;
        my_w[w] = (dense && trow < nr) ? (warm0[((i & 1) * kFwRowSlots + trow) * kWarmWords + w] & half) : 0ull;
        unsigned long long m = my_w[w];
        while (m) {
          const float v0 = hot[64 * (w + 1) + __builtin_ctzll(m)];
          m &= m - 1;
          const float v1 = m ? hot[64 * (w + 1) + __builtin_ctzll(m)] : 0.0f;
          m &= m - 1;
          rowsum += v0;
          rowsum += v1;
        }
      }
#pragma unroll
      for (int off = 1; off < kFwLanesPerRow; off <<= 1) rowsum += __shfl_xor(rowsum, off, 64);
      // Epilogue on the first lane of each row; everything a consumer wave needs afterwards (its
      // 32 rows' residuals and masks) stays inside the wave, so the consumers need no barrier
      // of their own.
      unsigned long long my_r = 0, my_r2 = 0;
      const unsigned long long my_m = (dense && trow < nr) ? mask_s[trow] : 0ull;
      float *out_s = out_s0 + (int)(i & 1) * kFwRowSlots, *out2_s = out2_s0 + (int)(i & 1) * kFwRowSlots;
      if (q == 0 && trow < nr) {
        const float m = rowsum;
        const float lab = lab0[slot * kFwRowSlots + trow];
        if (MODE == FWD_MARGIN) {
          out_s[trow] = m;
        } else if (MODE == FWD_EVAL) {
          // src/metrics.cc:104-105: double predict, -log p / -log(1-p)
          const double pd = 1.0f / (1 + exp(-(double)clamp35(m)));
          out_s[trow] = (float)pd;
          acc0 += lab > 0 ? -log(pd) : -log(1 - pd);
        } else if (MODE == FWD_LBFGS) {
          const float t = lab > 0 ? 1.0f : 0.0f;
          const float loss = log1pf(expf(-fabsf(m))) + (t > 0 ? fmaxf(-m, 0.0f) : fmaxf(m, 0.0f));
          const float res = 1.0f / (1.0f + expf(-m)) - t;
          out_s[trow] = res;
          out2_s[trow] = m;
          my_r = (unsigned long long)(long long)(res * a.scale);
          acc0 += (double)loss;
          acc1 += (double)res;
        } else {
          // src/ftrl.cc:72,76-78
          const float p = (float)(1.0 / (1 + exp(-(double)clamp35(m))));
          const int y = lab > 0 ? 1 : 0;
          const float g = p - y;
          out_s[trow] = g;
          my_r = (unsigned long long)(long long)(g * a.scale);
          my_r2 = (unsigned long long)(long long)(g * g * a.scale) | 1ull;  // non-zero marks "touched"
          acc0 += (double)g;
          acc1 += (double)(g * g);
        }
      }
      if (colsums) {
        // X^T r restricted to the mask columns: row r of this wave lives in lane r * lanes-per-row;
        // lane k adds the rows whose mask has bit k
        constexpr int kRowsPerWave = 64 / kFwLanesPerRow;
        const int r0w = wave * kRowsPerWave;
        const int nrw = nr - r0w < kRowsPerWave ? nr - r0w : kRowsPerWave;
        // (a row's mask and residual are broadcast with v_readlane: scalar results, no LDS trip;
        // the scalar mask then is the lane predicate itself)
        auto bcast = [](unsigned long long v, int src) {
          const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
          const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
          return ((unsigned long long)hi << 32) | lo;
        };
        for (int r = 0; r < nrw; ++r) {
          const unsigned long long m = bcast(my_m, r * kFwLanesPerRow);
          const unsigned long long rv = bcast(my_r, r * kFwLanesPerRow);
          const bool on = __builtin_amdgcn_inverse_ballot_w64(m);  // lane k <- bit k of the scalar mask
          h1 += on ? rv : 0ull;
          if (MODE == FWD_FTRL) {
            const unsigned long long rv2 = bcast(my_r2, r * kFwLanesPerRow);
            h2 += on ? rv2 : 0ull;
          }
        }
        // warm columns: few set bits per row, added to per-column LDS accumulators
        const unsigned long long rrow = __shfl(my_r, lane & ~(kFwLanesPerRow - 1), 64);
        const unsigned long long rrow2 = MODE == FWD_FTRL ? __shfl(my_r2, lane & ~(kFwLanesPerRow - 1), 64) : 0ull;
#pragma unroll
        for (int w = 0; w < kWarmWords; ++w) {
          for (unsigned long long m = my_w[w]; m; m &= m - 1) {
            const int c = 64 * w + __builtin_ctzll(m);
            atomicAdd(&wacc[c], rrow);
            if (MODE == FWD_FTRL) atomicAdd(&wacc2[c], rrow2);
          }
        }
      }
    }
    __syncthreads();
  }
  }
#undef FW_TILE_HEAD
  if (producer && nmine > 0) {
    const uint64_t ptile = tile_of(nmine - 1);
    const int ph = (int)((nmine - 1) & 1) * kFwRowSlots;
    if (ptid < tile_rows(ptile)) {
      a.out[a.r0 + ptile * rpt + ptid] = out_s0[ph + ptid];
      if (MODE == FWD_LBFGS && a.out2) a.out2[a.r0 + ptile * rpt + ptid] = out2_s0[ph + ptid];
    }
  }
  if (colsums) {
    __syncthreads();
    if (tid < 64 * kWarmWords) {
      const uint32_t o = a.perm[64 + tid];
      if (wacc[tid]) atomicAdd(a.g1 + o, wacc[tid]);
      if (MODE == FWD_FTRL && wacc2[tid]) atomicAdd(a.g2 + o, wacc2[tid]);
    }
    // combine the waves' column sums and add them to the global accumulators
    for (int pass = 0; pass < (MODE == FWD_FTRL ? 2 : 1); ++pass) {
      __syncthreads();
      hsum[wave][lane] = pass == 0 ? h1 : h2;
      __syncthreads();
      if (tid < 64) {
        unsigned long long t = 0;
        for (int w8 = 0; w8 < kFwThreads / 64; ++w8) t += hsum[w8][tid];
        if (t) atomicAdd((pass == 0 ? a.g1 : a.g2) + a.perm[tid], t);
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    acc0 += __shfl_down(acc0, off, 64);
    acc1 += __shfl_down(acc1, off, 64);
  }
  __syncthreads();
  if (lane == 0) shred[wave] = acc0;
  __syncthreads();
  if (tid == 0) {
    double t = 0;
    for (int w8 = 0; w8 < kFwThreads / 64; ++w8) t += shred[w8];
    a.partials[0 * kMaxPartialBlocks + blockIdx.x] = t;
  }
  __syncthreads();
  if (lane == 0) shred[wave] = acc1;
  __syncthreads();
  if (tid == 0) {
    double t = 0;
    for (int w8 = 0; w8 < kFwThreads / 64; ++w8) t += shred[w8];
    a.partials[1 * kMaxPartialBlocks + blockIdx.x] = t;
  }
}


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

// wsum[j] = sum over the weight vectors of w_g[perm[j]]  (perm == nullptr: identity)
__global__ void __launch_bounds__(256) k_wsum(const float *__restrict__ w0, const float *__restrict__ w1,
                                              const float *__restrict__ w2, const float *__restrict__ w3,
                                              const uint32_t *__restrict__ perm, float *__restrict__ wsum,
                                              uint64_t dim) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < dim; j += stride) {
    const uint64_t o = perm ? perm[j] : j;
    float s = w0[o];
    if (w1) s += w1[o];
    if (w2) s += w2[o];
    if (w3) s += w3[o];
    wsum[j] = s;
  }
}

// whole-set sweep over the row-panel copy
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
  admm_ctx *ctx = csr->ctx;
  ADMM_REQUIRE(nw >= 0 && nw <= 4, "forward sweep: up to 4 weight vectors, got %d", nw);
  ADMM_REQUIRE(r0 <= r1 && r1 <= csr->nrows, "forward sweep: bad row range");
  if (!csr->d_wsum) ADMM_HIP_CHECK(hipMalloc(&csr->d_wsum, sizeof(float) * csr->dim));
  if (csr->has_fp && nw > 0 && mode != FWD_FTRL && r0 == 0 && r1 == csr->nrows)
    return launch_forward_panels(csr, mode, r0, r1, w, nw, out_rows, nblocks_out, margins_out);
  // a mini-batch of the FTRL pass: a range of whole panels, weights already in csr->d_wsum
  if (csr->has_fp && nw == 0 && mode == FWD_FTRL && r0 < r1 && r0 % csr->fp_rows == 0 &&
      (r1 % csr->fp_rows == 0 || r1 == csr->nrows))
    return launch_forward_panels(csr, mode, r0, r1, w, nw, out_rows, nblocks_out, margins_out);
  ADMM_TRY(require_rows(csr, "forward sweep over row ranges"));
  if (nw > 0) {  // nw == 0: the caller has filled csr->d_wsum itself
    KernelTimer t(ctx, ADMM_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_wsum, dim3(grid_for(csr->dim, 256, ctx->cu_count * 8)), dim3(256), 0, ctx->stream, w[0],
                       nw > 1 ? w[1] : nullptr, nw > 2 ? w[2] : nullptr, nw > 3 ? w[3] : nullptr, csr->d_perm,
                       csr->d_wsum, csr->dim);
    ADMM_HIP_CHECK(hipGetLastError());
  }
  FwdArgs a{};
  a.rowptr = csr->hot_dense ? csr->d_crowptr : csr->d_rowptr;
  a.col = csr->hot_dense ? csr->d_ccol : csr->d_col;
  a.val = csr->hot_dense ? csr->d_cval : csr->d_val;
  a.label = csr->d_label;
  if (csr->hot_dense) {
    a.hmask = csr->d_hmask;
    a.hwarm = csr->d_hwarm;
    a.perm = csr->d_perm;
    if (mode != FWD_EVAL && mode != FWD_MARGIN) {
      ADMM_REQUIRE(csr->has_t, "forward sweep: the training forms need admm_csr_build_transposed");
      a.g1 = reinterpret_cast<unsigned long long *>(csr->d_g1);
      a.g2 = reinterpret_cast<unsigned long long *>(csr->d_g2);
      a.scale = std::ldexp(1.0f, csr->fx_shift);
    }
  }
  a.r0 = r0;
  a.r1 = r1;
  a.w[0] = csr->d_wsum;
  a.nw = 1;
  a.out = out_rows;
  a.out2 = margins_out;
  a.partials = ctx->d_partials;
  a.mode = (int)mode;
  a.nhot = (uint32_t)std::min<uint64_t>(kFwHot, csr->dim);
  a.dim = (uint32_t)csr->dim;
  static bool raised[64] = {};  // per device: function attributes belong to a device's copy of the code
  if (!raised[ctx->device & 63]) {
    for (const void *fn : {reinterpret_cast<const void *>(&k_forward<FWD_EVAL, true>),
                           reinterpret_cast<const void *>(&k_forward<FWD_LBFGS, true>),
                           reinterpret_cast<const void *>(&k_forward<FWD_FTRL, true>),
                           reinterpret_cast<const void *>(&k_forward<FWD_EVAL, false>),
                           reinterpret_cast<const void *>(&k_forward<FWD_LBFGS, false>),
                           reinterpret_cast<const void *>(&k_forward<FWD_FTRL, false>),
                           reinterpret_cast<const void *>(&k_forward<FWD_MARGIN, true>),
                           reinterpret_cast<const void *>(&k_forward<FWD_MARGIN, false>)})
      ADMM_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwLds));
    raised[ctx->device & 63] = true;
  }
  // rows per tile: as many as normally fit one chunk of stored entries, with some headroom
  {
    const uint64_t entries = csr->hot_dense ? csr->cnnz : csr->nnz;
    const double avg = csr->nrows ? (double)entries / (double)csr->nrows : 1.0;
    const double fit = 0.94 * kFwChunk / std::max(avg, 1.0);
    a.rows_per_tile = (uint32_t)std::min<double>(kFwRowSlots, std::max(8.0, std::floor(fit / 8.0) * 8.0));
  }
  const uint64_t ntiles = (r1 - r0 + a.rows_per_tile - 1) / a.rows_per_tile;
  const bool hasval = a.val != nullptr;
  const int threads = hasval ? fw_threads(true) : fw_threads(false);
  const int grid = (int)std::max<uint64_t>(1, std::min<uint64_t>(ntiles, (uint64_t)ctx->cu_count));  // one workgroup per CU
  *nblocks_out = grid;
  KernelTimer t(ctx, mode == FWD_EVAL ? ADMM_K_FORWARD_EVAL : ADMM_K_FORWARD);
#define FW_LAUNCH(M)                                                                                       \
  if (hasval) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward<M, true>), dim3(grid), dim3(threads), kFwLds, ctx->stream, a); \
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_forward<M, false>), dim3(grid), dim3(threads), kFwLds, ctx->stream, a);
  if (mode == FWD_EVAL) {
    FW_LAUNCH(FWD_EVAL)
  } else if (mode == FWD_LBFGS) {
    FW_LAUNCH(FWD_LBFGS)
  } else if (mode == FWD_MARGIN) {
    FW_LAUNCH(FWD_MARGIN)
  } else {
    FW_LAUNCH(FWD_FTRL)
  }
#undef FW_LAUNCH
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
  unsigned long long *g1, *g2;  // 64-bit fixed point, value * scale
  float scale;
  uint32_t nblocks, col_block;
  uint64_t panel0, npanels;  // panels [panel0, panel0 + npanels) are swept by this launch
  int no_gather;             // calibration runs only: bit 0 skips the per-row gathers, bit 1 the global adds (results are then meaningless)
  // FTRL form over one panel (= one mini-batch): fz != nullptr applies the batch's update to the
  // cell's columns right here (the cell holds every entry the batch stores in them) instead of adding
  // the sums to g1 / g2 for a separate pass
  float *fw, *fz, *fn;
  const float *freg, *foff;
  const uint32_t *finv;      // caller's id -> stored id (nullptr: identity)
  float *fwsum;              // the forward sweep's gathered vector (stored order), kept current column by column
  float *fabsmax;            // [nblocks] running max |fwsum| per column block (each written by that block's workgroup only)
  float falpha, fbeta, fl1, fl2;
  double finv_scale;
  long long *fmpart;         // the mask columns' sums of the forward sweep (kMaskSlots copies), nullptr: none
  const double *fpartials;   // bias column of a set without masks: the forward sweep's per-workgroup sums
  int fbias_blocks;
  uint64_t fdim;
};


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

static int launch_transposed_impl(admm_csr *csr, bool ftrl_mode, uint64_t p0, uint64_t p1, const TArgs *fuse);

int launch_transposed(admm_csr *csr, bool ftrl_mode, uint64_t p0, uint64_t p1) {
  return launch_transposed_impl(csr, ftrl_mode, p0, p1, nullptr);
}

// fuse: the f* fields of a mini-batch's fused update (FTRL form, one panel), or nullptr
static int launch_transposed_impl(admm_csr *csr, bool ftrl_mode, uint64_t p0, uint64_t p1, const TArgs *fuse) {
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

// ---------------------------------------------------------------------------------------------
// mini-batch FTRL
// ---------------------------------------------------------------------------------------------


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
