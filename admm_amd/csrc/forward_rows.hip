// Forward sweep over the row-major form: k_forward (producer / consumer waves), any row range.
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

// the same sweep over the row-major form (the compact rows of a relabelled set): any row range
int launch_forward_rows(admm_csr *csr, FwdMode mode, uint64_t r0, uint64_t r1, const float *const *w, int nw,
                        float *out_rows, int *nblocks_out, float *margins_out) {
  admm_ctx *ctx = csr->ctx;
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

}  // namespace admm_impl
