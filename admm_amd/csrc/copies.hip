// The copies of a sample set that the sweeps read, built on the device: column-block cells and column stream
// (transposed sweep), row panels with hot stream and column-ordered lists (forward sweep), the order of the cells'
// entries, the previous-occurrence table of the sequential FTRL pass; and what a lean set frees when.
#include "internal.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>

using namespace admm_impl;

namespace admm_impl {

// ---- column-block copy ------------------------------------------------------------------------
//
// Rows are cut into panels of P rows, columns into blocks of CB columns.  Cell (p, b) holds the
// entries of panel p whose column lies in block b, as three parallel arrays (row u32, local
// column u16, value f32).  Inside a cell the entries of one build chunk (kChunkRows consecutive
// rows) are contiguous, so a sweep over a cell reads the per-row vector almost sequentially while
// its per-column accumulators live in LDS.  The bias column (col 0 of a bias-shifted set) is left
// out: it is present in every row and is reduced by the forward sweep instead.

constexpr int kChunkRows = 512;

struct TBuildArgs {
  const uint64_t *rowptr;
  const uint32_t *col;
  const uint32_t *perm;  // relabelled sets: stored id -> original id (the copy keeps original ids)
  const float *val;
  uint64_t nrows, panel_rows;
  uint32_t col_block, nblocks;
  int skip_bias;
  int hot_dense;  // entries held by the per-row masks (relabelled id < kMaskCols, value 1) are left out
  uint32_t stream_cols;  // entries whose stored id is below this are in the column stream, not in the cells
  unsigned long long *cell_count;   // pass 1: entries per cell
  unsigned long long *cell_cursor;  // pass 2: next free slot per cell (starts at tptr[cell])
  uint32_t *trow;
  uint16_t *tcol;
  float *tval;
};

// LDS histogram over the column blocks of one chunk of rows; pass 1 adds it to the global cell
// counts, pass 2 reserves a contiguous range per (chunk, block) and scatters the entries.
template <bool SCATTER>
__global__ void __launch_bounds__(256) k_tbuild(TBuildArgs a) {
  extern __shared__ uint32_t sh[];  // [nblocks] counts, then (SCATTER) [nblocks*2] bases lo/hi
  uint32_t *cnt = sh;
  const uint64_t chunks_per_panel = (a.panel_rows + kChunkRows - 1) / kChunkRows;
  const uint64_t nchunks_total = ((a.nrows + a.panel_rows - 1) / a.panel_rows) * chunks_per_panel;
  for (uint64_t chunk = blockIdx.x; chunk < nchunks_total; chunk += gridDim.x) {
    const uint64_t panel = chunk / chunks_per_panel;
    const uint64_t r0 = panel * a.panel_rows + (chunk % chunks_per_panel) * kChunkRows;
    uint64_t r1 = r0 + kChunkRows;
    const uint64_t pend = (panel + 1) * a.panel_rows;
    if (r1 > pend) r1 = pend;
    if (r1 > a.nrows) r1 = a.nrows;
    if (r0 >= r1) continue;  // uniform per block
    for (uint32_t i = threadIdx.x; i < a.nblocks; i += 256) cnt[i] = 0;
    __syncthreads();
    const uint64_t kb = a.rowptr[r0], ke = a.rowptr[r1];
    for (uint64_t k = kb + threadIdx.x; k < ke; k += 256) {
      uint32_t c = a.col[k];
      if (c < a.stream_cols) continue;
      if (a.hot_dense && c < kMaskCols && (!a.val || a.val[k] == 1.0f)) continue;
      if (a.perm) c = a.perm[c];
      if (a.skip_bias && c == 0) continue;
      atomicAdd(&cnt[c / a.col_block], 1u);
    }
    __syncthreads();
    if (!SCATTER) {
      for (uint32_t i = threadIdx.x; i < a.nblocks; i += 256)
        if (cnt[i]) atomicAdd(&a.cell_count[panel * a.nblocks + i], (unsigned long long)cnt[i]);
      __syncthreads();
    } else {
      unsigned long long *base = reinterpret_cast<unsigned long long *>(sh + ((a.nblocks + 1) & ~1u));
      for (uint32_t i = threadIdx.x; i < a.nblocks; i += 256) {
        base[i] = cnt[i] ? atomicAdd(&a.cell_cursor[panel * a.nblocks + i], (unsigned long long)cnt[i]) : 0ull;
        cnt[i] = 0;
      }
      __syncthreads();
      // entries of one row keep their row id; find it by walking rows per thread group:
      // each 16-lane group takes rows r0+g, r0+g+16, ...
      const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
      for (uint64_t r = r0 + grp; r < r1; r += 16) {
        const uint64_t b = a.rowptr[r], e = a.rowptr[r + 1];
        for (uint64_t k = b + sub; k < e; k += 16) {
          uint32_t c = a.col[k];
          if (c < a.stream_cols) continue;
          if (a.hot_dense && c < kMaskCols && (!a.val || a.val[k] == 1.0f)) continue;
          if (a.perm) c = a.perm[c];
          if (a.skip_bias && c == 0) continue;
          const uint32_t blk = c / a.col_block;
          const uint32_t slot = atomicAdd(&cnt[blk], 1u);
          const uint64_t pos = base[blk] + slot;
          a.trow[pos] = (uint32_t)r;
          a.tcol[pos] = (uint16_t)(c - blk * a.col_block);
          if (a.tval) a.tval[pos] = a.val[k];
        }
      }
      __syncthreads();
    }
  }
}

__global__ void __launch_bounds__(256) k_absmax(const float *__restrict__ v, uint64_t n, double *__restrict__ partials) {
  __shared__ float sh[4];
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  float m = 0.0f;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float a = fabsf(v[i]);
    m = (a > m || a != a) ? a : m;  // NaN propagates
  }
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_down(m, off, 64);
    m = (o > m || o != o) ? o : m;
  }
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = sh[0];
    for (int i = 1; i < 4; ++i) r = (sh[i] > r || sh[i] != sh[i]) ? sh[i] : r;
    partials[blockIdx.x] = (double)r;
  }
}

// ---- row-panel copy for the forward sweep -----------------------------------------------------
//
// The entries outside the hot stream.  One workgroup per panel: a counting sort of the panel's
// entries by column bin in LDS (count, exclusive scan, scatter), so that 64 consecutive entries
// gather their weights from a few adjacent cache lines instead of 64 scattered ones.

constexpr int kFpBuildThreads = 1024;
constexpr uint32_t kFpMaxBins = 16384;

struct FpBuildArgs {
  const uint64_t *xptr;   // row pointers of the source form (compact or plain CSR)
  const uint32_t *col;    // its column ids
  const float *val;       // nullptr = implicit 1
  uint64_t nrows;
  uint32_t panel_rows, bin_shift, nbins;
  int skip_bias;          // leave out column 0
  uint32_t nhot;          // ids below this are in the hot stream, not here
  const uint64_t *fptr;   // per panel: where its entries start in fcol / frow / fval
  uint32_t *fcol;
  uint16_t *frow;
  float *fval;
};

__global__ void __launch_bounds__(kFpBuildThreads) k_fpbuild(FpBuildArgs a) {
  extern __shared__ uint32_t fp_sh[];  // [nbins] counts / cursors, [kFpBuildThreads] scan scratch
  uint32_t *cnt = fp_sh;
  uint32_t *scan = fp_sh + a.nbins;
  const uint64_t npanels = (a.nrows + a.panel_rows - 1) / a.panel_rows;
  const int tid = threadIdx.x;
  for (uint64_t p = blockIdx.x; p < npanels; p += gridDim.x) {
    const uint64_t r0 = p * a.panel_rows;
    const uint64_t r1 = r0 + a.panel_rows < a.nrows ? r0 + a.panel_rows : a.nrows;
    const uint64_t kb = a.xptr[r0], ke = a.xptr[r1];
    const uint64_t out0 = a.fptr[p];
    for (uint32_t i = tid; i < a.nbins; i += kFpBuildThreads) cnt[i] = 0;
    __syncthreads();
    for (uint64_t k = kb + tid; k < ke; k += kFpBuildThreads) {
      const uint32_t c = a.col[k];
      if (c < a.nhot || (a.skip_bias && c == 0)) continue;
      atomicAdd(&cnt[c >> a.bin_shift], 1u);
    }
    __syncthreads();
    // exclusive scan of cnt[0 .. nbins): every thread owns nbins / threads consecutive bins
    const uint32_t per = (a.nbins + kFpBuildThreads - 1) / kFpBuildThreads;
    uint32_t mine = 0;
    for (uint32_t j = 0; j < per; ++j) {
      const uint32_t i = tid * per + j;
      if (i < a.nbins) mine += cnt[i];
    }
    scan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < kFpBuildThreads; off <<= 1) {
      const uint32_t add = tid >= off ? scan[tid - off] : 0u;
      __syncthreads();
      scan[tid] += add;
      __syncthreads();
    }
    uint32_t run = scan[tid] - mine;
    for (uint32_t j = 0; j < per; ++j) {
      const uint32_t i = tid * per + j;
      if (i < a.nbins) {
        const uint32_t c = cnt[i];
        cnt[i] = run;  // becomes the bin's cursor
        run += c;
      }
    }
    __syncthreads();
    // scatter: a 16-lane group walks a row, so that a row's entries keep their order inside a bin
    // as far as the atomics allow (any order inside a bin is valid)
    const int sub = tid & 15, grp = tid >> 4;
    for (uint64_t r = r0 + grp; r < r1; r += kFpBuildThreads / 16) {
      const uint64_t b = a.xptr[r], e = a.xptr[r + 1];
      for (uint64_t k = b + sub; k < e; k += 16) {
        const uint32_t c = a.col[k];
        if (c < a.nhot || (a.skip_bias && c == 0)) continue;
        const uint64_t pos = out0 + atomicAdd(&cnt[c >> a.bin_shift], 1u);
        a.fcol[pos] = c;
        a.frow[pos] = (uint16_t)(r - r0);
        if (a.fval) a.fval[pos] = a.val[k];
      }
    }
    __syncthreads();
  }
}

// Second pass over the row-panel copy: inside every bin, order the entries by column.  Adjacent
// lanes of the forward sweep then gather from the same or the next cache line (the address unit
// takes a wave-wide gather four lanes at a time and pays per distinct line among the four).
// One workgroup per panel: it recounts the bins from the copy itself, then every wave takes bins in
// turn, one entry per lane, ranks them by all-pairs comparison and stores them back in order (all
// loads of a wave-instruction have landed before any lane's store issues: in place is safe).  Bins
// of more than 64 entries keep their order: any order inside a bin is valid.
__global__ void __launch_bounds__(kFpBuildThreads) k_fpsort(FpBuildArgs a) {
  extern __shared__ uint32_t fp_sh[];
  uint32_t *cnt = fp_sh;               // [nbins] counts, then bin starts
  uint32_t *scan = fp_sh + a.nbins;
  const uint64_t npanels = (a.nrows + a.panel_rows - 1) / a.panel_rows;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (uint64_t p = blockIdx.x; p < npanels; p += gridDim.x) {
    const uint64_t out0 = a.fptr[p], out1 = a.fptr[p + 1];
    for (uint32_t i = tid; i < a.nbins; i += kFpBuildThreads) cnt[i] = 0;
    __syncthreads();
    for (uint64_t k = out0 + tid; k < out1; k += kFpBuildThreads) atomicAdd(&cnt[a.fcol[k] >> a.bin_shift], 1u);
    __syncthreads();
    const uint32_t per = (a.nbins + kFpBuildThreads - 1) / kFpBuildThreads;
    uint32_t mine = 0;
    for (uint32_t j = 0; j < per; ++j) {
      const uint32_t i = tid * per + j;
      if (i < a.nbins) mine += cnt[i];
    }
    scan[tid] = mine;
    __syncthreads();
    for (int off = 1; off < kFpBuildThreads; off <<= 1) {
      const uint32_t add = tid >= off ? scan[tid - off] : 0u;
      __syncthreads();
      scan[tid] += add;
      __syncthreads();
    }
    uint32_t run = scan[tid] - mine;
    for (uint32_t j = 0; j < per; ++j) {
      const uint32_t i = tid * per + j;
      if (i < a.nbins) {
        const uint32_t c = cnt[i];
        cnt[i] = run;  // the bin's start
        run += c;
      }
    }
    __syncthreads();
    const uint32_t total = (uint32_t)(out1 - out0);
    for (uint32_t b = wave; b < a.nbins; b += kFpBuildThreads / 64) {
      const uint32_t s = cnt[b], e = b + 1 < a.nbins ? cnt[b + 1] : total;
      const uint32_t n = e - s;
      if (n < 2 || n > 64) continue;  // wave-uniform
      const bool on = (uint32_t)lane < n;
      const uint64_t at = out0 + s + (on ? lane : 0);
      const uint32_t c = on ? a.fcol[at] : 0xFFFFFFFFu;
      const uint32_t q = a.frow[at];
      const float v = a.fval ? a.fval[at] : 0.0f;
      uint32_t rank = 0;
      for (uint32_t j = 0; j < n; ++j) {
        const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)j);
        rank += (cj < c || (cj == c && j < (uint32_t)lane)) ? 1u : 0u;
      }
      if (on) {
        a.fcol[out0 + s + rank] = c;
        a.frow[out0 + s + rank] = (uint16_t)q;
        if (a.fval) a.fval[out0 + s + rank] = v;
      }
    }
    __syncthreads();
  }
}

// Bit transpose of the first mask word in tiles of 64 rows: out[64 t + c] bit k = in[64 t + k] bit c.
// One wavefront per tile; rows past the end count as empty.
__global__ void __launch_bounds__(256) k_mask_transpose(const uint64_t *__restrict__ hmask, uint64_t nrows,
                                                        uint64_t *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const uint64_t ntiles = (nrows + 63) / 64;
  const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * 256) >> 6;
  for (uint64_t t = wave; t < ntiles; t += nwaves) {
    const uint64_t r = t * 64 + lane;
    const unsigned long long m = r < nrows ? hmask[r] : 0ull;
    unsigned long long mine = 0;
    for (int c = 0; c < 64; ++c) {
      const unsigned long long b = __ballot((m >> c) & 1ull);
      if (lane == c) mine = b;
    }
    out[t * 64 + lane] = mine;
  }
}

// ---- hot stream of the row-panel copy ---------------------------------------------------------

struct HpBuildArgs {
  const uint64_t *xptr;
  const uint32_t *col;
  const float *val;
  uint64_t nrows;
  uint32_t panel_rows, nhot;
  int skip_bias;
  uint32_t *hlen, *clen;   // count pass: per panel, words of the hot stream / entries outside it
  const uint64_t *hbase;   // fill pass
  uint16_t *hcol;
  float *hval;
  uint16_t *hfirst;
};

constexpr int kHpBuildThreads = 1024;

// per panel: how many words its hot stream has (one per entry below nhot, one filler per row
// without any) and how many entries stay outside it
__global__ void __launch_bounds__(kHpBuildThreads) k_hpcount(HpBuildArgs a) {
  __shared__ uint32_t sh[2];
  const uint64_t npanels = (a.nrows + a.panel_rows - 1) / a.panel_rows;
  const int tid = threadIdx.x;
  for (uint64_t p = blockIdx.x; p < npanels; p += gridDim.x) {
    const uint64_t r0 = p * a.panel_rows;
    const uint64_t r1 = r0 + a.panel_rows < a.nrows ? r0 + a.panel_rows : a.nrows;
    if (tid < 2) sh[tid] = 0;
    __syncthreads();
    uint32_t hw = 0, cw = 0;
    for (uint64_t r = r0 + tid; r < r1; r += kHpBuildThreads) {
      uint32_t h = 0;
      for (uint64_t k = a.xptr[r], e = a.xptr[r + 1]; k < e; ++k) {
        const uint32_t c = a.col[k];
        if (a.skip_bias && c == 0) continue;
        if (c < a.nhot) ++h;
        else ++cw;
      }
      hw += h ? h : 1u;
    }
    atomicAdd(&sh[0], hw);
    atomicAdd(&sh[1], cw);
    __syncthreads();
    if (tid == 0) {
      a.hlen[p] = sh[0];
      a.clen[p] = sh[1];
    }
    __syncthreads();
  }
}

// writes the hot stream of every panel in the lane-interleaved layout, and the row each lane starts in
__global__ void __launch_bounds__(kHpBuildThreads) k_hpfill(HpBuildArgs a) {
  extern __shared__ uint32_t hp_sh[];  // [panel_rows + 1] word offsets of the rows, [kHpBuildThreads] scan scratch
  uint32_t *off = hp_sh;
  uint32_t *scan = hp_sh + a.panel_rows + 1;
  const uint64_t npanels = (a.nrows + a.panel_rows - 1) / a.panel_rows;
  const int tid = threadIdx.x;
  const uint32_t per = (a.panel_rows + kHpBuildThreads - 1) / kHpBuildThreads;  // consecutive rows per thread
  for (uint64_t p = blockIdx.x; p < npanels; p += gridDim.x) {
    const uint64_t r0 = p * a.panel_rows;
    const uint32_t nr = (uint32_t)(a.nrows - r0 < a.panel_rows ? a.nrows - r0 : a.panel_rows);
    const uint64_t hb = a.hbase[p];
    const uint32_t S = (uint32_t)((a.hbase[p + 1] - hb) / kHpLanes);  // words per lane, a multiple of 4
    uint32_t mine = 0;
    for (uint32_t j = 0; j < per; ++j) {
      const uint32_t r = tid * per + j;
      if (r < nr) {
        uint32_t h = 0;
        for (uint64_t k = a.xptr[r0 + r], e = a.xptr[r0 + r + 1]; k < e; ++k) {
          const uint32_t c = a.col[k];
          if (c < a.nhot && !(a.skip_bias && c == 0)) ++h;
        }
        h = h ? h : 1u;
        off[r] = h;
        mine += h;
      }
    }
    scan[tid] = mine;
    __syncthreads();
    for (int o = 1; o < kHpBuildThreads; o <<= 1) {
      const uint32_t add = tid >= o ? scan[tid - o] : 0u;
      __syncthreads();
      scan[tid] += add;
      __syncthreads();
    }
    const uint32_t total = scan[kHpBuildThreads - 1];
    uint32_t run = scan[tid] - mine;
    for (uint32_t j = 0; j < per; ++j) {
      const uint32_t r = tid * per + j;
      if (r < nr) {
        const uint32_t h = off[r];
        off[r] = run;
        run += h;
      }
    }
    if (tid == 0) off[nr] = total;
    __syncthreads();
    auto slot = [&](uint32_t i) -> uint64_t {
      const uint32_t t = i / S, k = i - t * S;
      return hb + ((uint64_t)(k >> 2) * kHpLanes + t) * 4 + (k & 3u);
    };
    for (uint32_t j = 0; j < per; ++j) {
      const uint32_t r = tid * per + j;
      if (r >= nr) continue;
      const uint32_t first = off[r], count = off[r + 1] - first;
      uint32_t h = 0;
      for (uint64_t k = a.xptr[r0 + r], e = a.xptr[r0 + r + 1]; k < e; ++k) {
        const uint32_t c = a.col[k];
        if (c < a.nhot && !(a.skip_bias && c == 0)) {
          const uint64_t at = slot(first + h);
          ++h;
          a.hcol[at] = (uint16_t)(c | (h == count ? 0x8000u : 0u));
          if (a.hval) a.hval[at] = a.val[k];
        }
      }
      if (h == 0) {  // a row without hot entries: one filler word that ends the row
        const uint64_t at = slot(first);
        a.hcol[at] = (uint16_t)(a.nhot | 0x8000u);
        if (a.hval) a.hval[at] = 0.0f;
      }
    }
    for (uint32_t i = total + tid; i < S * kHpLanes; i += kHpBuildThreads) {  // the lanes' runs are padded to equal length
      const uint64_t at = slot(i);
      a.hcol[at] = (uint16_t)a.nhot;
      if (a.hval) a.hval[at] = 0.0f;
    }
    if (tid < kHpLanes) {
      // the row that holds word tid * S: the last row whose offset is <= that word
      const uint32_t pos = tid * S;
      uint32_t lo = 0, hi = nr;  // off[lo] <= pos < off[hi] once pos < total
      if (pos >= total) lo = 0;
      else
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (off[mid] <= pos) lo = mid;
          else hi = mid;
        }
      a.hfirst[p * kHpLanes + tid] = (uint16_t)lo;
    }
    __syncthreads();
  }
}

// ---- column stream of the transposed sweep -----------------------------------------------------

struct TcBuildArgs {
  const uint64_t *xptr;   // source form: compact rows of a relabelled set, else the plain rows
  const uint32_t *col;
  const float *val;
  uint64_t nrows;
  uint32_t ncols;         // stored ids below this go to the stream
  int skip_bias;
  uint32_t *len;          // count pass: [npanels] words, then [npanels] entries
  const uint64_t *tbase;  // fill pass
  uint16_t *tw;
  float *tv;
  uint16_t *tfirst;
};

// One workgroup per panel of kTcRows rows: counts the panel's entries per column, lays the columns
// out one after another (an empty column takes one filler word) and scatters the entries.
template <bool FILL>
__global__ void __launch_bounds__(kTcLanes) k_tcbuild(TcBuildArgs a) {
  extern __shared__ uint32_t tc_sh[];  // [ncols] counts / cursors, [ncols + 1] word offsets, [kTcLanes] scan scratch
  uint32_t *cnt = tc_sh;
  uint32_t *off = tc_sh + a.ncols;
  uint32_t *scan = off + a.ncols + 1;
  const uint64_t npanels = (a.nrows + kTcRows - 1) / kTcRows;
  const int tid = threadIdx.x;
  const uint32_t per = (a.ncols + kTcLanes - 1) / kTcLanes;  // consecutive columns per thread
  for (uint64_t p = blockIdx.x; p < npanels; p += gridDim.x) {
    const uint64_t r0 = p * kTcRows;
    const uint32_t nr = (uint32_t)(a.nrows - r0 < kTcRows ? a.nrows - r0 : kTcRows);
    for (uint32_t c = tid; c < a.ncols; c += kTcLanes) cnt[c] = 0;
    __syncthreads();
    for (uint32_t r = tid; r < nr; r += kTcLanes)
      for (uint64_t k = a.xptr[r0 + r], e = a.xptr[r0 + r + 1]; k < e; ++k) {
        const uint32_t c = a.col[k];
        if (c < a.ncols && !(a.skip_bias && c == 0)) atomicAdd(&cnt[c], 1u);
      }
    __syncthreads();
    uint32_t mine = 0, mine_entries = 0;
    for (uint32_t j = 0; j < per; ++j) {
      const uint32_t c = tid * per + j;
      if (c < a.ncols) {
        mine += cnt[c] ? cnt[c] : 1u;
        mine_entries += cnt[c];
      }
    }
    scan[tid] = mine;
    __syncthreads();
    for (int o = 1; o < kTcLanes; o <<= 1) {
      const uint32_t add = tid >= o ? scan[tid - o] : 0u;
      __syncthreads();
      scan[tid] += add;
      __syncthreads();
    }
    const uint32_t total = scan[kTcLanes - 1];
    if (!FILL) {
      if (tid == 0) {
        a.len[p] = total;
        a.len[npanels + p] = 0;
      }
      __syncthreads();
      if (mine_entries) atomicAdd(&a.len[npanels + p], mine_entries);
      __syncthreads();
      continue;
    }
    const uint64_t tb = a.tbase[p];
    const uint32_t S = (uint32_t)((a.tbase[p + 1] - tb) / kTcLanes);  // words per lane, a multiple of 4
    auto slot = [&](uint32_t i) -> uint64_t {
      const uint32_t t = i / S, k = i - t * S;
      return tb + ((uint64_t)(k >> 2) * kTcLanes + t) * 4 + (k & 3u);
    };
    uint32_t run = scan[tid] - mine;
    for (uint32_t j = 0; j < per; ++j) {
      const uint32_t c = tid * per + j;
      if (c < a.ncols) {
        off[c] = run;
        if (cnt[c] == 0) {  // a column without entries in this panel: one filler word that ends the column
          const uint64_t at = slot(run);
          a.tw[at] = (uint16_t)(kTcRows | 0x8000u);
          if (a.tv) a.tv[at] = 0.0f;
          run += 1;
        } else {
          run += cnt[c];
          cnt[c] = 0;  // becomes the column's cursor
        }
      }
    }
    if (tid == 0) off[a.ncols] = total;
    __syncthreads();
    for (uint32_t r = tid; r < nr; r += kTcLanes)
      for (uint64_t k = a.xptr[r0 + r], e = a.xptr[r0 + r + 1]; k < e; ++k) {
        const uint32_t c = a.col[k];
        if (c < a.ncols && !(a.skip_bias && c == 0)) {
          const uint32_t i = atomicAdd(&cnt[c], 1u), n = off[c + 1] - off[c];
          const uint64_t at = slot(off[c] + i);
          a.tw[at] = (uint16_t)(r | (i + 1 == n ? 0x8000u : 0u));
          if (a.tv) a.tv[at] = a.val[k];
        }
      }
    for (uint32_t i = total + tid; i < S * kTcLanes; i += kTcLanes) {  // the lanes' runs are padded to equal length
      const uint64_t at = slot(i);
      a.tw[at] = (uint16_t)kTcRows;
      if (a.tv) a.tv[at] = 0.0f;
    }
    {
      // the column that holds word tid * S: the last column whose offset is <= that word
      const uint32_t pos = tid * S;
      uint32_t lo = 0, hi = a.ncols;
      if (pos < total)
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (off[mid] <= pos) lo = mid;
          else hi = mid;
        }
      a.tfirst[p * kTcLanes + tid] = (uint16_t)lo;
    }
    __syncthreads();
  }
}

int build_column_stream(admm_csr *csr, bool want) {
  admm_ctx *ctx = csr->ctx;
  for (void **p : {(void **)&csr->d_tcw, (void **)&csr->d_tcv, (void **)&csr->d_tcbase, (void **)&csr->d_tcfirst}) {
    if (*p) hipFree(*p);
    *p = nullptr;
  }
  csr->has_tc = false;
  csr->tc_cols = 0;
  csr->tc_words = csr->tc_entries = 0;
  if (!want || csr->nrows == 0 || csr->nnz == 0) return ADMM_OK;
  const uint32_t ncols = (uint32_t)std::min<uint64_t>(kFpHot, csr->dim);
  const uint64_t npanels = (csr->nrows + kTcRows - 1) / kTcRows;
  const int grid = (int)std::min<uint64_t>(npanels, (uint64_t)ctx->cu_count * 2);
  const size_t lds = sizeof(uint32_t) * (2 * (size_t)ncols + 1 + kTcLanes);
  static bool raised[64] = {};
  if (!raised[ctx->device & 63]) {
    for (const void *fn : {reinterpret_cast<const void *>(&k_tcbuild<false>), reinterpret_cast<const void *>(&k_tcbuild<true>)})
      ADMM_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(sizeof(uint32_t) * (2 * (size_t)kFpHot + 1 + kTcLanes))));
    raised[ctx->device & 63] = true;
  }
  TcBuildArgs a{};
  a.xptr = csr->hot_dense ? csr->d_crowptr : csr->d_rowptr;
  a.col = csr->hot_dense ? csr->d_ccol : csr->d_col;
  a.val = csr->hot_dense ? csr->d_cval : csr->d_val;
  a.nrows = csr->nrows;
  a.ncols = ncols;
  a.skip_bias = (csr->has_bias && !csr->hot_dense) ? 1 : 0;
  uint32_t *d_len = nullptr;
  ADMM_HIP_CHECK(hipMalloc(&d_len, sizeof(uint32_t) * 2 * npanels));
  a.len = d_len;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tcbuild<false>), dim3(grid), dim3(kTcLanes), lds, ctx->stream, a);
  std::vector<uint32_t> len(2 * npanels);
  hipError_t ce = hipGetLastError();
  if (ce == hipSuccess)
    ce = hipMemcpyAsync(len.data(), d_len, sizeof(uint32_t) * 2 * npanels, hipMemcpyDeviceToHost, ctx->stream);
  if (ce == hipSuccess) ce = hipStreamSynchronize(ctx->stream);
  hipFree(d_len);
  ADMM_HIP_CHECK(ce);
  std::vector<uint64_t> tbase(npanels + 1);
  tbase[0] = 0;
  uint64_t entries = 0;
  for (uint64_t p = 0; p < npanels; ++p) {
    const uint64_t per_lane = ((uint64_t)len[p] + kTcLanes - 1) / kTcLanes;
    tbase[p + 1] = tbase[p] + ((per_lane + 3) & ~3ull) * kTcLanes;
    entries += len[npanels + p];
  }
  const uint64_t words = tbase[npanels];
  ADMM_HIP_CHECK(hipMalloc(&csr->d_tcbase, sizeof(uint64_t) * (npanels + 1)));
  ADMM_HIP_CHECK(hipMemcpyAsync(csr->d_tcbase, tbase.data(), sizeof(uint64_t) * (npanels + 1), hipMemcpyHostToDevice, ctx->stream));
  ADMM_HIP_CHECK(hipMalloc(&csr->d_tcw, sizeof(uint16_t) * (words + 8)));
  if (a.val) ADMM_HIP_CHECK(hipMalloc(&csr->d_tcv, sizeof(float) * (words + 8)));
  ADMM_HIP_CHECK(hipMalloc(&csr->d_tcfirst, sizeof(uint16_t) * npanels * kTcLanes));
  a.len = nullptr;
  a.tbase = csr->d_tcbase;
  a.tw = csr->d_tcw;
  a.tv = csr->d_tcv;
  a.tfirst = csr->d_tcfirst;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tcbuild<true>), dim3(grid), dim3(kTcLanes), lds, ctx->stream, a);
  ADMM_HIP_CHECK(hipGetLastError());
  ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  csr->tc_cols = ncols;
  csr->tc_words = words;
  csr->tc_entries = entries;
  csr->has_tc = true;
  if (const char *vb = getenv("ADMM_VERBOSE"); vb && vb[0] != '0')
    fprintf(stderr, "[admm_hip] column stream: %llu words for %llu entries (ids < %u)\n", (unsigned long long)words,
            (unsigned long long)entries, ncols);
  return ADMM_OK;
}

__global__ void __launch_bounds__(256) k_maxrowlen(const uint64_t *__restrict__ rowptr, uint64_t nrows,
                                                   double *__restrict__ partials) {
  __shared__ unsigned long long sh[4];
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  unsigned long long m = 0;
  for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < nrows; r += stride) {
    const unsigned long long len = rowptr[r + 1] - rowptr[r];
    m = len > m ? len : m;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_down(m, off, 64);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long r = sh[0];
    for (int i = 1; i < 4; ++i) r = sh[i] > r ? sh[i] : r;
    partials[blockIdx.x] = (double)r;
  }
}

__global__ void __launch_bounds__(256) k_iota_cells(uint32_t *v, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) v[i] = (uint32_t)i;
}

// ---- order of the entries inside the cells of the column-block copy ----------------------------
//
// The builder leaves a cell's entries grouped by runs of 512 rows, in no particular order inside a
// run.  Any order gives the same sums; the one that makes the sweep's gathers of the per-row vector
// cheap is: rows ascending, and inside every aligned group of 256 entries (what one wavefront takes
// with its 16-byte loads, four entries per lane) entry 64 c + l at position 4 l + c -- so that
// component c of the 64 lanes is 64 consecutive entries, adjacent lanes gather adjacent rows, and
// the address unit, which takes a gather four lanes at a time, finds them in the same line.

__global__ void __launch_bounds__(256) k_cell_offsets(const uint64_t *__restrict__ tptr, uint64_t first_cell, uint32_t nblocks,
                                                      uint32_t *__restrict__ off) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i <= nblocks) off[i] = (uint32_t)(tptr[first_cell + i] - tptr[first_cell]);
}

__global__ void __launch_bounds__(256) k_cell_reorder(const uint32_t *__restrict__ sorted_row, const uint32_t *__restrict__ src,
                                                      const uint16_t *__restrict__ col_old, const float *__restrict__ val_old,
                                                      const uint32_t *__restrict__ off, uint32_t nblocks, uint64_t e0,
                                                      uint32_t n, uint32_t *__restrict__ trow, uint16_t *__restrict__ tcol,
                                                      float *__restrict__ tval) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    // the cell that holds panel-relative position i
    uint32_t lo = 0, hi = nblocks;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (off[mid] <= i) lo = mid;
      else hi = mid;
    }
    const uint64_t beg = e0 + off[lo], end = e0 + off[lo + 1];
    const uint64_t body0 = (beg + 3) & ~3ull, body1 = end & ~3ull;
    uint64_t k = e0 + i;
    if (k >= body0 && body0 < body1) {
      const uint64_t rel = k - body0, chunk = rel >> 8;
      if (body0 + ((chunk + 1) << 8) <= body1) {
        const uint32_t u = (uint32_t)(rel & 255u);
        k = body0 + (chunk << 8) + 4 * (u & 63u) + (u >> 6);
      }
    }
    const uint32_t from = src[i];
    trow[k] = sorted_row[i];
    tcol[k] = col_old[from];
    if (tval) tval[k] = val_old[from];
  }
}

static int sort_cells(admm_csr *csr) {
  admm_ctx *ctx = csr->ctx;
  {
    const char *env = getenv("ADMM_CELL_SORT");  // 0: leave the cells as the builder wrote them
    if (env && env[0] == '0') return ADMM_OK;
  }
  if (csr->tnnz == 0) return ADMM_OK;
  uint64_t cap = 0;
  for (uint64_t p = 0; p < csr->npanels; ++p)
    cap = std::max(cap, csr->h_tptr[(p + 1) * csr->nblocks] - csr->h_tptr[p * csr->nblocks]);
  if (cap == 0 || cap >= 0x7FFFFFFFull) return ADMM_OK;  // a panel beyond the sort's index range keeps its order
  int bits = 1;
  while ((1ull << bits) < csr->nrows && bits < 32) ++bits;
  uint32_t *keys = nullptr, *idx_in = nullptr, *idx_out = nullptr, *off = nullptr;
  uint16_t *col_old = nullptr;
  float *val_old = nullptr;
  void *tmp = nullptr;
  size_t tmp_bytes = 0;
  auto cleanup = [&]() {
    for (void *q : {(void *)keys, (void *)idx_in, (void *)idx_out, (void *)off, (void *)col_old, (void *)val_old, tmp})
      if (q) hipFree(q);
  };
#define SC_CHECK(expr)                                                            \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      set_error("admm_csr_build_transposed (cell order): %s: %s", #expr, hipGetErrorString(e_)); \
      cleanup();                                                                  \
      return e_ == hipErrorOutOfMemory ? ADMM_ERR_NOMEM : ADMM_ERR_HIP;           \
    }                                                                             \
  } while (0)
  SC_CHECK(hipMalloc(&keys, sizeof(uint32_t) * cap));
  SC_CHECK(hipMalloc(&idx_in, sizeof(uint32_t) * cap));
  SC_CHECK(hipMalloc(&idx_out, sizeof(uint32_t) * cap));
  SC_CHECK(hipMalloc(&off, sizeof(uint32_t) * (csr->nblocks + 1)));
  SC_CHECK(hipMalloc(&col_old, sizeof(uint16_t) * cap));
  if (csr->d_tval) SC_CHECK(hipMalloc(&val_old, sizeof(float) * cap));
  SC_CHECK(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, tmp_bytes, csr->d_trow, keys, idx_in, idx_out, (int)cap,
                                                       (int)csr->nblocks, off, off + 1, 0, bits, ctx->stream));
  SC_CHECK(hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16)));
  hipLaunchKernelGGL(k_iota_cells, dim3(grid_for(cap, 256, ctx->cu_count * 8)), dim3(256), 0, ctx->stream, idx_in, cap);
  for (uint64_t p = 0; p < csr->npanels; ++p) {
    const uint64_t first = p * csr->nblocks, e0 = csr->h_tptr[first];
    const uint32_t n = (uint32_t)(csr->h_tptr[first + csr->nblocks] - e0);
    if (n == 0) continue;
    hipLaunchKernelGGL(k_cell_offsets, dim3((csr->nblocks + 256) / 256), dim3(256), 0, ctx->stream, csr->d_tptr, first,
                       csr->nblocks, off);
    size_t bytes = tmp_bytes;
    SC_CHECK(hipcub::DeviceSegmentedRadixSort::SortPairs(tmp, bytes, csr->d_trow + e0, keys, idx_in, idx_out, (int)n,
                                                         (int)csr->nblocks, off, off + 1, 0, bits, ctx->stream));
    SC_CHECK(hipMemcpyAsync(col_old, csr->d_tcol + e0, sizeof(uint16_t) * n, hipMemcpyDeviceToDevice, ctx->stream));
    if (val_old) SC_CHECK(hipMemcpyAsync(val_old, csr->d_tval + e0, sizeof(float) * n, hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_cell_reorder, dim3(grid_for(n, 256, ctx->cu_count * 8)), dim3(256), 0, ctx->stream, keys, idx_out,
                       col_old, val_old, off, csr->nblocks, e0, n, csr->d_trow, csr->d_tcol, csr->d_tval);
    SC_CHECK(hipGetLastError());
  }
  SC_CHECK(hipStreamSynchronize(ctx->stream));
#undef SC_CHECK
  cleanup();
  return ADMM_OK;
}

// ---- previous-occurrence table of the sequential FTRL pass -------------------------------------
//
// dep[k] = (rows back << 16) | position in that row, of the last earlier row that stores entry k's
// column; 0 when there is none within 65534 rows (or its position does not fit 16 bits).  Built on
// the device: a stable sort of (column, entry index) pairs puts every entry next to the previous
// entry of its column; rows are handled in chunks with a halo of 65535 rows before each.

__global__ void __launch_bounds__(256) k_iota_from(uint32_t *v, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) v[i] = (uint32_t)i;
}

__device__ inline uint64_t dep_row_of(const uint64_t *__restrict__ rowptr, uint64_t lo, uint64_t hi, uint64_t k) {
  // last row r in [lo, hi) with rowptr[r] <= k
  while (hi - lo > 1) {
    const uint64_t mid = (lo + hi) >> 1;
    if (rowptr[mid] <= k) lo = mid;
    else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(256) k_dep_from_sorted(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ idx,
                                                         uint64_t n, uint64_t e0, uint64_t first_own,
                                                         const uint64_t *__restrict__ rowptr, uint64_t row_lo, uint64_t row_hi,
                                                         uint32_t *__restrict__ dep) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const uint64_t k = e0 + idx[i];
    if (k < first_own) continue;  // a halo entry: its own chunk has written it
    uint32_t d = 0;
    if (i > 0 && keys[i - 1] == keys[i]) {
      const uint64_t kp = e0 + idx[i - 1];
      const uint64_t r = dep_row_of(rowptr, row_lo, row_hi, k), rp = dep_row_of(rowptr, row_lo, row_hi, kp);
      const uint64_t back = r - rp, pos = kp - rowptr[rp];
      if (back < 0xFFFFull && pos < 0x10000ull) d = ((uint32_t)back << 16) | (uint32_t)pos;
    }
    dep[k] = d;
  }
}

__global__ void __launch_bounds__(256) k_map_cols(const uint32_t *__restrict__ col, const uint32_t *__restrict__ perm,
                                                  uint32_t *__restrict__ out, uint64_t nnz) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nnz; i += stride) out[i] = perm[col[i]];
}

int build_dep(admm_csr *csr) {
  if (csr->d_dep) return ADMM_OK;
  admm_ctx *ctx = csr->ctx;
  const uint64_t nrows = csr->nrows, nnz = csr->nnz;
  {  // longest row
    const int g = grid_for(std::max<uint64_t>(nrows, 1), 256, ctx->cu_count * 8);
    hipLaunchKernelGGL(k_maxrowlen, dim3(g), dim3(256), 0, ctx->stream, csr->d_rowptr, nrows, ctx->d_partials);
    ADMM_HIP_CHECK(hipGetLastError());
    std::vector<double> hp(g);
    ADMM_HIP_CHECK(hipMemcpyAsync(hp.data(), ctx->d_partials, sizeof(double) * g, hipMemcpyDeviceToHost, ctx->stream));
    ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    double m = 0;
    for (double v : hp) m = std::max(m, v);
    csr->max_row_len = (uint64_t)m;
  }
  if (csr->d_perm && nnz) {  // relabelled set: the pass addresses the state by original ids; translate once
    ADMM_HIP_CHECK(hipMalloc(&csr->d_seqcol, sizeof(uint32_t) * (nnz + 8)));
    hipLaunchKernelGGL(k_map_cols, dim3(grid_for(nnz, 256, ctx->cu_count * 16)), dim3(256), 0, ctx->stream, csr->d_col,
                       csr->d_perm, csr->d_seqcol, nnz);
    ADMM_HIP_CHECK(hipGetLastError());
  }
  ADMM_HIP_CHECK(hipMalloc(&csr->d_dep, sizeof(uint32_t) * (nnz + 8)));
  if (nnz == 0) return ADMM_OK;
  const uint64_t halo = 0xFFFF;
  const uint64_t avg = std::max<uint64_t>(1, nnz / std::max<uint64_t>(nrows, 1));
  uint64_t chunk_rows = std::max<uint64_t>(4 * halo, (192ull << 20) / avg);  // about 2e8 entries per sort
  if (const char *env = getenv("ADMM_DEP_CHUNK_ROWS")) chunk_rows = std::max<uint64_t>(1, std::strtoull(env, nullptr, 10));  // test aid
  int bits = 1;
  while ((1ull << bits) < csr->dim) ++bits;
  uint32_t *keys_out = nullptr, *idx_in = nullptr, *idx_out = nullptr;
  void *tmp = nullptr;
  size_t tmp_bytes = 0;
  uint64_t cap = 0;
  auto cleanup = [&]() {
    for (void *p : {(void *)keys_out, (void *)idx_in, (void *)idx_out, tmp})
      if (p) hipFree(p);
    keys_out = idx_in = idx_out = nullptr;
    tmp = nullptr;
  };
  auto fail = [&](hipError_t e, const char *what) {
    set_error("sequential FTRL pass, previous-occurrence table: %s: %s", what, hipGetErrorString(e));
    cleanup();
    hipFree(csr->d_dep);
    csr->d_dep = nullptr;
    return e == hipErrorOutOfMemory ? ADMM_ERR_NOMEM : ADMM_ERR_HIP;
  };
  for (uint64_t c0 = 0; c0 < nrows; c0 += chunk_rows) {
    const uint64_t c1 = std::min(nrows, c0 + chunk_rows), h0 = c0 > halo ? c0 - halo : 0;
    uint64_t edge[3];
    hipError_t e = hipMemcpyAsync(&edge[0], csr->d_rowptr + h0, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&edge[1], csr->d_rowptr + c0, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&edge[2], csr->d_rowptr + c1, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(e, "reading row pointers");
    const uint64_t e0 = edge[0], n = edge[2] - edge[0];
    if (n == 0) continue;
    if (n >= 0xFFFFFFFFull) return fail(hipErrorInvalidValue, "more than 2^32 entries in one chunk of rows");
    if (n > cap) {
      cleanup();
      cap = n;
      if ((e = hipMalloc(&keys_out, sizeof(uint32_t) * cap)) != hipSuccess) return fail(e, "hipMalloc");
      if ((e = hipMalloc(&idx_in, sizeof(uint32_t) * cap)) != hipSuccess) return fail(e, "hipMalloc");
      if ((e = hipMalloc(&idx_out, sizeof(uint32_t) * cap)) != hipSuccess) return fail(e, "hipMalloc");
      tmp_bytes = 0;
      e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, csr->d_col, keys_out, idx_in, idx_out, (int64_t)cap, 0, bits,
                                             ctx->stream);
      if (e != hipSuccess) return fail(e, "sort size query");
      if ((e = hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16))) != hipSuccess) return fail(e, "hipMalloc");
    }
    hipLaunchKernelGGL(k_iota_from, dim3(grid_for(n, 256, ctx->cu_count * 8)), dim3(256), 0, ctx->stream, idx_in, n);
    size_t bytes = tmp_bytes;
    e = hipcub::DeviceRadixSort::SortPairs(tmp, bytes, csr->d_col + e0, keys_out, idx_in, idx_out, (int64_t)n, 0, bits, ctx->stream);
    if (e != hipSuccess) return fail(e, "sort");
    hipLaunchKernelGGL(k_dep_from_sorted, dim3(grid_for(n, 256, ctx->cu_count * 8)), dim3(256), 0, ctx->stream, keys_out,
                       idx_out, n, e0, edge[1], csr->d_rowptr, h0, c1, csr->d_dep);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "launch");
  }
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return fail(e, "synchronize");
  cleanup();
  return ADMM_OK;
}

int require_rows(const admm_csr *csr, const char *what) {
  if (!csr->rows_released) return ADMM_OK;
  set_error("%s: this sample set keeps only the forms its whole-set sweeps read (admm_csr_set_lean); its rows are gone", what);
  return ADMM_ERR_STATE;
}

// longest stored row (mask-held entries included), from the row pointers while they exist
static int note_max_row_len(admm_csr *csr) {
  admm_ctx *ctx = csr->ctx;
  if (!csr->d_rowptr || csr->nrows == 0) return ADMM_OK;
  const int g = grid_for(csr->nrows, 256, ctx->cu_count * 8);
  hipLaunchKernelGGL(k_maxrowlen, dim3(g), dim3(256), 0, ctx->stream, csr->d_rowptr, csr->nrows, ctx->d_partials);
  ADMM_HIP_CHECK(hipGetLastError());
  std::vector<double> hp(g);
  ADMM_HIP_CHECK(hipMemcpyAsync(hp.data(), ctx->d_partials, sizeof(double) * g, hipMemcpyDeviceToHost, ctx->stream));
  ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  double m = 1;
  for (double v : hp) m = std::max(m, v);
  csr->max_row_len = std::max<uint64_t>(csr->max_row_len, (uint64_t)m);
  return ADMM_OK;
}

// a lean set with row masks: the row-major CSR is not read again once the compact form exists ...
int release_row_major(admm_csr *csr) {
  if (!csr->lean || !csr->hot_dense || !csr->d_rowptr) return ADMM_OK;
  ADMM_TRY(note_max_row_len(csr));
  ADMM_HIP_CHECK(hipStreamSynchronize(csr->ctx->stream));
  for (void **p : {(void **)&csr->d_rowptr, (void **)&csr->d_col, (void **)&csr->d_val, (void **)&csr->d_dep, (void **)&csr->d_seqcol}) {
    if (*p) ADMM_HIP_CHECK(hipFree(*p));
    *p = nullptr;
  }
  csr->rows_released = true;
  return ADMM_OK;
}

// ... and the compact form's entries not once the row-panel and column-block copies exist (the masks stay)
int release_compact(admm_csr *csr) {
  if (!csr->lean || !csr->hot_dense || !csr->has_fp || !csr->has_t || !csr->rows_released) return ADMM_OK;
  ADMM_HIP_CHECK(hipStreamSynchronize(csr->ctx->stream));
  for (void **p : {(void **)&csr->d_crowptr, (void **)&csr->d_ccol, (void **)&csr->d_cval}) {
    if (*p) ADMM_HIP_CHECK(hipFree(*p));
    *p = nullptr;
  }
  return ADMM_OK;
}

int build_forward_panels(admm_csr *csr) {
  admm_ctx *ctx = csr->ctx;
  for (void **p : {(void **)&csr->d_fcol, (void **)&csr->d_frow, (void **)&csr->d_fval, (void **)&csr->d_fptr,
                   (void **)&csr->d_hcol, (void **)&csr->d_hval, (void **)&csr->d_hbase, (void **)&csr->d_hfirst,
                   (void **)&csr->d_hmaskT}) {
    if (*p) hipFree(*p);
    *p = nullptr;
  }
  csr->has_fp = false;
  if (csr->nrows == 0 || csr->nnz == 0) return ADMM_OK;
  {
    const char *env = getenv("ADMM_FORWARD_PANELS");  // 0: keep the row-major forward sweep everywhere
    if (env && env[0] == '0') return ADMM_OK;
  }
  // mini-batch copies: small panels, so that the forward sweep of one batch is a few hundred workgroups
  const uint32_t R = (csr->t_named && csr->panel_rows <= (1u << 18) && csr->panel_rows % 256 == 0) ? 256 : 4096;
  uint32_t nhot = (uint32_t)std::min<uint64_t>(kFpHot, csr->dim);
  if (const char *env = getenv("ADMM_FORWARD_HOT"))  // test aid: a smaller hot range, so that small sets have entries outside it
    nhot = (uint32_t)std::min<uint64_t>(nhot, std::strtoull(env, nullptr, 10));
  const bool skip_bias = csr->has_bias && !csr->hot_dense;
  // longest stored row (mask-held entries included): bounds a margin's magnitude
  ADMM_TRY(note_max_row_len(csr));
  ADMM_REQUIRE(csr->max_row_len > 0, "forward panels: the longest row is not known");
  csr->fp_lenbits = 0;
  while (std::ldexp(1.0, csr->fp_lenbits) < (double)csr->max_row_len + 1.0) ++csr->fp_lenbits;
  const uint64_t npanels = (csr->nrows + R - 1) / R;
  const int grid = (int)std::min<uint64_t>(npanels, (uint64_t)ctx->cu_count * 4);
  const float *src_val = csr->hot_dense ? csr->d_cval : csr->d_val;
  // ---- sizes: words of every panel's hot stream, entries outside it
  HpBuildArgs h{};
  h.xptr = csr->hot_dense ? csr->d_crowptr : csr->d_rowptr;
  h.col = csr->hot_dense ? csr->d_ccol : csr->d_col;
  h.val = src_val;
  h.nrows = csr->nrows;
  h.panel_rows = R;
  h.nhot = nhot;
  h.skip_bias = skip_bias ? 1 : 0;
  uint32_t *d_len = nullptr;
  ADMM_HIP_CHECK(hipMalloc(&d_len, sizeof(uint32_t) * 2 * npanels));
  h.hlen = d_len;
  h.clen = d_len + npanels;
  hipLaunchKernelGGL(k_hpcount, dim3(grid), dim3(kHpBuildThreads), 0, ctx->stream, h);
  std::vector<uint32_t> len(2 * npanels);
  hipError_t ce = hipGetLastError();
  if (ce == hipSuccess)
    ce = hipMemcpyAsync(len.data(), d_len, sizeof(uint32_t) * 2 * npanels, hipMemcpyDeviceToHost, ctx->stream);
  if (ce == hipSuccess) ce = hipStreamSynchronize(ctx->stream);
  hipFree(d_len);
  ADMM_HIP_CHECK(ce);
  std::vector<uint64_t> hbase(npanels + 1), fptr(npanels + 1);
  hbase[0] = fptr[0] = 0;
  for (uint64_t p = 0; p < npanels; ++p) {
    const uint64_t per_lane = ((uint64_t)len[p] + kHpLanes - 1) / kHpLanes;
    hbase[p + 1] = hbase[p] + ((per_lane + 3) & ~3ull) * kHpLanes;
    fptr[p + 1] = fptr[p] + len[npanels + p];
  }
  const uint64_t hwords = hbase[npanels], total = fptr[npanels];
  ADMM_HIP_CHECK(hipMalloc(&csr->d_hbase, sizeof(uint64_t) * (npanels + 1)));
  ADMM_HIP_CHECK(hipMalloc(&csr->d_fptr, sizeof(uint64_t) * (npanels + 1)));
  ADMM_HIP_CHECK(hipMemcpyAsync(csr->d_hbase, hbase.data(), sizeof(uint64_t) * (npanels + 1), hipMemcpyHostToDevice, ctx->stream));
  ADMM_HIP_CHECK(hipMemcpyAsync(csr->d_fptr, fptr.data(), sizeof(uint64_t) * (npanels + 1), hipMemcpyHostToDevice, ctx->stream));
  ADMM_HIP_CHECK(hipMalloc(&csr->d_hcol, sizeof(uint16_t) * (hwords + 8)));
  if (src_val) ADMM_HIP_CHECK(hipMalloc(&csr->d_hval, sizeof(float) * (hwords + 8)));
  ADMM_HIP_CHECK(hipMalloc(&csr->d_hfirst, sizeof(uint16_t) * npanels * kHpLanes));
  ADMM_HIP_CHECK(hipMalloc(&csr->d_fcol, sizeof(uint32_t) * (total + 8)));
  ADMM_HIP_CHECK(hipMalloc(&csr->d_frow, sizeof(uint16_t) * (total + 8)));
  if (src_val) ADMM_HIP_CHECK(hipMalloc(&csr->d_fval, sizeof(float) * (total + 8)));
  static bool raised[64] = {};
  if (!raised[ctx->device & 63]) {
    for (const void *fn : {reinterpret_cast<const void *>(&k_fpbuild), reinterpret_cast<const void *>(&k_fpsort)})
      ADMM_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)(sizeof(uint32_t) * (kFpMaxBins + kFpBuildThreads))));
    raised[ctx->device & 63] = true;
  }
  if (csr->hot_dense) {  // the first mask word by column, for the sweep's column sums
    const uint64_t ntiles = (csr->nrows + 63) / 64;
    ADMM_HIP_CHECK(hipMalloc(&csr->d_hmaskT, sizeof(uint64_t) * ntiles * 64));
    hipLaunchKernelGGL(k_mask_transpose, dim3(grid_for(ntiles * 64, 256, ctx->cu_count * 8)), dim3(256), 0, ctx->stream,
                       csr->d_hmask, csr->nrows, csr->d_hmaskT);
    ADMM_HIP_CHECK(hipGetLastError());
  }
  // ---- hot stream
  h.hlen = h.clen = nullptr;
  h.hbase = csr->d_hbase;
  h.hcol = csr->d_hcol;
  h.hval = csr->d_hval;
  h.hfirst = csr->d_hfirst;
  hipLaunchKernelGGL(k_hpfill, dim3(grid), dim3(kHpBuildThreads), sizeof(uint32_t) * (R + 1 + kHpBuildThreads), ctx->stream, h);
  ADMM_HIP_CHECK(hipGetLastError());
  // ---- the entries outside it, in column order
  uint32_t bin_shift = 0;
  while (((csr->dim + (1ull << bin_shift) - 1) >> bin_shift) > kFpMaxBins) ++bin_shift;
  const uint32_t nbins = (uint32_t)((csr->dim + (1ull << bin_shift) - 1) >> bin_shift);
  if (total > 0) {
    FpBuildArgs a{};
    a.xptr = h.xptr;
    a.col = h.col;
    a.val = src_val;
    a.nrows = csr->nrows;
    a.panel_rows = R;
    a.bin_shift = bin_shift;
    a.nbins = nbins;
    a.nhot = nhot;
    a.fptr = csr->d_fptr;
    a.skip_bias = skip_bias ? 1 : 0;
    a.fcol = csr->d_fcol;
    a.frow = csr->d_frow;
    a.fval = csr->d_fval;
    const size_t lds = sizeof(uint32_t) * (nbins + kFpBuildThreads);
    hipLaunchKernelGGL(k_fpbuild, dim3(grid), dim3(kFpBuildThreads), lds, ctx->stream, a);
    ADMM_HIP_CHECK(hipGetLastError());
    if (bin_shift > 0) {  // bins of one column are in order already
      hipLaunchKernelGGL(k_fpsort, dim3(grid), dim3(kFpBuildThreads), lds, ctx->stream, a);
      ADMM_HIP_CHECK(hipGetLastError());
    }
  }
  ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  csr->fp_rows = R;
  csr->fp_bin_cols = 1u << bin_shift;
  csr->fp_hot = nhot;
  csr->fp_nnz = total;
  csr->hp_words = hwords;
  csr->has_fp = true;
  if (const char *vb = getenv("ADMM_VERBOSE"); vb && vb[0] != '0')
    fprintf(stderr, "[admm_hip] forward panels: %llu rows, hot stream %llu words (ids < %u), %llu entries outside it\n",
            (unsigned long long)csr->nrows, (unsigned long long)hwords, nhot, (unsigned long long)total);
  return ADMM_OK;
}

int ensure_rowbuf(admm_csr *csr) {
  if (csr->d_rowbuf || csr->nrows == 0) return ADMM_OK;
  ADMM_HIP_CHECK(hipMalloc(&csr->d_rowbuf, sizeof(float) * csr->nrows));
  return ADMM_OK;
}

int ensure_taccum(admm_csr *csr) {
  if (!csr->d_gs) {
    ADMM_HIP_CHECK(hipMalloc(&csr->d_gs, sizeof(long long) * kFpHot));
    ADMM_HIP_CHECK(hipMemsetAsync(csr->d_gs, 0, sizeof(long long) * kFpHot, csr->ctx->stream));
  }
  if (csr->d_g1) return ADMM_OK;
  const uint64_t padded = (uint64_t)csr->nblocks * csr->col_block;
  ADMM_HIP_CHECK(hipMalloc(&csr->d_g1, sizeof(long long) * padded));
  ADMM_HIP_CHECK(hipMalloc(&csr->d_g2, sizeof(long long) * padded));
  ADMM_HIP_CHECK(hipMemsetAsync(csr->d_g1, 0, sizeof(long long) * padded, csr->ctx->stream));
  ADMM_HIP_CHECK(hipMemsetAsync(csr->d_g2, 0, sizeof(long long) * padded, csr->ctx->stream));
  return ADMM_OK;
}

}  // namespace admm_impl

extern "C" int admm_csr_build_transposed(admm_csr *csr, uint64_t panel_rows, uint32_t col_block) {
  ADMM_REQUIRE(csr, "admm_csr_build_transposed: csr is NULL");
  // automatic panels = the copy is for whole-set sweeps: its most frequent columns go to the column
  // stream.  A caller that names the panel size sweeps panel ranges (mini-batches), which read cells only.
  bool stream = panel_rows == 0;
  csr->t_named = panel_rows != 0;
  if (const char *env = getenv("ADMM_COLUMN_STREAM"))  // 0: never; 2: also with a named panel size (test aid)
    stream = env[0] != '0' && (stream || env[0] == '2');
  return admm_impl::build_transposed(csr, panel_rows, col_block, stream);
}

namespace admm_impl {
int build_transposed(admm_csr *csr, uint64_t panel_rows, uint32_t col_block, bool stream) {
  admm_ctx *ctx = csr->ctx;
  ADMM_REQUIRE(!(csr->rows_released && !csr->d_crowptr), "admm_csr_build_transposed: this lean sample set has been built already and "
               "no longer holds the rows a rebuild would read");
  if (col_block == 0) col_block = 8192;
  ADMM_REQUIRE(col_block <= 16384 && col_block >= 64 && (col_block % 64) == 0,
               "admm_csr_build_transposed: col_block must be a multiple of 64 in [64, 16384]");
  if (panel_rows == 0) {
    // about a million rows per panel, and a multiple of 8 panels when the set is large enough:
    // the sweep hands each of the 8 XCDs its own panels
    // (a set of up to 13M rows gets one group of eight larger panels: 12.5M rows swept 6 % faster
    // in 8 panels of 1.56M than in 16 of 0.78M; at 100M rows 1.04M beats 1.56M by 5 %)
    panel_rows = 1u << 20;
    if (csr->nrows > (1u << 20)) {
      const uint64_t limit = csr->nrows <= 13000000ull ? 1625000ull : (1ull << 20);
      uint64_t groups = 1;
      while ((csr->nrows + 8 * groups - 1) / (8 * groups) > limit) ++groups;
      panel_rows = (csr->nrows + 8 * groups - 1) / (8 * groups);
    }
  }
  ADMM_HIP_CHECK(hipSetDevice(ctx->device));
  ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  // drop a previous copy
  for (void **p : {(void **)&csr->d_tptr, (void **)&csr->d_trow, (void **)&csr->d_tcol, (void **)&csr->d_tval,
                   (void **)&csr->d_g1, (void **)&csr->d_g2}) {
    if (*p) hipFree(*p);
    *p = nullptr;
  }
  csr->has_t = false;
  csr->cache_valid = false;
  ADMM_TRY(build_column_stream(csr, stream));
  csr->panel_rows = panel_rows;
  csr->col_block = col_block;
  csr->nblocks = (uint32_t)((csr->dim + col_block - 1) / col_block);
  csr->npanels = csr->nrows ? (csr->nrows + panel_rows - 1) / panel_rows : 0;
  const uint64_t ncells = csr->npanels * csr->nblocks;
  ADMM_REQUIRE(csr->nblocks <= 8192, "admm_csr_build_transposed: too many column blocks (%u)", csr->nblocks);
  csr->h_tptr.assign(ncells + 1, 0);
  unsigned long long *d_cnt = nullptr;
  ADMM_HIP_CHECK(hipMalloc(&d_cnt, sizeof(unsigned long long) * (ncells + 1)));
  ADMM_HIP_CHECK(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long) * (ncells + 1), ctx->stream));
  TBuildArgs a{};
  // (a set with row masks is read in its compact form, which leaves out what the masks hold)
  a.rowptr = csr->hot_dense ? csr->d_crowptr : csr->d_rowptr;
  a.col = csr->hot_dense ? csr->d_ccol : csr->d_col;
  a.perm = csr->d_perm;
  a.val = csr->hot_dense ? csr->d_cval : csr->d_val;
  a.nrows = csr->nrows;
  a.panel_rows = panel_rows;
  a.col_block = col_block;
  a.nblocks = csr->nblocks;
  a.skip_bias = (csr->has_bias && !csr->hot_dense) ? 1 : 0;
  a.hot_dense = 0;
  a.stream_cols = csr->tc_cols;
  a.cell_count = d_cnt;
  const uint64_t chunks = csr->npanels * ((panel_rows + kChunkRows - 1) / kChunkRows);
  const int grid = grid_for(chunks * 256, 256, ctx->cu_count * 8);
  const size_t lds1 = sizeof(uint32_t) * csr->nblocks;
  const size_t lds2 = sizeof(uint32_t) * ((csr->nblocks + 1) & ~1u) + sizeof(unsigned long long) * csr->nblocks;
  if (csr->nrows) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tbuild<false>), dim3(grid), dim3(256), lds1, ctx->stream, a);
    ADMM_HIP_CHECK(hipGetLastError());
  }
  std::vector<unsigned long long> h_cnt(ncells + 1, 0);
  ADMM_HIP_CHECK(hipMemcpyAsync(h_cnt.data(), d_cnt, sizeof(unsigned long long) * ncells, hipMemcpyDeviceToHost, ctx->stream));
  ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  uint64_t run = 0;
  for (uint64_t i = 0; i < ncells; ++i) {
    csr->h_tptr[i] = run;
    run += h_cnt[i];
  }
  csr->h_tptr[ncells] = run;
  csr->tnnz = run;
  const uint64_t expect = (csr->hot_dense ? csr->cnnz : (csr->has_bias ? csr->nnz - csr->nrows : csr->nnz)) - csr->tc_entries;
  if (run != expect) {
    hipFree(d_cnt);
    set_error("admm_csr_build_transposed: counted %llu entries, expected %llu",
              (unsigned long long)run, (unsigned long long)expect);
    return ADMM_ERR_STATE;
  }
  ADMM_HIP_CHECK(hipMalloc(&csr->d_tptr, sizeof(uint64_t) * (ncells + 1)));
  ADMM_HIP_CHECK(hipMalloc(&csr->d_trow, sizeof(uint32_t) * (run + 8)));
  ADMM_HIP_CHECK(hipMalloc(&csr->d_tcol, sizeof(uint16_t) * (run + 8)));
  if (csr->has_val) ADMM_HIP_CHECK(hipMalloc(&csr->d_tval, sizeof(float) * (run + 8)));
  ADMM_HIP_CHECK(hipMemcpyAsync(csr->d_tptr, csr->h_tptr.data(), sizeof(uint64_t) * (ncells + 1), hipMemcpyHostToDevice, ctx->stream));
  // cursors start at the cell bases
  ADMM_HIP_CHECK(hipMemcpyAsync(d_cnt, csr->h_tptr.data(), sizeof(uint64_t) * (ncells + 1), hipMemcpyHostToDevice, ctx->stream));
  a.cell_cursor = d_cnt;
  a.trow = csr->d_trow;
  a.tcol = csr->d_tcol;
  a.tval = csr->d_tval;
  if (csr->nrows) {
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tbuild<true>), dim3(grid), dim3(256), lds2, ctx->stream, a);
    ADMM_HIP_CHECK(hipGetLastError());
  }
  ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  hipFree(d_cnt);
  ADMM_TRY(sort_cells(csr));
  // fixed-point scale of the accumulators: |r * val| <= vmax for |r| <= 1, at most panel_rows
  // addends per column and cell, everything below 2^61
  csr->vmax = 1.0f;
  if (csr->has_val && csr->nnz) {
    // (what the row masks hold has value 1, and the maximum is at least 1 anyway)
    const float *vsrc = csr->hot_dense ? csr->d_cval : csr->d_val;
    const uint64_t vn = csr->hot_dense ? csr->cnnz : csr->nnz;
    const int g = grid_for(std::max<uint64_t>(vn, 1), 256, ctx->cu_count * 8);
    hipLaunchKernelGGL(k_absmax, dim3(g), dim3(256), 0, ctx->stream, vsrc, vn, ctx->d_partials);
    std::vector<double> hp(g);
    ADMM_HIP_CHECK(hipMemcpyAsync(hp.data(), ctx->d_partials, sizeof(double) * g, hipMemcpyDeviceToHost, ctx->stream));
    ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    double m = 0;
    for (double v : hp) m = std::max(m, v);
    ADMM_REQUIRE(std::isfinite(m), "admm_csr_build_transposed: non-finite feature value");
    csr->vmax = (float)std::max(1.0, m);
  }
  int need = 0;
  while ((1ull << need) < csr->nrows && need < 40) ++need;
  int vbits = 0;
  while (std::ldexp(1.0, vbits) < (double)csr->vmax) ++vbits;
  // a column's sum over the whole set stays below 2^61, and its sum over one panel of the column
  // stream below 2^50 (the stream adds those up in doubles, exactly)
  csr->fx_shift = std::max(8, std::min(50 - 13, 61 - need) - vbits);
  csr->has_t = true;
  ADMM_TRY(ensure_taccum(csr));
  ADMM_TRY(build_forward_panels(csr));
  ADMM_TRY(release_row_major(csr));  // (a lean set whose copy was asked for after its relabelling got here with them gone already)
  ADMM_TRY(release_compact(csr));
  return ADMM_OK;
}
}  // namespace admm_impl
