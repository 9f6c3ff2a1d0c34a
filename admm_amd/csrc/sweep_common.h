// Device helpers shared by the sweep kernels (forward, transposed, mini-batch FTRL).
#pragma once

#include "internal.h"

namespace admm_impl {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaskSlots = 16;  // copies of the mask columns' global sums in the mini-batch FTRL pass (see RpArgs::mpart)

__device__ inline float clamp35(float m) { return fmaxf(fminf(m, 35.0f), -35.0f); }

__device__ inline unsigned long long rp_to_fx(float x, float scale) { return (unsigned long long)(long long)(x * scale); }

// sum of a 64-bit value over the 16 lanes of a DPP row (row_ror by 8, 4, 2, 1); every lane gets the total
__device__ inline long long row16_sum(long long x) {
#define ROW16_STEP(CTRL)                                                                                      \
  {                                                                                                           \
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(unsigned long long)x, CTRL, 0xF, 0xF, false);     \
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)((unsigned long long)x >> 32), CTRL, 0xF, 0xF, false); \
    x += (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);                                \
  }
  ROW16_STEP(0x128)
  ROW16_STEP(0x124)
  ROW16_STEP(0x122)
  ROW16_STEP(0x121)
#undef ROW16_STEP
  return x;
}

// An integer-valued double of magnitude below 2^51 as a 64-bit integer: adding 1.5 * 2^52 leaves the
// integer (offset by 2^51) in the low mantissa bits.
__device__ inline unsigned long long rp_d2fx(double s) {
  return (unsigned long long)(__double_as_longlong(s + 6755399441055744.0) - 0x4338000000000000ll);
}

typedef unsigned long long fx_t;  // two's-complement fixed point held in an unsigned word

__device__ inline float ftrl_weight(float z, float n, float c, float alpha, float beta, float l1, float l2) {
  // src/ftrl.cc:61-66 (threshold on l1, shrink by the literal 1)
  const float sign = z < 0.0f ? -1.0f : 1.0f;
  if (sign * z <= l1) return 0.0f;
  const float num = sign * 1 - z + l2 * c;
  const double den = ((double)beta + sqrt((double)n)) / (double)alpha + (double)l2;
  return (float)((double)num / den);
}

__device__ inline fx_t to_fx(float x, float scale) { return (fx_t)(long long)(x * scale); }

// grid of the kernels that fill csr->d_wsum and leave max |wsum| per workgroup behind, and where those maxima go
inline int wsum_grid(admm_ctx *ctx, uint64_t dim) {
  const int cap = ctx->cu_count * 8 < kMaxPartialBlocks ? ctx->cu_count * 8 : kMaxPartialBlocks;
  return grid_for(dim, 256, cap);
}
inline float *wsum_blockmax(admm_ctx *ctx) { return reinterpret_cast<float *>(ctx->d_partials + 6 * (size_t)kMaxPartialBlocks); }

// arguments of the cell sweep (transposed.hip); the mini-batch FTRL pass fills the f* fields for its fused update
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

// forward sweep over the row-major form (forward_rows.hip); launch_forward (forward_panels.hip) falls back to it
int launch_forward_rows(admm_csr *csr, FwdMode mode, uint64_t r0, uint64_t r1, const float *const *w, int nw,
                        float *out_rows, int *nblocks_out, float *margins_out);
// cell sweep with the fused update of a mini-batch (fuse: the f* fields, or nullptr)
int launch_transposed_impl(admm_csr *csr, bool ftrl_mode, uint64_t p0, uint64_t p1, const TArgs *fuse);

}  // namespace admm_impl
