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

}  // namespace admm_impl
