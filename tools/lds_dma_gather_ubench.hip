// Development aid: checks how an exec-masked `buffer_load_dword ... lds` (a gather whose result the
// memory pipe writes straight into LDS) behaves on this GPU, and what it costs per active lane.
//   1. inactive lanes must leave their LDS words alone;
//   2. the LDS destination (M0) must reach addresses beyond 64 KiB;
//   3. cost in clocks per active lane against the same gather into registers.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/lds_dma_gather_ubench tools/lds_dma_gather_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                       \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__device__ inline void dma_gather(uint32_t voff, u32x4 res, uint32_t lds_byte_addr) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(lds_byte_addr), "s"(res)
               : "memory");
}

// correctness: one workgroup of 256 threads; lane gathers table[idx[tid]] when mask[tid] != 0
__global__ void k_check(const float *table, uint32_t n, const uint32_t *idx, const uint32_t *mask, float *out,
                        uint32_t lds_float_base) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 40000; i += blockDim.x) lds[i] = -1.0f;
  __syncthreads();
  const uint64_t base = reinterpret_cast<uint64_t>(table);
  const u32x4 res = {(uint32_t)base, (uint32_t)(base >> 32) & 0xFFFFu, n * 4u, 0x00020000u};
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t dst = (uint32_t)(uintptr_t)(lds + lds_float_base + wave * 64);
  const uint32_t c = idx[threadIdx.x];
  if (mask[threadIdx.x]) dma_gather(c << 2, res, __builtin_amdgcn_readfirstlane(dst));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 40000; i += blockDim.x) out[i] = lds[i];
}

// cost: every wave issues `reps` x 16 gathers, `active_pct` percent of lanes active
template <bool DMA>
__global__ void __launch_bounds__(512) k_cost(const float *table, uint32_t n, int reps, uint32_t active_thresh, float *sink) {
  extern __shared__ float lds[];
  const uint64_t base = reinterpret_cast<uint64_t>(table);
  const u32x4 res = {(uint32_t)base, (uint32_t)(base >> 32) & 0xFFFFu, n * 4u, 0x00020000u};
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t s = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
  float acc = 0.0f;
  for (int r = 0; r < reps; ++r) {
    float g[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s = s * 1664525u + 1013904223u;
      const uint32_t c = (s >> 8) % n;
      const bool on = ((s >> 3) & 1023u) < active_thresh;
      g[e] = 0.0f;
      if (DMA) {
        const uint32_t dst = (uint32_t)(uintptr_t)(lds + (e * 8 + wave) * 64);
        if (on) dma_gather(c << 2, res, __builtin_amdgcn_readfirstlane(dst));
      } else {
        const uint32_t off = c << 2;
        if (on) asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(g[e]) : "v"(off), "s"(res) : "memory");
      }
    }
    if (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc += lds[(r & 127) * 64 + (threadIdx.x & 63)];
    } else {
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]),
                     "+v"(g[8]), "+v"(g[9]), "+v"(g[10]), "+v"(g[11]), "+v"(g[12]), "+v"(g[13]), "+v"(g[14]), "+v"(g[15])
                   :
                   : "memory");
#pragma unroll
      for (int e = 0; e < 16; ++e) acc += g[e];
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

int main() {
  const uint32_t n = 4u << 20;  // 16 MB table
  std::vector<float> table(n);
  for (uint32_t i = 0; i < n; ++i) table[i] = (float)i;
  float *d_table, *d_out;
  uint32_t *d_idx, *d_mask;
  CHECK(hipMalloc(&d_table, n * 4ull));
  CHECK(hipMalloc(&d_out, 40000 * 4));
  CHECK(hipMalloc(&d_idx, 256 * 4));
  CHECK(hipMalloc(&d_mask, 256 * 4));
  CHECK(hipMemcpy(d_table, table.data(), n * 4ull, hipMemcpyHostToDevice));
  std::vector<uint32_t> idx(256), mask(256);
  for (int i = 0; i < 256; ++i) {
    idx[i] = (uint32_t)((i * 7919u + 13u) % n);
    mask[i] = (i % 3) != 0;
  }
  CHECK(hipMemcpy(d_idx, idx.data(), 1024, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_mask, mask.data(), 1024, hipMemcpyHostToDevice));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_check), hipFuncAttributeMaxDynamicSharedMemorySize, 160000));
  for (uint32_t lds_base : {0u, 1000u, 20000u, 39000u}) {
    hipLaunchKernelGGL(k_check, dim3(1), dim3(256), 160000, 0, d_table, n, d_idx, d_mask, d_out, lds_base);
    CHECK(hipDeviceSynchronize());
    std::vector<float> out(40000);
    CHECK(hipMemcpy(out.data(), d_out, 160000, hipMemcpyDeviceToHost));
    int bad_active = 0, bad_inactive = 0, bad_elsewhere = 0;
    for (int i = 0; i < 40000; ++i) {
      const int t = i - (int)lds_base;
      if (t >= 0 && t < 256) {
        if (mask[t]) bad_active += out[i] != (float)idx[t];
        else bad_inactive += out[i] != -1.0f;
      } else {
        bad_elsewhere += out[i] != -1.0f;
      }
    }
    printf("lds float base %5u (byte %6u): wrong active %d, touched inactive %d, touched elsewhere %d\n", lds_base,
           lds_base * 4, bad_active, bad_inactive, bad_elsewhere);
  }
  int dev_clock_khz = 0;
  CHECK(hipDeviceGetAttribute(&dev_clock_khz, hipDeviceAttributeClockRate, 0));
  float *d_sink;
  CHECK(hipMalloc(&d_sink, 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int reps = 400;
  for (int pct : {100, 50, 25}) {
    const uint32_t thresh = (uint32_t)(1024 * pct / 100);
    for (int dma = 0; dma < 2; ++dma) {
      float ms = 0;
      for (int pass = 0; pass < 2; ++pass) {
        CHECK(hipEventRecord(e0));
        if (dma) hipLaunchKernelGGL(k_cost<true>, dim3(256), dim3(512), 65536, 0, d_table, n, reps, thresh, d_sink);
        else hipLaunchKernelGGL(k_cost<false>, dim3(256), dim3(512), 65536, 0, d_table, n, reps, thresh, d_sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
      }
      const double lanes = 512.0 * reps * 16 * pct / 100.0;  // active lanes per CU
      printf("%3d%% active, %s: %.3f ms, %.2f clk per active lane per CU (at %.2f GHz)\n", pct,
             dma ? "to LDS      " : "to registers", ms, ms * 1e-3 * dev_clock_khz * 1e3 / lanes, dev_clock_khz * 1e-6);
    }
  }
  return 0;
}
