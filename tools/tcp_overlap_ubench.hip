// Development aid: do HBM streaming loads and L2-resident 4-byte gathers overlap inside one CU
// when different wavefronts issue them, and when the same wavefronts issue both?
//   hipcc -O3 --offload-arch=gfx950 tools/tcp_overlap_ubench.hip -o /tmp/tcp_overlap && /tmp/tcp_overlap
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
      std::exit(1);                                                            \
    }                                                                          \
  } while (0)

// mode bit 0: waves 0..7 stream; bit 1: waves 8..15 gather; mode 4: every wave does both in turn
__global__ void __launch_bounds__(1024) k(const u32x4 *__restrict__ big, uint64_t big_vec_per_wg,
                                          const float *__restrict__ tab, uint32_t tab_mask, int gathers_per_lane,
                                          int mode, float *out) {
  const int tid = threadIdx.x, wave = tid >> 6;
  uint32_t acc = 0;
  float facc = 0.0f;
  const u32x4 *base = big + (uint64_t)blockIdx.x * big_vec_per_wg;
  if (mode == 4) {
    uint32_t h = (blockIdx.x * 1024u + tid) * 2654435761u + 12345u;
    const int per = gathers_per_lane / 2;  // 16 waves share the gathers 8 waves would do
    const uint64_t nvec = big_vec_per_wg / 1024;
    const int g_per_it = (int)((per + nvec - 1) / nvec);
    int done = 0;
    for (uint64_t i = 0; i < nvec; ++i) {
      const u32x4 v = __builtin_nontemporal_load(base + i * 1024 + tid);
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
      for (int g = 0; g < g_per_it && done < per; ++g, ++done) {
        h = h * 1664525u + 1013904223u;
        facc += __builtin_nontemporal_load(tab + ((h >> 4) & tab_mask));
      }
    }
  } else if (wave < 8) {
    if (mode & 1) {
      const uint64_t nvec = big_vec_per_wg / 512;
      const int t = tid;  // 0..511
#pragma unroll 4
      for (uint64_t i = 0; i < nvec; ++i) {
        const u32x4 v = __builtin_nontemporal_load(base + i * 512 + t);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
      }
    }
  } else {
    if (mode & 2) {
      uint32_t h = (blockIdx.x * 1024u + tid) * 2654435761u + 12345u;
#pragma unroll 8
      for (int g = 0; g < gathers_per_lane; ++g) {
        h = h * 1664525u + 1013904223u;
        facc += __builtin_nontemporal_load(tab + ((h >> 4) & tab_mask));
      }
    }
  }
  if (acc == 0x12345678u || facc == 1.2345f) out[0] = facc + acc;
}

int main(int argc, char **argv) {
  const int wgs = 256;
  const uint64_t bytes_per_wg = 32ull << 20;  // 8 GiB in total
  const uint64_t vec_per_wg = bytes_per_wg / 16;
  const uint32_t tab_n = 1u << (argc > 1 ? atoi(argv[1]) : 22);  // floats in the gathered table
  u32x4 *big;
  float *tab, *out;
  CK(hipMalloc(&big, bytes_per_wg * wgs));
  CK(hipMemset(big, 1, bytes_per_wg * wgs));
  CK(hipMalloc(&tab, sizeof(float) * tab_n));
  CK(hipMemset(tab, 0, sizeof(float) * tab_n));
  CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int gathers = 8192;  // per lane of the 8 gather waves: 4.2e6 per CU
  for (int rep = 0; rep < 2; ++rep)
    for (int mode : {1, 2, 3, 4}) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(wgs), dim3(1024), 0, 0, big, vec_per_wg, tab, tab_n - 1, gathers, mode, out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double gb = (mode & 1 || mode == 4) ? bytes_per_wg * wgs / 1e9 : 0;
      const double ng = (mode & 2 || mode == 4) ? (double)gathers * 512 * wgs : 0;
      std::printf("mode %d: %.3f ms  stream %.1f GB (%.0f GB/s)  gathers %.3g (%.2f CU-clk each at 2.4 GHz)\n", mode,
                  ms, gb, gb / (ms * 1e-3), ng, ng > 0 ? ms * 1e-3 * 2.4e9 * 256 / ng : 0.0);
    }
  return 0;
}
