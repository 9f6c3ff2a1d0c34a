// Development aid: cost per lane of a divergent 4-byte gather from an L2-resident table for the
// cache-policy bits a load can carry (gfx950).
//   hipcc -O3 --offload-arch=gfx950 tools/gather_cost_ubench.hip -o tools/gather_cost_ubench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
      std::exit(1);                                                            \
    }                                                                          \
  } while (0)

template <int V>
__device__ inline float gload(const float *base, uint32_t byte_off) {
  float r;
  if (V == 0) asm volatile("global_load_dword %0, %1, %2" : "=v"(r) : "v"(byte_off), "s"(base) : "memory");
  if (V == 1) asm volatile("global_load_dword %0, %1, %2 nt" : "=v"(r) : "v"(byte_off), "s"(base) : "memory");
  if (V == 2) asm volatile("global_load_dword %0, %1, %2 sc0" : "=v"(r) : "v"(byte_off), "s"(base) : "memory");
  if (V == 3) asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(r) : "v"(byte_off), "s"(base) : "memory");
  if (V == 4) asm volatile("global_load_dword %0, %1, %2 sc0 sc1" : "=v"(r) : "v"(byte_off), "s"(base) : "memory");
  if (V == 5) asm volatile("global_load_dword %0, %1, %2 sc0 sc1 nt" : "=v"(r) : "v"(byte_off), "s"(base) : "memory");
  if (V == 6) asm volatile("global_load_dword %0, %1, %2 sc0 nt" : "=v"(r) : "v"(byte_off), "s"(base) : "memory");
  if (V == 7) asm volatile("global_load_dword %0, %1, %2 sc1 nt" : "=v"(r) : "v"(byte_off), "s"(base) : "memory");
  return r;
}

template <int V>
__global__ void __launch_bounds__(1024) k(const float *__restrict__ tab, uint32_t tab_mask, int iters, int active,
                                          float *out) {
  const int tid = threadIdx.x;
  uint32_t h = (blockIdx.x * 1024u + tid) * 2654435761u + 12345u;
  float facc = 0.0f;
  if ((tid & 63) < active) {
    for (int g = 0; g < iters; g += 8) {
      float r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        h = h * 1664525u + 1013904223u;
        r[u] = gload<V>(tab, ((h >> 4) & tab_mask) << 2);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; ++u) facc += r[u];
    }
  }
  if (facc == 1.2345f) out[0] = facc;
}

int main(int argc, char **argv) {
  const int bits = argc > 1 ? atoi(argv[1]) : 19;
  const uint32_t tab_n = 1u << bits;
  float *tab, *out;
  CK(hipMalloc(&tab, sizeof(float) * tab_n));
  CK(hipMemset(tab, 0, sizeof(float) * tab_n));
  CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int iters = 2048;
  const char *names[] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc0 nt", "sc1 nt"};
  for (int active : {64, 16}) {
    for (int v = 0; v < 8; ++v) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        switch (v) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, tab, tab_n - 1, iters, active, out); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, tab, tab_n - 1, iters, active, out); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, tab, tab_n - 1, iters, active, out); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 0, 0, tab, tab_n - 1, iters, active, out); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(1024), 0, 0, tab, tab_n - 1, iters, active, out); break;
          case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(1024), 0, 0, tab, tab_n - 1, iters, active, out); break;
          case 6: hipLaunchKernelGGL(k<6>, dim3(256), dim3(1024), 0, 0, tab, tab_n - 1, iters, active, out); break;
          default: hipLaunchKernelGGL(k<7>, dim3(256), dim3(1024), 0, 0, tab, tab_n - 1, iters, active, out); break;
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      const double ng = (double)iters * 16 * active * 256;
      std::printf("table 2^%d floats, %2d active lanes, %-11s %.3f ms  %.2f CU-clk per gathered lane (2.1 GHz)\n", bits,
                  active, names[v], best, best * 1e-3 * 2.1e9 * 256 / ng);
    }
  }
  return 0;
}
