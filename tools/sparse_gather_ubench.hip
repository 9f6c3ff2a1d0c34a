// Development aid: throughput of exec-masked divergent 4-byte gathers when only a random fraction
// of the lanes of each load is active, by number of waves per CU and loads in flight per wave.
//   hipcc -O3 --offload-arch=gfx950 tools/sparse_gather_ubench.hip -o tools/sparse_gather_ubench
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

template <int INFLIGHT>
__global__ void k(const float *__restrict__ tab, uint32_t tab_mask, int iters, uint32_t thresh, float *out,
                  unsigned long long *active_count) {
  const int tid = threadIdx.x;
  uint32_t h = (blockIdx.x * 1024u + tid) * 2654435761u + 12345u;
  float facc = 0.0f;
  unsigned long long act = 0;
  for (int g = 0; g < iters; g += INFLIGHT) {
    float r[INFLIGHT];
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u) {
      h = h * 1664525u + 1013904223u;
      r[u] = 0.0f;
      const uint32_t off = ((h >> 4) & tab_mask) << 2;
      if ((h >> 8 & 0xFFFFu) < thresh) {
        asm volatile("global_load_dword %0, %1, %2" : "+v"(r[u]) : "v"(off), "s"(tab) : "memory");
        ++act;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < INFLIGHT; ++u) facc += r[u];
  }
  if (facc == 1.2345f) out[0] = facc;
  atomicAdd(active_count, act);
}

int main(int argc, char **argv) {
  const int bits = argc > 1 ? atoi(argv[1]) : 19;
  const uint32_t tab_n = 1u << bits;
  float *tab, *out;
  unsigned long long *cnt;
  CK(hipMalloc(&tab, sizeof(float) * tab_n));
  CK(hipMemset(tab, 0, sizeof(float) * tab_n));
  CK(hipMalloc(&out, 64));
  CK(hipMalloc(&cnt, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (double frac : {1.0, 0.19}) {
    const uint32_t thresh = (uint32_t)(frac * 65536.0);
    for (int waves : {4, 8, 12, 16}) {
      for (int inflight : {8, 16, 32}) {
        const int iters = 4096;
        float best = 1e9f;
        unsigned long long act = 0;
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipMemset(cnt, 0, 8));
          CK(hipEventRecord(e0));
          if (inflight == 8) hipLaunchKernelGGL(k<8>, dim3(256), dim3(waves * 64), 0, 0, tab, tab_n - 1, iters, thresh, out, cnt);
          else if (inflight == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(waves * 64), 0, 0, tab, tab_n - 1, iters, thresh, out, cnt);
          else hipLaunchKernelGGL(k<32>, dim3(256), dim3(waves * 64), 0, 0, tab, tab_n - 1, iters, thresh, out, cnt);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          best = ms < best ? ms : best;
          CK(hipMemcpy(&act, cnt, 8, hipMemcpyDeviceToHost));
        }
        std::printf("table 2^%d, active fraction %.2f, %2d waves/CU, %2d loads in flight per wave: %.3f ms, %.2f CU-clk per gathered lane\n",
                    bits, frac, waves, inflight, best, best * 1e-3 * 2.1e9 * 256 / (double)act);
      }
    }
  }
  return 0;
}
