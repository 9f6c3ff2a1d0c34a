// Rate of no-return global integer adds on gfx950: 2^28 adds per launch, 64-bit or 32-bit, to
// addresses that are either scattered (a hash of the add's index) or consecutive over a wave
// (what a flush of per-column sums issues), over arrays of 4 MB .. 1 GB.
//   hipcc --offload-arch=gfx950 -O3 tools/global_atomic_ubench.hip -o tools/global_atomic_ubench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ inline uint64_t mix(uint64_t x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

template <typename T, bool SCATTER>
__global__ void __launch_bounds__(256) k_add(T *__restrict__ a, uint64_t words, uint64_t total) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    // consecutive: a wave's 64 adds fall on 64 adjacent words, waves on scattered runs
    const uint64_t w = SCATTER ? mix(i) % words : (mix(i >> 6) % (words >> 6)) * 64 + (i & 63);
    atomicAdd(a + w, (T)1);
  }
}

#define CHECK(e)                                              \
  do {                                                        \
    hipError_t r_ = (e);                                      \
    if (r_ != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); \
      return 1;                                               \
    }                                                         \
  } while (0)

template <typename T, bool SCATTER>
static int run(void *buf, uint64_t bytes, const char *what) {
  const uint64_t total = 1ull << 28;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_add<T, SCATTER>), dim3(256 * 16), dim3(256), 0, 0, (T *)buf, bytes / sizeof(T), total);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("%-28s array %6llu MB: %7.3f ms  %7.1f G adds/s\n", what, (unsigned long long)(bytes >> 20), best, total / best / 1e6);
  fflush(stdout);
  return 0;
}

int main() {
  void *buf = nullptr;
  const uint64_t maxb = 1ull << 30;
  CHECK(hipMalloc(&buf, maxb));
  CHECK(hipMemset(buf, 0, maxb));
  for (uint64_t bytes : {4ull << 20, 32ull << 20, 128ull << 20, 1ull << 30}) {
    if (run<unsigned long long, true>(buf, bytes, "64-bit scattered")) return 1;
    if (run<unsigned long long, false>(buf, bytes, "64-bit consecutive per wave")) return 1;
    if (run<unsigned int, true>(buf, bytes, "32-bit scattered")) return 1;
    if (run<unsigned int, false>(buf, bytes, "32-bit consecutive per wave")) return 1;
  }
  return 0;
}
