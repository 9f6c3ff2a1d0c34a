// Development aid: LDS float-atomic throughput on gfx950 for different address patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>  // 0 float atomic add, 1 plain rmw, 2 uint atomic add, 3 read only, 4 u64 atomic add (8 B slots)
__global__ void __launch_bounds__(512) k(const unsigned short *idx, int n_per_thread, float *out) {
  extern __shared__ float acc[];
  for (int c = threadIdx.x; c < 16384; c += 512) acc[c] = 0;
  __syncthreads();
  const unsigned short *p = idx + (size_t)blockIdx.x * 512 * n_per_thread;
  float s = 0;
  for (int i = 0; i < n_per_thread; ++i) {
    unsigned c = p[(size_t)i * 512 + threadIdx.x];
    if (MODE == 0) atomicAdd(&acc[c], 1.0f);
    else if (MODE == 1) acc[c] += 1.0f;
    else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned *>(&acc[c]), 1u);
    else if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long *>(acc) + (c & 8191), 0x100000001ull);
    else s += acc[c];
  }
  __syncthreads();
  float t = s;
  for (int c = threadIdx.x; c < 16384; c += 512) t += acc[c];
  if (t == 12345.678f) out[0] = t;
}

int main() {
  const int blocks = 2048, npt = 512;
  const size_t n = (size_t)blocks * 512 * npt;
  std::vector<unsigned short> h(n);
  unsigned short *d; float *out;
  CK(hipMalloc(&d, n * 2)); CK(hipMalloc(&out, 4));
  const char *names[] = {"uniform random", "zipf-ish (50% in 64 hot)", "all same address", "lane-linear (conflict free)", "50% one address"};
  for (int pat = 0; pat < 2; ++pat) {
    unsigned long long st = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) {
      st ^= st << 13; st ^= st >> 7; st ^= st << 17;
      unsigned r = (unsigned)(st >> 20);
      unsigned short c;
      if (pat == 0) c = r & 16383;
      else if (pat == 1) c = (r & 1) ? ((r >> 1) & 63) * 251 % 16384 : (r >> 1) & 16383;
      else if (pat == 2) c = 77;
      else if (pat == 3) c = (unsigned short)(i & 16383);
      else c = (r & 1) ? 77 : (r >> 1) & 16383;
      h[i] = c;
    }
    CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 5; ++mode) {
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        if (mode == 0) k<0><<<blocks, 512, 65536>>>(d, npt, out);
        if (mode == 1) k<1><<<blocks, 512, 65536>>>(d, npt, out);
        if (mode == 2) k<2><<<blocks, 512, 65536>>>(d, npt, out);
        if (mode == 3) k<3><<<blocks, 512, 65536>>>(d, npt, out);
        if (mode == 4) k<4><<<blocks, 512, 65536>>>(d, npt, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      }
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      const char *mn[] = {"atomic f32", "plain rmw", "atomic u32", "read only", "atomic u64"};
      printf("%-28s %-10s %.3f ms  %.1f Gops/s  (%.2f lanes/clk/CU @2.4GHz,256CU)\n", names[pat], mn[mode], ms, n / ms / 1e6,
             n / (ms * 1e-3) / 256 / 2.4e9);
    }
  }
  return 0;
}
