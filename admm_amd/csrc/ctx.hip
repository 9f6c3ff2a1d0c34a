// Context, device vectors, timing and the dim-sized elementwise kernels of the consensus step.
#include "internal.h"

#include <atomic>
#include <cstring>
#include <mutex>

namespace admm_impl {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

// ---- timing -------------------------------------------------------------------------------

int timing_begin(admm_ctx *ctx, int kernel) {
  TimingSlot s;
  if (!ctx->free_slots.empty()) {
    s = ctx->free_slots.back();
    ctx->free_slots.pop_back();
  } else {
    if (ctx->pending.size() >= 8192) timing_flush(ctx);
    if (!ctx->free_slots.empty()) {
      s = ctx->free_slots.back();
      ctx->free_slots.pop_back();
    } else {
      if (hipEventCreate(&s.start) != hipSuccess) return -1;
      if (hipEventCreate(&s.stop) != hipSuccess) return -1;
    }
  }
  s.kernel = kernel;
  hipEventRecord(s.start, ctx->stream);
  ctx->pending.push_back(s);
  return (int)ctx->pending.size() - 1;
}

void timing_end(admm_ctx *ctx, int slot) { hipEventRecord(ctx->pending[slot].stop, ctx->stream); }

int timing_flush(admm_ctx *ctx) {
  if (ctx->pending.empty()) return ADMM_OK;
  ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  for (auto &s : ctx->pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, s.start, s.stop) == hipSuccess) {
      ctx->k_ms[s.kernel] += ms;
      ctx->k_launches[s.kernel] += 1;
    }
    ctx->free_slots.push_back(s);
  }
  ctx->pending.clear();
  return ADMM_OK;
}

// ---- reductions ---------------------------------------------------------------------------

// Sums up to four rows of partials (rows first .. first + blockDim.x / 256 - 1; a row below `split`
// holds nblocks_a partials, the others nblocks_b), 256 threads a row, in a fixed tree order:
// deterministic for a given grid.  The sums go to scalars[] on the device and, with every scalar
// queued launches left there before, straight into the host's pinned copy; the launch's sequence
// number follows them, so the host sees the results the moment they exist, without a copy or an
// interrupt in between.
__global__ void __launch_bounds__(1024) k_finish_rows(const double *partials, int first, int split, int nblocks_a,
                                                      int nblocks_b, double *scalars, double *host_scalars,
                                                      unsigned long long *host_seq, unsigned long long seq) {
  __shared__ double sh[1024];
  __shared__ double res[4];
  const int grp = threadIdx.x >> 8, t = threadIdx.x & 255;
  const int row = first + grp;
  const int nblocks = row < split ? nblocks_a : nblocks_b;
  double acc = 0;
  for (int i = t; i < nblocks; i += 256) acc += partials[(size_t)row * kMaxPartialBlocks + i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (t < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (t == 0) {
    scalars[row] = sh[threadIdx.x];
    res[grp] = sh[threadIdx.x];
  }
  __syncthreads();
  if (threadIdx.x < kPartialScalars) {
    const int i = threadIdx.x, count = blockDim.x >> 8;
    const double v = (i >= first && i < first + count) ? res[i - first] : scalars[i];
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(host_scalars) + i,
                       (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
  }
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int reduce_rows_queue(admm_ctx *ctx, int first, int count, int split, int nblocks_a, int nblocks_b) {
  if (count <= 0) return ADMM_OK;
  ADMM_REQUIRE(count <= 4 && first >= 0 && first + count <= kPartialScalars, "reduce_rows_queue: rows %d..%d", first,
               first + count);
  hipLaunchKernelGGL(k_finish_rows, dim3(1), dim3(256 * count), 0, ctx->stream, ctx->d_partials, first, split, nblocks_a,
                     nblocks_b, ctx->d_scalars, ctx->h_scalars_dev, ctx->h_seq_dev, ctx->seq_issued + 1);
  ADMM_HIP_CHECK(hipGetLastError());
  ++ctx->seq_issued;
  return ADMM_OK;
}

// until ctx->h_scalars holds what the last queued reduction wrote.  The host polls the sequence number all the
// way (waking up from a blocking wait for the stream costs more than the reduction's own trip, and a line
// search pays it once per trial point); every few thousand polls it asks the stream whether it is still busy,
// which is how a fault on the stream, or a reduction that never ran, ends the wait.
int wait_scalars(admm_ctx *ctx) {
  const unsigned long long want = ctx->seq_issued;
  volatile unsigned long long *seq = ctx->h_seq;
  for (unsigned spins = 1; *seq < want; ++spins) {
    if ((spins & 8191u) == 0) {
      const hipError_t st = hipStreamQuery(ctx->stream);
      if (st == hipSuccess) {  // everything queued has run
        if (*seq < want) {
          set_error("wait_scalars: the stream is idle and reduction %llu has not arrived (at %llu)", want,
                    (unsigned long long)*seq);
          return ADMM_ERR_HIP;
        }
        break;
      }
      if (st != hipErrorNotReady) ADMM_HIP_CHECK(st);
    }
#if !defined(__HIP_DEVICE_COMPILE__)
    __builtin_ia32_pause();
#endif
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return check_device_reports(ctx);
}

int check_device_reports(admm_ctx *ctx) {
  volatile unsigned long long *word = ctx->h_seq + 1;
  if (*word == 0) return ADMM_OK;
  *word = 0;
  set_error("admm_ftrl_pass: the one-launch mini-batch pass gave up at a device-wide barrier (a workgroup was not "
            "resident or never arrived); its results are invalid -- ADMM_FTRL_PERSISTENT=0 selects the per-batch launches");
  return ADMM_ERR_STATE;
}

int finish_partials(admm_ctx *ctx, int nscalars, int nblocks) {
  ADMM_TRY(reduce_rows_queue(ctx, 0, nscalars, nscalars, nblocks, nblocks));
  return wait_scalars(ctx);
}

__device__ inline double block_sum_256(double v, double *sh) {
  // wave reduce (64 lanes), then 4 waves through LDS
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- elementwise kernels -------------------------------------------------------------------

__global__ void __launch_bounds__(256) k_payload(const float *__restrict__ x,
                                                 const float *__restrict__ y, float rho,
                                                 float *__restrict__ out, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    out[i] = x[i] * rho + y[i];
}

__device__ inline float soft_threshold(float s, float l1) {
  if (l1 > 0.0f) {
    if (s > l1) return s - l1;
    if (s < -l1) return s + l1;
    return 0.0f;
  }
  return s;
}

__global__ void __launch_bounds__(256) k_global_update(const float *__restrict__ sum, float denom,
                                                       float l1, float *__restrict__ z,
                                                       uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    z[i] = soft_threshold(sum[i], l1) / denom;
}

__global__ void __launch_bounds__(256) k_dual_update(float *__restrict__ y,
                                                     const float *__restrict__ x,
                                                     const float *__restrict__ z, float rho,
                                                     uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    y[i] += rho * (x[i] - z[i]);
}

// one coordinate of the consensus step: z-update (src/master.cc:17, with the optional soft
// threshold), dual update (src/workers.cc:117), and this coordinate's terms of the two residuals
__device__ inline void consensus_elem(const float *__restrict__ sum, float denom, float l1, float rho,
                                      const float *__restrict__ x, float *__restrict__ y,
                                      float *__restrict__ z, uint64_t i, double &r2, double &s2) {
  const float zo = z[i];
  const float zn = soft_threshold(sum[i], l1) / denom;
  const float xi = x[i];
  z[i] = zn;
  y[i] += rho * (xi - zn);
  const double d = (double)xi - (double)zn, e = (double)zn - (double)zo;
  r2 += d * d;
  s2 += e * e;
}

// z-update + dual update + both residual partial sums in one pass over the dim-sized vectors.
__global__ void __launch_bounds__(256) k_consensus(const float *__restrict__ sum, float denom,
                                                   float l1, float rho,
                                                   const float *__restrict__ x,
                                                   float *__restrict__ y, float *__restrict__ z,
                                                   uint64_t n, double *__restrict__ partials) {
  __shared__ double sh[4];
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  double r2 = 0, s2 = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    consensus_elem(sum, denom, l1, rho, x, y, z, i, r2, s2);
  r2 = block_sum_256(r2, sh);
  s2 = block_sum_256(s2, sh);
  if (threadIdx.x == 0) {
    partials[0 * kMaxPartialBlocks + blockIdx.x] = r2;
    partials[1 * kMaxPartialBlocks + blockIdx.x] = s2;
  }
}

// ---- the consensus step for callers that may not allocate ------------------------------------
//
// Same arithmetic per coordinate; the per-workgroup residual partials go to a fixed array in the
// code object's own device memory instead of a context's scratch allocation.  Workgroup b owns
// coordinates [b * kStaticChunk, (b + 1) * kStaticChunk), so the partition (and with it the order
// of the residual sums) does not depend on the GPU's CU count.
constexpr uint32_t kStaticChunk = 4096;        // coordinates per workgroup
constexpr uint32_t kStaticMaxBlocks = 65536;   // partial slots per residual: dim <= 2^28
static_assert((uint64_t)kStaticChunk * kStaticMaxBlocks == ADMM_CONSENSUS_STATIC_MAX_DIM, "header and kernel disagree");
// [0, kStaticMaxBlocks): partials of r2; [kStaticMaxBlocks, 2 kStaticMaxBlocks): partials of s2.
// 1 MiB per GPU the library runs on, held until the process ends.  Host code never passes this
// name to a kernel or a memory call: it is a host-side stand-in for the symbol, and each device
// has its own copy at its own address, resolved once per device by static_scratch() below.
static __device__ __align__(256) double scratch[2 * kStaticMaxBlocks];

__global__ void __launch_bounds__(256) k_consensus_chunked(const float *__restrict__ sum, float denom,
                                                           float l1, float rho,
                                                           const float *__restrict__ x,
                                                           float *__restrict__ y, float *__restrict__ z,
                                                           uint64_t n, double *__restrict__ partials) {
  __shared__ double sh[4];
  const uint64_t base = (uint64_t)blockIdx.x * kStaticChunk;
  const uint64_t end = base + kStaticChunk < n ? base + kStaticChunk : n;
  double r2 = 0, s2 = 0;
  for (uint64_t i = base + threadIdx.x; i < end; i += 256)
    consensus_elem(sum, denom, l1, rho, x, y, z, i, r2, s2);
  r2 = block_sum_256(r2, sh);
  s2 = block_sum_256(s2, sh);
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = r2;
    partials[kStaticMaxBlocks + blockIdx.x] = s2;
  }
}

// block 0: out[0] = sum of the r2 partials; block 1: out[1] = rho^2 * sum of the s2 partials.
// Only the first nblocks slots of each half are read: what an earlier, larger call left behind
// them is never looked at.
__global__ void __launch_bounds__(256) k_finish_static(const double *__restrict__ partials, uint32_t nblocks,
                                                       double rho2, double *__restrict__ out) {
  __shared__ double sh[256];
  const double *p = partials + (size_t)blockIdx.x * kStaticMaxBlocks;
  double acc = 0;
  for (uint32_t i = threadIdx.x; i < nblocks; i += 256) acc += p[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = blockIdx.x == 1 ? rho2 * sh[0] : sh[0];
}

// Device address of `scratch` on `device`, resolved once per device with that device current.
constexpr int kMaxDevices = 64;
static std::atomic<double *> g_scratch_ptr[kMaxDevices] = {};
static std::mutex g_scratch_mutex;

static int static_scratch(int device, double **out) {
  ADMM_REQUIRE(device >= 0 && device < kMaxDevices, "device %d: at most %d devices per process", device, kMaxDevices);
  // once resolved, a call reads the pointer and nothing else: no lock, no runtime query
  if (double *known = g_scratch_ptr[device].load(std::memory_order_acquire)) {
    *out = known;
    return ADMM_OK;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device].load(std::memory_order_relaxed)) {
    int cur = -1;
    ADMM_HIP_CHECK(hipGetDevice(&cur));
    if (cur != device) ADMM_HIP_CHECK(hipSetDevice(device));
    double *ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(reinterpret_cast<void **>(&ptr), HIP_SYMBOL(scratch));
    if (cur != device) hipSetDevice(cur);
    if (e != hipSuccess || !ptr) {
      set_error("hipGetSymbolAddress(scratch) on device %d failed: %s", device, hipGetErrorString(e));
      return ADMM_ERR_HIP;
    }
    g_scratch_ptr[device].store(ptr, std::memory_order_release);
  }
  *out = g_scratch_ptr[device].load(std::memory_order_relaxed);
  return ADMM_OK;
}

__global__ void __launch_bounds__(256) k_dot(const float *__restrict__ a,
                                             const float *__restrict__ b, uint64_t n,
                                             double *__restrict__ partials) {
  __shared__ double sh[4];
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  double acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    acc += (double)a[i] * (double)b[i];
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

__global__ void __launch_bounds__(256) k_axpy(float *__restrict__ y, float a,
                                              const float *__restrict__ x, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] += a * x[i];
}

__global__ void __launch_bounds__(256) k_scale(float *__restrict__ y, float a, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] *= a;
}

__global__ void __launch_bounds__(256) k_lincomb(float *__restrict__ out,
                                                 const float *__restrict__ a, float s,
                                                 const float *__restrict__ b, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    out[i] = a[i] + s * b[i];
}

static inline int ew_grid(admm_ctx *ctx, uint64_t n) { return grid_for(n, 256, ctx->cu_count * 8); }

int vec_dot(admm_ctx *ctx, const float *a, const float *b, uint64_t n, double *out_host) {
  const int grid = ew_grid(ctx, n);
  hipLaunchKernelGGL(k_dot, dim3(grid), dim3(256), 0, ctx->stream, a, b, n, ctx->d_partials);
  ADMM_HIP_CHECK(hipGetLastError());
  ADMM_TRY(finish_partials(ctx, 1, grid));
  *out_host = ctx->h_scalars[0];
  return ADMM_OK;
}

int vec_axpy(admm_ctx *ctx, float *y, float a, const float *x, uint64_t n) {
  hipLaunchKernelGGL(k_axpy, dim3(ew_grid(ctx, n)), dim3(256), 0, ctx->stream, y, a, x, n);
  ADMM_HIP_CHECK(hipGetLastError());
  return ADMM_OK;
}

int vec_scale(admm_ctx *ctx, float *y, float a, uint64_t n) {
  hipLaunchKernelGGL(k_scale, dim3(ew_grid(ctx, n)), dim3(256), 0, ctx->stream, y, a, n);
  ADMM_HIP_CHECK(hipGetLastError());
  return ADMM_OK;
}

int vec_lincomb(admm_ctx *ctx, float *out, const float *a, float s, const float *b, uint64_t n) {
  hipLaunchKernelGGL(k_lincomb, dim3(ew_grid(ctx, n)), dim3(256), 0, ctx->stream, out, a, s, b, n);
  ADMM_HIP_CHECK(hipGetLastError());
  return ADMM_OK;
}

int vec_sub(admm_ctx *ctx, float *out, const float *a, const float *b, uint64_t n) {
  return vec_lincomb(ctx, out, a, -1.0f, b, n);
}

}  // namespace admm_impl

using namespace admm_impl;

// =============================================================================================
// C ABI
// =============================================================================================

extern "C" const char *admm_last_error(void) { return g_last_error.c_str(); }
extern "C" int admm_abi_version(void) { return 1; }

extern "C" int admm_device_count(int *count) {
  ADMM_REQUIRE(count, "admm_device_count: NULL argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  *count = n;
  return ADMM_OK;
}

extern "C" int admm_ctx_create(int device, void *stream, admm_ctx **out) {
  ADMM_REQUIRE(out != nullptr, "admm_ctx_create: out is NULL");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    set_error("admm_ctx_create: no HIP device visible (%s); this library has no CPU path",
              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return ADMM_ERR_HIP;
  }
  ADMM_REQUIRE(device >= 0 && device < ndev, "admm_ctx_create: device %d out of range (%d)", device,
               ndev);
  ADMM_HIP_CHECK(hipSetDevice(device));
  {
    // resolve this device's copy of the static scratch array now, so that a later
    // admm_consensus_static on this device finds it without any runtime query
    double *unused = nullptr;
    ADMM_TRY(static_scratch(device, &unused));
  }
  admm_ctx *ctx = new admm_ctx();
  ctx->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->cu_count = prop.multiProcessorCount;
  if (stream) {
    ctx->stream = (hipStream_t)stream;
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      set_error("admm_ctx_create: hipStreamCreate failed");
      delete ctx;
      return ADMM_ERR_HIP;
    }
    ctx->own_stream = true;
  }
  if (hipMalloc(&ctx->d_partials, sizeof(double) * kPartialScalars * kMaxPartialBlocks) != hipSuccess ||
      hipMalloc(&ctx->d_scalars, sizeof(double) * kPartialScalars) != hipSuccess ||
      hipHostMalloc(&ctx->h_scalars, sizeof(double) * (kPartialScalars + 8), hipHostMallocMapped | hipHostMallocCoherent) !=
          hipSuccess) {
    set_error("admm_ctx_create: scratch allocation failed");
    admm_ctx_destroy(ctx);
    return ADMM_ERR_NOMEM;
  }
  // the reductions write their results and a sequence number into this pinned block themselves
  memset(ctx->h_scalars, 0, sizeof(double) * (kPartialScalars + 8));
  ctx->h_seq = reinterpret_cast<unsigned long long *>(ctx->h_scalars + kPartialScalars);
  void *dev = nullptr;
  if (hipHostGetDevicePointer(&dev, ctx->h_scalars, 0) != hipSuccess) {
    set_error("admm_ctx_create: the pinned scalars are not mapped into the device");
    admm_ctx_destroy(ctx);
    return ADMM_ERR_HIP;
  }
  ctx->h_scalars_dev = static_cast<double *>(dev);
  ctx->h_seq_dev = reinterpret_cast<unsigned long long *>(ctx->h_scalars_dev + kPartialScalars);
  *out = ctx;
  return ADMM_OK;
}

extern "C" int admm_ctx_destroy(admm_ctx *ctx) {
  if (!ctx) return ADMM_OK;
  hipSetDevice(ctx->device);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  for (auto &s : ctx->pending) {
    hipEventDestroy(s.start);
    hipEventDestroy(s.stop);
  }
  for (auto &s : ctx->free_slots) {
    hipEventDestroy(s.start);
    hipEventDestroy(s.stop);
  }
  comm_release(ctx);
  if (ctx->d_partials) hipFree(ctx->d_partials);
  if (ctx->d_scalars) hipFree(ctx->d_scalars);
  if (ctx->h_scalars) hipHostFree(ctx->h_scalars);
  if (ctx->lbfgs_ws) hipFree(ctx->lbfgs_ws);
  if (ctx->d_multi) hipFree(ctx->d_multi);
  if (ctx->d_ps) hipFree(ctx->d_ps);
  if (ctx->auc_ws) hipFree(ctx->auc_ws);
  if (ctx->row_ws) hipFree(ctx->row_ws);
  if (ctx->d_lb) hipFree(ctx->d_lb);
  if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return ADMM_OK;
}

extern "C" int admm_sync(admm_ctx *ctx) {
  ADMM_REQUIRE(ctx, "admm_sync: ctx is NULL");
  ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return check_device_reports(ctx);
}

extern "C" int admm_timing_enable(admm_ctx *ctx, int on) {
  ADMM_REQUIRE(ctx, "admm_timing_enable: ctx is NULL");
  if (!on) ADMM_TRY(timing_flush(ctx));
  ctx->timing = on != 0;
  return ADMM_OK;
}

extern "C" int admm_timing_classes(admm_ctx *ctx, unsigned mask) {
  ADMM_REQUIRE(ctx, "admm_timing_classes: ctx is NULL");
  ctx->timing_classes = mask;
  return ADMM_OK;
}

extern "C" int admm_timing_every(admm_ctx *ctx, unsigned n) {
  ADMM_REQUIRE(ctx && n >= 1, "admm_timing_every: ctx is NULL or n is 0");
  ctx->timing_every = n;
  for (int k = 0; k < ADMM_K_COUNT; ++k) ctx->k_seen[k] = 0;
  return ADMM_OK;
}

extern "C" int admm_timing_reset(admm_ctx *ctx) {
  ADMM_REQUIRE(ctx, "admm_timing_reset: ctx is NULL");
  ADMM_TRY(timing_flush(ctx));
  for (int k = 0; k < ADMM_K_COUNT; ++k) {
    ctx->k_ms[k] = 0;
    ctx->k_launches[k] = 0;
  }
  return ADMM_OK;
}

extern "C" int admm_timing_get(admm_ctx *ctx, int kernel, double *total_ms, uint64_t *launches) {
  ADMM_REQUIRE(ctx && kernel >= 0 && kernel < ADMM_K_COUNT, "admm_timing_get: bad arguments");
  ADMM_TRY(timing_flush(ctx));
  if (total_ms) *total_ms = ctx->k_ms[kernel];
  if (launches) *launches = ctx->k_launches[kernel];
  return ADMM_OK;
}

// ---- vectors --------------------------------------------------------------------------------

extern "C" int admm_vec_alloc(admm_ctx *ctx, uint64_t n, float **out) {
  ADMM_REQUIRE(ctx && out && n > 0, "admm_vec_alloc: bad arguments");
  ADMM_HIP_CHECK(hipSetDevice(ctx->device));
  float *p = nullptr;
  ADMM_HIP_CHECK(hipMalloc(&p, sizeof(float) * n));
  hipError_t e = hipMemsetAsync(p, 0, sizeof(float) * n, ctx->stream);
  if (e != hipSuccess) {
    hipFree(p);
    set_error("admm_vec_alloc: memset failed: %s", hipGetErrorString(e));
    return ADMM_ERR_HIP;
  }
  *out = p;
  return ADMM_OK;
}

extern "C" int admm_vec_free(admm_ctx *ctx, float *v) {
  ADMM_REQUIRE(ctx, "admm_vec_free: ctx is NULL");
  if (!v) return ADMM_OK;
  ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  ADMM_HIP_CHECK(hipFree(v));
  return ADMM_OK;
}

extern "C" int admm_vec_zero(admm_ctx *ctx, float *v, uint64_t n) {
  ADMM_REQUIRE(ctx && v, "admm_vec_zero: bad arguments");
  ADMM_HIP_CHECK(hipMemsetAsync(v, 0, sizeof(float) * n, ctx->stream));
  return ADMM_OK;
}

extern "C" int admm_vec_copy(admm_ctx *ctx, float *dst, const float *src, uint64_t n) {
  ADMM_REQUIRE(ctx && dst && src, "admm_vec_copy: bad arguments");
  ADMM_HIP_CHECK(hipMemcpyAsync(dst, src, sizeof(float) * n, hipMemcpyDeviceToDevice, ctx->stream));
  return ADMM_OK;
}

extern "C" int admm_vec_upload(admm_ctx *ctx, float *dst, const float *src_host, uint64_t n) {
  ADMM_REQUIRE(ctx && dst && src_host, "admm_vec_upload: bad arguments");
  ADMM_HIP_CHECK(hipMemcpyAsync(dst, src_host, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream));
  ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // the host buffer is only borrowed
  return ADMM_OK;
}

extern "C" int admm_vec_download(admm_ctx *ctx, float *dst_host, const float *src, uint64_t n) {
  ADMM_REQUIRE(ctx && dst_host && src, "admm_vec_download: bad arguments");
  ADMM_HIP_CHECK(hipMemcpyAsync(dst_host, src, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
  ADMM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return check_device_reports(ctx);
}

// ---- consensus algebra ------------------------------------------------------------------------

extern "C" int admm_payload(admm_ctx *ctx, const float *x, const float *ydual, float rho, float *out,
                            uint64_t dim) {
  ADMM_REQUIRE(ctx && x && ydual && out && dim > 0, "admm_payload: bad arguments");
  KernelTimer t(ctx, ADMM_K_ELEMENTWISE);
  hipLaunchKernelGGL(k_payload, dim3(ew_grid(ctx, dim)), dim3(256), 0, ctx->stream, x, ydual, rho, out,
                     dim);
  ADMM_HIP_CHECK(hipGetLastError());
  return ADMM_OK;
}

extern "C" int admm_global_update(admm_ctx *ctx, const float *sum, int num_worker, float rho,
                                  float global_var, float l1, float *z, uint64_t dim) {
  ADMM_REQUIRE(ctx && sum && z && dim > 0 && num_worker > 0, "admm_global_update: bad arguments");
  const float denom = num_worker * rho + global_var;  // src/master.cc:17, float arithmetic
  KernelTimer t(ctx, ADMM_K_ELEMENTWISE);
  hipLaunchKernelGGL(k_global_update, dim3(ew_grid(ctx, dim)), dim3(256), 0, ctx->stream, sum, denom,
                     l1, z, dim);
  ADMM_HIP_CHECK(hipGetLastError());
  return ADMM_OK;
}

extern "C" int admm_dual_update(admm_ctx *ctx, float *ydual, const float *x, const float *z, float rho,
                                uint64_t dim) {
  ADMM_REQUIRE(ctx && ydual && x && z && dim > 0, "admm_dual_update: bad arguments");
  KernelTimer t(ctx, ADMM_K_ELEMENTWISE);
  hipLaunchKernelGGL(k_dual_update, dim3(ew_grid(ctx, dim)), dim3(256), 0, ctx->stream, ydual, x, z,
                     rho, dim);
  ADMM_HIP_CHECK(hipGetLastError());
  return ADMM_OK;
}

extern "C" int admm_consensus(admm_ctx *ctx, const float *sum, int num_worker, float rho,
                              float global_var, float l1, const float *x, float *ydual, float *z,
                              uint64_t dim, double *r2, double *s2) {
  ADMM_REQUIRE(ctx && sum && x && ydual && z && dim > 0 && num_worker > 0,
               "admm_consensus: bad arguments");
  const float denom = num_worker * rho + global_var;
  const int grid = grid_for(dim, 256, ctx->cu_count * 8);
  {
    KernelTimer t(ctx, ADMM_K_ELEMENTWISE);
    hipLaunchKernelGGL(k_consensus, dim3(grid), dim3(256), 0, ctx->stream, sum, denom, l1, rho, x,
                       ydual, z, dim, ctx->d_partials);
    ADMM_HIP_CHECK(hipGetLastError());
  }
  if (r2 || s2) {
    ADMM_TRY(finish_partials(ctx, 2, grid));
    if (r2) *r2 = ctx->h_scalars[0];
    if (s2) *s2 = (double)rho * (double)rho * ctx->h_scalars[1];
  }
  return ADMM_OK;
}

extern "C" int admm_consensus_static(void *stream, const float *sum, int num_worker, float rho,
                                     float global_var, float l1, const float *x, float *ydual,
                                     float *z, uint64_t dim, double *residuals_out) {
  ADMM_REQUIRE(sum && x && ydual && z && dim > 0 && num_worker > 0, "admm_consensus_static: bad arguments");
  ADMM_REQUIRE(dim <= ADMM_CONSENSUS_STATIC_MAX_DIM,
               "admm_consensus_static: dim %llu exceeds the static scratch (at most %llu coordinates); "
               "use admm_consensus",
               (unsigned long long)dim, (unsigned long long)ADMM_CONSENSUS_STATIC_MAX_DIM);
  int device = -1;
  ADMM_HIP_CHECK(hipGetDevice(&device));
  double *ptr = nullptr;  // this device's copy of `scratch`
  ADMM_TRY(static_scratch(device, &ptr));
  const float denom = num_worker * rho + global_var;  // src/master.cc:17, float arithmetic
  const uint32_t nblocks = (uint32_t)((dim + kStaticChunk - 1) / kStaticChunk);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_consensus_chunked, dim3(nblocks), dim3(256), 0, s, sum, denom, l1, rho, x, ydual, z, dim,
                     ptr);
  ADMM_HIP_CHECK(hipGetLastError());
  if (residuals_out) {
    hipLaunchKernelGGL(k_finish_static, dim3(2), dim3(256), 0, s, static_cast<const double *>(ptr), nblocks,
                       (double)rho * (double)rho, residuals_out);
    ADMM_HIP_CHECK(hipGetLastError());
  }
  return ADMM_OK;
}
