// The one collective of the path: fp32 sum all-reduce of the dim-sized payload
// (rabit::Allreduce<op::Sum>, src/admm_allreduce.cpp:69) on RCCL.  librccl is opened lazily so
// that single-GPU use and hosts that bring their own process group never load it.
#include "internal.h"

#include <dlfcn.h>

#include <cstring>

using namespace admm_impl;

namespace {

// the few RCCL entry points used, with the NCCL ABI (ncclUniqueId is 128 opaque bytes)
struct UniqueId {
  char bytes[128];
};
typedef int (*fn_get_unique_id)(UniqueId *);
typedef int (*fn_comm_init_rank)(void **comm, int nranks, UniqueId id, int rank);
typedef int (*fn_comm_destroy)(void *comm);
typedef int (*fn_all_reduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm,
                             hipStream_t stream);
typedef const char *(*fn_get_error_string)(int);

struct Rccl {
  void *handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_get_error_string get_error_string = nullptr;
};

Rccl g_rccl;

int load_rccl() {
  if (g_rccl.handle) return ADMM_OK;
  // a copy the process has loaded already (a host framework's own RCCL carries the same soname)
  // is found by the first plain name; otherwise the usual search path, then ROCm's directory
  const char *names[] = {getenv("ADMM_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *nm : names) {
    if (!nm || !*nm) continue;
    g_rccl.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) {
    set_error("RCCL not found (set ADMM_RCCL_LIB): %s", dlerror());
    return ADMM_ERR_COMM;
  }
  g_rccl.get_unique_id = (fn_get_unique_id)dlsym(g_rccl.handle, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (fn_comm_init_rank)dlsym(g_rccl.handle, "ncclCommInitRank");
  g_rccl.comm_destroy = (fn_comm_destroy)dlsym(g_rccl.handle, "ncclCommDestroy");
  g_rccl.all_reduce = (fn_all_reduce)dlsym(g_rccl.handle, "ncclAllReduce");
  g_rccl.get_error_string = (fn_get_error_string)dlsym(g_rccl.handle, "ncclGetErrorString");
  if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_reduce) {
    set_error("RCCL library lacks ncclGetUniqueId/ncclCommInitRank/ncclAllReduce");
    dlclose(g_rccl.handle);
    g_rccl = Rccl();
    return ADMM_ERR_COMM;
  }
  return ADMM_OK;
}

const char *rccl_err(int code) { return g_rccl.get_error_string ? g_rccl.get_error_string(code) : "?"; }

}  // namespace

namespace admm_impl {
void comm_release(admm_ctx *ctx) {
  if (ctx->comm && g_rccl.comm_destroy) g_rccl.comm_destroy(ctx->comm);
  ctx->comm = nullptr;
}
}  // namespace admm_impl

extern "C" int admm_comm_unique_id(void *unique_id_128) {
  ADMM_REQUIRE(unique_id_128, "admm_comm_unique_id: NULL argument");
  ADMM_TRY(load_rccl());
  UniqueId id;
  const int rc = g_rccl.get_unique_id(&id);
  if (rc != 0) {
    set_error("ncclGetUniqueId failed: %s", rccl_err(rc));
    return ADMM_ERR_COMM;
  }
  memcpy(unique_id_128, id.bytes, 128);
  return ADMM_OK;
}

extern "C" int admm_comm_init(admm_ctx *ctx, int rank, int world, const void *unique_id_128) {
  ADMM_REQUIRE(ctx && world >= 1 && rank >= 0 && rank < world, "admm_comm_init: bad rank/world");
  ctx->rank = rank;
  ctx->world = world;
  if (world == 1) return ADMM_OK;
  ADMM_REQUIRE(unique_id_128, "admm_comm_init: unique id is NULL");
  ADMM_TRY(load_rccl());
  ADMM_HIP_CHECK(hipSetDevice(ctx->device));
  UniqueId id;
  memcpy(id.bytes, unique_id_128, 128);
  const int rc = g_rccl.comm_init_rank(&ctx->comm, world, id, rank);
  if (rc != 0) {
    set_error("ncclCommInitRank failed: %s", rccl_err(rc));
    ctx->comm = nullptr;
    return ADMM_ERR_COMM;
  }
  return ADMM_OK;
}

extern "C" int admm_allreduce_sum_f32(admm_ctx *ctx, float *buf, uint64_t n) {
  ADMM_REQUIRE(ctx && buf, "admm_allreduce_sum_f32: NULL argument");
  if (ctx->world == 1) return ADMM_OK;
  ADMM_REQUIRE(ctx->comm, "admm_allreduce_sum_f32: world %d but no communicator (admm_comm_init)", ctx->world);
  // ncclFloat32 = 7, ncclSum = 0
  const int rc = g_rccl.all_reduce(buf, buf, (size_t)n, 7, 0, ctx->comm, ctx->stream);
  if (rc != 0) {
    set_error("ncclAllReduce failed: %s", rccl_err(rc));
    return ADMM_ERR_COMM;
  }
  return ADMM_OK;
}

namespace {
__global__ void __launch_bounds__(256) k_comm_fill(float *buf, uint32_t n, float base) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) buf[i] = base + (float)(i & 7);
}
}  // namespace

// Self-test of the RCCL binding (stands where the reference keeps its rabit smoke program,
// src/admm_main.cpp:37-52: a Max and a Sum all-reduce and a broadcast between the ranks).
// Always: a one-rank communicator of its own is created on the context's device, a sum all-reduce
// of 4096 floats runs through it in place and must leave them unchanged, and the communicator is
// destroyed -- this exercises library loading, the unique id, ncclCommInitRank (the id struct by
// value) and ncclAllReduce's datatype / operator codes without needing a second GPU.  With a
// communicator over more than one rank (admm_comm_init) it then all-reduces rank-dependent values
// through that one and checks the sums; that part is collective: every rank must call it.
extern "C" int admm_comm_selftest(admm_ctx *ctx) {
  ADMM_REQUIRE(ctx, "admm_comm_selftest: ctx is NULL");
  ADMM_TRY(load_rccl());
  ADMM_HIP_CHECK(hipSetDevice(ctx->device));
  constexpr uint32_t n = 4096;
  float *buf = nullptr;
  ADMM_HIP_CHECK(hipMalloc(&buf, sizeof(float) * n));
  std::vector<float> host(n);
  int rc_out = ADMM_OK;
  auto run = [&](void *comm, float base, double expect_scale, double expect_base_sum, const char *what) -> int {
    hipLaunchKernelGGL(k_comm_fill, dim3(n / 256), dim3(256), 0, ctx->stream, buf, n, base);
    const int rc = g_rccl.all_reduce(buf, buf, (size_t)n, 7, 0, comm, ctx->stream);
    if (rc != 0) {
      set_error("admm_comm_selftest (%s): ncclAllReduce failed: %s", what, rccl_err(rc));
      return ADMM_ERR_COMM;
    }
    if (hipMemcpyAsync(host.data(), buf, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
      set_error("admm_comm_selftest (%s): copy back failed", what);
      return ADMM_ERR_HIP;
    }
    for (uint32_t i = 0; i < n; ++i) {
      const double want = expect_base_sum + expect_scale * (double)(i & 7);
      if ((double)host[i] != want) {
        set_error("admm_comm_selftest (%s): element %u is %g, expected %g", what, i, (double)host[i], want);
        return ADMM_ERR_COMM;
      }
    }
    return ADMM_OK;
  };
  {
    UniqueId id;
    void *self = nullptr;
    int rc = g_rccl.get_unique_id(&id);
    if (rc == 0) rc = g_rccl.comm_init_rank(&self, 1, id, 0);
    if (rc != 0) {
      set_error("admm_comm_selftest: one-rank communicator: %s", rccl_err(rc));
      rc_out = ADMM_ERR_COMM;
    } else {
      rc_out = run(self, 3.0f, 1.0, 3.0, "one rank");
      if (g_rccl.comm_destroy) g_rccl.comm_destroy(self);
    }
  }
  if (rc_out == ADMM_OK && ctx->world > 1) {
    if (!ctx->comm) {
      set_error("admm_comm_selftest: world %d but no communicator (admm_comm_init)", ctx->world);
      rc_out = ADMM_ERR_INVALID;
    } else {
      // rank r contributes (r + 1) + (i & 7): the sum is world (world + 1) / 2 + world (i & 7)
      const double w = (double)ctx->world;
      rc_out = run(ctx->comm, (float)(ctx->rank + 1), w, w * (w + 1.0) / 2.0, "all ranks");
    }
  }
  hipFree(buf);
  return rc_out;
}
