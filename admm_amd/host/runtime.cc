// Process set-up, device vectors and the all-reduce for the host programs.
#include <chrono>
#include <cstdarg>
#include <cstring>
#include <fstream>
#include <thread>

#include "admm_host.h"

namespace admm {

static int env_int(const char *a, const char *b, int dflt) {
  for (const char *name : {a, b}) {
    if (!name) continue;
    const char *v = std::getenv(name);
    if (v && *v) return std::atoi(v);
  }
  return dflt;
}

Runtime &Runtime::Get() {
  static Runtime r;
  return r;
}

void Runtime::Init(int, char *[]) {
  // launcher contract of dmlc_local.py / tracker.py: DMLC_TASK_ID, DMLC_NUM_WORKER
  rank_ = env_int("DMLC_TASK_ID", "RANK", 0);
  world_ = env_int("DMLC_NUM_WORKER", "WORLD_SIZE", 1);
  int ndev = 0;
  ADMM_CALL(admm_device_count(&ndev));
  const int dev = env_int("LOCAL_RANK", nullptr, rank_) % (ndev > 0 ? ndev : 1);
  ADMM_CALL(admm_ctx_create(dev, nullptr, &ctx_));
  if (world_ > 1) {
    // rank 0 publishes the communicator id in a file every rank can read
    const char *dir = std::getenv("ADMM_RENDEZVOUS_DIR");
    const char *job = std::getenv("DMLC_TRACKER_PORT");
    std::string path = std::string(dir && *dir ? dir : "/tmp") + "/admm_uid_" + (job && *job ? job : "local");
    char id[128];
    if (rank_ == 0) {
      ADMM_CALL(admm_comm_unique_id(id));
      std::ofstream f(path + ".tmp", std::ios::binary);
      f.write(id, sizeof(id));
      f.close();
      std::rename((path + ".tmp").c_str(), path.c_str());
    } else {
      for (int tries = 0;; ++tries) {
        std::ifstream f(path, std::ios::binary);
        if (f && f.read(id, sizeof(id))) break;
        ADMM_CHECK(tries < 6000);  // ten minutes
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
      }
    }
    ADMM_CALL(admm_comm_init(ctx_, rank_, world_, id));
    if (rank_ == 0) std::remove(path.c_str());
  }
}

void Runtime::Finalize() {
  if (ctx_) admm_ctx_destroy(ctx_);
  ctx_ = nullptr;
}

void Runtime::TrackerPrintf(const char *fmt, ...) const {
  va_list ap;
  va_start(ap, fmt);
  std::vprintf(fmt, ap);
  va_end(ap);
  std::fflush(stdout);
}

void Runtime::AllreduceSum(float *device_buf, size_t n) { ADMM_CALL(admm_allreduce_sum_f32(ctx_, device_buf, n)); }

DeviceVector &DeviceVector::operator=(const DeviceVector &o) {
  if (this == &o) return *this;
  if (n_ != o.n_) {
    clear();
    if (o.n_) resize(o.n_);
  }
  if (n_) ADMM_CALL(admm_vec_copy(Runtime::Get().ctx(), ptr_, o.ptr_, n_));
  return *this;
}

void DeviceVector::resize(size_t n) {
  if (n == n_) {
    fill_zero();
    return;
  }
  clear();
  if (n) ADMM_CALL(admm_vec_alloc(Runtime::Get().ctx(), n, &ptr_));
  n_ = n;
}

void DeviceVector::clear() {
  if (ptr_ && Runtime::Get().ctx()) admm_vec_free(Runtime::Get().ctx(), ptr_);
  ptr_ = nullptr;
  n_ = 0;
}

void DeviceVector::fill_zero() {
  if (n_) ADMM_CALL(admm_vec_zero(Runtime::Get().ctx(), ptr_, n_));
}

std::vector<float> DeviceVector::to_host() const {
  std::vector<float> h(n_);
  if (n_) ADMM_CALL(admm_vec_download(Runtime::Get().ctx(), h.data(), ptr_, n_));
  return h;
}

void DeviceVector::from_host(const std::vector<float> &h) {
  if (h.size() != n_) resize(h.size());
  if (n_) ADMM_CALL(admm_vec_upload(Runtime::Get().ctx(), ptr_, h.data(), n_));
}

}  // namespace admm
