// Process set-up, device vectors and the all-reduce for the host programs.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
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
  const char *dir = std::getenv("ADMM_RENDEZVOUS_DIR");
  const char *job = std::getenv("DMLC_TRACKER_PORT");
  const std::string base = std::string(dir && *dir ? dir : "/tmp") + "/admm_";
  const std::string jobid = job && *job ? job : "local";
  const char *mode = std::getenv("ADMM_ALLREDUCE");
  use_shm_ = world_ > 1 && mode && std::strcmp(mode, "shm") == 0;
  shm_path_ = base + "shm_" + jobid;
  if (world_ > 1 && !use_shm_) {
    // rank 0 publishes the communicator id in a file every rank can read
    std::string path = base + "uid_" + jobid;
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
  if (shm_) {
    ShmBarrier();
    munmap(shm_, shm_bytes_);
    shm_ = nullptr;
    if (rank_ == 0) unlink(shm_path_.c_str());
  }
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

void Runtime::AllreduceSum(float *device_buf, size_t n) {
  if (world_ == 1) return;
  if (use_shm_) ShmAllreduce(device_buf, n);
  else ADMM_CALL(admm_allreduce_sum_f32(ctx_, device_buf, n));
}

// ---- all-reduce through host shared memory ------------------------------------------------------
//
// Segment layout: a header of two counters (arrivals, generation), then one slot of n floats per
// rank.  Rank 0 creates the file at the first call (slots sized for that call's length, at least
// 64 Ki floats); the others wait for it to reach its full size.  A longer vector goes through in
// slot-sized pieces.

namespace {
struct ShmHeader {
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  char pad[56];
};
static_assert(sizeof(ShmHeader) == 64, "header is one cache line");
}  // namespace

void Runtime::ShmBarrier() {
  ShmHeader *h = static_cast<ShmHeader *>(shm_);
  const uint32_t gen = h->generation.load(std::memory_order_acquire);
  if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world_) {
    h->arrived.store(0, std::memory_order_relaxed);
    h->generation.store(gen + 1, std::memory_order_release);
  } else {
    for (uint64_t spins = 0; h->generation.load(std::memory_order_acquire) == gen; ++spins) {
      if (spins > 1000) std::this_thread::sleep_for(std::chrono::microseconds(50));
      ADMM_CHECK(spins < 20000000ull);  // a rank died: give up after about a quarter of an hour
    }
  }
}

void Runtime::ShmAllreduce(float *device_buf, size_t n) {
  if (!shm_) {
    shm_n_ = std::max<size_t>(n, 65536);
    shm_bytes_ = sizeof(ShmHeader) + sizeof(float) * shm_n_ * (size_t)world_;
    int fd = -1;
    if (rank_ == 0) {
      const std::string tmp = shm_path_ + ".tmp";
      fd = open(tmp.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0600);
      ADMM_CHECK(fd >= 0);
      ADMM_CHECK(ftruncate(fd, (off_t)shm_bytes_) == 0);  // zero-filled: counters start at 0
      ADMM_CHECK(std::rename(tmp.c_str(), shm_path_.c_str()) == 0);
    } else {
      for (int tries = 0;; ++tries) {
        struct stat st;
        fd = open(shm_path_.c_str(), O_RDWR);
        if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size == shm_bytes_) break;
        if (fd >= 0) close(fd);
        ADMM_CHECK(tries < 6000);  // ten minutes
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
      }
    }
    shm_ = mmap(nullptr, shm_bytes_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    ADMM_CHECK(shm_ != MAP_FAILED);
  }
  float *slots = reinterpret_cast<float *>(static_cast<char *>(shm_) + sizeof(ShmHeader));
  for (size_t done = 0; done < n; done += shm_n_) {
    const size_t m = std::min(shm_n_, n - done);
    ADMM_CALL(admm_vec_download(ctx_, slots + (size_t)rank_ * shm_n_, device_buf + done, m));
    ShmBarrier();
    std::vector<float> sum(slots, slots + m);  // rank 0's slot, then the others in rank order
    for (int r = 1; r < world_; ++r) {
      const float *s = slots + (size_t)r * shm_n_;
      for (size_t i = 0; i < m; ++i) sum[i] += s[i];
    }
    ShmBarrier();  // nobody refills its slot before every rank has read all of them
    ADMM_CALL(admm_vec_upload(ctx_, device_buf + done, sum.data(), m));
  }
}

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
