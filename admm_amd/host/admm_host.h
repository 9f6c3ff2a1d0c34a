// Host-side classes of the ADMM / FTRL trainer on top of libadmm_hip.so.
//
// They carry the names and call order of the reference's C++ surface (src/sample_set.h,
// src/ftrl.h, src/workers.h, src/master.h, src/metrics.h, src/arg_parser.h, src/config.h,
// src/admm.h of wuyunhao/admm) so that its three programs (admm, ftrl, evan) can be written
// against them almost line for line -- but every vector lives in HBM and every loop over samples
// is a kernel behind the C ABI of include/admm_hip.h.  Failures follow the reference's
// conventions: bool / message to stdout for I/O problems, abort for broken invariants (CHECK).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "admm_hip.h"

#define ADMM_CHECK(cond)                                                                 \
  do {                                                                                   \
    if (!(cond)) {                                                                       \
      std::fprintf(stderr, "Check failed: %s (%s:%d) %s\n", #cond, __FILE__, __LINE__,   \
                   admm_last_error());                                                   \
      std::abort();                                                                      \
    }                                                                                    \
  } while (0)
#define ADMM_CALL(expr) ADMM_CHECK((expr) == ADMM_OK)

namespace admm {

// ---- process / device ------------------------------------------------------------------------

// What rabit provides to the reference's programs: rank, world size, the tracker's print channel
// and the sum all-reduce (rabit::Init/Finalize/GetRank/GetWorldSize/TrackerPrintf/Allreduce).
// Rank and world come from the launcher's environment (DMLC_TASK_ID / DMLC_NUM_WORKER, or
// RANK / WORLD_SIZE); one process drives one GPU (LOCAL_RANK, else rank modulo visible GPUs).
class Runtime {
 public:
  static Runtime &Get();
  void Init(int argc, char *argv[]);
  void Finalize();
  int GetRank() const { return rank_; }
  int GetWorldSize() const { return world_; }
  admm_ctx *ctx() const { return ctx_; }
  void TrackerPrintf(const char *fmt, ...) const;
  // in-place float sum over ranks of a device buffer: RCCL, or -- when several ranks share one GPU
  // (ADMM_ALLREDUCE=shm, set by the local launcher) -- a host shared-memory segment in which every
  // rank adds the ranks' buffers in rank order
  void AllreduceSum(float *device_buf, size_t n);

 private:
  void ShmAllreduce(float *device_buf, size_t n);
  void ShmBarrier();
  int rank_ = 0, world_ = 1;
  admm_ctx *ctx_ = nullptr;
  bool use_shm_ = false;
  std::string shm_path_;
  void *shm_ = nullptr;
  size_t shm_bytes_ = 0, shm_n_ = 0;
  uint32_t shm_sense_ = 0;
};

// dim floats in HBM with value semantics close enough to std::vector<float> for the mains
class DeviceVector {
 public:
  DeviceVector() = default;
  explicit DeviceVector(size_t n) { resize(n); }
  DeviceVector(const DeviceVector &o) { *this = o; }
  DeviceVector &operator=(const DeviceVector &o);
  DeviceVector(DeviceVector &&o) noexcept { swap(o); }
  DeviceVector &operator=(DeviceVector &&o) noexcept {
    swap(o);
    return *this;
  }
  ~DeviceVector() { clear(); }
  void swap(DeviceVector &o) noexcept {
    std::swap(ptr_, o.ptr_);
    std::swap(n_, o.n_);
  }
  void resize(size_t n);  // zero-filled, as resize + std::fill(…, 0) in the reference
  void clear();
  void fill_zero();
  size_t size() const { return n_; }
  float *data() { return ptr_; }
  const float *data() const { return ptr_; }
  std::vector<float> to_host() const;
  void from_host(const std::vector<float> &h);

 private:
  float *ptr_ = nullptr;
  size_t n_ = 0;
};

// ---- configuration (src/config.h) --------------------------------------------------------------

class AdmmConfig {
 public:
  typedef float real_t;
  real_t step_size = 0;
  real_t global_var = 0;
  real_t bias_var = 0;
  real_t ftrl_alpha = 0;
  std::size_t dim = 0;
  DeviceVector global_weights;
  int passes = 0;
  std::string train_path, test_path, output_path;
  // keys this build adds; absent keys keep the reference's behaviour
  real_t global_l1 = 0;                // soft threshold of the consensus update
  std::string local_solver = "ftrl";   // ftrl | lbfgs
  uint64_t ftrl_batch = 1;             // 1 = the reference's row-by-row FTRL
  int global_metrics = 0;              // 1: rank 0 also reports logloss / AUC over all ranks' test rows
  admm_lbfgs_params lbfgs = {6, 10, 20, 1e-5f, 1e-4f, 0.9f, 0};
};

class FtrlConfig {
 public:
  typedef float real_t;
  FtrlConfig() {}
  explicit FtrlConfig(const AdmmConfig &a) : alpha(a.ftrl_alpha), beta(1), dim(a.dim) {}
  real_t alpha = 0, beta = 0, l_1 = 0, l_2 = 0;
  std::size_t dim = 0;
  int passes = 0;
  std::string train_path, test_path, output_path;
  uint64_t ftrl_batch = 1;
};

// src/arg_parser.h
class ArgParser {
 public:
  void ADMMParse(const char *file, AdmmConfig &admm_params);
  void FTRLParse(const char *file, FtrlConfig &ftrl_params);
  std::vector<std::string> psid;
  std::vector<int> num_part;
  std::string load_path;
};

// ---- data (src/sample_set.h) -------------------------------------------------------------------

struct Row {  // the fields of dmlc::Row the reference touches
  float label = 0, weight = 1;
  size_t length = 0;
  const size_t *index = nullptr;
  const float *value = nullptr;
};

class SampleSet {
 public:
  typedef std::size_t IndexType;
  SampleSet();
  virtual ~SampleSet();
  // Parses part `part_index` of `num_parts` of a libsvm file and uploads it (bias column added,
  // indices shifted by one, src/sample_set.cc:70-88) into HBM.  dim = 0 sizes the set by its
  // largest index.  Returns false when the file cannot be opened.
  bool Initialize(const std::string &uri, uint32_t part_index, uint32_t num_parts, size_t dim = 0);
  // The host half of Initialize: part `part_index` of `num_parts` of a libsvm text file as CSR
  // arrays without the bias column (several threads; ADMM_PARSE_THREADS, default min(cores, 16)).
  // With ADMM_CSR_CACHE_DIR set the arrays are also kept there as a binary file that later runs
  // read instead of parsing, for as long as the text file's size and modification time match.
  static bool ParseText(const std::string &uri, uint32_t part_index, uint32_t num_parts,
                        std::vector<uint64_t> &rowptr, std::vector<uint32_t> &col, std::vector<float> &val,
                        std::vector<float> &label, uint32_t *max_index);
  bool Next();
  Row GetData();
  void Rewind();
  Row TranslateData(const Row &x);  // caller owns index/value (delete[]), as in the reference
  Row GetLastData();
  int Size();
  // device side
  admm_csr *device() const { return csr_; }
  void PrepareTransposed(uint64_t panel_rows = 0, uint32_t col_block = 0, bool relabel = true, bool lean = false);
  uint64_t nnz_with_bias() const { return nnz_ + rows_; }

 protected:
  void Release();
  void EnsureHost();  // the row cursor's host arrays, fetched back from the device when first needed
  admm_csr *csr_ = nullptr;
  std::vector<uint64_t> rowptr_;
  std::vector<uint32_t> col_;
  std::vector<size_t> col_wide_;
  std::vector<float> val_, label_;
  uint64_t rows_ = 0, nnz_ = 0;
  int rb_size_ = -1, current_index_ = -1;
  bool block_served_ = false;
};

// ---- solvers ----------------------------------------------------------------------------------

}  // namespace admm

namespace ftrl {

// src/ftrl.h.  weight_/mid_weight_/squared_sum_ are device vectors.
class FtrlSolver {
 public:
  typedef float real_t;
  typedef ::admm::SampleSet SampleSet;
  FtrlSolver() {}
  FtrlSolver(real_t lambda_1, real_t lambda_2, real_t alpha_init, real_t beta_init, std::size_t dim_init);
  void Init(::admm::FtrlConfig &params);
  void Assign(const ::admm::DeviceVector &x, const ::admm::DeviceVector &y);
  // one pass over train_set (FtrlSolver::Run, src/ftrl.cc:101-145); test_set is unused, as there
  void Run(SampleSet &train_set, SampleSet &test_set, const ::admm::DeviceVector &offset,
           const ::admm::DeviceVector &reg_offset);
  // one row at a time, as FtrlSolver::Run does inside (src/ftrl.cc:54-88): x is a translated row
  // (SampleSet::TranslateData); one launch and one host wait per call
  real_t Predict(::admm::Row &x, const ::admm::DeviceVector &offset, const ::admm::DeviceVector &reg_offset);
  void Update(real_t predict, const ::admm::Row &x, const ::admm::DeviceVector &reg_offset);
  ::admm::DeviceVector weight() const { return weight_; }

  ::admm::DeviceVector weight_, mid_weight_, squared_sum_;
  real_t l_1_ = 0, l_2_ = 0, alpha_ = 0, beta_ = 0;
  std::size_t dim_ = 0;
  uint64_t batch_ = 1;
  int num_part_ = 1;
  std::string psid_;
};

}  // namespace ftrl

namespace admm {

// src/workers.h
class Worker {
 public:
  typedef float real_t;
  Worker() {}
  virtual ~Worker() {}
  void InitWorker(std::size_t fdim);
  void BaseUpdate(SampleSet &train_set, SampleSet &test_set, const AdmmConfig &admm_params);
  void BiasUpdate(SampleSet &train_set, SampleSet &test_set, const AdmmConfig &admm_params);
  void LangrangeUpdate(const AdmmConfig &admm_params);
  void GetWeights(AdmmConfig &admm_params, DeviceVector &out) const;
  void LoadParts(const AdmmConfig &admm_params);
  // resident training parts (parsed once, not once per iteration as the reference does)
  std::vector<std::unique_ptr<SampleSet>> parts_;
  admm_lbfgs_result last_solve_ = {0, 0, 0, 0, 0};

  std::string psid_;
  int num_part_ = 1;
  int partid_ = 0;
  DeviceVector base_vec_, bias_vec_, langr_vec_, nt_vec_, zt_vec_;
  bool load = false;
};

// src/master.h
class Master {
 public:
  typedef float real_t;
  bool GlobalUpdate(const DeviceVector &workers, AdmmConfig &admm_params, int num_worker);
  // GlobalUpdate + Worker::LangrangeUpdate + residual norms in one kernel
  bool Consensus(const DeviceVector &workers, AdmmConfig &admm_params, int num_worker, Worker &worker,
                 double *r2, double *s2);
};

}  // namespace admm

// src/metrics.h (single-model forms)
class Metrics {
 public:
  typedef float real_t;
  // tied rank function (src/metrics.cc:14-46), computed on the device
  void Sort(const std::vector<real_t> &input, std::vector<real_t> &output);
  real_t Auc(::admm::SampleSet &sample_set, std::vector<const ::admm::DeviceVector *> &weights, bool T);
  real_t LogLoss(::admm::SampleSet &sample_set, std::vector<const ::admm::DeviceVector *> &weights, bool T);
  // both in one sweep
  void Both(::admm::SampleSet &sample_set, std::vector<const ::admm::DeviceVector *> &weights, real_t *loss,
            real_t *auc);
  // Logloss and AUC over the test rows of all ranks (the reference reports them per rank only):
  // every rank predicts into its slice of a zero-filled buffer, a sum all-reduce concatenates the
  // slices, and the device AUC runs once over the whole.  Collective: every rank must call it.
  void AllRanks(::admm::Runtime &rt, ::admm::SampleSet &sample_set, std::vector<const ::admm::DeviceVector *> &weights,
                double *loss, double *auc, uint64_t *rows);
};

// src/admm.h: text (de)serialisation of the worker and global state
class LocalModel {
 public:
  ::admm::Worker worker_processor_;
  bool Load(const std::string &path);
  void Save(const std::string &path) const;        // <psid>.bin
  void SaveState(const std::string &path) const;   // admm_weight_<psid>
  void InitModel(std::size_t fdim, ::admm::ArgParser &arg_parser);
};

class GlobalModel {
 public:
  ::admm::AdmmConfig admm_params_;
  bool Load(const std::string &path);
  void Save(const std::string &path) const;  // global_params_<psid>
  void InitModel(const char *conf, ::admm::ArgParser &arg_parser) { arg_parser.ADMMParse(conf, admm_params_); }
};
